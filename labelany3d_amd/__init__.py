"""labelany3d_amd — MI355X-native (gfx950) implementation of LabelAny3D's geometric hot path:
depth back-projection (reference src/util.py:52-75) and oriented 3D box fitting (reference
src/util_3dbox.py:106-178), behind the reference's own function signatures.

    from labelany3d_amd import fit_instances            # batched: depth + masks -> (B,39) boxes on the GPU
    from labelany3d_amd.util_3dbox import estimate_bbox # drop-in scalar signature
    from labelany3d_amd.util import depth_to_points

All compute goes through libla3d.so (hand-written HIP, C-ABI in include/la3d.h); there is no CPU path.
"""
from ._lib import REC, AUX, NSAMPLE, La3dError  # noqa: F401
from .batched import (  # noqa: F401
    InstanceFitter,
    draw_sample_idx,
    fit_instances,
    fit_points,
    mask_counts,
    unproject,
    unpack_boxes,
)

from .consumers import correspondences_to_world, hungarian_matching, iou2d_matrix, project_boxes, unproject_matches  # noqa: E402,F401
from .masks import (annotation_areas, filter_annotations, fit_annotations, fit_annotations_all, fit_instances_ex, fit_instances_poly, fit_instances_rle, keep_instances, mask_stats, mask_stats_poly,  # noqa: E402,F401
                    mask_stats_rle, masked_ratio_median, pack_polygons, pack_rle, pad_depth_rows, padded_width, poly_decode, rle_decode, rle_from_string,
                    segmentations_to_masks)  # noqa: E402,F401

from .depth_align import align_apply, align_depth, align_select, align_select_batch, depth_match_transform  # noqa: E402,F401

from .pipeline import fit_batches  # noqa: E402,F401
from .options import scheduling  # noqa: E402,F401

__all__ = ["fit_instances_ex", "fit_annotations", "fit_annotations_all", "annotation_areas", "correspondences_to_world", "fit_batches", "scheduling", "align_depth", "align_select", "align_select_batch", "align_apply", "depth_match_transform", "fit_instances_poly", "pack_polygons", "poly_decode", "mask_stats_poly", "segmentations_to_masks", "unproject_matches", "masked_ratio_median", "project_boxes", "iou2d_matrix", "hungarian_matching", "fit_instances_rle", "rle_decode", "filter_annotations", "mask_stats", "mask_stats_rle", "keep_instances", "pack_rle", "pad_depth_rows", "padded_width", "rle_from_string","fit_instances", "fit_points", "mask_counts", "unproject", "draw_sample_idx", "unpack_boxes",
           "InstanceFitter", "La3dError", "REC", "AUX", "NSAMPLE"]
