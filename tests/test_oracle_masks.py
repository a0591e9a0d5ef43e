"""Pins the mask-ingestion part of the oracle (RLE codec, instance filters) to outputs of the reference's own
binary_mask_to_rle / analyze_mask / get_maximum_height (tests/golden/g8_masks.npz).  CPU only."""
import numpy as np

from oracle import la3d_oracle as O


def _split(g):
    offs = np.concatenate([[0], np.cumsum(g["lens"])])
    return [g["counts"][offs[i]:offs[i + 1]] for i in range(len(g["lens"]))]


def test_rle_encode_matches_reference(golden):
    g = golden("g8_masks.npz")
    for m, want in zip(g["masks"], _split(g)):
        rle = O.rle_encode(m)
        assert rle["size"] == list(m.shape)
        assert rle["counts"] == want.tolist()


def test_rle_decode_inverts_reference_encoder(golden):
    g = golden("g8_masks.npz")
    H, W = g["masks"].shape[1:]
    for m, counts in zip(g["masks"], _split(g)):
        np.testing.assert_array_equal(O.rle_decode(counts, H, W), m.astype(bool))
        s = O.rle_to_string(counts)                       # compressed COCO string form round-trips
        assert O.rle_from_string(s) == counts.tolist()
        assert all(48 <= ch < 48 + 64 for ch in s)
    # known string from the COCO API documentation style: counts [6, 1, 40, 4, 5, 4, 5, 4, 21]
    assert O.rle_from_string(O.rle_to_string([6, 1, 40, 4, 5, 4, 5, 4, 21])) == [6, 1, 40, 4, 5, 4, 5, 4, 21]
    # truncated / over-long run lists are clipped to the frame
    assert O.rle_decode([5, 10 ** 6], 4, 4).sum() == 11


def test_mask_stats_match_reference_filters(golden):
    g = golden("g8_masks.npz")
    H, W = g["masks"].shape[1:]
    for m, ref in zip(g["masks"], g["ref_stats"]):
        area, rows, span, trunc = O.mask_stats(m)
        assert (area, rows, span) == tuple(ref[:3])
        assert (trunc >= 10) == bool(ref[3])              # analyze_mask's is_truncated
        assert (area >= 100) == bool(ref[4])              # analyze_mask's is_scaleable
        for from_rle in (True, False):
            height = rows if from_rle else span
            want = (height / H > 0.0625) and not bool(ref[3]) and bool(ref[4])   # src/util.py:375
            assert O.keep_instance((area, rows, span, trunc), H, from_rle) == want


def test_box_consumers_match_reference(golden):
    """project_to_2d / bbox2D_proj / bbox2D_trunc / iou2D of the reference (src/tools/combine_results.py)."""
    g = golden("g9_consumers.npz")
    np.testing.assert_allclose(O.project_boxes(g["records"], g["K"], tuple(g["image_size"])), g["boxes2d"], rtol=1e-14)
    np.testing.assert_allclose(O.iou2d_matrix(g["iou_a"], g["iou_b"]), g["iou"], rtol=1e-14, atol=1e-15)
