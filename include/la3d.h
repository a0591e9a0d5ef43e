/* la3d.h — C-ABI of the MI355X-native LabelAny3D geometric hot path (libla3d.so, gfx950).
 *
 * The reference has no FFI layer: the path is three module-level Python/NumPy functions
 * resolved by name (reference src/batch_scripts/whole.py:10,15-16).  This ABI is what a
 * binding for that path calls; labelany3d_amd/{util,util_3dbox}.py bind it with ctypes and
 * keep the reference's Python signatures.
 *
 * Conventions
 *  - every `const T* dev` / `T* dev` argument is a DEVICE pointer (tensor.data_ptr());
 *    arguments documented as HOST are read synchronously before the call returns.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *    asynchronous with respect to the host; the caller owns every buffer.
 *  - no call throws, allocates device memory or synchronises the device.
 *  - return value: LA3D_SUCCESS or a negative LA3D_ERR_*; la3d_last_error() gives text.
 *  - per-box failures (the reference's ValueError cases) are reported in `status[B]`,
 *    never as a failed call.
 */
#ifndef LA3D_H
#define LA3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LA3D_ABI_VERSION 2
#define LA3D_REC 39      /* doubles per box: center_cam[3] dimensions[3]=(dz,dy,dx) R_cam[9] bbox3D_cam[8][3] */
#define LA3D_AUX 4       /* doubles per box: yaw, n_valid, n_in (mask pixels / cloud points), eigen-gap (l1-l2)/l1;
                            with LA3D_METHOD_CONVEX_HULL aux[3] = -(hull vertices) when the hull decided the yaw
                            and stays >= 0 when the reference's PCA fallback was taken (:222-224).
                            The PCA axis is conditioned like 1 / gap.  gap == 0 says "axis unresolved - any yaw is as good":
                            an exact tie of the two eigenvalues or a footprint without any spread (every point at the same
                            (x', z'): the reference's own axis is then the SVD of rounding noise).  A footprint whose spread
                            is below ~1/360 of its distance from the origin of the sums - the raw second moments cancel to
                            rounding noise, kappa = (sum x^2 + sum z^2) / (n l1) > 2^17 - is resolved by a second moments
                            pass about the mean (every engine, every point-cloud kernel) and reports its true gap
                             */
#define LA3D_NSAMPLE 500 /* reference src/util_3dbox.py:123-125 */

/* call status */
#define LA3D_SUCCESS 0
#define LA3D_ERR_ARG (-1)
#define LA3D_ERR_UNSUPPORTED (-2)
#define LA3D_ERR_HIP (-3)

/* per-box status (int32) — the reference's error behaviour, src/util_3dbox.py */
#define LA3D_BOX_OK 0         /* a box was fitted                                                        */
#define LA3D_BOX_EMPTY 1      /* ValueError("No valid points after removing NaN values")  (:142-143)     */
#define LA3D_BOX_BAD_GROUND 2 /* ground parallel/antiparallel to [0,-1,0] or zero: NaN rotation (:37-55)   */
#define LA3D_BOX_TOO_FEW 3    /* one valid point: scikit-learn PCA(2) ValueError (:183-184)              */
#define LA3D_BOX_NONFINITE 4  /* +-inf coordinate reaches PCA: scikit-learn ValueError (:183-184)        */
#define LA3D_BOX_UNSUPPORTED 5 /* convex_hull on more than 2048 valid points (the reference feeds it <= 500) */
#define LA3D_BOX_FILTERED 6   /* dropped by the instance filter of the *_filtered entry points (src/util.py:375)        */

/* yaw method (reference src/util_3dbox.py:146-151) */
#define LA3D_METHOD_PCA 0
#define LA3D_METHOD_CONVEX_HULL 1
/* OR-ed into `method` of la3d_fit_points: the caller promises that no cloud has more than a few thousand rows to visit (after
 * sampling) - true for the reference's own calls (500 mesh samples per object).  PCA method: one WAVE per cloud instead of one
 * workgroup - no LDS, no barriers, four clouds per workgroup.  Larger clouds stay correct, only slow. */
#define LA3D_HINT_SMALL_CLOUDS 0x100
#define LA3D_HINT_HULL_512 0x200     /* OR into `method` (convex hull): no cloud holds more than 512 valid rows - the reference's own call path
                                        (<= 500 mesh samples) - so the kernel takes its small-LDS form (eight workgroups per CU instead of
                                        three); implied by a sample_idx array.  A cloud that breaks the promise gets LA3D_BOX_UNSUPPORTED. */

int la3d_version(void);
/* "LA3D_BUILD_INFO:<sha256 of the sources this library was compiled from>:<sha256 of the compile command>" - static storage. */
const char* la3d_build_info(void);
const char* la3d_last_error(void);

/* Replaces depth_to_points(depth, K, R, t) — reference src/util.py:52-75 (caller
 * src/batch_scripts/depth.py:154).  depth: dev f32 [H*W] (batch element 0, as the reference
 * returns only that, :75).  K9: HOST f64[9] row-major pixel intrinsics.  Rt12: HOST f64[12]
 * = R row-major (9) then t (3), or NULL for identity.  out: dev [H*W*3], f64 when
 * out_is_f64 != 0 (the reference's dtype) else f32.  u = column, v = row, no half-pixel. */
int la3d_unproject(const float* depth, const double* K9, const double* Rt12, int H, int W,
                   void* out, int out_is_f64, void* stream);

/* The same for P frames in one launch (a dataset stage unprojects every image: the reference loops over them,
 * src/batch_scripts/depth.py:138-160).  depth: dev f32 [P][H*W]; K: DEVICE f64, 9 per frame (k_stride >= 9) or one shared
 * matrix (k_stride 0), inverted in the kernel with the same elimination as the host routine of la3d_unproject;
 * Rt12 as above (shared by all frames) or NULL; out: dev [P][H*W*3]. */
int la3d_unproject_batch(const float* depth, const double* K, int32_t k_stride, const double* Rt12, int P, int H, int W,
                         void* out, int out_is_f64, void* stream);

/* Number of True pixels per mask plane: what the reference sees as in_pc.shape[0]
 * (src/util_3dbox.py:123) when fed pts[mask].  mask: dev u8 [B][H*W] (non-zero = True);
 * counts: dev i32 [B].  The host needs it to draw np.random.randint(0, N, 500). */
int la3d_mask_counts(const uint8_t* mask, int B, int H, int W, int32_t* counts, void* stream);

/* Depth rows padded on the right with zeros: src dev f32 [rows][W] -> dst dev f32 [rows][Wp] (Wp >= W, Wp % 4 == 0, dst 16-byte
 * aligned; rows = planes x H).  What la3d_fit_args::frame_width wants for frames whose width is not a multiple of 32 (run-length /
 * polygon masks on COCO's 427 / 500 / 375 / 333-wide images): Wp = the next multiple of 32. */
int la3d_pad_rows(const float* src, int64_t rows, int W, int Wp, float* dst, void* stream);

/* The library keeps NO mutable process state (re-entrant, no global state).  How a call is scheduled - which
 * engine, whether the size-balanced launch order runs, which build of the instance kernel - is decided per call from its
 * arguments; the three `opt_*` fields of la3d_fit_args override the decision for one call (0 = the library's choice).  The
 * LA3D_* environment variables the measurement scripts use (LA3D_ENGINE, LA3D_BALANCE, LA3D_BUILD, ...) are read ONCE, at
 * the first call, into an immutable table of process defaults; they never change records, only speed.
 * (ABI 2: la3d_set_launch_order / la3d_get_launch_order of ABI 1 - a process-wide switch - are gone; use opt_launch_order.) */
#define LA3D_ENGINE_DEFAULT 0
#define LA3D_ENGINE_INSTANCE 1        /* one workgroup per instance */
#define LA3D_ENGINE_SPLIT 2           /* band scan + tile-range-balanced passes (falls back to the instance engine where it does not apply) */
#define LA3D_ENGINE_BAND 3            /* two, four or eight workgroups per instance, one per band of tile rows (u8 planes, tiled frames; falls back likewise) */
#define LA3D_ENGINE_ROWS 4            /* up to sixteen workgroups per instance, one per band of rows (u8 planes, no ground array, at most 512
                                         instances; falls back likewise).  Round 6: ONE launch - the band that finishes last merges its
                                         instance's partial sums and writes the record */
#define LA3D_ENGINE_ROWS2 5           /* the row engine in its round-5 form: the partial sums merged by a second short launch (what a call
                                         captured into a HIP graph takes anyway) */
#define LA3D_ORDER_DEFAULT 0          /* size-balanced launch order for 256 < B <= 3 resident sets; the sort keys are estimated inside the
                                         fit kernel and handed over through the workspace, so ONE workspace serves ONE call at a time, and
                                         several ORDERED calls running concurrently on different streams slow each other down (a call whose
                                         workgroups are not all resident waits ~0.25 ms before computing its neighbours' keys itself):
                                         pipelined callers pass LA3D_ORDER_OFF, as labelany3d_amd/pipeline.py does */
#define LA3D_ORDER_OFF 1              /* a caller pipelining independent batches on several streams wants it off (measured +20 %) */
#define LA3D_ORDER_ON 2
#define LA3D_BUILD_DEFAULT 0          /* 64 VGPRs, four workgroups per CU; un-grounded, skew-free cameras take the separable SINGLE pass
                                         (round 5: one walk over the depth, extents from per-column depth ranges), every other call two passes */
#define LA3D_BUILD_PLAIN 1            /* the same build pinned to its two-pass form (pass-B tile culling) for every camera */
#define LA3D_BUILD_NOCULL 2           /* the two-pass form that walks EVERY active tile in pass B (no culling plan): the reference the culling
                                         tests compare with, never faster */
#define LA3D_BUILD_RETAINING LA3D_BUILD_NOCULL   /* rounds 2-5: a 128-VGPR build that kept depth tiles in registers between the passes -
                                         fastest nowhere since round 4 (106 vs 81 us per 1024 instances) and deleted in round 6; the value
                                         stays accepted and now selects the no-cull build (the same records: it never culled) */

/* Bytes of device scratch la3d_fit_instances needs for (B,H,W); may be 0. */
size_t la3d_workspace_bytes(int B, int H, int W);

/* The composed hot path, batched:
 *     for n in range(B):
 *         img  = image_index[n] if image_index else n
 *         pts  = depth_to_points(depth[img][None], K[img])[mask[n]]     # src/util.py:52-75, :480-481
 *         out[n] = estimate_bbox(pts, None, ground[n], 'pca')          # src/util_3dbox.py:106-178
 * depth        dev f32, planes of H*W floats, plane p at depth + p*depth_plane_stride
 *              (stride in floats; 0 = one shared plane)
 * image_index  dev i32 [B] or NULL (instance n uses plane n)
 * mask         dev u8 [B][H*W], non-zero = True (np.bool_ layout, src/util.py:367,382)
 * K            dev f64, 9 per image, image p at K + p*k_stride (k_stride 0 = shared, else >= 9)
 * ground       dev f64 [B][4] or NULL; only [:3] is used (src/util_3dbox.py:128-134); a row whose
 *              first component is NaN means "no ground" for that instance
 * sample_idx   dev i32 [B][500] or NULL.  NULL = full-mask mode (every masked pixel is used).
 *              Non-NULL = reference-subsample mode: for an instance with N > 500 masked pixels
 *              the 500 ranks (0 <= r < N, row-major order of True pixels) the reference would
 *              draw at :124 select the points; rows of instances with N <= 500 are ignored.
 * out          dev f64 [B][39]  (NaN where status != 0)
 * status       dev i32 [B]
 * aux          dev f64 [B][4] or NULL
 * workspace    dev scratch of la3d_workspace_bytes(B,H,W) bytes (may be NULL when that is 0) */
int la3d_fit_instances(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                       const uint8_t* mask, const double* K, int32_t k_stride,
                       const double* ground, const int32_t* sample_idx,
                       int B, int H, int W, double* out, int32_t* status, double* aux,
                       void* workspace, void* stream);

/* ---- mask ingestion: the data format immediately upstream of the path -------------------
 * The reference decodes COCO / COCONut annotations to (H,W) bool arrays on the CPU with pycocotools
 * (mask_utils.decode, src/util.py:367,401-402; encoder src/download_coconut.py:167-175) and filters instances by
 * area / height / border truncation (src/util.py:291-335, :375).  Run lengths are over the (H,W) mask in
 * COLUMN-major order, alternating zeros / ones, zeros first. */

/* la3d_fit_instances with the masks given as run lengths: rle_counts dev i32 [total], rle_offsets dev i64 [B+1].
 * The runs are decoded straight into the kernel's LDS bit image: no 1 B/px plane is ever materialised or read
 * (H*W <= 1048576).  All other arguments as la3d_fit_instances. */
int la3d_fit_instances_rle(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                           const int32_t* rle_counts, const int64_t* rle_offsets, const double* K, int32_t k_stride,
                           const double* ground, const int32_t* sample_idx, int B, int H, int W,
                           double* out, int32_t* status, double* aux, void* workspace, void* stream);

/* mask_utils.decode for a batch: run lengths -> u8 planes mask_out dev [B][H*W] (0/1), H*W <= 1048576. */
int la3d_rle_decode(const int32_t* counts, const int64_t* offsets, int B, int H, int W, uint8_t* mask_out, void* stream);

/* Per mask plane (dev u8 [B][H*W]) the quantities of the reference's instance filter: stats dev i32 [B][4] =
 * area, rows holding a pixel (the RLE branch's height, :368-369), last-first+1 rows (get_maximum_height, :328-335),
 * pixels inside the four `boundary`-px border strips with corners counted twice (analyze_mask, :303-322). */
int la3d_mask_stats(const uint8_t* mask, int B, int H, int W, int boundary, int32_t* stats, void* stream);

/* The same four quantities straight from COCO run lengths (layout as la3d_fit_instances_rle), by interval arithmetic on
 * the runs: no mask plane is decoded.  Replaces mask_utils.decode + np.any/np.sum + analyze_mask for RLE annotations
 * (src/util.py:364-376).  H <= 32768, H*W <= 2^30. */
int la3d_mask_stats_rle(const int32_t* counts, const int64_t* offsets, int B, int H, int W, int boundary, int32_t* stats,
                        void* stream);

/* ---- polygon segmentations: the branch every kept COCONut instance takes in the reference -----------------------
 * create_boolean_mask_from_polygon (src/util.py:386-400; producer src/download_coconut.py:178-199, :275-280): every part
 * of a segmentation is truncated to int32 vertices (np.array(polygon).reshape(-1,2).astype(np.int32) — done by the
 * caller) and filled on its own with cv2.fillPoly(mask, [points], 1): OpenCV's drawing.cpp rule for 8-bit images,
 * LINE_8, shift 0 (sides drawn with the 8-connected LineIterator + even-odd scanline fill in 16.16 fixed point; parts
 * are OR-ed).  Layout: poly_xy dev i32 [total_points][2] (x, y); ring_offsets dev i64 [R+1] point offsets of the parts;
 * inst_rings dev i64 [B+1] part offsets of the instances (instance n = parts inst_rings[n]..inst_rings[n+1]).
 * Vertices may lie outside the frame (clipped like cv::clipLine).  H*W <= 1048576. */

/* la3d_fit_instances with the masks given as polygon parts, rasterised straight into the kernel's LDS bit image (no
 * 1 B/px plane exists anywhere).  All other arguments as la3d_fit_instances. */
int la3d_fit_instances_poly(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                            const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings,
                            const double* K, int32_t k_stride, const double* ground, const int32_t* sample_idx,
                            int B, int H, int W, double* out, int32_t* status, double* aux, void* workspace, void* stream);

/* ---- instance filter fused into the fit (reference read_bounding_boxes_segmentations, src/util.py:336-383) ----------
 * The reference keeps an annotation when  height / H > 0.0625  and fewer than `max_edge` (10) mask pixels lie in the
 * `boundary`-px (10) border strips  and  area >= `min_area` (100)  (:375; analyze_mask :291-326), with height = rows
 * holding a pixel for RLE annotations (:368-369) and last row - first row + 1 for polygons (get_maximum_height,
 * :328-335).  These entry points evaluate that rule on the bit image the fit kernel has just built - no second decode /
 * rasterisation pass - write the four statistics (area, rows, span, edge; as la3d_mask_stats*) to stats (dev i32 [B][4],
 * may be NULL) and fit only the kept instances; a dropped instance gets status LA3D_BOX_FILTERED and a NaN record. */
int la3d_fit_instances_rle_filtered(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                                    const int32_t* rle_counts, const int64_t* rle_offsets, const double* K, int32_t k_stride,
                                    const double* ground, const int32_t* sample_idx, int B, int H, int W, int boundary,
                                    int min_area, int max_edge, double* out, int32_t* status, double* aux, int32_t* stats,
                                    void* workspace, void* stream);
int la3d_fit_instances_poly_filtered(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                                     const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings,
                                     const double* K, int32_t k_stride, const double* ground, const int32_t* sample_idx, int B,
                                     int H, int W, int boundary, int min_area, int max_edge, double* out, int32_t* status,
                                     double* aux, int32_t* stats, void* workspace, void* stream);

/* ---- one extensible entry: every option of the fit calls above, plus the 2-D boxes of the records -----------------
 * Exactly one of mask / rle_counts(+rle_offsets) / poly_xy(+ring_offsets, inst_rings) gives the masks.  filter_boundary >= 0
 * together with filter_max_edge > 0 switches the fused instance filter on (run-length / polygon masks; see the *_filtered
 * entry points); a zero-initialised block (`la3d_fit_args a = {0}`) therefore means NO filter, and so does
 * filter_boundary = -1.  proj != NULL adds
 * la3d_project_boxes' output for every record - bbox2D_proj (4) and bbox2D_trunc (4), reference
 * src/tools/combine_results.py:105-108, :238-252 - written by the same epilogue that writes the record (rejected / dropped
 * instances: 8 NaNs); image_width / image_height are the clamp limits.  struct_size = sizeof(la3d_fit_args) of the caller:
 * fields beyond it are taken as zero, so the struct can grow. */
typedef struct la3d_fit_args {
  int32_t struct_size;
  int32_t B, H, W;
  const float* depth; int64_t depth_plane_stride; const int32_t* image_index;
  const uint8_t* mask;
  const int32_t* rle_counts; const int64_t* rle_offsets;
  const int32_t* poly_xy; const int64_t* ring_offsets; const int64_t* inst_rings;
  const double* K; int32_t k_stride;
  int32_t filter_boundary, filter_min_area, filter_max_edge;   /* filter_boundary < 0 or filter_max_edge <= 0: no filter */
  const double* ground; const int32_t* sample_idx;
  int32_t* stats;                                              /* [B][4] | NULL (filter) */
  double* proj; double image_width, image_height;              /* [B][8] | NULL */
  double* out; int32_t* status; double* aux;
  void* workspace; void* stream;
  /* --- fields added after the first publication (struct_size tells which the caller has) --- */
  const int32_t* area_hint;   /* dev i32 [B] | NULL: mask areas in pixels the caller already knows (annotation "area", the statistics of
                                 a preceding filter): the size-balanced launch order then needs no estimate pass over the masks.  A
                                 hint only orders the work - wrong values cost speed, never correctness. */
  /* --- ABI 2: per-call scheduling overrides (speed only, never records; 0 = the library's choice) --- */
  int32_t opt_engine;         /* LA3D_ENGINE_* */
  int32_t opt_launch_order;   /* LA3D_ORDER_* */
  int32_t opt_build;          /* LA3D_BUILD_* */
  int32_t frame_width;        /* round 5 (this field was `opt_reserved, must be 0`): 0 = W.  0 < frame_width < W: the planes are W pixels wide
                                 IN MEMORY, but only the first frame_width columns are image - rows padded on the right, e.g. to a
                                 multiple of 32, which is what the tiled / single-pass forms need (a 640 x 427 frame runs 4-5 x
                                 faster as H = 640, W = 448, frame_width = 427).  Run-length / polygon masks only: polygon sides are
                                 clipped to frame_width (cv2.fillPoly on the unpadded frame), run lengths are column-major and need
                                 nothing, the fused filter takes its right border from frame_width; K and the pixel coordinates are
                                 those of the unpadded frame.  With u8 planes the caller pads the planes with zeros and leaves 0. */
} la3d_fit_args;
int la3d_fit_instances_ex(const la3d_fit_args* args);

/* create_boolean_mask_from_polygon for a batch: polygon parts -> u8 planes mask_out dev [B][H*W] (0/1). */
int la3d_poly_decode(const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings, int B, int H, int W,
                     uint8_t* mask_out, void* stream);

/* The four filter quantities of la3d_mask_stats for polygon annotations, rasterised in LDS (no plane is written):
 * replaces create_boolean_mask_from_polygon + get_maximum_height + analyze_mask (src/util.py:291-335, :371-375). */
int la3d_mask_stats_poly(const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings, int B, int H, int W,
                         int boundary, int32_t* stats, void* stream);

/* HOST helper: COCO compressed RLE string (pycocotools rleFrString) -> run lengths.  Returns the number of
 * counts written, or -1 (malformed string / cap too small). */
int la3d_rle_from_string_host(const char* s, int64_t len, int32_t* counts, int cap);

/* ---- box consumers: the step immediately downstream of the path ----------------------------
 * Reference src/tools/combine_results.py: the 8 corners of each record are projected with the image's K
 * (project_to_2d, :105-108), bbox2D_proj = [min_x, min_y, max_x, max_y] and bbox2D_trunc = its clamp to
 * [0,width] x [0,height] (:238-252).  records dev f64 [B][39] (as written by the fit calls), K dev f64 (9 per
 * image, k_stride 0 = shared), image_index dev i32 [B] | NULL; out dev f64 [B][8] = proj(4), trunc(4); a box with a
 * NaN projection gives 8 NaNs. */
int la3d_project_boxes(const double* records, const double* K, int32_t k_stride, const int32_t* image_index, int B,
                       double width, double height, double* out, void* stream);

/* iou2D (:111-124) of every pair of xyxy boxes: a dev f64 [na][4], b dev f64 [nb][4] -> out dev f64 [na][nb]
 * (negated = the Hungarian cost matrix of :131-135). */
int la3d_iou_matrix(const double* boxes_a, int na, const double* boxes_b, int nb, double* out, void* stream);

/* ---- masked depth statistics ------------------------------------------------------------------
 * Reference src/util.py:476-486 (align_to_depth_match): overlap = mask & render_mask;
 * scale = np.median(depth_map[overlap] / depth_render[overlap])  (float32 arithmetic, float32 result).
 * num   dev f32 planes (plane of instance n = image_index[n] or n; stride in floats, 0 = one shared plane)
 * den   dev f32 [B][H*W];  mask_a dev u8 [B][H*W];  mask_b dev u8 [B][H*W] | NULL (non-zero = True)
 * median dev f32 [B] (NaN when the overlap is empty or a ratio is NaN, as np.median);  count dev i32 [B].
 * H*W <= 819200 (the overlap bit image, the chunk list and the key buffer share one CU's LDS); larger frames: LA3D_ERR_UNSUPPORTED. */
int la3d_masked_ratio_median(const float* num, int64_t num_plane_stride, const int32_t* image_index, const float* den,
                             const uint8_t* mask_a, const uint8_t* mask_b, int B, int H, int W, float* median,
                             int32_t* count, void* stream);

/* align_depth (reference src/batch_scripts/depth.py:52-92), the data-parallel parts around its scikit-learn RANSAC fit:
 * la3d_align_select compacts, in row-major order, the pixels with ~isinf(relative) & (metric < max_valid_depth) [& mask]
 * into relative_out / metric_out (dev f32, capacity n) — what the reference feeds regressor.fit (:69-78) — and writes
 * their number to count (dev i64).  workspace: dev scratch of la3d_align_workspace_bytes(n) bytes.
 * la3d_align_apply writes depth = full(fill); depth[sel] = relative[sel] * coef + intercept in float32 with
 * sel = mask when given, else ~isinf(relative) (:82-90; fill is 10000.0 there). */
size_t la3d_align_workspace_bytes(int64_t n);
int la3d_align_select(const float* relative, const float* metric, const uint8_t* mask, int64_t n, float max_valid_depth,
                      float* relative_out, float* metric_out, int64_t* count, void* workspace, void* stream);
int la3d_align_apply(const float* relative, const uint8_t* mask, int64_t n, float coef, float intercept, float fill,
                     float* out, void* stream);
/* la3d_align_select for P frames of n pixels each in ONE call (three launches for the whole batch instead of three per frame plus an
 * 8-byte read-back each; the reference loops over images, src/batch_scripts/depth.py:138-160): relative / metric dev f32 [P][n],
 * mask dev u8 [P][n] | NULL; relative_out / metric_out dev f32 [P][n] (frame p's selection at [p][0 .. counts[p]), row-major order
 * kept); counts dev i64 [P] - left on the device; workspace: P * la3d_align_workspace_bytes(n) bytes.  P <= 65535. */
int la3d_align_select_batch(const float* relative, const float* metric, const uint8_t* mask, int P, int64_t n,
                            float max_valid_depth, float* relative_out, float* metric_out, int64_t* counts, void* workspace,
                            void* stream);

/* ---- sparse unprojection at match points -------------------------------------------------------
 * Reference src/matching/matcher.py:70-91: depth dev f32 [H][W] looked up at (int(v), int(u)) of each match
 * uv dev f64 [N][2]; matches whose depth is -1 (or that fall outside the frame) get valid = 0 and NaNs;
 * p = ((u'-cx) d/fx, (v'-cy) d/fy, d) with u' = flip-u, v' = flip-v when use_flip (the reference uses 512);
 * with R9 / T3 (HOST, both or neither): world = R (p - T)  (:88-89).  out dev f64 [N][3], valid dev i32 [N]. */
int la3d_unproject_matches(const float* depth, int H, int W, const double* uv, int N, double fx, double fy, double cx,
                           double cy, int use_flip, double flip, const double* R9, const double* T3, double* out,
                           int32_t* valid, void* stream);

/* Replaces estimate_bbox(in_pc, cat_name, ground_equ, method) for B point clouds at once —
 * reference src/util_3dbox.py:106-178 (caller :273-278, 500 mesh samples per object).
 * points   dev f64 [total][3];  offsets dev i64 [B+1] (cloud n = rows offsets[n]..offsets[n+1])
 * ground / sample_idx / out / status / aux as above (sample ranks index the cloud's rows).
 * method   LA3D_METHOD_PCA or LA3D_METHOD_CONVEX_HULL (_estimate_yaw_convex_hull, :189-224; at most 2048
 *          valid rows per cloud after sampling, else that box gets LA3D_BOX_UNSUPPORTED). */
int la3d_fit_points(const double* points, const int64_t* offsets, const double* ground,
                    const int32_t* sample_idx, int method, int B,
                    double* out, int32_t* status, double* aux, void* stream);

/* Host-pointer single calls (round 5): the reference's own calling pattern is one object / one image per call on NumPy arrays.
 * One C call = upload + kernel + download, synchronous, on a private stream of the calling thread; the staging memory (pinned and
 * device-mapped for the cloud, device scratch for the frame) belongs to the library, is per thread and per device, grows on demand
 * and is kept until la3d_host_release() / process exit.  ALL pointers are HOST pointers.
 *
 * la3d_estimate_bbox_host replaces estimate_bbox(in_pc, cat_name, ground_equ, method) for ONE cloud — reference
 * src/util_3dbox.py:106-178, call site :273-278.  points f64 [n][3] (the caller has already drawn its 500 rows when n > 500,
 * exactly where the reference draws them, :123-125); ground4 NULL or a NaN first entry = "ground_equ is None"; out39 / aux4 /
 * status as la3d_fit_points writes them (aux4 may be NULL).  PCA: the arithmetic of la3d_fit_points with
 * LA3D_HINT_SMALL_CLOUDS, bit for bit.
 *
 * la3d_unproject_host replaces depth_to_points(depth[None], K) for ONE frame — reference src/util.py:52-75, call site
 * src/batch_scripts/depth.py:154: depth f32 [H][W] -> out f64 / f32 [H][W][3], same arithmetic as la3d_unproject. */
int la3d_estimate_bbox_host(const double* points, int64_t n, const double* ground4, int method, double* out39, double* aux4,
                            int32_t* status);
int la3d_unproject_host(const float* depth, const double* K9, const double* Rt12, int H, int W, void* out, int out_is_f64);
void la3d_host_release(void);   /* frees the calling thread's staging memory and stream (optional) */

/* The reference's per-IMAGE pattern as one foreign call (round 5): the annotations of an image as run lengths or polygon parts ->
 * decode + the reference's keep rule + fit (la3d_fit_instances_ex with the filter fields) -> records.  `args` is a la3d_fit_args in which
 * `depth` is a DEVICE pointer (the image's depth plane(s), resident) and EVERY OTHER pointer is a HOST pointer: rle_counts / rle_offsets
 * or poly_xy / ring_offsets / inst_rings, K, image_index, ground, area_hint in; out / status / aux / stats out.  mask, sample_idx, proj
 * and workspace must be NULL / are ignored (the library uses the calling thread's staging memory).
 * `stream` = the stream the depth plane(s) were PRODUCED on (NULL = the legacy default stream): the call's upload, fit and completion
 * flag are enqueued on THAT stream, behind everything it holds, so a depth map a model / an upload / la3d_pad_rows has only enqueued
 * there is complete before the fit reads it.  Depth produced on any other stream must be complete before the call.  The depth must
 * live on the CURRENT device of the calling thread (hipSetDevice before the call).
 * Synchronous.  Replaces read_bounding_boxes_segmentations + the per-object fit for one image: src/util.py:336-383, util_3dbox.py:250-281. */
int la3d_fit_annotations_host(const la3d_fit_args* args);

/* The reference's per-scene box file from packed records, on the HOST (round 5; labelany3d_amd/csrc/la3d_json.cpp): the text
 * json.dump([{"obj_id", "category_name", "center_cam", "R_cam", "dimensions", "bbox3D_cam"}, ...], f) writes - reference
 * src/util_3dbox.py:283-292 - byte for byte (floats as float.__repr__ prints them), for S scenes in one call, without a Python
 * object per record.  records host f64 [*][39]; scene s owns entries [scene_off[s], scene_off[s+1]) of rows (record row) / obj_ids /
 * name_ids (index into names_json: UTF-8, already JSON-escaped and quoted); out: host buffer of `cap` bytes
 * (la3d_3dbbox_json_bound(entries, total name bytes, S)); text_off [S+1]: scene s's text = out[text_off[s] .. text_off[s+1]).
 * Returns the bytes written, -1 if cap is too small or an argument is missing. */
int64_t la3d_3dbbox_json_bound(int64_t n, int64_t name_bytes, int64_t S);
int64_t la3d_format_3dbbox_json(const double* records, const int64_t* rows, const int32_t* obj_ids, const int32_t* name_ids,
                                const int64_t* scene_off, int32_t S, const char* const* names_json, char* out, int64_t cap,
                                int64_t* text_off);

/* Host-side staging helper (no device work): n pageable source planes of bytes_each bytes -> consecutive slots of dst (a pinned
 * buffer), copied by `threads` native threads in one call (labelany3d_amd/fit_scenes.py: the depth_map.npy planes of a batch,
 * reference src/batch_scripts/whole.py:63-67).  0 on success, -1 on a bad argument. */
int la3d_gather_planes_host(const void* const* src, int64_t n, int64_t bytes_each, void* dst, int threads);

/* Host-side helper exported for tests: float64 -> float16 (round-to-nearest-even, as NumPy's
 * astype(float16), reference src/util_3dbox.py:165) -> float64, the same routine the kernels use. */
double la3d_f16_round_host(double x);

#ifdef __cplusplus
}
#endif
#endif /* LA3D_H */
