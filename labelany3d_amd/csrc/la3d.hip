// la3d.hip — the C-ABI of la3d_fit_instances* and its dispatcher (MI355X, gfx950 / CDNA4): the LabelAny3D geometric hot path -
// pinhole back-projection of masked depth pixels -> per-object moments -> closed-form PCA yaw -> extents along the principal
// axes -> 39-double box record.
//
// Reference semantics (behaviour only; nothing is copied):
//   depth_to_points   /root/reference/src/util.py:52-75
//   estimate_bbox     /root/reference/src/util_3dbox.py:106-178   (+ helpers :20-103, PCA yaw :181-186)
//
// Memory-bound integer/byte + fp64 reduction work: no MFMA.  Layout, kernel design and measurements: DESIGN.md.  Round 6 split the
// former 3 000-line file by engine, one translation unit each (compiled side by side):
//   la3d_instance.hip   one workgroup per instance: every call the others do not take (fit_instances_kernel)
//   la3d_rows.hip       up to sixteen workgroups per instance, one per band of rows: un-grounded u8 batches up to 160 instances
//   la3d_band.hip       two / four / eight workgroups per instance that meet through the workspace: grounded u8 batches of 1..160
//   la3d_split.hip      scan -> plan -> walk -> axis -> walk -> final over tile ranges: grounded run-length / polygon batches up to 160
//   la3d_walks.hpp      the walks over an instance's pixels (generic, tiled two-pass, separable single pass)
//   la3d_stages.hpp     moments -> axis, extents -> record, the in-kernel launch order, the culling plan, workgroup hand-off primitives
//   la3d_points.hip     explicit point clouds (la3d_fit_points) and the host-pointer single calls
//   la3d_masks.hip      whole-frame depth_to_points, mask decode / statistics;  la3d_consumers.hip  box consumers, depth statistics, matcher geometry
//   la3d_json.cpp       the host-side writer of 3dbbox.json + the build identity
// This file: the process defaults read once from the environment (config), workspace sizing, argument checks, and which engine
// fits a call (fit_dispatch; profiles/r06/r06_engines_by_batch.txt has the row in which each engine is the fastest).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "la3d_device.hpp"
#include "la3d_poly.hpp"
#include "la3d_engines.hpp"

namespace la3d {
thread_local char g_err[256] = "";
// the ONE place that reads the environment: a function-local static, initialised once (thread-safe since C++11)
const Config& config() {
  static const Config c = [] {
    Config k;
    const char* e = getenv("LA3D_ENGINE");
    k.engine = (e && !strcmp(e, "instance")) ? LA3D_ENGINE_INSTANCE : (e && !strcmp(e, "split")) ? LA3D_ENGINE_SPLIT
             : (e && !strcmp(e, "band")) ? LA3D_ENGINE_BAND : (e && !strcmp(e, "rows")) ? LA3D_ENGINE_ROWS
             : (e && !strcmp(e, "rows2")) ? LA3D_ENGINE_ROWS2 : LA3D_ENGINE_DEFAULT;
    e = getenv("LA3D_BANDS");
    k.bands = (e && (atoi(e) == 8 || atoi(e) == 4 || atoi(e) == 2)) ? atoi(e) : 0;
    e = getenv("LA3D_BAND_DEFAULT");      // 0: the band engine only when asked for (LA3D_ENGINE=band / opt_engine)
    k.band_default = !(e && e[0] == '0');
    e = getenv("LA3D_BAND_MAXB");
    k.band_maxb = (e && atoi(e) > 0) ? atoi(e) : 160;
    e = getenv("LA3D_ROWS_MAXB");
    k.rows_maxb = e ? atoi(e) : 160;   // largest batch the row engine takes by default (0: never); above, one workgroup per instance is as fast
                                       // (round 6, us per call, instance | rows: B = 128: 37.1 | 30.3; 192: 39.8 | 41.1 - profiles/r06/r06_rows_engine.txt)
    e = getenv("LA3D_ROWS_FUSED");        // 0: the two-launch form of the row engine (fit_rows_kernel + merge_rows_kernel)
    k.rows_fused = !(e && e[0] == '0');
    e = getenv("LA3D_ROWS_WGS");          // workgroups the row engine spreads a batch over, at most
    k.rows_wgs = (e && atoi(e) > 0) ? atoi(e) : 640;
    e = getenv("LA3D_BALANCE");
    k.balance = !(e && e[0] == '0');
    e = getenv("LA3D_BALANCE_ROUNDS");
    k.balance_rounds = (e && atoi(e) > 0) ? atoi(e) : 3;
    e = getenv("LA3D_BUILD");             // plain | nocull -> LA3D_BUILD_PLAIN / LA3D_BUILD_NOCULL for every call (LA3D_RETAIN=0 / 1: the old spelling)
    k.build = (e && !strcmp(e, "plain")) ? LA3D_BUILD_PLAIN : (e && !strcmp(e, "nocull")) ? LA3D_BUILD_NOCULL : LA3D_BUILD_DEFAULT;
    e = getenv("LA3D_RETAIN");
    if (e && k.build == LA3D_BUILD_DEFAULT) k.build = atoi(e) > 0 ? LA3D_BUILD_NOCULL : LA3D_BUILD_PLAIN;
    e = getenv("LA3D_CULL_MIN");          // pass-B culling threshold (active tiles) for every input; unset: 224, u8 planes LA3D_CULL_MIN_U8
    k.cull_min = e ? atoi(e) : 0;
    e = getenv("LA3D_CULL_MIN_U8");
    k.cull_min_u8 = (e && atoi(e) > 0) ? atoi(e) : 128;
    e = getenv("LA3D_ORDER_SELF");        // 0: helper kernel in front of every ordered launch; 2: test mode of the fallback
    k.order_self = e ? atoi(e) : 1;
    e = getenv("LA3D_STAGGER_US");
    k.stagger_us = e ? atof(e) : -1.0;
    e = getenv("LA3D_STAGGER_NOMASK_US");   // run-length / polygon input: stagger period of the resident groups (0 / unset: none)
    k.stagger_nomask_us = e ? atof(e) : 0.0;
    e = getenv("LA3D_SPLIT_GRID");
    k.split_grid = (e && atoi(e) > 0) ? atoi(e) : 0;
    e = getenv("LA3D_SPLIT_SUB");
    k.split_sub = (e && atoi(e) > 0) ? atoi(e) : 0;
    e = getenv("LA3D_BAND_TEST");
    k.band_test = e ? atoi(e) : 0;
    e = getenv("LA3D_SEP");               // 0: no separable single pass (the two-pass plain build everywhere)
    k.sep = !(e && e[0] == '0');
    return k;
  }();
  return c;
}
}

using namespace la3d;


// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int la3d_version(void) { return LA3D_ABI_VERSION; }

const char* la3d_last_error(void) { return g_err; }

double la3d_f16_round_host(double x) { return f16_round(x); }


// Workspace layout (one per concurrently running call; contents need not be initialised or preserved):
//   instance engine: [B] u32 sort keys of the size-balanced launch order (4*B bytes)
//   band engine:     [B] u32 sort keys | [B][4] i32 arrival counters | [B][88] f64 exchange records (band_workspace_bytes)
//   split engine:    [B][GEO_D] f64 geometry, then bit images, tile lists and partial-sum slots (split_workspace_bytes)
//   row engine:      [B][nb][10] f64 partial records | [B][nb][2 W] u32 per-column depth ranges (rows_workspace_bytes)
size_t la3d_workspace_bytes(int B, int H, int W) {
  if (B <= 0) return 0;
  const size_t inst = (size_t)B * GEO_D * sizeof(double);  // kept as the minimum (older callers size by it)
  const size_t split = split_workspace_bytes(B, H, W);     // split engine: + bit image, tile lists, partial slots
  const size_t band = band_frame_ok(H, W, 2) ? band_workspace_bytes(B) : 0;   // band engine: keys, arrival counters, exchange records
  const size_t rows = rows_workspace_bytes(B, H, W);         // row engine: partial records and per-column ranges of every band
  size_t m = split > inst ? split : inst;
  if (band > m) m = band;
  return rows > m ? rows : m;
}

struct PolyArgs { const int32_t* xy; const int64_t* ring_off; const int64_t* inst_rings; };
struct FilterArgs { int boundary, min_area, max_edge; int32_t* stats; };
struct ProjArgs { double* out; double width, height; };
struct CallOpts { int engine, order, build, frame_w; };

static int fit_dispatch(const float* depth, int64_t depth_plane_stride, const int32_t* image_index, const uint8_t* mask,
                        const int32_t* rle_counts, const int64_t* rle_offsets, const double* K, int32_t k_stride,
                        const double* ground, const int32_t* sample_idx, int B, int H, int W, double* out,
                        int32_t* status, double* aux, void* workspace, void* stream, const char* who,
                        const PolyArgs* poly = nullptr, const FilterArgs* filter = nullptr, const ProjArgs* proj = nullptr,
                        const int32_t* area_hint = nullptr, const CallOpts* opts = nullptr) {
  const bool rle = rle_counts != nullptr || poly != nullptr;   // "no u8 plane": the mask is decoded into the LDS bit image
  if (!depth || (!mask && !rle) || (rle_counts && !rle_offsets) || (poly && (!poly->ring_off || !poly->inst_rings)) || !K ||
      !out || !status || B < 0 || H <= 0 || W <= 0 ||
      depth_plane_stride < 0 || (k_stride != 0 && k_stride < 9) || (long long)H * W > (1LL << 28)) {
    snprintf(g_err, sizeof(g_err), "%s: bad argument", who);
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
    snprintf(g_err, sizeof(g_err), "%s: workspace of la3d_workspace_bytes() bytes (8-aligned) required", who);
    return LA3D_ERR_ARG;
  }
  FitParams p;
  p.geo = static_cast<double*>(workspace);
  p.depth = depth; p.depth_plane_stride = depth_plane_stride; p.image_index = image_index;
  p.mask = mask; p.K = K; p.k_stride = k_stride; p.ground = ground; p.sample_idx = sample_idx;
  p.rle_counts = rle_counts; p.rle_offsets = reinterpret_cast<const long long*>(rle_offsets);
  p.poly_xy = poly ? poly->xy : nullptr;
  p.poly_ring_off = poly ? reinterpret_cast<const long long*>(poly->ring_off) : nullptr;
  p.poly_inst_rings = poly ? reinterpret_cast<const long long*>(poly->inst_rings) : nullptr;
  p.B = B; p.H = H; p.W = W; p.HW = H * W;
  p.nwords = (p.HW + 31) / 32;
  p.rows_aligned = (W % 4 == 0);
  p.rcpW = 1.0f / (float)W;
  p.out = out; p.status = status; p.aux = aux;
  p.ntx = p.nty = p.tiles_per_wave = p.list_cap = 0;
  p.rcp_ntx = 1.0f;
  p.order_nch = 0; p.order_keys = nullptr; p.order_resident = 0; p.order_shift = 0;
  p.order_self = 0; p.order_flags = nullptr; p.order_nonce = 0; p.est_step = 1;
  p.stagger_ticks = 0;
  p.band_test = config().band_test;
  // build of the call: the default (separable single pass where it applies), PLAIN = the two-pass form for every camera, NOCULL = the
  // two-pass form that also walks EVERY active tile in pass B (no culling plan): the reference build the culling tests compare with
  const int build = (opts && opts->build != LA3D_BUILD_DEFAULT) ? opts->build : config().build;
  p.sep_off = (config().sep == 0 || build != LA3D_BUILD_DEFAULT) ? 1 : 0;
  p.band_trows = 0; p.band_arrive = nullptr; p.band_tag = 0; p.band_xch = nullptr;
  p.cull_min = build == LA3D_BUILD_NOCULL ? 0x7fffffff : config().cull_min > 0 ? config().cull_min : (mask != nullptr ? config().cull_min_u8 : CULL_MIN);
  p.filter_boundary = -1; p.filter_min_area = 0; p.filter_max_edge = 0; p.filter_stats = nullptr;
  p.proj = proj ? proj->out : nullptr; p.proj_w = proj ? proj->width : 0; p.proj_h = proj ? proj->height : 0;
  p.area_hint = area_hint;
  p.opt_engine = opts ? opts->engine : 0; p.opt_order = opts ? opts->order : 0; p.opt_build = opts ? opts->build : 0;
  p.frame_w = W;
  if (opts && opts->frame_w != 0 && opts->frame_w != W) {
    // rows padded on the right (la3d_fit_args::frame_width): run-length / polygon masks, word-aligned rows
    if (opts->frame_w < 0 || opts->frame_w > W || mask != nullptr || W % 32 != 0) {
      snprintf(g_err, sizeof(g_err), "%s: frame_width must be 0 or in (0, W], with run-length / polygon masks and W %% 32 == 0", who);
      return LA3D_ERR_ARG;
    }
    p.frame_w = opts->frame_w;
  }
  if (filter) {
    if (!rle || filter->boundary < 0) {
      snprintf(g_err, sizeof(g_err), "%s: the fused filter needs run-length or polygon masks and boundary >= 0", who);
      return LA3D_ERR_ARG;
    }
    p.filter_boundary = filter->boundary; p.filter_min_area = filter->min_area; p.filter_max_edge = filter->max_edge;
    p.filter_stats = filter->stats;
  }
  const int bit_bytes = ((((p.HW + 15) / 16 + 1) / 2) * 4 + 15) & ~15;  // u16 per 16 px, padded to u32, 16-aligned
  const bool ldsmask = bit_bytes <= MAX_MASK_LDS;
  p.mask_lds_bytes = ldsmask ? bit_bytes : 0;
  if (rle && !ldsmask) {
    snprintf(g_err, sizeof(g_err), "%s: run-length / polygon masks need the bit image in LDS (H*W <= 1048576)", who);
    return LA3D_ERR_UNSUPPORTED;
  }
  // 16-byte vector path: every plane base 16-aligned (the u8 mask only when it is read at all)
  const bool vec = (p.HW % 16 == 0) && (rle || (reinterpret_cast<uintptr_t>(mask) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(depth) & 15) == 0) && (depth_plane_stride % 4 == 0);
  const bool sample = sample_idx != nullptr;
  size_t lds = (size_t)p.mask_lds_bytes + sizeof(Shared);
  // polygons: the side stage sits behind Shared, where the tile list / rank prefix go later (disjoint in time)
  const size_t poly_stage = poly ? (size_t)POLY_STAGE_BYTES : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Round 5: a call WITHOUT a ground array on a frame the one-pass tile list covers takes the instance engine at EVERY batch size - its
  // separable single pass (no pass B, no cull plan: a chain of three short phases per workgroup) is faster than the chain of six
  // launches of the split engine and than the band engine's exchange from B = 1 on, for all three mask formats
  // (profiles/r05/r05_small_batches.txt: B = 1 / 16 / 64 / 256, u8 planes: 27.7 / 32.1 / 36.8 / 45.1 us vs 31.7 / 34.8 / 37.9 / 59.8;
  // run lengths 31.2 / 35.6 / 36.4 / 40.0 vs 34.3 / 37.9 / 43.3 / 57.4).  A skewed K (not separable) still takes this route - the
  // kernel then runs its two passes -; a call WITH a ground array keeps the old choice below (two passes either way).
  // Small batches of u8 planes (up to 192 instances by default) go one step further: the same single pass split over up to sixteen
  // workgroups per instance, one per band of rows, and a short merge launch (row engine: B = 1 / 16 / 64 / 128 18.2 / 19.3 / 24.0 /
  // 30.0 us per call, profiles/r05/r05_rows_engine.txt).
  const int eng = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;
  const bool single_pass_call = ground == nullptr && !sample && !p.sep_off && ldsmask && vec && W % 32 == 0 && W / 32 <= 255 &&
                                (H + 7) / 8 <= 255 && ((W / 32) * ((H + 7) / 8) + NWAVE - 1) / NWAVE <= 256 &&
                                (eng == LA3D_ENGINE_DEFAULT || eng == LA3D_ENGINE_ROWS || eng == LA3D_ENGINE_ROWS2);   // (rows pinned but not applicable: as by default)
  {
    int rc = LA3D_SUCCESS;
    if (rows_fit_if_eligible(p, vec, sample, s, workspace, &rc)) return rc;
  }
  if (!single_pass_call && band_eligible(p, vec, sample)) return band_fit(p, s, workspace);   // grounded u8 planes, 1 <= B <= 160 (or pinned): two / four / eight workgroups per instance, ONE launch
  if (!single_pass_call && !sample && p.frame_w == W && split_eligible(p, vec, ldsmask)) {   // (the split engine's decoders know no padded rows)
    const int rc = split_fit(p, workspace, s);   // (the split engine's final kernel does not project: one small follow-up launch)
    if (rc != LA3D_SUCCESS || !p.proj) return rc;
    return la3d_project_boxes(out, K, k_stride, image_index, B, p.proj_w, p.proj_h, p.proj, stream);   // (la3d_consumers.hip)
  }
  return instance_fit(p, vec, ldsmask, sample, lds, poly_stage, s, workspace, who);
}

int la3d_fit_instances(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                       const uint8_t* mask, const double* K, int32_t k_stride, const double* ground,
                       const int32_t* sample_idx, int B, int H, int W, double* out, int32_t* status, double* aux,
                       void* workspace, void* stream) {
  return fit_dispatch(depth, depth_plane_stride, image_index, mask, nullptr, nullptr, K, k_stride, ground, sample_idx, B, H,
                      W, out, status, aux, workspace, stream, "la3d_fit_instances");
}

int la3d_fit_instances_rle(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                           const int32_t* rle_counts, const int64_t* rle_offsets, const double* K, int32_t k_stride,
                           const double* ground, const int32_t* sample_idx, int B, int H, int W, double* out,
                           int32_t* status, double* aux, void* workspace, void* stream) {
  if (!rle_counts && B > 0) {
    set_err("la3d_fit_instances_rle: bad argument");
    return LA3D_ERR_ARG;
  }
  return fit_dispatch(depth, depth_plane_stride, image_index, nullptr, rle_counts, rle_offsets, K, k_stride, ground,
                      sample_idx, B, H, W, out, status, aux, workspace, stream, "la3d_fit_instances_rle");
}

int la3d_fit_instances_poly(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                            const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings, const double* K,
                            int32_t k_stride, const double* ground, const int32_t* sample_idx, int B, int H, int W, double* out,
                            int32_t* status, double* aux, void* workspace, void* stream) {
  if ((!poly_xy || !ring_offsets || !inst_rings) && B > 0) {
    set_err("la3d_fit_instances_poly: bad argument");
    return LA3D_ERR_ARG;
  }
  const PolyArgs pa{poly_xy, ring_offsets, inst_rings};
  return fit_dispatch(depth, depth_plane_stride, image_index, nullptr, nullptr, nullptr, K, k_stride, ground, sample_idx, B, H, W,
                      out, status, aux, workspace, stream, "la3d_fit_instances_poly", &pa);
}

int la3d_fit_instances_rle_filtered(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                                    const int32_t* rle_counts, const int64_t* rle_offsets, const double* K, int32_t k_stride,
                                    const double* ground, const int32_t* sample_idx, int B, int H, int W, int boundary,
                                    int min_area, int max_edge, double* out, int32_t* status, double* aux, int32_t* stats,
                                    void* workspace, void* stream) {
  if (!rle_counts && B > 0) {
    set_err("la3d_fit_instances_rle_filtered: bad argument");
    return LA3D_ERR_ARG;
  }
  const FilterArgs fa{boundary, min_area, max_edge, stats};
  return fit_dispatch(depth, depth_plane_stride, image_index, nullptr, rle_counts, rle_offsets, K, k_stride, ground, sample_idx, B,
                      H, W, out, status, aux, workspace, stream, "la3d_fit_instances_rle_filtered", nullptr, &fa);
}

int la3d_fit_instances_poly_filtered(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                                     const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings,
                                     const double* K, int32_t k_stride, const double* ground, const int32_t* sample_idx, int B,
                                     int H, int W, int boundary, int min_area, int max_edge, double* out, int32_t* status,
                                     double* aux, int32_t* stats, void* workspace, void* stream) {
  if ((!poly_xy || !ring_offsets || !inst_rings) && B > 0) {
    set_err("la3d_fit_instances_poly_filtered: bad argument");
    return LA3D_ERR_ARG;
  }
  const PolyArgs pa{poly_xy, ring_offsets, inst_rings};
  const FilterArgs fa{boundary, min_area, max_edge, stats};
  return fit_dispatch(depth, depth_plane_stride, image_index, nullptr, nullptr, nullptr, K, k_stride, ground, sample_idx, B, H, W,
                      out, status, aux, workspace, stream, "la3d_fit_instances_poly_filtered", &pa, &fa);
}

int la3d_fit_instances_ex(const la3d_fit_args* args) {
  constexpr int32_t V1_SIZE = (int32_t)offsetof(la3d_fit_args, area_hint);   // the block as first published: every field up to `stream`
  if (!args || args->struct_size < V1_SIZE) {   // (a longer block from a newer caller is fine, fields it lacks are taken as zero)
    set_err("la3d_fit_instances_ex: bad struct_size");
    return LA3D_ERR_ARG;
  }
  la3d_fit_args a;
  memset(&a, 0, sizeof(a));
  memcpy(&a, args, (size_t)args->struct_size < sizeof(a) ? (size_t)args->struct_size : sizeof(a));
  const int kinds = (a.mask ? 1 : 0) + (a.rle_counts ? 1 : 0) + (a.poly_xy ? 1 : 0);
  if (kinds != 1 && a.B > 0) {
    set_err("la3d_fit_instances_ex: give exactly one of mask / rle_counts / poly_xy");
    return LA3D_ERR_ARG;
  }
  if (a.poly_xy && (!a.ring_offsets || !a.inst_rings)) {
    set_err("la3d_fit_instances_ex: polygon masks need ring_offsets and inst_rings");
    return LA3D_ERR_ARG;
  }
  if (a.proj && !(a.image_width > 0 && a.image_height > 0)) {
    set_err("la3d_fit_instances_ex: proj needs image_width / image_height > 0");
    return LA3D_ERR_ARG;
  }
  const PolyArgs pa{a.poly_xy, a.ring_offsets, a.inst_rings};
  // the fused filter is on when filter_boundary >= 0 AND filter_max_edge > 0: a zero-initialised block (the natural C idiom, and
  // what "missing fields are zero" gives) means NO filter - max_edge == 0 would reject every instance (edge < 0 never holds)
  const bool filter_on = a.filter_boundary >= 0 && a.filter_max_edge > 0;
  const FilterArgs fa{a.filter_boundary, a.filter_min_area, a.filter_max_edge, a.stats};
  const ProjArgs pr{a.proj, a.image_width, a.image_height};
  if (a.opt_engine < 0 || a.opt_engine > LA3D_ENGINE_ROWS2 || a.opt_launch_order < 0 || a.opt_launch_order > LA3D_ORDER_ON ||
      a.opt_build < 0 || a.opt_build > LA3D_BUILD_NOCULL) {
    set_err("la3d_fit_instances_ex: bad opt_engine / opt_launch_order / opt_build");
    return LA3D_ERR_ARG;
  }
  const CallOpts co{a.opt_engine, a.opt_launch_order, a.opt_build, a.frame_width};
  return fit_dispatch(a.depth, a.depth_plane_stride, a.image_index, a.mask, a.rle_counts, a.rle_offsets, a.K, a.k_stride, a.ground,
                      a.sample_idx, a.B, a.H, a.W, a.out, a.status, a.aux, a.workspace, a.stream, "la3d_fit_instances_ex",
                      a.poly_xy ? &pa : nullptr, filter_on ? &fa : nullptr, a.proj ? &pr : nullptr, a.area_hint, &co);
}

}  // extern "C"
