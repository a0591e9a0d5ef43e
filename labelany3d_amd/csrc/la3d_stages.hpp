// la3d_stages.hpp - the workgroup stages shared by the fit engines (moments -> axis, extents -> record), the size-balanced launch order decided
// inside the kernel, the pass-B culling plan, and the primitives workgroups of ONE launch use to hand data to each other.
#pragma once
#include "la3d_device.hpp"
#include "la3d_walks.hpp"

using namespace la3d;

namespace {
// ------------------------------------------------------------------------------------------
// workgroup stages shared by the fit kernels (every thread of the workgroup must call them)
// ------------------------------------------------------------------------------------------
// moments of all waves -> wave 0 (fixed xor tree: bit-reproducible) -> status, yaw axis.
// On return sh->st / sh->cyaw / sh->syaw are valid for every thread.  The aux record (with its atan2) is
// written at the end of the kernel (stage_status_aux), off everybody's critical path.
// allow_redo: the sums come from the optimistic pass (quad_math<0, false>); if they are not finite - or ill-conditioned
// (axis_from_sums) - set sh->redo (1 / 2) and return without deciding anything: the caller re-runs the checked pass (1) or the
// moments about the pivot left in LDS (2: pivot_pass) and calls again with allow_redo = false.
// The pivot of an ill-conditioned instance's second moments pass lives in two doubles of `part` that no stage uses between the axis
// stage and the checked pass A (the moments stage writes part[w][0..4], the extents stages part[w][0..5] later on).
__device__ inline void set_pivot(Shared* sh, double x, double z) { sh->part[NWAVE - 1][5] = x; sh->part[NWAVE - 1][6] = z; }
__device__ inline const double* pivot_ptr(const Shared* sh) { return &sh->part[NWAVE - 1][5]; }
__device__ inline void get_pivot(const Shared* sh, double* piv) { piv[0] = uniform_f64(sh->part[NWAVE - 1][5]); piv[1] = uniform_f64(sh->part[NWAVE - 1][6]); }

__device__ inline void stage_moments_to_axis(Shared* sh, const FitParams& p, int inst, const double* acc, int cnt,
                                             int nmask, int tid, int wave, int lane, bool allow_redo = false) {
  {
    const double r0 = wave_sum(acc[0]), r1 = wave_sum(acc[1]), r2 = wave_sum(acc[2]), r3 = wave_sum(acc[3]),
                 r4 = wave_sum(acc[4]);
    const int rc = wave_sum_i(cnt), rn = wave_sum_i(nmask);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4;
      sh->cnt[wave] = rc;
      sh->nmask[wave] = rn;
    }
  }
  __syncthreads();
  LA3D_SUBSTAMP(sh, 9);
  if (wave == 0) {
    // the NWAVE partials: one per lane, then a fixed xor tree over those lanes (bit-reproducible)
    double s[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = lane < NWAVE ? sh->part[lane][k] : 0.0;
    int n = lane < NWAVE ? sh->cnt[lane] : 0, nm = lane < NWAVE ? sh->nmask[lane] : 0;
    static_assert(NWAVE == 8, "the tree below combines lanes 0..7");
#pragma unroll
    for (int k = 0; k < 5; ++k) {   // xor 1, xor 2, then the other quad of the first eight lanes: DPP moves, no LDS round trips
      s[k] += dpp_f64<DPP_XOR1>(s[k]); s[k] += dpp_f64<DPP_XOR2>(s[k]); s[k] += dpp_f64<DPP_HALF_MIRROR>(s[k]);
    }
    n += dpp_i32<DPP_XOR1>(n); n += dpp_i32<DPP_XOR2>(n); n += dpp_i32<DPP_HALF_MIRROR>(n);
    nm += dpp_i32<DPP_XOR1>(nm); nm += dpp_i32<DPP_XOR2>(nm); nm += dpp_i32<DPP_HALF_MIRROR>(nm);
    if (lane == 0) {
    double gap = NAN;
    int st = LA3D_BOX_OK;
    if (sh->bad_ground) st = LA3D_BOX_BAD_GROUND;
    else if (n == 0) st = LA3D_BOX_EMPTY;
    else if (n == 1) st = LA3D_BOX_TOO_FEW;
    const double chk = (s[0] + s[1]) + (s[2] + s[3]) + s[4];
    const bool nonfinite = !(fabs(chk) <= 1.79769313486231570815e308);
    double cy = NAN, sy = NAN;
    bool ill = false;
    if (st == LA3D_BOX_OK) ill = axis_from_sums((double)n, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap);
    // redo: non-finite sums -> the checked pass; ill-conditioned sums (axis_from_sums) -> the checked pass about the pivot left here
    sh->redo = (allow_redo && !sh->bad_ground) ? (nonfinite ? 1 : (ill ? 2 : 0)) : 0;
    set_pivot(sh, (ill && !nonfinite) ? s[0] / (double)n : 0.0, (ill && !nonfinite) ? s[1] / (double)n : 0.0);
    if (ill && !allow_redo) gap = 0.0;   // unresolved about its own mean, or a path without a second pass: the "don't care" value
    sh->cyaw = cy; sh->syaw = sy;
    sh->qhead = 0u;   // pass B's work queue starts at the first tile
    sh->st = st;
    sh->n_valid = n;
    sh->gap = gap;
    sh->nm = nm;
    }
  }
  LA3D_SUBSTAMP(sh, 10);
  __syncthreads();
  if (sh->redo) return;  // uniform
  if (tid == 0 && sh->st != LA3D_BOX_OK) {  // rejected instance: the workgroup returns right after this call
    if (p.aux) {
      double* a = p.aux + (long long)inst * LA3D_AUX;
      a[0] = atan2(sh->syaw, sh->cyaw); a[1] = (double)sh->n_valid; a[2] = (double)sh->nm; a[3] = sh->gap;
    }
    p.status[inst] = sh->st;
    write_nan_box(p.out + (long long)inst * LA3D_REC);
    if (p.proj) { for (int j = 0; j < 8; ++j) p.proj[(long long)inst * 8 + j] = NAN; }
  }
}

// status and aux record of an accepted instance: written at the very end by lane 0 of wave 1, next to wave 0 writing the
// box - the atan2 of the reported yaw is the only trigonometry of the kernel and nobody waits for it
__device__ inline void stage_status_aux(const Shared* sh, const FitParams& p, int inst, int tid) {
  if (tid != 64) return;
  if (p.aux) {
    double* a = p.aux + (long long)inst * LA3D_AUX;
    a[0] = atan2(sh->syaw, sh->cyaw); a[1] = (double)sh->n_valid; a[2] = (double)sh->nm; a[3] = sh->gap;
  }
  p.status[inst] = LA3D_BOX_OK;
}

// extents (x,y,z : lo,hi) of all waves -> wave 0 -> the 39-double record, written lane-parallel
__device__ inline void stage_extents_to_box(Shared* sh, const FitParams& p, int inst, const double* ext, int tid,
                                            int wave, int lane) {
  {
    const double r0 = wave_min(ext[0]), r1 = wave_max(ext[1]), r2 = wave_min(ext[2]), r3 = wave_max(ext[3]),
                 r4 = wave_min(ext[4]), r5 = wave_max(ext[5]);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; pp[5] = r5;
    }
  }
  __syncthreads();
  LA3D_SUBSTAMP(sh, 11);
  if (wave == 0) {
    double lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = lane < NWAVE ? sh->part[lane][2 * k] : INFINITY;
      hi[k] = lane < NWAVE ? sh->part[lane][2 * k + 1] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // only the first NWAVE (8) lanes hold data: xor 1, xor 2, other quad - DPP moves
      lo[k] = fmin(lo[k], dpp_f64<DPP_XOR1>(lo[k])); lo[k] = fmin(lo[k], dpp_f64<DPP_XOR2>(lo[k])); lo[k] = fmin(lo[k], dpp_f64<DPP_HALF_MIRROR>(lo[k]));
      hi[k] = fmax(hi[k], dpp_f64<DPP_XOR1>(hi[k])); hi[k] = fmax(hi[k], dpp_f64<DPP_XOR2>(hi[k])); hi[k] = fmax(hi[k], dpp_f64<DPP_HALF_MIRROR>(hi[k]));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = readlane_f64(lo[k], 0); hi[k] = readlane_f64(hi[k], 0); }   // write_box_wave wants them in every lane
    double Rg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rg[i] = sh->Rg[i];
    if (p.proj) {   // uniform: the 2-D boxes of the record in the same epilogue (la3d_fit_instances_ex)
      const int img = p.image_index ? p.image_index[inst] : inst;
      write_box_wave(p.out + (long long)inst * LA3D_REC, Rg, sh->cyaw, sh->syaw, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lane,
                     p.proj + (long long)inst * 8, p.K + (long long)img * p.k_stride, p.proj_w, p.proj_h);
    } else {
      write_box_wave(p.out + (long long)inst * LA3D_REC, Rg, sh->cyaw, sh->syaw, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lane);
    }
  }
  LA3D_SUBSTAMP(sh, 12);
}

// rows 0 and 2 of rotate_y(yaw) @ M (reference :154) as wave-uniform SGPR values; row 1 is M's row 1
__device__ inline void yaw_rows(const Shared* sh, const double* Mg, double* N0, double* N2) {
  const double cy = uniform_f64(sh->cyaw), sy = uniform_f64(sh->syaw);
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    N0[jj] = uniform_f64(cy * Mg[jj] + sy * Mg[6 + jj]);
    N2[jj] = uniform_f64(-sy * Mg[jj] + cy * Mg[6 + jj]);
  }
}

// ------------------------------------------------------------------------------------------
// size-balanced launch order, decided inside the fit kernel (round 3: the ranking kernel of rounds 1-2 is gone).
// Measured on MI355X (profiles/microbench/wg_census.hip, profiles/exp_chain.py): workgroup b of a fresh grid starts on
// CU b % 256, so with G workgroups resident per CU the instances of blocks {c, c+256, ..} share CU c for their whole
// life and the launch lasts as long as the most loaded CU (random sizes: ~2x the mean; 152 us unordered vs 113 us
// ordered on the same multiset).  Rank r of the descending size order -> group r/256; group 0 goes to CUs 0..255 in
// order, every later group in reverse (the CU with the largest instance gets the smallest member of every other group);
// ranks beyond the resident set follow in descending order (longest-first list scheduling of the dynamic remainder).
// The ranking is CHUNK-LOCAL: the batch is cut into nch = ceil(B/64) chunks of consecutive instances (the first B % nch
// one longer), "rank in chunk * nch + chunk" stands in for the global rank (a round-robin merge of the chunk orders: a
// bijection onto 0..B-1, and what the exact merge gives for equally distributed chunks; per-CU load max/mean 1.27 vs
// 1.17 for the exact ranking on the config-2 sizes).  So workgroup b inverts the map - block -> rank -> (chunk, rank in
// chunk) - loads the <= 64 keys of that chunk (L2-resident, written by size_estimate_kernel, or built from the caller's
// area_hint: then NO helper launch at all) and finds the instance with that rank by register broadcast on one wave.
// Measured (profiles/r03/r03_launch_order.txt): chunks of 64 / 128 / 256 -> 106.0 / 106.5 / 110.1 us per 1024-instance
// call against 107.3 with the ranking kernel: the selection sits on every workgroup's critical path, so the cheapest
// one wins although its balance is the coarsest.
// The order only steers speed: records do not depend on it (tests/test_gpu_parity.py::test_launch_order_is_invisible).
// ------------------------------------------------------------------------------------------
constexpr int ORDER_CHUNK = 64;   // instances ranked together: 64 keys per wave on ORDER_CHUNK / 64 waves
constexpr int KEY_IDX_BITS = 14;    // sort key = (area quantised to 18 bits) << 14 | (16383 - instance): unique, and a
                                    // plain unsigned compare orders by area descending, then index ascending
constexpr int ORDER_MAX_B = 1 << KEY_IDX_BITS;

__device__ inline unsigned make_order_key(int area, int shift, int inst) {
  unsigned q = (unsigned)(area < 0 ? 0 : area) >> shift;
  if (q > 0x3ffffu) q = 0x3ffffu;
  return (q << KEY_IDX_BITS) | (unsigned)((1 << KEY_IDX_BITS) - 1 - inst);
}

// the estimate of ONE instance by ONE wave (every lane returns the wave's sum): shoelace area of the polygon parts, the exact sum of
// the ones-runs, or the popcount of every step-th 128-byte line of the u8 plane
__device__ inline int estimate_wave(const unsigned char* __restrict__ mask, const int* __restrict__ rle_counts,
                                    const long long* __restrict__ rle_offsets, const int* __restrict__ poly_xy,
                                    const long long* __restrict__ poly_ring_off, const long long* __restrict__ poly_inst_rings,
                                    int inst, int HW, int step, int lane) {
  int c = 0;
  if (poly_xy) {  // shoelace area of every part (an estimate: parts may overlap or leave the frame)
    long long tot = 0;
    for (long long r = poly_inst_rings[inst]; r < poly_inst_rings[inst + 1]; ++r) {
      const long long p0 = poly_ring_off[r], n = poly_ring_off[r + 1] - p0;
      long long a2 = 0;
      for (long long i = lane; i < n; i += 64) {
        const long long j = (i + 1 == n) ? 0 : i + 1;
        a2 += (long long)poly_xy[2 * (p0 + i)] * poly_xy[2 * (p0 + j) + 1] - (long long)poly_xy[2 * (p0 + j)] * poly_xy[2 * (p0 + i) + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a2 += __shfl_xor(a2, o);
      tot += (a2 < 0 ? -a2 : a2) / 2;
    }
    c = (int)(tot > (long long)HW ? HW : tot);
    if (lane != 0) c = 0;   // the wave sum below adds the lanes
  } else if (rle_counts) {  // exact: the sum of the ones-runs (odd positions)
    const long long lo = rle_offsets[inst], hi = rle_offsets[inst + 1];
    for (long long k = lo + 1 + 2 * lane; k < hi; k += 128) {
      const int v = rle_counts[k];
      c += v > 0 ? v : 0;
    }
  } else {
    // whole 128-byte lines (HBM delivers nothing smaller): every step-th line of the plane, eight lanes per line,
    // eight lines per lane in flight (VGA: 65 of 2400 lines, one batch)
    const u32x4* src = reinterpret_cast<const u32x4*>(mask + (long long)inst * HW);
    const int nlines = HW >> 7, sub = lane & 7;
    for (int l0 = (lane >> 3) * step; l0 < nlines; l0 += 64 * step) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int l = l0 + k * 8 * step;
        v[k] = (l < nlines) ? src[l * 8 + sub] : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) c += __popc(nz4(v[k].x)) + __popc(nz4(v[k].y)) + __popc(nz4(v[k].z)) + __popc(nz4(v[k].w));
    }
  }
  return wave_sum_i(c);
}

// Self-estimating launch (round 4): instead of a helper kernel in front of the fit, wave 0 of workgroup b estimates instance b (natural
// index) in the kernel's prologue and publishes the key together with a per-call nonce (publish_key_word below: agent-scope stores).  The
// nonce is new for every call, so nothing has to be cleared: a record that does not carry it is "not yet".  order_select waits for
// the 64 records of its chunk; if they do not show up (a workgroup of this launch is not resident
// because something else holds the chip) it computes the missing keys itself - the estimate is a pure function of the mask, so
// everybody sees the same keys whoever wrote them, and nobody waits for ever.
constexpr unsigned ORDER_SPIN_MAX = 256;    // x (s_sleep(8) + two loads) ~ 1 us each: a quarter of a millisecond before the fallback
__device__ inline void st_agent_u32(unsigned* q, unsigned v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent_u64(unsigned long long* q, unsigned long long v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned ld_agent_u32(const unsigned* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long ld_agent_u64(const unsigned long long* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// this thread's share of the estimate of instance inst when NTH threads work on it (NTH = 64: one wave, NT: the workgroup): the
// SAME integer whoever computes it - every step-th 128-byte line of the plane, all eight 16-byte groups of a line (u8 planes);
// the ones-runs (run lengths).  One load in flight per thread: few registers (this code sits in the prologue of the fit kernel).
template <int NTH>
__device__ inline int estimate_share(const FitParams& p, int inst, int t) {
  int c = 0;
  if (p.rle_counts) {
    const long long lo = p.rle_offsets[inst], hi = p.rle_offsets[inst + 1];
#pragma unroll 1
    for (long long k = lo + 1 + 2 * t; k < hi; k += 2 * NTH) {
      const int v = p.rle_counts[k];
      c += v > 0 ? v : 0;
    }
  } else {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.mask + (long long)inst * p.HW);
    const int nlines = p.HW >> 7, sub = t & 7;
#pragma unroll 1
    for (int l = (t >> 3) * p.est_step; l < nlines; l += (NTH / 8) * p.est_step) {
      const u32x4 v = src[l * 8 + sub];
      c += __popc(nz4(v.x)) + __popc(nz4(v.y)) + __popc(nz4(v.z)) + __popc(nz4(v.w));
    }
  }
  return c;
}
// Publication needs NO ordering between stores: the key travels inside both words of its record, each next to one half of the call's
// 64-bit nonce - w0 = nonce.lo : key, w1 = nonce.hi : key.  A reader takes the key only when both words carry the nonce and the same
// key; any other state - stale words of an earlier call, one word of two arrived - reads as "not yet".  (A first version published
// key, fence, flag in separate words: across XCDs the flag could become visible before the key, and a workgroup ranked with the key
// of the PREVIOUS call - one skipped and one duplicated instance in one run of the full suite.)
__device__ inline void publish_key_word(const FitParams& p, int inst, unsigned key) {   // one lane
  const_cast<unsigned*>(p.order_keys)[inst] = key;   // (the plain table: what the helper kernel leaves - tests and tools read it)
  st_agent_u64(p.order_flags + 2 * inst, ((p.order_nonce & 0xffffffffull) << 32) | key);
  st_agent_u64(p.order_flags + 2 * inst + 1, (p.order_nonce & 0xffffffff00000000ull) | key);
}
// the key of instance inst if its record is complete for this call, else 0 (no key is 0: the index bits of an instance < 16383 are not)
__device__ inline unsigned published_key(const FitParams& p, int inst) {
  const unsigned long long w0 = ld_agent_u64(p.order_flags + 2 * inst), w1 = ld_agent_u64(p.order_flags + 2 * inst + 1);
  const bool ok = (w0 >> 32) == (p.order_nonce & 0xffffffffull) && (w1 >> 32) == (p.order_nonce >> 32) && (unsigned)w0 == (unsigned)w1;
  return ok ? (unsigned)w0 : 0u;
}
// one wave estimates (polygon input in the prologue - its shoelace sums are per ring -, and the fallback of order_select); every lane
// returns the key
__device__ inline unsigned estimate_key_wave(const FitParams& p, int inst, int lane) {
  int c;
  if (p.poly_xy) c = estimate_wave(nullptr, nullptr, nullptr, p.poly_xy, p.poly_ring_off, p.poly_inst_rings, inst, p.HW, p.est_step, lane);
  else c = wave_sum_i(estimate_share<64>(p, inst, lane));
  return make_order_key(c, p.order_shift, inst);
}
__device__ inline void estimate_publish_wave(const FitParams& p, int inst, int lane) {
  const unsigned key = estimate_key_wave(p, inst, lane);
  if (lane == 0) publish_key_word(p, inst, key);
}
// the prologue: workgroup b estimates instance b with all its threads (every thread of the workgroup calls it; one barrier)
__device__ inline void estimate_publish_wg(const FitParams& p, int inst, Shared* sh, int tid, int wave, int lane) {
  if (p.poly_xy) {   // uniform
    if (wave == 0) estimate_publish_wave(p, inst, lane);
    return;
  }
  const int c = wave_sum_i(estimate_share<NT>(p, inst, tid));
  if (lane == 0) sh->scan[wave] = (unsigned)c;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) tot += (int)sh->scan[w];
    publish_key_word(p, inst, make_order_key(tot, p.order_shift, inst));
  }
}

// every thread of the workgroup calls it (one barrier); returns the instance of block b, wave-uniform
__device__ inline int order_select(const FitParams& p, int b, Shared* sh, int wave, int lane) {
  const int R = p.B < p.order_resident ? p.B : p.order_resident;
  int grank = b;
  if (b < R) {   // invert: group 0 ascending CU index, every later group descending
    const int g = b >> 8, ng = (R - (g << 8)) < 256 ? (R - (g << 8)) : 256;
    grank = (g << 8) + (g >= 1 ? ng - 1 - (b & 255) : (b & 255));
  }
  const int nch = p.order_nch;
  const int lr = grank / nch, c = grank - lr * nch;        // rank in chunk, chunk
  const int per = p.B / nch, rem = p.B - per * nch;
  const int start = c * per + (c < rem ? c : rem), size = per + (c < rem ? 1 : 0);
  if (wave < ORDER_CHUNK / 64) {
    static_assert(ORDER_CHUNK == 64, "the self-estimating launch waits with one wave per chunk");
    unsigned self_key = 0u;
    if (p.order_self) {   // uniform: the keys of this chunk are being written by workgroups start .. start + size - 1 of this launch
      unsigned spins = 0;
      unsigned long long missing;
      while (true) {
        if (lane < size && self_key == 0u) self_key = published_key(p, start + lane);
        missing = __ballot(lane < size && self_key == 0u);
        if (missing == 0ull || spins >= ORDER_SPIN_MAX) break;
        __builtin_amdgcn_s_sleep(8);
        ++spins;
      }
      // (diagnostics: the word behind the records counts the keys computed here; nobody clears it - tests zero the workspace first)
      if (missing && lane == 0) atomicAdd(p.order_flags + 2 * (long long)p.B, (unsigned long long)__popcll(missing));
      while (missing) {   // (fallback, normally never: see above)
        const int m = __ffsll((long long)missing) - 1;
        missing &= missing - 1ull;
        const unsigned km = estimate_key_wave(p, start + m, lane);   // (every lane gets the key; not published: the owner will)
        if (lane == m) self_key = km;
      }
    }
    unsigned k[ORDER_CHUNK / 64];
#pragma unroll
    for (int h = 0; h < ORDER_CHUNK / 64; ++h) {
      const int l = h * 64 + lane;
      k[h] = 0u;   // key 0 never counts as larger
      if (l < size) k[h] = p.area_hint ? make_order_key(p.area_hint[start + l], p.order_shift, start + l)
                                       : (p.order_self ? self_key : p.order_keys[start + l]);
    }
    unsigned mine = k[0];
#pragma unroll
    for (int h = 1; h < ORDER_CHUNK / 64; ++h) mine = wave == h ? k[h] : mine;
    int rank = 0;
#pragma unroll
    for (int h = 0; h < ORDER_CHUNK / 64; ++h)
#pragma unroll
      for (int t = 0; t < 64; ++t) rank += ((unsigned)__builtin_amdgcn_readlane((int)k[h], t) > mine) ? 1 : 0;
    // (keys are unique, so exactly one lane matches; the default and the clamp below only matter if the key table was
    // clobbered - a workspace shared by two concurrent calls - and turn a wild instance index into a duplicated fit)
    if (wave == 0 && lane == 0) sh->order_inst = start;
    if (wave * 64 + lane < size && rank == lr) sh->order_inst = start + wave * 64 + lane;
  }
  __syncthreads();
  const int inst = __builtin_amdgcn_readfirstlane(sh->order_inst);
  return inst < 0 ? 0 : (inst >= p.B ? p.B - 1 : inst);
}

// ------------------------------------------------------------------------------------------
// launch order (see order_select above): the one helper kernel left estimates every instance's mask area - one wave per
// instance, spread over the whole chip (eight workgroups pulling the samples through eight CUs take 2x longer than the
// fit saves: profiles/r03/r03_launch_order.txt) - and writes a sort key per instance.
// ------------------------------------------------------------------------------------------
constexpr int EST_STEP = 37;        // area estimate: every 37th 128-byte line of the plane (37 is coprime to W/128 = 5, 10,
                                    // 15: the lattice visits every column block); small frames take a smaller prime so
                                    // that at least 64 lines are sampled

__global__ __launch_bounds__(256) void size_estimate_kernel(const unsigned char* __restrict__ mask,
                                                            const int* __restrict__ rle_counts,
                                                            const long long* __restrict__ rle_offsets,
                                                            const int* __restrict__ poly_xy, const long long* __restrict__ poly_ring_off,
                                                            const long long* __restrict__ poly_inst_rings, int B, int HW,
                                                            int step, int shift, unsigned* __restrict__ keys, int* __restrict__ band_arrive) {
  const int lane = threadIdx.x & 63;
  const int inst = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (inst >= B) return;
  const int c = estimate_wave(mask, rle_counts, rle_offsets, poly_xy, poly_ring_off, poly_inst_rings, inst, HW, step, lane);
  if (lane == 0) {
    keys[inst] = make_order_key(c, shift, inst);
  }
}

// size-balanced launch order on for this call?  (per-call opt_order, else the process default)
inline bool balance_enabled(const FitParams& p) {
  if (p.opt_order == LA3D_ORDER_OFF) return false;
  if (p.opt_order == LA3D_ORDER_ON) return true;
  return config().balance != 0;
}

inline int balance_max_rounds() {
  return config().balance_rounds;  // measured: +21 % at one resident set, +9 % at two, +3 % at three, none at four, negative beyond
}

// ------------------------------------------------------------------------------------------
// pass-B tile culling (plain build, round 4).  The six extents are min / max over the points, so a tile that provably cannot
// move any of them need not be visited - the records stay bit-identical.  Pass A leaves [dlo, dhi], the range of the valid
// depths of every active tile (tile_range).  A coordinate of the yaw frame is q = d * rho(u, v) with rho affine in the pixel,
// so over a tile q lies between the extremes of the four products {dlo, dhi} x {rho_min, rho_max} (rho at the tile corners),
// widened by a slack far above the rounding of the pixel math (2^-40 of the largest product the tile could form; the pixel
// math differs from the corner evaluation by a few ulp).  Stage 1 picks, per direction, the tile with the most extreme bound
// (six "champions": where the true extreme most likely sits, interior tiles included - the nearest point of a convex object
// is not on its silhouette) and runs the exact pixel math on them: their extents E are achieved values.  Stage 2 keeps the
// tiles whose bounds reach beyond E in some direction (ties cannot change a min / max) and compacts them into the survivor
// list the work queue of pass B walks.  Config 2 (random depth): 40 % of the active tiles survive (26 % of the large
// instances', which are the launch's critical path); smooth depth: 15-30 %.
// ------------------------------------------------------------------------------------------
// CULL_MIN (la3d_device.hpp): active tiles below which the plan costs more than it saves (measured: profiles/r04/r04_cull.txt); the
// per-call value is FitParams::cull_min: 128 for u8 planes, whose launches are bandwidth-bound - after the cheaper tile range
// B = 1024 / 1536 / 2048 run 98.6 / 133.1 / 170.1 -> 96.8 / 130.3 / 164.6 us, config-5 masks unchanged, run lengths 68.0 -> 69.1
// (hence 224 there); profiles/r04/r04_cull_threshold.txt
constexpr int CULL_MAXT = 2 * NT;   // tiles the plan handles (two per thread)

// bounds L <= q <= U of one yaw-frame coordinate q = d * rho, rho = a[0] u + a[1] v + a[2], over tile (tx, ty) for depths in
// [dlo, dhi] >= 0 (a negative / infinite / NaN depth makes the tile unbounded).  rho over the tile = centre +- radius; the
// slack (2^-40 of the largest product the tile could form) is far above the rounding of the pixel math.
__device__ inline void cull_bound1(double u0, double v0, double dlo, double dhi, bool unbounded, const double* a, double* L, double* U) {
  const double rc = fma(a[0], u0 + 15.5, fma(a[1], v0 + 3.5, a[2]));
  const double rad = fabs(a[0]) * 15.5 + fabs(a[1]) * 3.5;
  const double rmin = rc - rad, rmax = rc + rad;
  const double slack = dhi * 9.094947017729282e-13 * fma(fabs(a[0]), u0 + 32.0, fma(fabs(a[1]), v0 + 8.0, fabs(a[2])));
  *L = unbounded ? -INFINITY : fmin(dlo * rmin, dhi * rmin) - slack;
  *U = unbounded ? INFINITY : fmax(dlo * rmax, dhi * rmax) + slack;
}

// Every thread of the workgroup calls it (four barriers).  On return ext[] holds the champions' extents (the start values of
// pass B, the same in every lane) and the survivor list sits in the range area; returns the number of survivors (uniform).
template <bool CHK>
__device__ inline int cull_plan(Shared* sh, const FitParams& p, const float* __restrict__ dpl, unsigned* bits,
                                const unsigned short* list, int nactive, int rng_words, const double* N0, const double* M1,
                                const double* N2, int tid, int wave, int lane, double* ext) {
  const unsigned* rng = bits + nactive * 8;
  unsigned short* surv = reinterpret_cast<unsigned short*>(bits + nactive * 8);
  float* cval = reinterpret_cast<float*>(bits + nactive * 8 + rng_words - CULL_SCRATCH_WORDS);   // [NWAVE][6]
  unsigned* cidx = reinterpret_cast<unsigned*>(cval + NWAVE * 6);                                 // [NWAVE][6]
  // ---- stage 1: champions ----
  float bv[6];
  int bi[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { bv[k] = -INFINITY; bi[k] = 0; }
  for (int t = tid; t < nactive; t += NT) {
    const unsigned tt = list[t];
    const uint2 rg = *reinterpret_cast<const uint2*>(rng + 2 * t);
    const bool unbounded = rg.y >= 0x7f800000u;
    const double dlo = (double)__uint_as_float(rg.x), dhi = (double)__uint_as_float(rg.y);
    const double u0 = (double)((tt & 0xffu) * 32u), v0 = (double)((tt >> 8) * 8u);
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // directions 2k: the minimum (as -L), 2k + 1: the maximum
      double L, U;
      cull_bound1(u0, v0, dlo, dhi, unbounded, k == 0 ? N0 : (k == 1 ? M1 : N2), &L, &U);
      const float a = -(float)L, b = (float)U;
      if (a > bv[2 * k]) { bv[2 * k] = a; bi[2 * k] = t; }
      if (b > bv[2 * k + 1]) { bv[2 * k + 1] = b; bi[2 * k + 1] = t; }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float m = bv[k];
    m = fmaxf(m, __int_as_float(dpp_i32<DPP_XOR1>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<DPP_XOR2>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<DPP_HALF_MIRROR>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<DPP_MIRROR>(__float_as_int(m))));
    const float w = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0)),
                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 16))),
                          fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32)),
                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 48))));
    const unsigned long long hit = __ballot(bv[k] == w);
    const int src = hit ? (int)__builtin_ctzll(hit) : 0;
    const int idx = __builtin_amdgcn_readlane(bi[k], src);
    if (lane == 0) { cval[wave * 6 + k] = w; cidx[wave * 6 + k] = (unsigned)idx; }
  }
  __syncthreads();
  int champ[6];
  {
    float v = -INFINITY;
    int i = 0;
    if (lane < 6) {
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) {
        const float cv = cval[w * 6 + lane];
        if (cv > v) { v = cv; i = (int)cidx[w * 6 + lane]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) champ[k] = __builtin_amdgcn_readlane(i, k);
  }
  // the exact pixel math on the champions: wave w < 6 takes champion w, all lanes over its pixels
  TileCtx c;
  c.W = p.W; c.H = p.H; c.ntx = p.ntx; c.r = lane >> 3; c.cq = lane & 7;
  c.compact = 1;
  {
    const int k = (p.mask_lds_bytes - nactive * 32 - rng_words * 4) >> 10;
    c.keepn = k > 0 ? k : 0;
  }
  c.keep = reinterpret_cast<uint4*>(bits + nactive * 8 + rng_words);
  c.rng = nullptr; c.surv = nullptr;
  c.a00 = N0[0]; c.a01 = N0[1]; c.a02 = N0[2];
  c.a10 = M1[0]; c.a11 = M1[1]; c.a12 = M1[2];
  c.a20 = N2[0]; c.a21 = N2[1]; c.a22 = N2[2];
  if (wave < 6) {
    int e = champ[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) e = wave == k ? champ[k] : e;
    const unsigned tt = __builtin_amdgcn_readfirstlane((unsigned)list[e]);
    const int tx = (int)(tt & 0xffu), ty = (int)(tt >> 8);
    const unsigned nib = (bits[e * 8 + c.r] >> (c.cq * 4)) & 0xFu;
    uint4 dq = make_uint4(0u, 0u, 0u, 0u);
    if (e < c.keepn) dq = c.keep[e * 64 + lane];
    else if (nib) dq = *reinterpret_cast<const uint4*>(dpl + (long long)(ty * 8 + c.r) * c.W + tx * 32 + c.cq * 4);
    const unsigned db[4] = {dq.x, dq.y, dq.z, dq.w};
    const double vd = (double)(ty * 8 + c.r), ud = (double)(tx * 32 + c.cq * 4);
    const double r0 = fma(c.a00, ud, fma(c.a01, vd, c.a02));
    const double r1 = fma(c.a10, ud, fma(c.a11, vd, c.a12));
    const double r2 = fma(c.a20, ud, fma(c.a21, vd, c.a22));
    int dummy = 0;
    double cx[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
    quad_math<1, CHK>(nib, db, r0, r1, r2, c.a00, c.a10, c.a20, cx, &dummy);
    const double e0 = wave_min(cx[0]), e1 = wave_max(cx[1]), e2 = wave_min(cx[2]), e3 = wave_max(cx[3]),
                 e4 = wave_min(cx[4]), e5 = wave_max(cx[5]);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = e0; pp[1] = e1; pp[2] = e2; pp[3] = e3; pp[4] = e4; pp[5] = e5;
    }
  }
  __syncthreads();
  // ---- stage 2: survivors ----
  // E = the champions' extents (achieved values), combined per wave like stage_extents_to_box does and moved to SGPRs
  double Elo[3], Ehi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double lo = lane < 6 ? sh->part[lane][2 * k] : INFINITY, hi = lane < 6 ? sh->part[lane][2 * k + 1] : -INFINITY;
    lo = fmin(lo, dpp_f64<DPP_XOR1>(lo)); lo = fmin(lo, dpp_f64<DPP_XOR2>(lo)); lo = fmin(lo, dpp_f64<DPP_HALF_MIRROR>(lo));
    hi = fmax(hi, dpp_f64<DPP_XOR1>(hi)); hi = fmax(hi, dpp_f64<DPP_XOR2>(hi)); hi = fmax(hi, dpp_f64<DPP_HALF_MIRROR>(hi));
    Elo[k] = readlane_f64(lo, 0); Ehi[k] = readlane_f64(hi, 0);
    ext[2 * k] = Elo[k]; ext[2 * k + 1] = Ehi[k];   // every lane starts pass B from the champions' extents
  }
  bool sv[2] = {false, false};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int t = tid + h * NT;
    if (t < nactive) {
      const unsigned tt = list[t];
      const uint2 rg = *reinterpret_cast<const uint2*>(rng + 2 * t);
      const bool unbounded = rg.y >= 0x7f800000u;
      const double dlo = (double)__uint_as_float(rg.x), dhi = (double)__uint_as_float(rg.y);
      const double u0 = (double)((tt & 0xffu) * 32u), v0 = (double)((tt >> 8) * 8u);
      bool inside = true;   // (written so that a NaN bound keeps the tile)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double L, U;
        cull_bound1(u0, v0, dlo, dhi, unbounded, k == 0 ? N0 : (k == 1 ? M1 : N2), &L, &U);
        inside = inside && (L >= Elo[k]) && (U <= Ehi[k]);
      }
      bool is_champ = false;
#pragma unroll
      for (int k = 0; k < 6; ++k) is_champ = is_champ || (t == champ[k]);
      sv[h] = !inside && !is_champ && rg.x <= rg.y;   // (rg.x > rg.y: the tile has no valid pixel)
    }
  }
  const unsigned long long b0 = __ballot(sv[0]), b1 = __ballot(sv[1]);
  if (lane == 0) sh->scan[wave] = (unsigned)(__popcll(b0) + __popcll(b1));
  __syncthreads();   // (every thread has also read its ranges by now: the survivors may overwrite them)
  int base = 0, nsurv = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) {
    const int cw = (int)sh->scan[w];
    if (w < wave) base += cw;
    nsurv += cw;
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  if (sv[0]) surv[base + __popcll(b0 & below)] = (unsigned short)tid;
  if (sv[1]) surv[base + __popcll(b0) + __popcll(b1 & below)] = (unsigned short)(tid + NT);
  __syncthreads();
  return nsurv;
}

// the thread index rebuilt from the wave's scalar index and the lane, opaque to common-subexpression elimination (every use gets
// its own short-lived register)
__device__ inline int tid_here(int wave, int lane) {
  int t = (wave << 6) | lane;
  asm volatile("" : "+v"(t));
  return t;
}


// Ordering of the exchange: every exchanged word is written and read with AGENT-scope relaxed atomics - single instructions that
// are coherent at the L2 / memory side by themselves (sc1) - so what is needed between "my record" and "my arrival" is that the
// record's stores have been ACKNOWLEDGED before the arrival is issued, and between "their arrival" and "their record" that the
// poll's load has returned before the record's loads are issued.  Round 5 makes both explicit: band_release() = s_waitcnt
// vmcnt(0) (gfx9 counts stores in vmcnt too) in front of the arrival, band_acquire() = the same wait behind the poll; both are
// compiler barriers as well.  (Round 4 had a workgroup-scope FENCE here, which on gfx950 does not wait for outstanding global
// stores: the order held only because tagged_arrive's own load in front of its CAS forced a vmcnt(0) - ADVICE round 4.)  An
// agent-scope fence / release would also write back and invalidate the XCD's whole L2 - in the middle of everybody's streams:
// measured 409 us instead of 107 us per 1024-instance call with four of them per workgroup (profiles/r04/r04_band.txt) - and is
// not needed: nothing here relies on PLAIN stores becoming visible.  No assumption about which XCD a block lands on is made
// (LA3D_BAND_TEST=2 permutes the blocks so that the bands of an instance sit on different XCDs: tests/test_gpu_band.py).
__device__ inline void band_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
__device__ inline void band_acquire() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// internal states of a band workgroup after a watchdog timeout (never written to p.status)
constexpr int BAND_ST_TAKEOVER = 100;   // this band claimed the instance: it fits the WHOLE instance on its own (band_takeover)
constexpr int BAND_ST_ABANDON = 101;    // another band of the instance claimed it: leave without writing anything
// Arrival counters that nobody has to clear (round 4, late: the band engine is ONE launch - no memset in front): a word holds the
// call's 48-bit tag and a 16-bit count; the first arrival of a call finds another tag and starts the count at one.  (Calls captured
// into a HIP graph replay with the same tag: there the words are cleared by a memset node, as before.)
__device__ inline unsigned tagged_arrive(unsigned long long* w, unsigned long long tag) {   // returns the count including this arrival
  unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    const unsigned long long want = ((old >> 16) == tag ? old : (tag << 16)) + 1ull;
    const unsigned long long prev = atomicCAS(w, old, want);
    if (prev == old) return (unsigned)(want & 0xffffull);
    old = prev;
  }
}
// The same counter for MANY arrivals per word at about the same time (the row engine: up to sixteen bands of an instance finish
// together; the CAS loop above then retries once per competitor - measured 20 us for sixteen): once the word carries this call's
// tag an arrival is ONE atomic add; only the arrivals that still see a foreign tag compete for the reset.
__device__ inline unsigned tagged_arrive_many(unsigned long long* w, unsigned long long tag) {
  while (true) {
    const unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 16) == tag) return (unsigned)(atomicAdd(w, 1ull) & 0xffffull) + 1u;   // (the tag stays for the rest of the call)
    if (atomicCAS(w, old, (tag << 16) + 1ull) == old) return 1u;                       // this arrival opened the call's count
  }
}
__device__ inline unsigned tagged_count(const unsigned long long* w, unsigned long long tag) {
  const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v >> 16) == tag ? (unsigned)(v & 0xffffull) : 0u;
}
__device__ inline void st_agent(double* q, double v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_agent(const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte write-through store (sc0 sc1): what one workgroup hands another through global memory without a release fence - scalar
// sc1 stores are one fabric write each (a dword costs ~6 x the time per byte of a dwordx4: MI355X_MICROARCH.md, "stores of each
// flavour"), so exchanged arrays go out in 16-byte granules.  (The s_nop keeps the data registers untouched while the store reads them.)
__device__ inline void st16_through(void* q, uint4 v) {
  const u32x4 w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(q), "v"(w) : "memory");
}
// Four independent 16-byte loads of such data in flight at once, then one wait: the compiler puts an s_waitcnt vmcnt(0) behind EVERY
// agent-scope atomic load (measured: the merge of sixteen bands through __hip_atomic_load took 20 us), and it cannot see into inline
// assembly, so the wait is part of the block.  (Early-clobber outputs: no result register doubles as a later address.)
// one such load (and its wait): where the loads of a hand-off can be spread over lanes instead of batched in one
__device__ inline u32x4 ld16_through(const void* q) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(q) : "memory");
  return r;
}
__device__ inline void ld16x4_through(const void* p0, const void* p1, const void* p2, const void* p3, u32x4* a, u32x4* b, u32x4* c, u32x4* d) {
  u32x4 r0, r1, r2, r3;
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
               "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
               "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  *a = r0; *b = r1; *c = r2; *d = r3;
}

}  // namespace
