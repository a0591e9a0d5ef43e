// la3d_rows.hip - the ROW ENGINE of la3d_fit_instances (rounds 5-6): fit_rows_kernel (one launch: the band that arrives last merges
// its instance), merge_rows_kernel (the two-launch form) and their launcher.  Un-grounded u8 batches up to 160 instances by default.
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
#include "la3d_walks.hpp"
#include "la3d_stages.hpp"


namespace {
// ------------------------------------------------------------------------------------------
// row engine (round 5): NB workgroups per instance, one per band of tile rows, for SMALL batches of u8 planes without a ground
// array - the separable single pass split by rows.  Everything the single pass accumulates is a sum or a min / max, so the bands
// need no exchange and no co-residency: every band workgroup streams only ITS rows of the mask plane, builds its tile list, runs
// sweep_sep over its rows (band-local bit image / list / depth pointer, the frame row of its first tile row handed in) and leaves a
// partial record - five sums, the y extent, the mask count, a flag - and its per-column depth ranges in the workspace; a second,
// short launch (merge_rows_kernel: one workgroup per instance) adds the partials in a fixed order and runs the SAME axis / extent /
// box stages as the instance engine.  A lone workgroup needs ~27 us for one 640x480 instance (5 us to stream the plane, 12-17 us
// in a pass that has only its own 16 loads per wave in flight); sixteen bands need a sixteenth of each.  A band that cannot take
// the single pass (skewed K, a NaN / inf / negative depth under the mask) raises its flag and the merge workgroup fits the whole
// instance with the generic row-linear two-pass walk (what the band engine's take-over uses): slow, rare, never a dropped box.
// ------------------------------------------------------------------------------------------
constexpr int ROWS_NB_MAX = 16;
constexpr int ROWS_PART_D = 20;   // doubles per (instance, band): Sx, Sz, Sxx, Sxz, Szz | ymin, ymax | mask pixels (+ ROWS_FLAG) | pad[2] | M[9] | pad
constexpr int ROWS_MAX_B = 512;
constexpr double ROWS_FLAG = 1099511627776.0;   // 2^40, added to a band's pixel count: "this band could not take the single pass"

struct RowsArgs {
  int nb;           // bands per instance (the last ones may be shorter; every band holds at least one tile row)
  int trows;        // tile rows per band
  int bits_bytes;   // band bit image + per-column ranges (16-aligned): LDS in front of Shared
  double* part;     // [B][nb][ROWS_PART_D]
  unsigned* col;    // [B][nb][2 W]: colmin | colmax of the band
  // round 6, the ONE-launch form: the band workgroup that arrives LAST at its instance's counter merges the instance (nobody ever
  // waits); arrive = [B] tagged arrival words (tagged_arrive: never cleared), null = the two-launch form (merge_rows_kernel)
  unsigned long long* arrive;
  unsigned long long tag;
};

// host: bands for a batch of B instances on an H x W frame; false = the row engine does not apply
inline bool rows_plan(int B, int H, int W, RowsArgs* ra) {
  if (B < 1 || B > ROWS_MAX_B || W % 32 != 0 || W / 32 > 255 || H < 16) return false;
  const int nty = (H + 7) / 8;   // (a frame height that is not a multiple of 8 - COCO's 427 - leaves the last band a partial tile row)
  int nb = ROWS_NB_MAX;
  // two-launch form (round 5): about 500-640 workgroups in all (profiles/r05/r05_rows_engine.txt: B = 64 / 128 / 192, us per call with
  // at most 256 | 512 | 1024 | 2048 workgroups: 24.4 | 24.0 | 27.9 | 27.9; 32.7 | 30.0 | 33.1 | 36.8; 38.2 | 38.0 | 40.7 | 39.8).
  // The one-launch form (round 6) keeps the plan: with a full resident round (1024 workgroups) B = 64 takes sixteen bands per instance
  // and 39.4 us instead of 25.0, and above ~170 instances more bands per instance do not help at all - B = 256 as 2 / 4 / 8 bands:
  // 46.2 / 47.6 / 60.7 us against 44.3 with one workgroup per instance (profiles/r06/r06_rows_engine.txt): the bands of a call stream,
  // list, walk and merge in lockstep, so the call lasts (chain of one band) + (bytes / bandwidth) however fine the bands are.
  const int wg_cap = config().rows_wgs;
  while (nb > 2 && B * nb > wg_cap) nb >>= 1;
  int trows = (nty + nb - 1) / nb;
  if (trows < 2) trows = 2;                          // (a band of one tile row is all fixed cost)
  nb = (nty + trows - 1) / trows;
  if (nb < 2 || (long long)(W / 32) * trows > 256 * NWAVE) return false;   // one-pass tile list: <= 256 tiles per wave
  const long long bits = ((long long)trows * W + sep_col_words(W) * 4 + 15) & ~15LL;
  if (bits + (long long)sizeof(Shared) + (long long)(W / 32) * trows * 2 + 64 > 64 * 1024) return false;
  ra->nb = nb; ra->trows = trows; ra->bits_bytes = (int)bits;
  return true;
}
inline size_t rows_workspace_bytes_impl(int B, int H, int W) {
  RowsArgs ra;
  if (!rows_plan(B, H, W, &ra)) return 0;
  return (((size_t)B * ra.nb * ROWS_PART_D * 8 + 255) & ~(size_t)255) + (((size_t)B * ra.nb * 2 * W * 4 + 255) & ~(size_t)255) + (size_t)B * 8 + 256;
}

// The plainest walk over one instance: thread t visits pixels t, t + NT, ... of the u8 plane, one at a time.  PASS 0: count and
// moments of (x', z'); PASS 1: the six extents (A0 / A1 / A2 as in `sweep`).  Used where a path is rare and registers are scarce.
// piv (pass 0, LDS): the pivot of the moments - zeros, or the mean an ill-conditioned first pass left (axis_from_sums); read per pixel
// (x + -0.0 is x, bit for bit: the walk about zero gives the sums it always gave).
template <int PASS>
__device__ inline void sweep_plain(const FitParams& p, const float* __restrict__ dpl, const unsigned char* __restrict__ mpl,
                                   const double* A0, const double* A1, const double* A2, int tid, double* acc, int* cnt, int* nmask,
                                   const double* piv = nullptr) {
#pragma clang loop unroll(disable) vectorize(disable)
  for (int i = tid; i < p.HW; i += NT) {
    if (!mpl[i]) continue;
    if (PASS == 0) *nmask += 1;
    const float df = dpl[i];
    if (!finite_f32(df)) continue;
    unsigned u, v;
    pix_uv((unsigned)i, p.W, p.rcpW, &u, &v);
    const double ud = (double)u, vd = (double)v, d = (double)df;
    double x, z;
    if (PASS == 0) { x = fma(d, fma(A0[0], ud, fma(A0[1], vd, A0[2])), -piv[0]); z = fma(d, fma(A2[0], ud, fma(A2[1], vd, A2[2])), -piv[1]); }
    else { x = d * fma(A0[0], ud, fma(A0[1], vd, A0[2])); z = d * fma(A2[0], ud, fma(A2[1], vd, A2[2])); }
    if (PASS == 0) {
      acc[0] += x; acc[1] += z;
      acc[2] = fma(x, x, acc[2]); acc[3] = fma(x, z, acc[3]); acc[4] = fma(z, z, acc[4]);
      *cnt += 1;
    } else {
      const double y = d * fma(A1[0], ud, fma(A1[1], vd, A1[2]));
      acc[0] = dmin(acc[0], x); acc[1] = dmax(acc[1], x);
      acc[2] = dmin(acc[2], y); acc[3] = dmax(acc[3], y);
      acc[4] = dmin(acc[4], z); acc[5] = dmax(acc[5], z);
    }
  }
}

// The merge of one instance by one workgroup: the partial records of its nb bands (band b in lane b of wave 0: the fixed tree of
// stage_moments_to_axis adds them - the same operands in the same order whichever workgroup merges) and the bands' per-column
// depth ranges (min / max INTO mcol: LDS, 2 W words, holding either the merging band's own ranges or the identities) -> status,
// axis, extents, record: the instance engine's stages.  Everything another workgroup wrote is read with agent-scope loads.
template <bool XCH>
__device__ inline void rows_merge(Shared* sh, const FitParams& p, const RowsArgs& ra, int inst, int img, unsigned* mcol, int tid,
                                  int wave, int lane) {
  // XCH: the data was written by other workgroups of THIS launch with 16-byte write-through stores and is loaded the same way,
  // four granules per thread in flight (ld16x4_through); else (the merge launch) plain loads
  const int W = p.W;
  const double* part = ra.part + (long long)inst * ra.nb * ROWS_PART_D;
  const unsigned* gc = ra.col + (long long)inst * ra.nb * 2 * W;
  // the per-column ranges first (the longest chain of loads).  Work item = (16-byte granule g of [colmin W | colmax W], four bands):
  // min / max of the four, folded into mcol with LDS atomics (a band index past the last band repeats the last one: harmless)
  const int ngran = W / 2, nq = (ra.nb + 3) >> 2;
  for (int it = tid; it < ngran * nq; it += NT) {
    const int bq = it / ngran, g = it - bq * ngran;
    const bool is_max = 2 * g >= ngran;
    const u32x4* src = reinterpret_cast<const u32x4*>(gc) + g;
    const long long bs = (long long)W / 2;   // granules per band
    const int b0 = bq * 4, b1 = min(b0 + 1, ra.nb - 1), b2 = min(b0 + 2, ra.nb - 1), b3 = min(b0 + 3, ra.nb - 1);
    u32x4 x0, x1, x2, x3;
    if (XCH) ld16x4_through(src + b0 * bs, src + b1 * bs, src + b2 * bs, src + b3 * bs, &x0, &x1, &x2, &x3);
    else { x0 = src[b0 * bs]; x1 = src[b1 * bs]; x2 = src[b2 * bs]; x3 = src[b3 * bs]; }
    unsigned* dst = mcol + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (is_max) atomicMax(dst + k, max(max(x0[k], x1[k]), max(x2[k], x3[k])));
      else atomicMin(dst + k, min(min(x0[k], x1[k]), min(x2[k], x3[k])));
    }
  }
  if (!XCH && tid < 9) sh->M[tid] = part[10 + tid];              // band 0's camera (every band computed the same one; a band
                                                                  // that merges keeps its own: the same expression of the same K)
  else if (tid >= 9 && tid < 18) sh->Rg[tid - 9] = ((tid - 9) % 4 == 0) ? 1.0 : 0.0;   // no ground array: the identity (ground_rotation(nullptr))
  if (tid == 18) { sh->bad_ground = 0; sh->order_inst = inst; sh->sep_bad = 0; set_pivot(sh, 0.0, 0.0); }
  double acc[5] = {0, 0, 0, 0, 0}, ylo = INFINITY, yhi = -INFINITY;
  int nm = 0, flag = 0;
  if (tid < ra.nb) {   // band b's record in lane b of wave 0: its first four granules
    const u32x4* q = reinterpret_cast<const u32x4*>(part + (long long)tid * ROWS_PART_D);
    u32x4 x0, x1, x2, x3;
    if (XCH) ld16x4_through(q, q + 1, q + 2, q + 3, &x0, &x1, &x2, &x3);
    else { x0 = q[0]; x1 = q[1]; x2 = q[2]; x3 = q[3]; }
    acc[0] = __hiloint2double((int)x0[1], (int)x0[0]); acc[1] = __hiloint2double((int)x0[3], (int)x0[2]);
    acc[2] = __hiloint2double((int)x1[1], (int)x1[0]); acc[3] = __hiloint2double((int)x1[3], (int)x1[2]);
    acc[4] = __hiloint2double((int)x2[1], (int)x2[0]); ylo = __hiloint2double((int)x2[3], (int)x2[2]);
    yhi = __hiloint2double((int)x3[1], (int)x3[0]);
    double cntf = __hiloint2double((int)x3[3], (int)x3[2]);
    if (cntf >= ROWS_FLAG) { flag = 1; cntf -= ROWS_FLAG; }
    nm = (int)cntf;
  }
  bool generic = __syncthreads_or(flag) != 0;   // (also publishes M / Rg / mcol)
  // generic: a band could not take the single pass (a NaN / inf / negative depth under the mask, a skewed K) - the whole instance
  // by this workgroup, pixel by pixel straight from the planes (sweep_plain: rare, written for few registers, not for speed)
  if (!generic) {   // uniform
    stage_moments_to_axis(sh, p, inst, acc, nm, nm, tid, wave, lane, true);
    if (sh->redo) { generic = true; __syncthreads(); }   // uniform: non-finite or ill-conditioned sums (then the stage left a pivot)
  }
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
  const unsigned char* mpl = p.mask + (long long)inst * p.HW;
  if (generic) {
    // at most two walks: about the pivot at hand (zero, or the mean of ill-conditioned merged sums), then - should THESE sums be
    // ill-conditioned (the first walk ran about zero) - about their mean
    double gacc[5] = {0, 0, 0, 0, 0};
    int cnt = 0, nmask = 0;
    sweep_plain<0>(p, dpl, mpl, sh->M, sh->M + 3, sh->M + 6, tid, gacc, &cnt, &nmask, pivot_ptr(sh));
    stage_moments_to_axis(sh, p, inst, gacc, cnt, nmask, tid, wave, lane, true);
    if (sh->redo) {   // uniform
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 5; ++i) gacc[i] = 0;
      cnt = 0; nmask = 0;
      sweep_plain<0>(p, dpl, mpl, sh->M, sh->M + 3, sh->M + 6, tid, gacc, &cnt, &nmask, pivot_ptr(sh));
      stage_moments_to_axis(sh, p, inst, gacc, cnt, nmask, tid, wave, lane, false);
    }
  }
  if (sh->st != LA3D_BOX_OK) return;
  double ext[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  if (generic) {
    if (tid < 3) {   // rows 0 and 2 of rotate_y(yaw) @ M through LDS: this route keeps nothing wave-uniform in registers
      sh->part[0][tid] = sh->cyaw * sh->M[tid] + sh->syaw * sh->M[6 + tid];
      sh->part[1][tid] = -sh->syaw * sh->M[tid] + sh->cyaw * sh->M[6 + tid];
    }
    __syncthreads();
    int d0 = 0, d1 = 0;
    sweep_plain<1>(p, dpl, mpl, sh->part[0], sh->M + 3, sh->part[1], tid, ext, &d0, &d1);
    __syncthreads();
  } else {
    double Mg[9], N0[3], N2[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Mg[i] = uniform_f64(sh->M[i]);
    yaw_rows(sh, Mg, N0, N2);
    sep_col_extents(mcol, W, N0, N2, tid, ext);
    ext[2] = ylo; ext[3] = yhi;
  }
  stage_extents_to_box(sh, p, inst, ext, tid, wave, lane);
  stage_status_aux(sh, p, inst, tid);
}

__global__ __launch_bounds__(NT, NT / 64) void fit_rows_kernel(const FitParams p, const RowsArgs ra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  Shared* sh = reinterpret_cast<Shared*>(smem + ra.bits_bytes);
  unsigned short* list = reinterpret_cast<unsigned short*>(smem + ra.bits_bytes + sizeof(Shared));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v = (int)blockIdx.x;
  const int inst = v / ra.nb, band = v - inst * ra.nb;
  const int img = p.image_index ? p.image_index[inst] : inst;
  const int row0 = band * ra.trows * 8;
  const int trows = min(ra.trows, (p.H + 7) / 8 - band * ra.trows);   // >= 1 (rows_plan)
  const int rows = min(trows * 8, p.H - row0);                        // pixel rows of the band that lie inside the frame
  const int W = p.W, ntx = W / 32;
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride + (long long)row0 * W;
  const unsigned char* mpl = p.mask + (long long)inst * p.HW + (long long)row0 * W;
  if (tid == NT - 1) {   // M = K^-1 (no ground array: Rg is the identity; the same expression as the instance engine's)
    double Kinv[9], Rg[9];
    inv3_camera(p.K + (long long)img * p.k_stride, Kinv);
    (void)ground_rotation(nullptr, Rg);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) sh->M[i * 3 + j] = Rg[i] * Kinv[j] + Rg[3 + i] * Kinv[3 + j] + Rg[6 + i] * Kinv[6 + j];
  }
  // ---- the band's rows of the u8 plane -> bit image (the instance engine's optimistic 0 / 1 form, general form behind it) ----
  unsigned short* b16 = reinterpret_cast<unsigned short*>(bits);
  const int ngroups = rows * W / 16;
  for (int g = ngroups + tid; g < trows * 8 * W / 16; g += NT) b16[g] = 0;   // (rows of the last tile row past the frame: no pixels)
  const u32x4* m4 = reinterpret_cast<const u32x4*>(mpl);
  int nmask = 0;
  {
    unsigned seen = 0;
#pragma unroll 4
    for (int g = tid; g < ngroups; g += NT) {
      const u32x4 w = __builtin_nontemporal_load(m4 + g);
      const unsigned lo = __builtin_amdgcn_udot4(w.y, 0x80402010u, __builtin_amdgcn_udot4(w.x, 0x08040201u, 0u, false), false);
      const unsigned hi = __builtin_amdgcn_udot4(w.w, 0x80402010u, __builtin_amdgcn_udot4(w.z, 0x08040201u, 0u, false), false);
      const unsigned pat = lo | (hi << 8);
      seen |= (w.x | w.y) | (w.z | w.w);
      b16[g] = (unsigned short)pat;
      nmask += __popc(pat);
    }
    const unsigned long long odd = __ballot((seen & 0xfefefefeu) != 0);
    if (lane == 0) sh->scan[wave] = odd != 0 ? 1u : 0u;
    __syncthreads();
    unsigned general = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) general |= sh->scan[w];
    if (general) {   // uniform: some byte is neither 0 nor 1
      nmask = 0;
#pragma unroll 4
      for (int g = tid; g < ngroups; g += NT) {
        const u32x4 w = m4[g];
        const unsigned pat = nz16(w.x, w.y, w.z, w.w);
        b16[g] = (unsigned short)pat;
        nmask += __popc(pat);
      }
    }
  }
  __syncthreads();
  double Mg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Mg[i] = uniform_f64(sh->M[i]);
  const bool sep_cam = Mg[1] == 0.0 && Mg[3] == 0.0 && Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform
  // ---- active tiles of the band: list + the eight row words of every active tile, compacted in place ----
  const int ntiles = ntx * trows, per = (ntiles + NWAVE - 1) / NWAVE;
  const int tbeg = wave * per, tend = min(tbeg + per, ntiles);
  unsigned long long bal[4];
  unsigned wrd[4][8];
  int wcount = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = tbeg + k * 64 + lane;
    unsigned any = 0;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) wrd[k][rr] = 0u;
    if (t < tend) {
      const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * ntx;
      const unsigned* bw = bits + (ty * 8) * ntx + tx;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const unsigned w = bw[rr * ntx];
        any |= w;
        wrd[k][rr] = w;
      }
    }
    bal[k] = __ballot(any != 0);
    wcount += __popcll(bal[k]);
  }
  if (lane == 0) sh->scan[wave] = (unsigned)wcount;
  __syncthreads();   // (every wave has read its row words: the image region can be overwritten)
  int base = 0, nactive = 0;
  for (int w = 0; w < NWAVE; ++w) {
    const int c = (int)sh->scan[w];
    if (w < wave) base += c;
    nactive += c;
  }
  {
    int off = base;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((bal[k] >> lane) & 1ull) {
        const int t = tbeg + k * 64 + lane;
        const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * ntx;
        const int idx = off + __popcll(bal[k] & ((1ull << lane) - 1ull));
        list[idx] = (unsigned short)((ty << 8) | tx);
        uint4* e = reinterpret_cast<uint4*>(bits) + 2 * idx;
        e[0] = make_uint4(wrd[k][0], wrd[k][1], wrd[k][2], wrd[k][3]);
        e[1] = make_uint4(wrd[k][4], wrd[k][5], wrd[k][6], wrd[k][7]);
      }
      off += __popcll(bal[k]);
    }
  }
  unsigned* col = bits + nactive * 8;   // behind the entries: rows_plan sized the region for a band with every tile active
  for (int u = tid; u < W; u += NT) { col[u] = 0xffffffffu; col[W + u] = 0u; }
  __syncthreads();
  // ---- the single pass over the band ----
  double sacc[5] = {0, 0, 0, 0, 0}, yx[2] = {INFINITY, -INFINITY};
  unsigned unsafe = 0u;
  if (sep_cam) {
    if (rows == trows * 8) sweep_sep<false>(p, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe, row0);   // uniform
    else {   // the frame's last, partial tile row is in this band: the walk that loads row by row there (band-local frame height)
      FitParams pb = p;
      pb.H = rows;
      sweep_sep<true>(pb, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe, row0);
    }
  }
  {
    const double r0 = wave_sum(sacc[0]), r1 = wave_sum(sacc[1]), r2 = wave_sum(sacc[2]), r3 = wave_sum(sacc[3]), r4 = wave_sum(sacc[4]);
    const double ylo = wave_min(yx[0]), yhi = wave_max(yx[1]);
    const int rn = wave_sum_i(nmask);
    const bool bad = __ballot(unsafe >= 0x7f800000u) != 0ull;
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; pp[5] = ylo; pp[6] = yhi;
      sh->nmask[wave] = rn;
      sh->cnt[wave] = bad ? 1 : 0;
    }
  }
  __syncthreads();   // (also: every ds_min / ds_max of the pass has landed)
  // the band's partial record and per-column ranges -> workspace.  One-launch form: 16-byte write-through stores - the workgroup
  // that merges the instance may sit on another XCD (another L2); two-launch form: plain stores (the kernel boundary publishes them)
  const bool xch = ra.arrive != nullptr;   // uniform
  if (tid == 0) {
    double t[7] = {0, 0, 0, 0, 0, INFINITY, -INFINITY};
    int nm = 0, bad = sep_cam ? 0 : 1;
    for (int w = 0; w < NWAVE; ++w) {   // fixed order: reproducible
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] += sh->part[w][k];
      t[5] = fmin(t[5], sh->part[w][5]); t[6] = fmax(t[6], sh->part[w][6]);
      nm += sh->nmask[w];
      bad |= sh->cnt[w];
    }
    // through LDS (sh->part is free again: every wave's partials have been read): [0..6] sums and y extent | mask pixels | flag | - |
    // M[9] (the merge takes the camera from band 0: no second inversion) | -
    double* z = &sh->part[0][0];
#pragma unroll
    for (int k = 0; k < 7; ++k) z[k] = t[k];
    z[7] = (double)nm + (bad ? ROWS_FLAG : 0.0); z[8] = 0.0; z[9] = 0.0;   // (mask pixels < 2^28: the sum is exact)
#pragma unroll
    for (int k = 0; k < 9; ++k) z[10 + k] = sh->M[k];
    z[19] = 0.0;
  }
  __syncthreads();
  {
    double* q = ra.part + (long long)v * ROWS_PART_D;
    unsigned* gcol = ra.col + (long long)v * 2 * W;
    const uint4* zq = reinterpret_cast<const uint4*>(&sh->part[0][0]);
    const uint4* cq = reinterpret_cast<const uint4*>(col);
    if (xch) {
      if (tid < ROWS_PART_D / 2) st16_through(reinterpret_cast<uint4*>(q) + tid, zq[tid]);
      for (int g = tid; g < W / 2; g += NT) st16_through(reinterpret_cast<uint4*>(gcol) + g, cq[g]);
    } else {
      if (tid < ROWS_PART_D / 2) reinterpret_cast<uint4*>(q)[tid] = zq[tid];
      for (int g = tid; g < W / 2; g += NT) reinterpret_cast<uint4*>(gcol)[g] = cq[g];
    }
  }
  if (ra.arrive == nullptr) return;   // uniform: the two-launch form - merge_rows_kernel follows on the stream
  // ---- one launch: the band that arrives last merges the instance (nobody waits for anybody) ----
  band_release();      // every thread: its own stores have been acknowledged ...
  __syncthreads();     // ... all of them, before thread 0 announces the band
  if (tid == 0) sh->scan[0] = tagged_arrive_many(ra.arrive + inst, ra.tag) == (unsigned)ra.nb ? 1u : 0u;
  __syncthreads();
  if (!sh->scan[0]) return;   // uniform
  band_acquire();      // the other bands' data is loaded after their arrivals were seen
  rows_merge<true>(sh, p, ra, inst, img, col, tid, wave, lane);
}

// one workgroup per instance: partials of its bands -> status, axis, extents, record (the instance engine's stages)
__global__ __launch_bounds__(NT, NT / 64) void merge_rows_kernel(const FitParams p, const RowsArgs ra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Shared* sh = reinterpret_cast<Shared*>(smem);
  unsigned* mcol = reinterpret_cast<unsigned*>(smem + sizeof(Shared));   // [2 W]: the bands' per-column ranges merged
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int inst = (int)blockIdx.x;
  const int img = p.image_index ? p.image_index[inst] : inst;
  for (int u = tid; u < p.W; u += NT) { mcol[u] = 0xffffffffu; mcol[p.W + u] = 0u; }   // the identities: every band is merged in
  rows_merge<false>(sh, p, ra, inst, img, mcol, tid, wave, lane);
}

// u8 planes, 16-byte aligned, full-mask mode, no ground array, B <= ROWS_MAX_B: LA3D_ENGINE=rows / opt_engine pins it, by default
// it takes the batches up to config().rows_maxb
inline bool rows_eligible(const FitParams& p, bool vec, bool sample, RowsArgs* ra) {
  const int e = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;
  if (e != LA3D_ENGINE_DEFAULT && e != LA3D_ENGINE_ROWS && e != LA3D_ENGINE_ROWS2) return false;
  if (!vec || sample || p.mask == nullptr || p.ground != nullptr || p.sep_off || p.filter_boundary >= 0) return false;
  if (!rows_plan(p.B, p.H, p.W, ra)) return false;
  return e == LA3D_ENGINE_ROWS || e == LA3D_ENGINE_ROWS2 || p.B <= config().rows_maxb;
}

int launch_fit_rows(const FitParams& p_in, RowsArgs ra, hipStream_t s, void* workspace) {
  FitParams p = p_in;
  p.ntx = p.W / 32; p.nty = p.H / 8;
  p.rcp_ntx = 1.0f / (float)p.ntx;
  unsigned char* w = static_cast<unsigned char*>(workspace);
  const size_t part_bytes = ((size_t)p.B * ra.nb * ROWS_PART_D * 8 + 255) & ~(size_t)255;
  const size_t col_bytes = ((size_t)p.B * ra.nb * 2 * p.W * 4 + 255) & ~(size_t)255;
  ra.part = reinterpret_cast<double*>(w);
  ra.col = reinterpret_cast<unsigned*>(w + part_bytes);
  // One launch (round 6): the last band to arrive merges its instance.  Two launches - fit_rows_kernel, then merge_rows_kernel -
  // when pinned (LA3D_ENGINE_ROWS2 / LA3D_ROWS_FUSED=0) and for a call captured into a HIP graph (it would replay with the same tag).
  const int e = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;
  bool fused = e != LA3D_ENGINE_ROWS2 && config().rows_fused != 0;
  if (fused) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) fused = false;
    (void)hipGetLastError();
  }
  ra.arrive = nullptr; ra.tag = 0;
  if (fused) {
    ra.arrive = reinterpret_cast<unsigned long long*>(w + part_bytes + col_bytes);
    const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    ra.tag = (((t * 0x9E3779B97F4A7C15ull) >> 13) ^ (unsigned long long)reinterpret_cast<uintptr_t>(workspace)) & 0xffffffffffffull;
    if (ra.tag == 0) ra.tag = 1;
  }
  const size_t lds = (size_t)ra.bits_bytes + sizeof(Shared) + (size_t)p.ntx * ra.trows * 2 + 16;
  allow_big_lds(reinterpret_cast<const void*>(fit_rows_kernel));
  hipLaunchKernelGGL(fit_rows_kernel, dim3(p.B * ra.nb), dim3(NT), lds, s, p, ra);
  const int rc = check_launch("fit_rows_kernel");
  if (rc != LA3D_SUCCESS || fused) return rc;
  hipLaunchKernelGGL(merge_rows_kernel, dim3(p.B), dim3(NT), sizeof(Shared) + (size_t)2 * p.W * 4, s, p, ra);
  return check_launch("merge_rows_kernel");
}

}  // namespace

namespace la3d {
bool rows_fit_if_eligible(const FitParams& p, bool vec, bool sample, hipStream_t s, void* workspace, int* rc) {
  RowsArgs ra;
  if (!rows_eligible(p, vec, sample, &ra)) return false;
  *rc = launch_fit_rows(p, ra, s, workspace);
  return true;
}
size_t rows_workspace_bytes(int B, int H, int W) { return ::rows_workspace_bytes_impl(B, H, W); }
}  // namespace la3d
