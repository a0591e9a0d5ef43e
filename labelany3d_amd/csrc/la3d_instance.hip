// la3d_instance.hip - the INSTANCE ENGINE of la3d_fit_instances: one 512-thread workgroup per instance (mask -> bit image in LDS ->
// active-tile list -> separable single pass, or pass A / axis / pass B -> record), its launch order helper kernel and its launcher.
// Design and measurements: DESIGN.md section 4.1; the walks and stages it is built from: la3d_walks.hpp, la3d_stages.hpp.
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
// instance engine: one workgroup per instance
// ------------------------------------------------------------------------------------------
// SRC: where the mask comes from - 0 = u8 plane, 1 = COCO run lengths, 2 = polygon parts (both decoded into the LDS bit image)
template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED, int SRC>
__global__ __launch_bounds__(NT, NT / 64) void fit_instances_kernel(const FitParams p) {
  constexpr bool RLE = SRC == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  Shared* sh = reinterpret_cast<Shared*>(smem + p.mask_lds_bytes);
  unsigned* prefix = reinterpret_cast<unsigned*>(smem + p.mask_lds_bytes + sizeof(Shared));  // SAMPLE only
  // TILED only: compacted list of active tile ids
  unsigned short* list = reinterpret_cast<unsigned short*>(smem + p.mask_lds_bytes + sizeof(Shared));

  // (builds that carry the separable pass take the lane from the execution mask, not from threadIdx.x - the workgroup's waves are
  // full -, and rebuild the thread index where it is used: neither then keeps the kernel's input register alive across the passes)
  constexpr bool REBUILD_TID = TILED && !SAMPLE;
  const int tid_in = threadIdx.x;
  const int lane = REBUILD_TID ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (tid_in & 63);
  const int wave = __builtin_amdgcn_readfirstlane(tid_in >> 6);  // wave-uniform: lives in an SGPR
  const int tid = REBUILD_TID ? ((wave << 6) | lane) : tid_in;
#ifdef LA3D_TIMELINE
  const unsigned long long t_entry = wall_clock64();   // before the first memory access of the workgroup
#endif
  // (measured, profiles/timeline.py: all workgroups of a launch ENTER within 0.7 us, but VMEM issue is arbitrated by age, so the
  // youngest of the four workgroups of a CU gets its first load - this perm entry - back only when an older one has finished
  // its mask stream, ~25 us in; warming the table through L1 does not help, and s_setprio by dispatch group only moves the
  // starvation to the oldest group, which holds the largest instances: DESIGN.md section 5.2)
  // (self-estimating launch; order_self == 2 is the test mode of the fallback: every seventh workgroup keeps its key to itself)
  // (batches above one resident set: the workgroups of the FIRST set - the only ones certain to run without waiting for anybody -
  // estimate instances b, b + R, b + 2R, ...)
  if (!SAMPLE && p.order_self && (int)blockIdx.x < p.order_resident && !(p.order_self == 2 && blockIdx.x % 7 == 3)) {
    for (int ie = (int)blockIdx.x; ie < p.B; ie += p.order_resident) {   // uniform
      if (ie != (int)blockIdx.x) __syncthreads();   // (the block totals of the previous estimate have been read)
      estimate_publish_wg(p, ie, sh, tid, wave, lane);
    }
  }
  const int inst = p.order_nch > 0 ? order_select(p, (int)blockIdx.x, sh, wave, lane) : xcd_remap(blockIdx.x, p.B);
  if (tid == 0) { sh->order_inst = inst; sh->sep_bad = 0; }   // (the instance is re-read after the mask stage, see below)
  const int img = p.image_index ? p.image_index[inst] : inst;
  const int HW = p.HW;
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
  const unsigned char* mpl = p.mask ? p.mask + (long long)inst * HW : nullptr;

  if (tid == NT - 1) {
    // per-instance geometry (reference src/util.py:56, src/util_3dbox.py:128-134), one lane, overlapped with the
    // mask stream of everyone else: Kinv, Rg, M = Rg^T Kinv
    double Kinv[9], Rg[9];
    inv3_camera(p.K + (long long)img * p.k_stride, Kinv);
    sh->bad_ground = ground_rotation(p.ground ? p.ground + (long long)inst * 4 : nullptr, Rg);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) sh->M[i * 3 + j] = Rg[i] * Kinv[j] + Rg[3 + i] * Kinv[3 + j] + Rg[6 + i] * Kinv[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) sh->Rg[i] = Rg[i];
  }

#ifdef LA3D_TIMELINE
  // measurement build only (profiles/timeline.py): wall-clock stamps (100 MHz) per workgroup at the phase boundaries,
  // into the workspace behind the launch-order arrays
  double* tl = p.geo + 1024 + (long long)inst * 16;
#define LA3D_STAMP(k) do { if (tid == 0) tl[k] = (double)wall_clock64(); } while (0)
  if (tid == 0) { tl[7] = (double)blockIdx.x; tl[8] = (double)t_entry; sh->tl = tl; }
#else
#define LA3D_STAMP(k) do { } while (0)
#endif
  LA3D_STAMP(0);
  if (!SAMPLE && p.stagger_ticks > 0 && p.order_nch > 0 && blockIdx.x < 1024) {
    // Plain build, u8 planes, size-ordered launch (round 4): the four groups of 256 workgroups that fill the chip start one
    // stagger period apart, the group of the 256 LARGEST instances first (group g of the launch order = blocks [256 g, 256 g + 256)).
    // Every instance streams the same H*W mask bytes whatever its size; started together, the 1024 streams share the bandwidth and
    // nobody's passes begin before ~50 us.  Staggered, the large instances stream at four times the share and are in their (long)
    // passes - VALU work - while the smaller ones, which have the slack, stream.  Measured (helper-kernel build), us per call, without / with 10 us
    // (profiles/r04/r04_stagger.txt): config-2 masks B = 448 / 640 / 1024 / 1280 / 2048: 71.6 / 80.6 / 103.6 / 125.6 / 176.2 ->
    // 66.9 / 74.8 / 99.7 / 118.2 / 170.0; config-5 masks B = 512 / 1024 / 2048: 75.1 / 91.4 / 144.1 -> 69.9 / 83.2 / 139.6; neutral
    // from 4096 up.  Speed only: records do not depend on it.  (Run-length / polygon input has no stream to spread: slower there.)
    const unsigned long long t0 = wall_clock64();
    // (delays 0 / 0.81 / 1.81 / 2.94 periods: the later - smaller - groups wait a little longer each; against equal steps of one
    // period: config 2 at B = 1024 95.1 -> 94.0 us, at 1536 125.6 -> 124.5, config 5 at 1024 equal; equal steps of 10 us are as good on
    // config 2 and 2.7 us worse on config 5 - profiles/r04/r04_stagger.txt, run 5)
    const unsigned g = blockIdx.x >> 8;
    const unsigned long long w = (unsigned long long)p.stagger_ticks * (g == 1 ? 13u : (g == 2 ? 29u : (g == 3 ? 47u : 0u))) / 16u;
    while (wall_clock64() - t0 < w) __builtin_amdgcn_s_sleep(32);
  }
  // reference-subsample mode: this thread's drawn index, requested before the mask stream so that it is not a dependent
  // round trip afterwards (unused when the mask turns out to have <= 500 pixels)
  int my_draw = 0;
  if (SAMPLE && tid < LA3D_NSAMPLE) my_draw = p.sample_idx[(long long)inst * LA3D_NSAMPLE + tid];
  // ---- phase 0: u8 mask plane -> bit image in LDS --------------------------------------
  int nmask = 0;
  if (LDSMASK && RLE) {
    // masks arrive as COCO run lengths: decode straight into the LDS bit image — no u8 plane is ever read
    const long long o0 = p.rle_offsets[inst];
    // (the block totals of the column scan borrow the LDS of the tile list, which is built afterwards)
    nmask = rle_to_bits<NT>(p.rle_counts + o0, (int)(p.rle_offsets[inst + 1] - o0), bits, p.nwords, p.H, p.W, sh->scan, tid,
                            reinterpret_cast<unsigned*>(smem + p.mask_lds_bytes + sizeof(Shared)), TILED ? p.list_cap / 2 : 0, p.frame_w);
  } else if (LDSMASK && SRC == 2) {
    // masks arrive as polygon parts (the reference's create_boolean_mask_from_polygon, src/util.py:386-400): rasterised with
    // cv2.fillPoly's rule straight into the LDS bit image; the side stage borrows the space of the tile list
    nmask = poly_to_bits<NT>(p.poly_xy, p.poly_ring_off, p.poly_inst_rings[inst], p.poly_inst_rings[inst + 1],
                             reinterpret_cast<PolySide*>(smem + p.mask_lds_bytes + sizeof(Shared)), sh->scan, bits, p.nwords, p.H,
                             p.W, tid, p.frame_w);
  } else if (LDSMASK) {
    unsigned short* b16 = reinterpret_cast<unsigned short*>(bits);
    const int ngroups = (HW + 15) >> 4;
    if (VEC) {
      const u32x4* m4 = reinterpret_cast<const u32x4*>(mpl);
      // Optimistic form: np.bool_ planes (the reference's layout, src/util.py:367,382) hold only 0 and 1, and then the
      // 16-bit pattern of a 16-byte group is four dot products (sum byte_j * 2^j) - 11 VALU instructions per group instead
      // of 27 for the general non-zero test.  Every word is ORed into `seen`; a byte above 1 anywhere in the plane sends the
      // whole workgroup through the general loop below (same bit image either way).
      constexpr int P0U = 4;   // 16-byte loads in flight per lane
      unsigned seen = 0;
#pragma unroll P0U
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
      if (general) {   // uniform: some byte is neither 0 nor 1 (e.g. 255-valued masks)
        nmask = 0;
#pragma unroll 4
        for (int g = tid; g < ngroups; g += NT) {
          const u32x4 w = m4[g];
          const unsigned pat = nz16(w.x, w.y, w.z, w.w);
          b16[g] = (unsigned short)pat;
          nmask += __popc(pat);
        }
      }
    } else {
      for (int g = tid; g < ngroups; g += NT) {
        unsigned pat = 0;
        for (int k = 0; k < 16; ++k) {
          const int i = g * 16 + k;
          if (i < HW && mpl[i]) pat |= 1u << k;
        }
        b16[g] = (unsigned short)pat;
        nmask += __popc(pat);
      }
    }
    if ((ngroups & 1) && tid == 0) b16[ngroups] = 0;  // upper half of the last 32-bit word
  }
  __syncthreads();
  LA3D_STAMP(1);
  // (from here on the instance index is re-read from LDS: live across the decode stage it costs the polygon build a spilled
  // register pair)
  const int inst_p = __builtin_amdgcn_readfirstlane(sh->order_inst);
  // (and from here on the thread index is rebuilt where it is used - one v_lshl_or from the wave's scalar index and the lane -
  // instead of staying live from kernel entry: with the separable pass in the kernel the allocator otherwise spills it to scratch,
  // and a kernel with scratch launches its waves visibly slower: round 5, B = 8192 590 -> 670 us)
  const int tid_plain = tid;
#define tid (REBUILD_TID ? tid_here(wave, lane) : tid_plain)
  if (SRC != 0 && LDSMASK && p.filter_boundary >= 0) {   // uniform
    // the reference's instance filter (src/util.py:375) on the bit image just built: a dropped instance costs no passes
    int st4[4];
    bits_filter_stats<NT>(bits, p.H, p.frame_w, p.filter_boundary, reinterpret_cast<int*>(sh->part), tid, st4, p.W);   // (frame_w == W unless the rows are padded)
    if (p.filter_stats && tid < 4) (p.filter_stats + (long long)inst_p * 4)[tid] = st4[tid];   // (uniform base: scalar address arithmetic)
    const int height = SRC == 1 ? st4[1] : st4[2];   // run lengths: rows holding a pixel (:368-369); polygons: last - first + 1 (:328-335)
    const bool keep = 16 * height > p.H && st4[3] < p.filter_max_edge && st4[0] >= p.filter_min_area;   // height / H > 0.0625
    if (!keep) {
      if (tid == 0) {
        if (p.aux) {
          double* a = p.aux + (long long)inst_p * LA3D_AUX;
          a[0] = NAN; a[1] = 0.0; a[2] = (double)st4[0]; a[3] = NAN;
        }
        p.status[inst_p] = LA3D_BOX_FILTERED;
        write_nan_box(p.out + (long long)inst_p * LA3D_REC);
        if (p.proj) { for (int j = 0; j < 8; ++j) p.proj[(long long)inst_p * 8 + j] = NAN; }
      }
      return;
    }
  }
  double Mg[9];   // wave-uniform: moved to SGPRs
#pragma unroll
  for (int i = 0; i < 9; ++i) Mg[i] = uniform_f64(sh->M[i]);
  LA3D_STAMP(13);

  // reference-subsample mode: the reference subsamples when in_pc.shape[0] > 500 (src/util_3dbox.py:123) - needs N first.
  // Sampled instances need no tile list (their 500 points are picked through the block prefix, which shares its LDS).
  bool sampled = false;
  int ntot = 0;
  if (SAMPLE) {
    const int wsum = wave_sum_i(nmask);
    if (lane == 0) sh->nmask[wave] = wsum;
    __syncthreads();
    for (int w = 0; w < NWAVE; ++w) ntot += sh->nmask[w];
    sampled = ntot > LA3D_NSAMPLE;
  }

  // ---- active-tile list (deterministic two-pass compaction: count, prefix, write) ----------------
  int nactive = 0, nfull = 0;
  // plain build: the bit image is compacted to the active tiles (eight row words per list entry) and the LDS that frees keeps
  // depth tiles between the passes (sweep_tiled)
  constexpr bool LK = TILED && !SAMPLE;
  int compact = 0;
  // separable single pass (sweep_sep): no ground rotation, no skew - x ray by column, y ray by row, z = depth
  bool sep = false;
  const bool sep_cam = LK && !p.sep_off && Mg[1] == 0.0 && Mg[3] == 0.0 && Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform
  if (TILED && !sampled) {
    const int ntiles = p.ntx * p.nty, per = p.tiles_per_wave;
    const int tbeg = wave * per, tend = min(tbeg + per, ntiles);
    int base = 0;
    if (per <= 256) {
      // one pass: a wave looks at up to 4 x 64 tiles; the ballots stay in SGPRs across the barrier, the eight row words
      // of a tile are read back to back (rows past the frame re-read the last one), no integer division
      unsigned long long bal[4];
      unsigned wrd[LK ? 4 : 1][8];
      int wcount = 0, fcount = 0;
      unsigned fl = 0;   // bit k: this lane's k-th tile lies completely inside the mask (all eight row words all ones)
      if ((p.H & 7) == 0) {   // uniform: every tile row is complete (the common frame heights) - no row clamp, no select per word
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tbeg + k * 64 + lane;
          unsigned any = 0, all = 0xffffffffu;
          if (t < tend) {
            const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;  // exact: see fit_dispatch
            const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const unsigned w = bw[rr * p.ntx];
              any |= w;
              if constexpr (LK) { wrd[k][rr] = w; all &= w; }
            }
          }
          bal[k] = __ballot(any != 0);
          wcount += __popcll(bal[k]);
          if constexpr (LK) {
            const bool f = t < tend && all == 0xffffffffu;
            fcount += __popcll(__ballot(f));
            fl |= (f ? 1u : 0u) << k;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tbeg + k * 64 + lane;
          unsigned any = 0, all = 0xffffffffu;
          if (t < tend) {
            const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;  // exact: see fit_dispatch
            const int rmax = p.H - 1 - ty * 8;                                        // >= 0
            const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const unsigned w = bw[min(rr, rmax) * p.ntx];
              any |= w;
              if constexpr (LK) { wrd[k][rr] = rr <= rmax ? w : 0u; all &= wrd[k][rr]; }
            }
          }
          bal[k] = __ballot(any != 0);
          wcount += __popcll(bal[k]);
          if constexpr (LK) {
            const bool f = t < tend && all == 0xffffffffu;   // (a tile row past the frame is stored as zeros: never "full")
            fcount += __popcll(__ballot(f));
            fl |= (f ? 1u : 0u) << k;
          }
        }
      }
      if (lane == 0) sh->scan[wave] = (unsigned)wcount | ((unsigned)fcount << 16);   // (at most 256 tiles per wave)
      __syncthreads();
      LA3D_STAMP(14);
      // Round 6: the list holds the tiles that lie completely inside the mask FIRST ([0, nfull): the separable pass walks them with
      // a body that needs no mask bits), the others behind them - each class in tile order.  Any order of the list gives a valid
      // walk; the order only decides how the fp64 partial sums are grouped.
      int fbase = 0;
      for (int w = 0; w < NWAVE; ++w) {
        const int c = (int)(sh->scan[w] & 0xffffu), f = (int)(sh->scan[w] >> 16);
        if (w < wave) { base += c - f; fbase += f; }
        nactive += c;
        nfull += f;
      }
      if (nactive > p.list_cap) {
        nactive = -1;  // uniform: every thread sees the same total
        nfull = 0;
      } else {
        // (with pass-B culling the compact image also holds the survivor list / the depth ranges behind the entries)
        // every wave read its row words before the barrier above: the image region can be overwritten in place
        if constexpr (LK) compact = (nactive * 32 + cull_rng_words(nactive) * 4 <= p.mask_lds_bytes) ? 1 : 0;   // uniform
        int foff = fbase, poff = nfull + base;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool isf = LK && ((fl >> k) & 1u);
          const unsigned long long bf = LK ? __ballot(isf) : 0ull, bp = bal[k] & ~bf, lt = (1ull << lane) - 1ull;
          if ((bal[k] >> lane) & 1ull) {
            const int t = tbeg + k * 64 + lane;
            const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;
            const int idx = isf ? foff + __popcll(bf & lt) : poff + __popcll(bp & lt);
            list[idx] = (unsigned short)((ty << 8) | tx);
            if constexpr (LK) if (compact) {   // uniform
              uint4* e = reinterpret_cast<uint4*>(bits) + 2 * idx;
              e[0] = make_uint4(wrd[k][0], wrd[k][1], wrd[k][2], wrd[k][3]);
              e[1] = make_uint4(wrd[k][4], wrd[k][5], wrd[k][6], wrd[k][7]);
            }
          }
          foff += __popcll(bf); poff += __popcll(bp);
        }
        if constexpr (LK) if (compact) {   // uniform
          if (sep_cam && nactive * 32 + sep_col_words(p.W) * 4 <= p.mask_lds_bytes) {   // uniform
            sep = true;   // per-column depth range behind the entries: [min | max], the identities of unsigned min / max
            unsigned* col = bits + nactive * 8;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (int u = tid; u < p.W; u += NT) { col[u] = 0xffffffffu; col[p.W + u] = 0u; }
          }
        }
      }
    } else {
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        int wcount = 0;
        for (int t0 = tbeg; t0 < tend; t0 += 64) {   // wave-uniform trip count
          const int t = t0 + lane;
          unsigned any = 0, packed = 0;
          if (t < tend) {
            const int ty = t / p.ntx, tx = t - ty * p.ntx;
            const int rows = min(8, p.H - ty * 8);
            const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
            for (int rr = 0; rr < rows; ++rr) any |= bw[rr * p.ntx];
            packed = ((unsigned)ty << 8) | (unsigned)tx;
          }
          const unsigned long long bal = __ballot(any != 0);
          if (pass == 1 && any) list[base + wcount + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)packed;
          wcount += __popcll(bal);
        }
        if (pass == 0) {
          if (lane == 0) sh->scan[wave] = (unsigned)wcount;
          __syncthreads();
          for (int w = 0; w < NWAVE; ++w) {
            const int c = (int)sh->scan[w];
            if (w < wave) base += c;
            nactive += c;
          }
          if (nactive > p.list_cap) { nactive = -1; break; }  // uniform: every thread sees the same total
        }
      }
    }
    LA3D_STAMP(15);
    __syncthreads();
  }

  // ---- separable single pass: moments, y extent and per-column depth ranges in ONE walk; x / z extents from the ranges -------
  if constexpr (LK) {
    if (sep) {   // uniform
      LA3D_STAMP(2);
      unsigned* col = bits + nactive * 8;
      double sacc[5] = {0, 0, 0, 0, 0}, yx[2] = {INFINITY, -INFINITY};
      unsigned unsafe = 0u;
      if (p.H & 7) sweep_sep<true>(p, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe, 0, nfull);   // uniform
      else sweep_sep<false>(p, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe, 0, nfull);
      if (__ballot(unsafe >= 0x7f800000u) != 0ull && lane == 0) sh->sep_bad = 1;   // NaN / inf / negative depth under the mask
      // (the wave's y extent waits in scalar registers while the axis is computed: four vector registers fewer across that stage)
      const double ylo_w = uniform_f64(wave_min(yx[0])), yhi_w = uniform_f64(wave_max(yx[1]));
      LA3D_STAMP(3);
      stage_moments_to_axis(sh, p, inst_p, sacc, nmask, nmask, tid, wave, lane, true);
      LA3D_STAMP(4);
      if (!(sh->redo || sh->sep_bad)) {   // uniform
        if (sh->st != LA3D_BOX_OK) return;
        double N0[3], N2[3], ext[6];
        yaw_rows(sh, Mg, N0, N2);
        sep_col_extents(col, p.W, N0, N2, tid, ext);
        ext[2] = ylo_w; ext[3] = yhi_w;
        LA3D_STAMP(5);
        stage_extents_to_box(sh, p, inst_p, ext, tid, wave, lane);
        stage_status_aux(sh, p, inst_p, tid);
        LA3D_STAMP(6);
        return;
      }
      __syncthreads();   // everyone has read redo / sep_bad and the partials: on to the general two-pass path
    }
  }

  // pass-B tile culling (see cull_plan): instances with enough active tiles record every tile's depth range in pass A
  bool cull = false;
  int rng_words = 0;
  if constexpr (LK) {
    // (every compact instance reserves the area: pass B always walks a survivor list - the identity when nothing is culled)
    if (compact) {   // uniform
      rng_words = cull_rng_words(nactive);
      cull = nactive >= p.cull_min && nactive <= CULL_MAXT;
      if (!cull) {
        unsigned short* surv = reinterpret_cast<unsigned short*>(bits + nactive * 8);
        // (vectorised, the index vector tid + {0, 512, 1024, 1536} becomes a 128-bit register tuple that lives from kernel entry: a spill)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int t = tid; t < nactive; t += NT) surv[t] = (unsigned short)t;   // (visible after the barriers of the axis stage)
      }
    }
  }

  LA3D_STAMP(2);
  // ---- pass A: moments ------------------------------------------------------------------
  double acc[5] = {0, 0, 0, 0, 0};
  int cnt = 0;
  // sampled-point state (SAMPLE only): the point of this thread in the ground-aligned frame
  double px = 0, py = 0, pz = 0;
  bool pok = false;

  if (SAMPLE) {
    if (sampled) {
      // exclusive prefix of the popcounts of 32-word blocks (1024 px): thread t owns block t.  One word of LDS per block
      // keeps the workgroup at a quarter of the CU's LDS (four workgroups per CU, like the full-mask build).
      const int nblk = (p.nwords + 31) >> 5;
      unsigned run0 = 0;   // blocks of earlier rounds (frames above NT * 1024 px)
      for (int b0 = 0; b0 < nblk; b0 += NT) {
        const int blk = b0 + tid;
        unsigned local = 0;
        if (blk < nblk) {
          const int w0 = blk << 5, wn = min(32, p.nwords - w0);
          if (wn == 32) {
            const uint4* q = reinterpret_cast<const uint4*>(bits + w0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const uint4 v = q[i]; local += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
          } else {
            for (int i = 0; i < wn; ++i) local += __popc(bits[w0 + i]);
          }
        }
        unsigned incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        if (lane == 63) sh->scan[wave] = incl;
        __syncthreads();
        unsigned base = run0, tot = 0;
        for (int w = 0; w < NWAVE; ++w) { const unsigned c = sh->scan[w]; if (w < wave) base += c; tot += c; }
        if (blk < nblk) prefix[blk] = base + incl - local;
        run0 += tot;
        __syncthreads();
      }
      if (tid < LA3D_NSAMPLE) {
        int r = my_draw;
        r = r < 0 ? 0 : (r >= ntot ? ntot - 1 : r);
        int lo = 0, hi = nblk - 1;
        while (lo < hi) {  // last block whose exclusive prefix is <= r
          const int mid = (lo + hi + 1) >> 1;
          if (prefix[mid] <= (unsigned)r) lo = mid; else hi = mid - 1;
        }
        int k = r - (int)prefix[lo];          // rank inside the block
        lo <<= 5;
        unsigned w = 0;
        if (lo + 32 <= p.nwords) {
          // the word of the block that holds set bit k: all 32 words read at once, then a register scan
          uint4 q[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) q[i] = reinterpret_cast<const uint4*>(bits + lo)[i];
          int sel = 0;
          bool found = false;
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            const unsigned wi = (i & 3) == 0 ? q[i >> 2].x : (i & 3) == 1 ? q[i >> 2].y : (i & 3) == 2 ? q[i >> 2].z : q[i >> 2].w;
            const int c = __popc(wi);
            const bool here = !found && k < c;
            if (here) { w = wi; sel = i; }
            found = found || here;
            if (!found) k -= c;
          }
          lo += sel;
        } else {
          const int wend = p.nwords - 1;
          for (; lo < wend; ++lo) {
            const int c = __popc(bits[lo]);
            if (k < c) break;
            k -= c;
          }
          w = bits[lo];
        }
        for (; k > 0; --k) w &= w - 1;  // drop k lowest set bits
        const unsigned i = (unsigned)lo * 32u + (unsigned)(__ffs((int)w) - 1);
        const float df = dpl[i];
        unsigned u, v;
        pix_uv(i, p.W, p.rcpW, &u, &v);
        const double ud = (double)u, vd = (double)v;
        pok = finite_f32(df);
        const double d = pok ? (double)df : 0.0;
        px = d * fma(Mg[0], ud, fma(Mg[1], vd, Mg[2]));
        py = d * fma(Mg[3], ud, fma(Mg[4], vd, Mg[5]));
        pz = d * fma(Mg[6], ud, fma(Mg[7], vd, Mg[8]));
        if (pok) {
          acc[0] = px; acc[1] = pz; acc[2] = px * px; acc[3] = px * pz; acc[4] = pz * pz;
          cnt = 1;
        }
      }
    }
  }
  // TILED: optimistic pass first (no per-pixel finite test); a non-finite masked depth shows up as non-finite sums and
  // the workgroup falls back to the checked passes.  Same records either way.
  bool checked = !TILED;
  if (!sampled) {
    if (TILED) {
      // (the un-grounded, skew-free forms of the pixel math where they apply: same records, fewer instructions - quad_math)
      // (not in the subsample build, which walks tiles only for its small masks and has no registers to spare)
      constexpr bool SP = !SAMPLE;
      const bool specA = SP && Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform
      if (LK && cull) {
        if (specA) sweep_tiled<0, false, true, SP>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
        else sweep_tiled<0, false, true>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
      } else {
        if (specA) sweep_tiled<0, false, false, SP>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
        else sweep_tiled<0, false>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
      }
      cnt = nmask;   // the optimistic pass does not count: with every masked depth finite, valid pixels = mask pixels
    }
    else sweep<VEC, LDSMASK, 0>(p, dpl, mpl, bits, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, &nmask);
  }

  LA3D_STAMP(3);
  stage_moments_to_axis(sh, p, inst_p, acc, cnt, nmask, tid, wave, lane, true);
  if (sh->redo) {  // uniform: non-finite sums (the optimistic tiled pass) or ill-conditioned ones (axis_from_sums) - the checked pass
                   // about the pivot the stage left (zero unless the sums were ill-conditioned)
    __syncthreads();        // everyone has read sh->redo and the partials before they are rewritten (the pivot stays where it is)
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = 0;
    cnt = 0;
    if (sampled) {
      if (pok) {
        const double x = px - pivot_ptr(sh)[0], z = pz - pivot_ptr(sh)[1];
        acc[0] = x; acc[1] = z; acc[2] = x * x; acc[3] = x * z; acc[4] = z * z;
        cnt = 1;
      }
    } else if (TILED) {
      if (sh->redo == 2) {   // ill-conditioned sums: the moments about the pivot; the optimistic pass's tile ranges and pass B stand
        pivot_pass(p, dpl, bits, list, nactive, Mg, Mg + 6, wave, lane, compact, pivot_ptr(sh), acc, &cnt);
      } else {
        checked = true;
        if (LK && cull) sweep_tiled<0, true, true>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
        else sweep_tiled<0, true>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words, -1, nfull);
      }
    } else {
      double piv[2];
      get_pivot(sh, piv);
      sweep<VEC, LDSMASK, 0, true>(p, dpl, mpl, bits, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, &nmask, piv);
    }
    stage_moments_to_axis(sh, p, inst_p, acc, cnt, nmask, tid, wave, lane, false);
  }
  LA3D_STAMP(4);
  if (sh->st != LA3D_BOX_OK) return;

  // ---- pass B: extents along the principal axes -----------------------------------------
  double ext[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};  // x, y, z : lo, hi
  if (sampled) {
    if (pok) {  // exactly the reference's arithmetic: rotate_y(yaw) applied to the stored point
      const double x2 = sh->cyaw * px + sh->syaw * pz;
      const double z2 = -sh->syaw * px + sh->cyaw * pz;
      ext[0] = ext[1] = x2;
      ext[2] = ext[3] = py;
      ext[4] = ext[5] = z2;
    }
  } else {
    double N0[3], N2[3];
    yaw_rows(sh, Mg, N0, N2);
    int d0 = 0, d1 = 0;
    if (TILED) {
      // (pass B pulls its tiles from an LDS work queue: run-length input 74.8 -> 71.3 us, B = 512 88.7 -> 85.5, config 5 at 16 k
      // 945 -> 927; profiles/r03/r03_pass_b_queue.txt)
      unsigned* qh = &sh->qhead;
      int nsurv = -1;
      if constexpr (LK) {
        if (compact) nsurv = nactive;   // the identity list written before the axis stage
        if (cull) {   // uniform
          nsurv = checked ? cull_plan<true>(sh, p, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext)
                          : cull_plan<false>(sh, p, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext);
        }
      }
      if (checked) sweep_tiled<1, true>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, qh, compact, rng_words, nsurv);
      else if (!SAMPLE && Mg[3] == 0.0) sweep_tiled<1, false, false, !SAMPLE>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, qh, compact, rng_words, nsurv);
      else sweep_tiled<1, false>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, qh, compact, rng_words, nsurv);
    }
    else sweep<VEC, LDSMASK, 1>(p, dpl, mpl, bits, N0, Mg + 3, N2, wave, lane, ext, &d0, &d1);
  }
  LA3D_STAMP(5);
  stage_extents_to_box(sh, p, inst_p, ext, tid, wave, lane);
  stage_status_aux(sh, p, inst_p, tid);
  LA3D_STAMP(6);
}
#undef tid


template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED, int SRC>
int launch_fit_inst(const FitParams& p_in, size_t lds, hipStream_t s, void* workspace) {
  auto kern = fit_instances_kernel<VEC, LDSMASK, SAMPLE, TILED, SRC>;
  allow_big_lds(reinterpret_cast<const void*>(kern));
  FitParams p = p_in;
  p.order_nch = 0; p.order_keys = nullptr; p.order_resident = 0; p.order_shift = 0;
  p.order_self = 0; p.order_flags = nullptr; p.order_nonce = 0; p.est_step = 1;
  // size-balanced launch order: needs the 16-byte mask groups (VEC), more than one workgroup per CU, and a batch
  // the O(B^2) ranking is cheap for
  if (workspace && VEC && !SAMPLE && p.B > 256 && p.B <= ORDER_MAX_B && balance_enabled(p)) {
    const int max_rounds = balance_max_rounds();
    int wg_per_cu = 2048 / NT;  // wave slots: 32 per CU at 64 VGPRs
    const int by_lds = (int)((160 * 1024) / (lds ? lds : 1));
    if (by_lds < wg_per_cu) wg_per_cu = by_lds;
    if (wg_per_cu >= 1 && p.B <= max_rounds * wg_per_cu * 256) {
      p.order_nch = (p.B + ORDER_CHUNK - 1) / ORDER_CHUNK;
      p.order_resident = wg_per_cu * 256;
      p.order_shift = 0;
      p.order_keys = nullptr;
      if (p.area_hint) {   // the caller knows the mask areas (annotation metadata, a preceding filter): no helper launch at all
        while (((long long)p.HW >> p.order_shift) > 0x3ffff) ++p.order_shift;
      } else {
        unsigned* est = static_cast<unsigned*>(workspace);  // [B] sort keys
        // quantise the area to 18 bits: run lengths give the exact area (<= HW), the byte lattice about HW / 67
        int step = 1;
        for (int cand : {EST_STEP, 31, 17, 7, 3})
          if ((p.HW >> 7) / cand >= 64) { step = cand; break; }
        long long amax = (p.rle_counts || p.poly_xy) ? (long long)p.HW : (long long)p.HW / step + 128;
        int shift = 0;
        while ((amax >> shift) > 0x3ffff) ++shift;
        p.order_keys = est;
        bool self = config().order_self != 0 && wg_per_cu * 256 >= 256;
#ifdef LA3D_TIMELINE
        self = false;   // (the stamp rows of the measurement build live where the nonces would)
#endif
        if (self) {
          // a call captured into a HIP graph would replay with the SAME nonce: the records of the previous replay would read as
          // complete while this replay's keys are still on their way - with new masks in the same buffers, two workgroups could rank
          // with different keys.  Captured calls keep the helper kernel.
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) self = false;
          (void)hipGetLastError();
        }
        if (self) {
          // no helper launch: the fit kernel estimates in its prologue (estimate_publish); nonces behind the keys, 256-byte aligned
          p.order_self = config().order_self; p.est_step = step; p.order_shift = shift;
          p.order_flags = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(workspace) + (((size_t)p.B * 4 + 255) & ~(size_t)255));
          const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
          p.order_nonce = (t * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)reinterpret_cast<uintptr_t>(workspace) ^ 0xA5A5A5A55A5A5A5Aull;
        } else {
          hipLaunchKernelGGL(size_estimate_kernel, dim3((p.B + 3) / 4), dim3(256), 0, s, p.mask, p.rle_counts, p.rle_offsets, p.poly_xy,
                             p.poly_ring_off, p.poly_inst_rings, p.B, p.HW, step, shift, est, nullptr);
        }
      }
    }
  }
  hipLaunchKernelGGL(kern, dim3(p.B), dim3(NT), lds, s, p);
  return check_launch("fit_instances_kernel");
}


// run-length input is its own instantiation (it needs the LDS bit image), so the u8 kernels carry no decode code
template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED = false>
int launch_fit(const FitParams& p, size_t lds, hipStream_t s, void* workspace = nullptr) {
  if (LDSMASK && p.rle_counts != nullptr) return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, LDSMASK ? 1 : 0>(p, lds, s, workspace);
  if (LDSMASK && p.poly_xy != nullptr) return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, LDSMASK ? 2 : 0>(p, lds, s, workspace);
  return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, 0>(p, lds, s, workspace);
}

}  // namespace

namespace la3d {
int instance_fit(FitParams p, bool vec, bool ldsmask, bool sample, size_t lds, size_t poly_stage, hipStream_t s, void* workspace,
                 const char* who) {
  const int H = p.H, W = p.W, B = p.B;
  const unsigned char* mask = p.mask;
  if (sample) {
    if (!ldsmask) {
      snprintf(g_err, sizeof(g_err), "%s: reference-subsample mode needs the bit image in LDS (H*W <= 1048576)", who);
      return LA3D_ERR_UNSUPPORTED;
    }
    const size_t blocks = (size_t)((p.nwords + 31) / 32) * 4 + 16;   // one prefix word per 32-word block of the bit image
    lds += blocks > poly_stage ? blocks : poly_stage;
    if (lds > 160 * 1024 - 256) {
      snprintf(g_err, sizeof(g_err), "%s: reference-subsample mode: frame too large for LDS", who);
      return LA3D_ERR_UNSUPPORTED;
    }
    if (vec && W % 32 == 0 && W / 32 <= 255 && (H + 7) / 8 <= 255) {
      // masks of <= 500 px (not sampled) walk their active tiles; the list shares the LDS of the block prefix
      p.ntx = W / 32; p.nty = (H + 7) / 8;
      p.rcp_ntx = 1.0f / (float)p.ntx;
      p.tiles_per_wave = (p.ntx * p.nty + NWAVE - 1) / NWAVE;
      const size_t fixed = lds - (blocks > poly_stage ? blocks : poly_stage);
      size_t budget = (160 * 1024 / 4) & ~(size_t)15;          // four workgroups per CU if the frame allows
      while (budget < fixed + (blocks > 128 ? blocks : 128)) budget += 8 * 1024;
      long cap = (long)(budget - fixed) / 2;
      if (cap > (long)p.ntx * p.nty) cap = (long)p.ntx * p.nty;
      if (cap >= 64 && budget <= 160 * 1024 - 256) {
        p.list_cap = (int)cap;
        const size_t tail = (size_t)cap * 2 > blocks ? (size_t)cap * 2 : blocks;
        return launch_fit<true, true, true, true>(p, fixed + (tail > poly_stage ? tail : poly_stage), s);
      }
    }
    return vec ? launch_fit<true, true, true>(p, lds, s) : launch_fit<false, true, true>(p, lds, s);
  }
  // tiled fast path: 32-px-wide tiles map to exactly one bit-image word / one 128-B depth line per row
  p.ntx = W / 32; p.nty = (H + 7) / 8;
  // ty = int((t + 0.5f) * rcp_ntx) is exact for t < 65536: the fraction of (t+0.5)/ntx stays at least 0.5/ntx away from
  // an integer and the float error is below (65536/ntx) * 1.2e-7
  p.rcp_ntx = 1.0f / (float)(p.ntx > 0 ? p.ntx : 1);
  p.tiles_per_wave = (p.ntx * p.nty + NWAVE - 1) / NWAVE;
  if (ldsmask && vec && W % 32 == 0 && p.ntx <= 255 && p.nty <= 255) {
    // LDS per workgroup: the largest number of workgroups per CU (160 KiB LDS) that still leaves room
    // for a useful list; masks with more active tiles than the cap take the dense walk
    const size_t fixed = lds;
    const long ntiles = (long)p.ntx * p.nty;
    const long want = ntiles < 256 ? ntiles : 256;
    long cap = 0;
    for (int wg_per_cu = 4; wg_per_cu >= 1 && cap < want; --wg_per_cu) {
      const long budget = (160 * 1024 / wg_per_cu) & ~15L;
      cap = (budget - (long)fixed) / 2;
    }
    if (cap > ntiles) cap = ntiles;
    if (cap >= 64) {
      p.list_cap = (int)cap;
      if (mask != nullptr && B > 256) {
        // u8 planes: the resident groups start one group's stream time apart - 256 x H*W bytes at the ~6.4 TB/s a pure reader gets:
        // 12.3 us for 640x480 (the kernel applies it only under the size-ordered launch; LA3D_STAGGER_US overrides, 0 switches it
        // off).  Measured with the self-estimating launch (profiles/r04/r04_stagger.txt, run 4), us per call at 6 / 8 / 10 / 12 / 14 us:
        // config-2 masks B = 1024: 96.0 / 94.8 / 93.4 / 94.5 / 95.4, B = 1536: 130.4 / 127.4 / 124.9 / 125.0 / 124.3; config-5 masks
        // B = 1024: 84.8 / 82.5 / 80.4 / 78.1 / 77.5 - the config-2 optimum is 10, the skewed mix wants more: one stream time is between.
        double us = 256.0 * (double)p.HW / 6.4e6;
        if (config().stagger_us >= 0) us = config().stagger_us;
        p.stagger_ticks = (int)(us * 100.0);
      }
      if (mask == nullptr && B > 256 && config().stagger_nomask_us > 0) p.stagger_ticks = (int)(config().stagger_nomask_us * 100.0);   // (experiment switch)
      return launch_fit<true, true, false, true>(p, fixed + ((size_t)cap * 2 > poly_stage ? (size_t)cap * 2 : poly_stage), s, workspace);
    }
  }
  lds += poly_stage;
  if (ldsmask) return vec ? launch_fit<true, true, false>(p, lds, s, workspace) : launch_fit<false, true, false>(p, lds, s);
  return vec ? launch_fit<true, false, false>(p, lds, s) : launch_fit<false, false, false>(p, lds, s);
}
}  // namespace la3d
