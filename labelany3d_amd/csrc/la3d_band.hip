// la3d_band.hip - the BAND ENGINE of la3d_fit_instances (round 4): fit_bands_kernel<2|4> and its launcher.  Since round 5 it takes
// only GROUNDED u8 batches by default - 1..160 instances since round 6, eight bands per instance up to 128 (profiles/r06/r06_engines_by_batch.txt).
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
// band engine (round 4): NB workgroups per instance, one per band of tile rows - the work item finer than an instance
// that BASELINE config 5 / SURVEY section 7 name.  T(B) = 35 us + 72 us * B / 1024 fits the instance engine at B = 1024 / 2048 /
// 8192 (profiles/r04/r04_cull.txt): the 35 us are ramp-up (nothing to compute until a whole mask plane is streamed) and tail (the
// chain of the last instance: stream, list, pass A, axis, pass B, box on ONE workgroup), and both shrink with the work item.
// A band workgroup streams its rows of the mask plane, lists its active tiles, runs pass A on them and publishes its partial
// moments; the NB workgroups of an instance meet through global memory (release / acquire at agent scope: one fence pair per
// workgroup and exchange), every one sums the NB partials in band order - the same numbers in the same order, hence the same
// axis - and runs pass B on its own tiles; the workgroup that arrives LAST with its extents combines them and writes the record.
// Waiting happens once per instance (for the partner's moments) and is deadlock free as long as the partner is resident or
// will become resident without anybody waiting for this workgroup: partners are 8 blocks apart in dispatch order (same XCD: the
// exchange stays in one L2), so at any time all but the last few dispatched workgroups have their partners on the chip and
// finish.  A watchdog turns a partner that never shows up into a takeover (round 5): the band that timed out first fits the whole instance
// itself (band_takeover) - never a hang, never a dropped box.
// Records: deterministic run to run and under any launch order; the fp64 partial sums are grouped by band, so they agree with
// the instance engine to rounding (like the split engine), not bit for bit.  u8 planes, tiled frames, full-mask mode only.
// ------------------------------------------------------------------------------------------
constexpr int BAND_XD = 8;                 // doubles per published moment record: Sx, Sz, Sxx, Sxz, Szz, n_valid, n_mask, -
constexpr unsigned BAND_SPIN_MAX = 1u << 21;

// per-instance exchange area in the workspace: [NB][2 rounds][BAND_XD] moments, [NB][6] extents
template <int NB>
__device__ inline double* band_xch(const FitParams& p, int inst) { return p.band_xch + (long long)inst * (NB * (2 * BAND_XD + 6)); }


// Band moments -> instance moments -> status / axis, for every band of the instance alike.  Thread 0 publishes this band's
// partial record, waits for the other bands of the instance, then sums the NB records IN BAND ORDER (its own re-read from
// memory like the others: identical operands in identical order in every band).  Returns false on a watchdog timeout.
template <int NB>
__device__ inline void band_moments_to_axis(Shared* sh, const FitParams& p, int inst, int h, int round, const double* acc, int cnt,
                                            int nmask, int tid, int wave, int lane, bool allow_redo) {
  {
    const double r0 = wave_sum(acc[0]), r1 = wave_sum(acc[1]), r2 = wave_sum(acc[2]), r3 = wave_sum(acc[3]), r4 = wave_sum(acc[4]);
    const int rc = wave_sum_i(cnt), rn = wave_sum_i(nmask);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4;
      sh->cnt[wave] = rc; sh->nmask[wave] = rn;
    }
  }
  __syncthreads();
  if (wave == 0) {
    double s[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = lane < NWAVE ? sh->part[lane][k] : 0.0;
    int n = lane < NWAVE ? sh->cnt[lane] : 0, nm = lane < NWAVE ? sh->nmask[lane] : 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { s[k] += dpp_f64<DPP_XOR1>(s[k]); s[k] += dpp_f64<DPP_XOR2>(s[k]); s[k] += dpp_f64<DPP_HALF_MIRROR>(s[k]); }
    n += dpp_i32<DPP_XOR1>(n); n += dpp_i32<DPP_XOR2>(n); n += dpp_i32<DPP_HALF_MIRROR>(n);
    nm += dpp_i32<DPP_XOR1>(nm); nm += dpp_i32<DPP_XOR2>(nm); nm += dpp_i32<DPP_HALF_MIRROR>(nm);
    // Round 6: the record goes out as four 16-byte write-through stores and comes back as ONE batch of four 16-byte loads per band,
    // band b's record into lane b, all bands at once (rounds 4-5: seven 8-byte agent-scope stores, then 7 x NB agent-scope loads
    // one after the other - the compiler waits for every atomic load -: ~25 of the ~35 us of a small grounded call)
    double* x = band_xch<NB>(p, inst);
    int timeout_i = 0;
    if (lane == 0) {
      uint4* mine = reinterpret_cast<uint4*>(x + (h * 2 + round) * BAND_XD);
      auto pair = [](double a, double b) {
        return make_uint4((unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b));
      };
      st16_through(mine + 0, pair(s[0], s[1]));
      st16_through(mine + 1, pair(s[2], s[3]));
      st16_through(mine + 2, pair(s[4], (double)n));
      st16_through(mine + 3, pair((double)nm, 0.0));
      band_release();                                      // the record's stores are acknowledged before the arrival is issued
      unsigned long long* arrive = p.band_arrive + (long long)inst * 4 + round;
      const unsigned spin_max = p.band_test == 1 ? (1u << 10) : BAND_SPIN_MAX;
      unsigned spins = 0;
      if (tagged_arrive_many(arrive, p.band_tag) < (unsigned)NB)
      while (tagged_count(arrive, p.band_tag) < (unsigned)NB && spins < spin_max) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
      }
      timeout_i = spins >= spin_max ? 1 : 0;
    }
    const bool timeout = __builtin_amdgcn_readfirstlane(timeout_i) != 0;
    band_acquire();                                        // the other bands' records are read after their arrivals were seen
    // lane 4 b + q holds granule q of band b's record: ONE 16-byte load per lane, all of them in flight together
    double d0 = 0.0, d1 = 0.0;
    if (lane < 4 * NB) {
      const u32x4 g = ld16_through(reinterpret_cast<const u32x4*>(x + ((lane >> 2) * 2 + round) * BAND_XD) + (lane & 3));
      d0 = __hiloint2double((int)g[1], (int)g[0]); d1 = __hiloint2double((int)g[3], (int)g[2]);
    }
    double t[5] = {0, 0, 0, 0, 0}, tn = 0, tm = 0;
#pragma unroll
    for (int hb = 0; hb < NB; ++hb) {                      // band order: the same sum in every band of the instance
      t[0] += readlane_f64(d0, 4 * hb); t[1] += readlane_f64(d1, 4 * hb);
      t[2] += readlane_f64(d0, 4 * hb + 1); t[3] += readlane_f64(d1, 4 * hb + 1);
      t[4] += readlane_f64(d0, 4 * hb + 2); tn += readlane_f64(d1, 4 * hb + 2);
      tm += readlane_f64(d0, 4 * hb + 3);
    }
    if (lane == 0) {
      const int nt = (int)tn;
      double gap = NAN;
      int st = LA3D_BOX_OK;
      // A partner that never showed up (round 5): the first band to time out claims the instance through the fourth arrival
      // word and fits it on its own - band_takeover; the others leave.  The claimer never arrives at the extents counter, so no
      // other band can write the record.
      if (timeout) st = tagged_arrive(p.band_arrive + (long long)inst * 4 + 3, p.band_tag) == 1u ? BAND_ST_TAKEOVER : BAND_ST_ABANDON;
      else if (sh->bad_ground) st = LA3D_BOX_BAD_GROUND;
      else if (nt == 0) st = LA3D_BOX_EMPTY;
      else if (nt == 1) st = LA3D_BOX_TOO_FEW;
      const double chk = (t[0] + t[1]) + (t[2] + t[3]) + t[4];
      const bool nonfinite = !(fabs(chk) <= 1.79769313486231570815e308);
      double cy = NAN, sy = NAN;
      bool ill = false;
      if (st == LA3D_BOX_OK) ill = axis_from_sums((double)nt, t[0], t[1], t[2], t[3], t[4], &cy, &sy, &gap);
      // (the summed moments are the same in every band: all of them take the same decision and the same pivot - stage_moments_to_axis)
      sh->redo = (allow_redo && !timeout && !sh->bad_ground) ? (nonfinite ? 1 : (ill ? 2 : 0)) : 0;
      set_pivot(sh, (ill && !nonfinite) ? t[0] / (double)nt : 0.0, (ill && !nonfinite) ? t[1] / (double)nt : 0.0);
      if (ill && !allow_redo) gap = 0.0;
      sh->cyaw = cy; sh->syaw = sy;
      sh->qhead = 0u;
      sh->st = st;
      sh->n_valid = nt;
      sh->gap = gap;
      sh->nm = (int)tm;
    }
  }
  __syncthreads();
  if (sh->redo) return;   // uniform
  // rejected instance: band 0 writes the outputs, every band returns (a band that took the instance over after a watchdog
  // timeout writes them itself: band_takeover)
  if (tid == 0 && h == 0 && sh->st != LA3D_BOX_OK && sh->st < BAND_ST_TAKEOVER) {
    if (p.aux) {
      double* a = p.aux + (long long)inst * LA3D_AUX;
      a[0] = atan2(sh->syaw, sh->cyaw); a[1] = (double)sh->n_valid; a[2] = (double)sh->nm; a[3] = sh->gap;
    }
    p.status[inst] = sh->st;
    write_nan_box(p.out + (long long)inst * LA3D_REC);
    if (p.proj) { for (int j = 0; j < 8; ++j) p.proj[(long long)inst * 8 + j] = NAN; }
  }
}

// Watchdog fallback of the band engine (round 5; ADVICE / VERDICT round 4: a timeout used to drop a fittable box as
// LA3D_BOX_UNSUPPORTED): the band that claimed the instance fits ALL of it with the generic row-linear walk - mask bytes and depth
// straight from memory, no bit image, no tile list, so the band's LDS layout does not matter - and writes the record.  Slow
// (one workgroup re-reads the whole plane twice) and practically never taken: partners are dispatched within a few blocks of
// each other.  The sums are grouped like the untiled instance engine's, so the record agrees with the other engines to rounding.
__device__ inline void band_takeover(Shared* sh, const FitParams& p, int inst, int tid, int wave, int lane) {
  __syncthreads();
  const int img = p.image_index ? p.image_index[inst] : inst;
  if (tid == NT - 1) {   // M in FRAME rows again (the band kernel keeps band-local rows); Rg and bad_ground stand
    double Kinv[9];
    inv3_camera(p.K + (long long)img * p.k_stride, Kinv);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) sh->M[i * 3 + j] = sh->Rg[i] * Kinv[j] + sh->Rg[3 + i] * Kinv[3 + j] + sh->Rg[6 + i] * Kinv[6 + j];
  }
  __syncthreads();
  double Mg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Mg[i] = uniform_f64(sh->M[i]);
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
  const unsigned char* mpl = p.mask + (long long)inst * p.HW;
  double acc[5] = {0, 0, 0, 0, 0};
  int cnt = 0, nmask = 0;
  sweep<true, false, 0>(p, dpl, mpl, nullptr, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, &nmask);
  stage_moments_to_axis(sh, p, inst, acc, cnt, nmask, tid, wave, lane, false);
  if (sh->st != LA3D_BOX_OK) return;
  double ext[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  double N0[3], N2[3];
  yaw_rows(sh, Mg, N0, N2);
  int d0 = 0, d1 = 0;
  sweep<true, false, 1>(p, dpl, mpl, nullptr, N0, Mg + 3, N2, wave, lane, ext, &d0, &d1);
  stage_extents_to_box(sh, p, inst, ext, tid, wave, lane);
  stage_status_aux(sh, p, inst, tid);
}

template <int NB>
__global__ __launch_bounds__(NT, NT / 64) void fit_bands_kernel(const FitParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  Shared* sh = reinterpret_cast<Shared*>(smem + p.mask_lds_bytes);
  unsigned short* list = reinterpret_cast<unsigned short*>(smem + p.mask_lds_bytes + sizeof(Shared));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (instance slot, band): partners are 8 blocks apart, i.e. on the same XCD (the dispatcher places block b on XCD b % 8)
  int bx = (int)blockIdx.x;
  if (p.band_test == 2) bx ^= (bx >> 3) & 7;   // test hook: a bijection of the grid that puts the bands of an instance on different XCDs
  const int slot = ((bx >> 3) / NB) * 8 + (bx & 7), h = (bx >> 3) % NB;
  if (slot >= p.B) return;   // (grid padded to a multiple of 8 * NB)
  if (p.band_test == 1 && h == 1 && slot % 3 == 0) return;   // test hook: a partner that never shows up (the others take over)
  const int inst = p.order_nch > 0 ? order_select(p, slot, sh, wave, lane) : xcd_remap(slot, p.B);
  if (tid == 0) sh->order_inst = inst;
  const int img = p.image_index ? p.image_index[inst] : inst;
  // this band: tile rows [ty0, ty0 + ntyb) of the frame, pixel rows [row0, row0 + rows_b)
  const int ty0 = h * p.band_trows, ntyb = (h == NB - 1) ? p.nty - ty0 : p.band_trows;
  const int row0 = ty0 * 8, rows_b = min(ntyb * 8, p.H - row0);
  const int HWb = rows_b * p.W;
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride + (long long)row0 * p.W;
  const unsigned char* mpl = p.mask + (long long)inst * p.HW + (long long)row0 * p.W;

  if (tid == NT - 1) {
    // per-instance geometry as in the instance engine, in BAND-LOCAL pixel rows: v = v' + row0 folds into the constant column
    double Kinv[9], Rg[9];
    inv3_camera(p.K + (long long)img * p.k_stride, Kinv);
    sh->bad_ground = ground_rotation(p.ground ? p.ground + (long long)inst * 4 : nullptr, Rg);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double m[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) m[j] = Rg[i] * Kinv[j] + Rg[3 + i] * Kinv[3 + j] + Rg[6 + i] * Kinv[6 + j];
      sh->M[i * 3] = m[0]; sh->M[i * 3 + 1] = m[1]; sh->M[i * 3 + 2] = fma(m[1], (double)row0, m[2]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) sh->Rg[i] = Rg[i];
  }

  // ---- phase 0: the band's rows of the u8 plane -> bit image in LDS (same forms as the instance engine) ----
  int nmask = 0;
  {
    unsigned short* b16 = reinterpret_cast<unsigned short*>(bits);
    const int ngroups = HWb >> 4;
    const u32x4* m4 = reinterpret_cast<const u32x4*>(mpl);
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
  }
  __syncthreads();
  const int inst_p = __builtin_amdgcn_readfirstlane(sh->order_inst);
  double Mg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Mg[i] = uniform_f64(sh->M[i]);

  // a FitParams of the band: the walk functions see a frame of rows_b rows
  FitParams pb = p;
  pb.H = rows_b; pb.nty = ntyb; pb.HW = HWb;

  // ---- active-tile list of the band (one pass, ballots in SGPRs, image compacted in place) ----
  int nactive = 0, compact = 0;
  {
    const int ntiles = p.ntx * ntyb, per = (ntiles + NWAVE - 1) / NWAVE;   // per <= 256: fit_dispatch checks
    const int tbeg = wave * per, tend = min(tbeg + per, ntiles);
    int base = 0, wcount = 0;
    unsigned long long bal[4];
    unsigned wrd[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = tbeg + k * 64 + lane;
      unsigned any = 0;
      if (t < tend) {
        const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;
        const int rmax = rows_b - 1 - ty * 8;
        const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const unsigned w = bw[min(rr, rmax) * p.ntx];
          any |= w;
          wrd[k][rr] = rr <= rmax ? w : 0u;
        }
      }
      bal[k] = __ballot(any != 0);
      wcount += __popcll(bal[k]);
    }
    if (lane == 0) sh->scan[wave] = (unsigned)wcount;
    __syncthreads();
    for (int w = 0; w < NWAVE; ++w) {
      const int c = (int)sh->scan[w];
      if (w < wave) base += c;
      nactive += c;
    }
    if (nactive > p.list_cap) {
      nactive = -1;   // dense walk of the band
    } else {
      int off = base;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((bal[k] >> lane) & 1ull) {
          const int t = tbeg + k * 64 + lane;
          const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;
          list[off + __popcll(bal[k] & ((1ull << lane) - 1ull))] = (unsigned short)((ty << 8) | tx);
        }
        off += __popcll(bal[k]);
      }
      if (nactive * 32 + cull_rng_words(nactive) * 4 <= p.mask_lds_bytes) {   // uniform
        compact = 1;
        off = base;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((bal[k] >> lane) & 1ull) {
            uint4* e = reinterpret_cast<uint4*>(bits) + 2 * (off + __popcll(bal[k] & ((1ull << lane) - 1ull)));
            e[0] = make_uint4(wrd[k][0], wrd[k][1], wrd[k][2], wrd[k][3]);
            e[1] = make_uint4(wrd[k][4], wrd[k][5], wrd[k][6], wrd[k][7]);
          }
          off += __popcll(bal[k]);
        }
      }
    }
    __syncthreads();
  }
  bool cull = false;
  int rng_words = 0;
  if (compact) {   // uniform
    rng_words = cull_rng_words(nactive);
    cull = nactive >= p.cull_min && nactive <= CULL_MAXT;
    if (!cull) {
      unsigned short* surv = reinterpret_cast<unsigned short*>(bits + nactive * 8);
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
      for (int t = tid; t < nactive; t += NT) surv[t] = (unsigned short)t;
    }
  }

  // ---- pass A on the band, exchange, axis ----
  double acc[5] = {0, 0, 0, 0, 0};
  int cnt = 0;
  bool checked = false;
  const bool specA = Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform (quad_math: the un-grounded forms, same records)
  if (cull) {
    if (specA) sweep_tiled<0, false, true, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
    else sweep_tiled<0, false, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
  } else {
    if (specA) sweep_tiled<0, false, false, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
    else sweep_tiled<0, false>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
  }
  cnt = nmask;   // optimistic pass: valid pixels = mask pixels
  band_moments_to_axis<NB>(sh, p, inst_p, h, 0, acc, cnt, nmask, tid, wave, lane, true);
  if (sh->st >= BAND_ST_TAKEOVER) {   // uniform: watchdog timeout
    if (sh->st == BAND_ST_TAKEOVER) band_takeover(sh, p, inst_p, tid, wave, lane);
    return;
  }
  if (sh->redo) {   // uniform, and the same in every band of the instance: the summed moments decide
    __syncthreads();   // (the pivot - zero unless the summed moments were ill-conditioned: axis_from_sums - stays in LDS)
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = 0;
    cnt = 0;
    if (sh->redo == 2) {   // ill-conditioned sums: the moments about the pivot; the optimistic pass's tile ranges and pass B stand
      pivot_pass(pb, dpl, bits, list, nactive, Mg, Mg + 6, wave, lane, compact, pivot_ptr(sh), acc, &cnt);
    } else {
      checked = true;
      if (cull) sweep_tiled<0, true, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
      else sweep_tiled<0, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, nullptr, compact, rng_words);
    }
    band_moments_to_axis<NB>(sh, p, inst_p, h, 1, acc, cnt, nmask, tid, wave, lane, false);
    if (sh->st >= BAND_ST_TAKEOVER) {   // uniform
      if (sh->st == BAND_ST_TAKEOVER) band_takeover(sh, p, inst_p, tid, wave, lane);
      return;
    }
  }
  if (sh->st != LA3D_BOX_OK) return;

  // ---- pass B on the band ----
  double ext[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  {
    double N0[3], N2[3];
    yaw_rows(sh, Mg, N0, N2);
    int d0 = 0;
    int nsurv = compact ? nactive : -1;
    if (cull) {   // uniform
      nsurv = checked ? cull_plan<true>(sh, pb, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext)
                      : cull_plan<false>(sh, pb, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext);
    }
    if (checked) sweep_tiled<1, true>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, &sh->qhead, compact, rng_words, nsurv);
    else if (Mg[3] == 0.0) sweep_tiled<1, false, false, true>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, &sh->qhead, compact, rng_words, nsurv);
    else sweep_tiled<1, false>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, &sh->qhead, compact, rng_words, nsurv);
  }
  // ---- extents of the band -> exchange -> the last band to arrive writes the record ----
  {
    const double r0 = wave_min(ext[0]), r1 = wave_max(ext[1]), r2 = wave_min(ext[2]), r3 = wave_max(ext[3]), r4 = wave_min(ext[4]), r5 = wave_max(ext[5]);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; pp[5] = r5;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  double lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = lane < NWAVE ? sh->part[lane][2 * k] : INFINITY;
    hi[k] = lane < NWAVE ? sh->part[lane][2 * k + 1] : -INFINITY;
    lo[k] = fmin(lo[k], dpp_f64<DPP_XOR1>(lo[k])); lo[k] = fmin(lo[k], dpp_f64<DPP_XOR2>(lo[k])); lo[k] = fmin(lo[k], dpp_f64<DPP_HALF_MIRROR>(lo[k]));
    hi[k] = fmax(hi[k], dpp_f64<DPP_XOR1>(hi[k])); hi[k] = fmax(hi[k], dpp_f64<DPP_XOR2>(hi[k])); hi[k] = fmax(hi[k], dpp_f64<DPP_HALF_MIRROR>(hi[k]));
  }
  double* xe = band_xch<NB>(p, inst_p) + NB * 2 * BAND_XD;   // [NB][6]
  int last = 0;
  if (lane == 0) {   // three 16-byte write-through stores: (lo, hi) of x, y, z
#pragma unroll
    for (int k = 0; k < 3; ++k)
      st16_through(reinterpret_cast<uint4*>(xe + h * 6) + k, make_uint4((unsigned)__double2loint(lo[k]), (unsigned)__double2hiint(lo[k]),
                                                                        (unsigned)__double2loint(hi[k]), (unsigned)__double2hiint(hi[k])));
    band_release();
    last = tagged_arrive_many(p.band_arrive + (long long)inst_p * 4 + 2, p.band_tag) == (unsigned)NB ? 1 : 0;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  band_acquire();
  {   // lane 4 b + k holds (lo, hi) of axis k of band b: one 16-byte load per lane, then min / max over the bands
    double el = INFINITY, eh = -INFINITY;
    if (lane < 4 * NB && (lane & 3) < 3) {
      const u32x4 g = ld16_through(reinterpret_cast<const u32x4*>(xe + (lane >> 2) * 6) + (lane & 3));
      el = __hiloint2double((int)g[1], (int)g[0]); eh = __hiloint2double((int)g[3], (int)g[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double l = readlane_f64(el, k), u = readlane_f64(eh, k);
#pragma unroll
      for (int hb = 1; hb < NB; ++hb) { l = fmin(l, readlane_f64(el, 4 * hb + k)); u = fmax(u, readlane_f64(eh, 4 * hb + k)); }
      lo[k] = uniform_f64(l); hi[k] = uniform_f64(u);
    }
  }
  double Rg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rg[i] = sh->Rg[i];
  if (p.proj) {
    const int im = p.image_index ? p.image_index[inst_p] : inst_p;
    write_box_wave(p.out + (long long)inst_p * LA3D_REC, Rg, sh->cyaw, sh->syaw, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lane,
                   p.proj + (long long)inst_p * 8, p.K + (long long)im * p.k_stride, p.proj_w, p.proj_h);
  } else {
    write_box_wave(p.out + (long long)inst_p * LA3D_REC, Rg, sh->cyaw, sh->syaw, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lane);
  }
  if (lane == 63) {
    if (p.aux) {
      double* a = p.aux + (long long)inst_p * LA3D_AUX;
      a[0] = atan2(sh->syaw, sh->cyaw); a[1] = (double)sh->n_valid; a[2] = (double)sh->nm; a[3] = sh->gap;
    }
    p.status[inst_p] = LA3D_BOX_OK;
  }
}


// ---- band engine (fit_bands_kernel) ----
constexpr int BAND_NB_MAX = 8;
constexpr int BAND_MINB = 1;    // smallest batch the band engine takes by default (round 6: with eight bands per instance it leads from one
                                // instance on - 23 vs 32 us through the split engine; rounds 4-5: 16)
constexpr size_t band_xch_doubles(int nb) { return (size_t)nb * (2 * BAND_XD + 6); }

// workspace of the band engine: [B] u32 sort keys | [B][4] u64 tagged arrival words | [B][NB_MAX * 22] f64 exchange records
inline size_t band_keys_bytes(int B) { return ((size_t)B * 4 + 255) & ~(size_t)255; }
inline size_t band_workspace_bytes_impl(int B) { return band_keys_bytes(B) + (size_t)B * 32 + (size_t)B * band_xch_doubles(BAND_NB_MAX) * 8 + 256; }

inline bool band_frame_ok_impl(int H, int W, int nb) {
  if (W % 32 != 0 || (long long)H * W % 16 != 0) return false;
  const int ntx = W / 32, nty = (H + 7) / 8;
  if (ntx > 255 || nty > 255 || nty < nb) return false;
  const int tb = nty / nb, tmax = nty - (nb - 1) * tb;   // the last band takes the remainder
  return (long long)ntx * tmax <= 256 * NWAVE;          // one-pass tile list: <= 256 tiles per wave
}

// Bands per instance: LA3D_BANDS pins 2, 4 or 8; otherwise eight up to 128 instances (round 6), four up to 288, two beyond.  Measured, us
// per call, u8 planes - round 4 (profiles/r04/r04_band.txt; split | instance | two bands | four bands): B = 4: 34 | 37 | 44 | 31;
// 64: 43 | 56 | 51 | 38; 256: 67 | 67 | 64 | 63; 320: 80 | 76 | 68 | 69; 384: - | 75 | 73 | 77; 512: - | 80 | 85 | 95; 1024: - | 107 | 134 | 168;
// round 6, grounded calls after the exchange rewrite (profiles/r06/r06_engines_by_batch.txt; four | eight | sixteen bands):
// B = 1: 33.3 | 23.2 | 30.0; 4: - | 28.3 | 35.0; 16: 33.7 | 27.6 | 36.3; 32: - | 31.8 | 38.8; 64: 35.2 | 34.3 | 58.4; 128: 45.3 | 45.1 | -;
// 192: 57.2 | 64.8 | -.
inline int band_count(const FitParams& p) {
  int nb = config().bands ? config().bands : (p.B <= 128 ? 8 : p.B <= 288 ? 4 : 2);
  if (nb == 8 && !band_frame_ok_impl(p.H, p.W, 8)) nb = 4;
  if (nb == 4 && !band_frame_ok_impl(p.H, p.W, 4)) nb = 2;
  return nb;
}

// u8 planes, 16-byte aligned, full-mask mode.  By default the band engine takes the GROUNDED calls of 1 <= B <= 160 instances (round
// 6: eight bands per instance put it ahead of the split engine from one instance on - B = 1 / 16 / 64: 23 / 28 / 34 us vs 32 / 36 / 43 -,
// and above ~160 one workgroup per instance is as fast or faster: B = 192: 57 vs 54 us).  History - round 4, us per call, split | four
// bands once the band launch lost its memset: B = 1: 32.2 | 32.3; 16: 34.9 | 33.9; 64: 42.1 | 36.6; instance | two | four bands:
// B = 256: 59.7 | 58.6 | 60.0; 384: 63.4 | 66.2 | 78.1.  LA3D_ENGINE=band / opt_engine pins it for any batch, LA3D_BAND_MAXB moves the limit.
inline bool band_eligible_impl(const FitParams& p, bool vec, bool sample) {
  const int e = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;
  if (e == LA3D_ENGINE_INSTANCE || e == LA3D_ENGINE_SPLIT) return false;
  if (!vec || sample || p.mask == nullptr || !band_frame_ok_impl(p.H, p.W, band_count(p))) return false;
  if (e == LA3D_ENGINE_BAND) return true;
  return config().band_default && p.B >= BAND_MINB && p.B <= config().band_maxb;
}

template <int NB>
int launch_fit_bands(const FitParams& p_in, hipStream_t s, void* workspace) {
  FitParams p = p_in;
  auto kern = fit_bands_kernel<NB>;
  allow_big_lds(reinterpret_cast<const void*>(kern));
  p.ntx = p.W / 32; p.nty = (p.H + 7) / 8;
  p.rcp_ntx = 1.0f / (float)p.ntx;
  p.band_trows = p.nty / NB;
  const int tmax = p.nty - (NB - 1) * p.band_trows;
  p.list_cap = p.ntx * tmax;
  p.tiles_per_wave = (p.list_cap + NWAVE - 1) / NWAVE;
  // LDS: four workgroups per CU by wave slots, so each may use a quarter of the CU's LDS: the region behind the band's bit image
  // keeps depth tiles between the passes
  const size_t img = (((size_t)tmax * 8 * p.W / 8) + 15) & ~(size_t)15;
  const size_t fixed = sizeof(Shared) + (((size_t)p.list_cap * 2 + 15) & ~(size_t)15);
  size_t region = ((160 * 1024 / 4) - fixed) & ~(size_t)15;
  if (region < img) region = img;
  if (region + fixed > 160 * 1024 - 256) return LA3D_ERR_UNSUPPORTED;   // (band_frame_ok keeps frames far below this)
  p.mask_lds_bytes = (int)region;
  unsigned char* w = static_cast<unsigned char*>(workspace);
  unsigned* keys = reinterpret_cast<unsigned*>(w);
  p.band_arrive = reinterpret_cast<unsigned long long*>(w + band_keys_bytes(p.B));
  p.band_xch = reinterpret_cast<double*>(w + band_keys_bytes(p.B) + (size_t)p.B * 32);
  {
    const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    p.band_tag = (((t * 0x9E3779B97F4A7C15ull) >> 13) ^ (unsigned long long)reinterpret_cast<uintptr_t>(workspace)) & 0xffffffffffffull;
    if (p.band_tag == 0) p.band_tag = 1;   // (zeroed words - a captured call's memset - never look like this call's)
  }
  p.order_nch = 0; p.order_keys = nullptr; p.order_resident = 0; p.order_shift = 0;
  p.order_self = 0; p.order_flags = nullptr; p.order_nonce = 0; p.est_step = 1;
  if (p.B > 256 && p.B <= ORDER_MAX_B && balance_enabled(p) && p.B <= balance_max_rounds() * 4 * 256) {
    // largest instances first (chunk-local ranking as in the instance engine; no per-CU pairing: an instance's bands sit on NB CUs)
    p.order_nch = (p.B + ORDER_CHUNK - 1) / ORDER_CHUNK;
    if (p.area_hint) {
      while (((long long)p.HW >> p.order_shift) > 0x3ffff) ++p.order_shift;
    } else {
      int step = 1;
      for (int cand : {EST_STEP, 31, 17, 7, 3})
        if ((p.HW >> 7) / cand >= 64) { step = cand; break; }
      const long long amax = (long long)p.HW / step + 128;
      int shift = 0;
      while ((amax >> shift) > 0x3ffff) ++shift;
      hipLaunchKernelGGL(size_estimate_kernel, dim3((p.B + 3) / 4), dim3(256), 0, s, p.mask, nullptr, nullptr, nullptr, nullptr, nullptr,
                         p.B, p.HW, step, shift, keys, nullptr);
      p.order_keys = keys;
    }
  }
  {
    // a call captured into a HIP graph replays with the same tag: its arrival words are cleared by a memset node of the graph
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (capturing && hipMemsetAsync(p.band_arrive, 0, (size_t)p.B * 32, s) != hipSuccess) return check_launch("band engine memset");
  }
  const int grid = ((p.B + 7) / 8) * 8 * NB;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), region + fixed, s, p);
  return check_launch("fit_bands_kernel");
}

}  // namespace

namespace la3d {
bool band_eligible(const FitParams& p, bool vec, bool sample) { return ::band_eligible_impl(p, vec, sample); }
bool band_frame_ok(int H, int W, int nb) { return ::band_frame_ok_impl(H, W, nb); }
size_t band_workspace_bytes(int B) { return ::band_workspace_bytes_impl(B); }
int band_fit(const FitParams& p, hipStream_t s, void* workspace) {
  const int nb = band_count(p);
  return nb == 8 ? launch_fit_bands<8>(p, s, workspace) : nb == 4 ? launch_fit_bands<4>(p, s, workspace) : launch_fit_bands<2>(p, s, workspace);
}
}  // namespace la3d
