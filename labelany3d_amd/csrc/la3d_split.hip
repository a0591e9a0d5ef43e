// la3d_split.hip — the split engine of la3d_fit_instances (full-mask mode, W % 32 == 0, 16-B aligned planes).
//
// Why: with one workgroup per instance (la3d.hip) a 1024-instance batch is exactly one resident round,
// so (i) the CU that drew the largest masks sets the kernel time (max/mean tile load 2.2x on BASELINE
// config 2) and (ii) the memory-bound mask stream and the VALU-bound passes run as two chip-wide phases
// that never overlap.  Here the work is cut the other way:
//
//   scan_kernel      one 256-thread workgroup per (instance, band of 64 rows): streams the u8 mask once
//                    (128 px x 8 rows = 1 KB per wave-load, non-temporal), packs it to a TILE-MAJOR bit
//                    image in the workspace (32 B per 32x8-px tile; only tiles with a set bit are written),
//                    and emits the band's compacted list of active tiles + mask pixel count.
//   plan_kernel      one workgroup: the per-instance geometry (K^-1, Rg, M), prefix of the tile counts + every walking wave's start position.
//   moments_kernel   every wave of the grid walks an EQUAL range of the batch's concatenated active-tile
//                    list (4 tiles = 4 depth lines in flight per step), flushing one partial per instance
//                    it touches into slot (global wave index + instance index) — unique, ordered, static.
//   axis_kernel      one wave per instance: sums that instance's partial slots in fixed order -> status,
//                    yaw axis, aux.
//   extents_kernel   same walk, six extents in the yaw frame -> partial slots.
//   final_kernel     one wave per instance: min/max over its slots -> 39-double record.
//
// The batch is cut into sub-batches whose chains run on alternating internal streams (forked from and
// joined back to the caller's stream with events), so the scan of sub-batch j+1 overlaps the passes of j.
// Results are bit-identical run to run: the tile partition depends only on the tile counts, and every
// reduction has a fixed order.
//
// Reference semantics: depth_to_points src/util.py:52-75, estimate_bbox src/util_3dbox.py:106-178 (as in
// la3d.hip; the arithmetic per pixel is the shared quad_math<>).
#include <stdlib.h>

#include "la3d_device.hpp"
#include "la3d_poly.hpp"

namespace la3d {

namespace {

constexpr int SNT = 256;            // threads per workgroup in every split kernel
constexpr int SNW = SNT / 64;
constexpr int BAND_TROWS = 8;       // tile rows (of 8 px) per band -> 64 image rows
constexpr int PSTRIDE = 8;          // doubles per partial slot
constexpr int MAX_SUB = 8;          // sub-batches per call
constexpr int MAX_SEG = 4096;       // (instances x bands) segments per sub-batch held in LDS by the walkers

struct SplitParams {
  FitParams f;
  int b0, nb;           // sub-batch: instances [b0, b0+nb)
  int nband;            // bands per instance
  int tpb;              // tile slots per band = BAND_TROWS * ntx
  int nwaves;           // waves of the walking kernels' grid (grid * SNW)
  // workspace views (whole batch)
  unsigned* tbits;      // [B*nband][tpb][8]: row-words of the segment's active tiles, in list order
  int* tcount;          // [B*nband]
  int* nmaskb;          // [B*nband]
  unsigned short* tlist;  // [B*nband][tpb]
  int* toff;            // per sub-batch: [nb*nband + 1] exclusive prefix of tcount (plan_kernel)
  int* wstart;          // per sub-batch: [nwaves][2] = segment, index where each walking wave starts (plan_kernel)
  double* partA;        // per sub-batch: [(nwaves + nb)][PSTRIDE]
  double* partB;        // per sub-batch: [(nwaves + nb)][PSTRIDE]
  double* axis;         // [B][4] = cos yaw, sin yaw, status, pad
};

// per-instance geometry (reference src/util.py:56, src/util_3dbox.py:128-134): one thread per instance.  Runs inside plan_kernel
// (round 3: it was a launch of its own at the head of the chain - one dependent launch less per call, ~3 us at small batches);
// kept out of the scan kernels so that its 3x3 elimination does not cost the streaming waves registers
__device__ inline void geo_one(const FitParams& p, int inst) {
  const int img = p.image_index ? p.image_index[inst] : inst;
  double Kinv[9], Rg[9];
  inv3(p.K + (long long)img * p.k_stride, Kinv);
  const int bad = ground_rotation(p.ground ? p.ground + (long long)inst * 4 : nullptr, Rg);
  double* g = p.geo + (long long)inst * GEO_D;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) g[i * 3 + j] = Rg[i] * Kinv[j] + Rg[3 + i] * Kinv[3 + j] + Rg[6 + i] * Kinv[6 + j];
  for (int i = 0; i < 9; ++i) g[9 + i] = Rg[i];
  g[18] = bad ? 1.0 : 0.0;
  g[19] = 0.0;
}

// ------------------------------------------------------------------------------------------
// scan: mask -> per (instance, band): compacted active-tile ids + their 8 row-words of mask bits
// ------------------------------------------------------------------------------------------
// dynamic LDS: bandbits u32 [trows*ntx][8] (tile-major), then pos u16 [tpb], act u32 [64], wsum int [SNW]
__global__ __launch_bounds__(SNT) void scan_kernel(const SplitParams sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const FitParams& p = sp.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
  const int band = blockIdx.x % sp.nband;
  const int inst = sp.b0 + blockIdx.x / sp.nband;
  const int W = p.W, H = p.H, ntx = p.ntx, nty = p.nty;
  const int nsx = (W + 127) >> 7;
  unsigned* bandbits = reinterpret_cast<unsigned*>(smem);
  unsigned short* bb16 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* pos = reinterpret_cast<unsigned short*>(smem + (size_t)sp.tpb * 32);
  unsigned* act_bits = reinterpret_cast<unsigned*>(smem + (size_t)sp.tpb * 34);
  int* wsum = reinterpret_cast<int*>(act_bits + 64);
  const unsigned char* mpl = p.mask + (long long)inst * p.HW;
  const int r = lane >> 3, c = lane & 7;

  for (int i = tid; i < 64; i += SNT) act_bits[i] = 0;
  __syncthreads();

  const int trow0 = band * BAND_TROWS;
  const int trows = min(BAND_TROWS, nty - trow0);
  const int nst = trows * nsx;  // super-tiles (128 px x 8 rows) of this band
  int nmask = 0;
  // all of a wave's super-tile loads are independent: issue them in groups of 4 (4 KB in flight per wave)
  for (int e0 = wave; e0 < nst; e0 += SNW * 4) {
    u32x4 m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * SNW;
      m[q] = (u32x4){0u, 0u, 0u, 0u};
      if (e < nst) {
        const int g = e / nsx, sx = e - g * nsx;
        const int row = (trow0 + g) * 8 + r, col = sx * 128 + c * 16;
        if (row < H && col < W)
          m[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(mpl + (long long)row * W + col));
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * SNW;
      if (e >= nst) break;
      const int g = e / nsx, sx = e - g * nsx;
      const unsigned pat = nz4(m[q].x) | (nz4(m[q].y) << 4) | (nz4(m[q].z) << 8) | (nz4(m[q].w) << 12);
      nmask += __popc(pat);
      // tile-major: word (tile slot, row r) = 16-px chunk 2k in the low half, 2k+1 in the high half
      const int tx = sx * 4 + (c >> 1);
      if (tx < ntx) bb16[((g * ntx + tx) * 8 + r) * 2 + (c & 1)] = (unsigned short)pat;
      const unsigned long long bal = __ballot(pat != 0);
      if (bal != 0 && lane == 0) {
        unsigned a = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (bal & (0x0303030303030303ull << (2 * k))) a |= 1u << k;
        const int slot = g * ntx + sx * 4;  // first tile slot of this super-tile within the band
        atomicOr(&act_bits[slot >> 5], a << (slot & 31));
        if ((slot & 31) > 28) atomicOr(&act_bits[(slot >> 5) + 1], a >> (32 - (slot & 31)));
      }
    }
  }
  nmask = wave_sum_i(nmask);
  if (lane == 0) wsum[wave] = nmask;
  __syncthreads();

  // compaction of the active tile slots in ascending slot order (deterministic), by wave 0
  const int seg = inst * sp.nband + band;
  const int nslots = trows * ntx;
  if (wave == 0) {
    unsigned short* out = sp.tlist + (long long)seg * sp.tpb;
    int base = 0;
    for (int s0 = 0; s0 < nslots; s0 += 64) {
      const int sl = s0 + lane;
      bool on = false;
      int ty = 0, tx = 0;
      if (sl < nslots) {
        ty = sl / ntx; tx = sl - ty * ntx;
        on = (act_bits[sl >> 5] >> (sl & 31)) & 1u;
      }
      const unsigned long long bal = __ballot(on);
      const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (sl < nslots) pos[sl] = on ? (unsigned short)at : (unsigned short)0xffff;
      if (on) out[at] = (unsigned short)(((trow0 + ty) << 8) | tx);
      base += __popcll(bal);
    }
    if (lane == 0) {
      sp.tcount[seg] = base;
      int nm = 0;
      for (int w = 0; w < SNW; ++w) nm += wsum[w];
      sp.nmaskb[seg] = nm;
    }
  }
  __syncthreads();
  // the active tiles' 8 row-words, compacted in list order: entry e of the segment is 32 contiguous bytes
  unsigned* outb = sp.tbits + (long long)seg * sp.tpb * 8;
  for (int it = tid; it < nslots * 8; it += SNT) {
    const unsigned short at = pos[it >> 3];
    if (at != 0xffff) outb[(int)at * 8 + (it & 7)] = bandbits[it];
  }
}

// ------------------------------------------------------------------------------------------
// scan for masks that are not u8 planes (SRC 1 = COCO run lengths, 2 = polygon parts): one 512-thread workgroup per instance
// decodes into a row-major LDS bit image with the instance engine's own decoders (rle_to_bits / poly_to_bits: same bits), then
// every wave turns whole bands of it into the scan_kernel's output - compacted active-tile ids, their eight row words, the
// band's pixel count.  Small batches of the reference's own mask formats thereby get the split engine's parallelism: one big
// instance no longer sits on one CU (measured, largest mask 60-95 k px, B = 1 / 8 / 64 / 128: 47 / 46 / 59 / 59 us per call with one
// workgroup per instance; profiles/r03/r03_small_batches.txt).
// dynamic LDS: bit image (mask_lds_bytes) | 16 flag words | decode scratch (column-scan block totals / polygon side stage)
// ------------------------------------------------------------------------------------------
constexpr int DNT = 512;
#ifndef LA3D_SPLIT_MAXB_NOMASK
#define LA3D_SPLIT_MAXB_NOMASK 160   // measured crossover with the instance engine (see split_eligible; round 6: grounded run lengths, split | instance: B = 128: 48 | 53, 192: 52.6 | 51.2, 256: 58 | 51)
#endif
constexpr int DEC_SCRATCH_BYTES = POLY_STAGE_BYTES > 8192 ? POLY_STAGE_BYTES : 8192;
template <int SRC>
__global__ __launch_bounds__(DNT) void scan_bits_kernel(const SplitParams sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const FitParams& p = sp.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int inst = sp.b0 + blockIdx.x;
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  unsigned* flags = reinterpret_cast<unsigned*>(smem + p.mask_lds_bytes);
  unsigned char* scratch = smem + p.mask_lds_bytes + 64;
  if (SRC == 1) {
    const long long o0 = p.rle_offsets[inst];
    (void)rle_to_bits<DNT>(p.rle_counts + o0, (int)(p.rle_offsets[inst + 1] - o0), bits, p.nwords, p.H, p.W, flags, tid,
                           reinterpret_cast<unsigned*>(scratch), DEC_SCRATCH_BYTES / 4);
  } else {
    (void)poly_to_bits<DNT>(p.poly_xy, p.poly_ring_off, p.poly_inst_rings[inst], p.poly_inst_rings[inst + 1],
                            reinterpret_cast<PolySide*>(scratch), flags, bits, p.nwords, p.H, p.W, tid);
  }
  __syncthreads();
  const int ntx = p.ntx, nty = p.nty, H = p.H;
  for (int band = wave; band < sp.nband; band += DNT / 64) {
    const int trow0 = band * BAND_TROWS;
    const int trows = min(BAND_TROWS, nty - trow0);
    const int nslots = trows * ntx;
    const long long seg = (long long)inst * sp.nband + band;
    unsigned short* tl = sp.tlist + seg * sp.tpb;
    uint4* tb = reinterpret_cast<uint4*>(sp.tbits + seg * sp.tpb * 8);
    int base = 0, nm = 0;
    for (int s0 = 0; s0 < nslots; s0 += 64) {   // ascending slot order, like scan_kernel's compaction
      const int sl = s0 + lane;
      unsigned w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      unsigned any = 0;
      int ty = 0, tx = 0;
      if (sl < nslots) {
        ty = sl / ntx; tx = sl - ty * ntx;
        const int row0 = (trow0 + ty) * 8;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          if (row0 + rr < H) w[rr] = bits[(row0 + rr) * ntx + tx];
          any |= w[rr];
          nm += __popc(w[rr]);
        }
      }
      const unsigned long long bal = __ballot(any != 0);
      if (any) {
        const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
        tl[at] = (unsigned short)(((trow0 + ty) << 8) | tx);
        tb[at * 2] = make_uint4(w[0], w[1], w[2], w[3]);
        tb[at * 2 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
      }
      base += __popcll(bal);
    }
    nm = wave_sum_i(nm);
    if (lane == 0) { sp.tcount[seg] = base; sp.nmaskb[seg] = nm; }
  }
}

// ------------------------------------------------------------------------------------------
// plan: one workgroup per sub-batch — exclusive prefix of the per-segment tile counts (toff) and, for every
// wave of the walking grid, the segment / index where its equal share of the concatenated list starts
// ------------------------------------------------------------------------------------------
constexpr int PNT = 1024;
__global__ __launch_bounds__(PNT) void plan_kernel(const SplitParams sp) {
  __shared__ int prefix[MAX_SEG + 1];
  __shared__ int wtot[PNT / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
  const int nseg = sp.nb * sp.nband;
  const int* cnt = sp.tcount + (long long)sp.b0 * sp.nband;
  for (int i = tid; i < sp.nb; i += PNT) geo_one(sp.f, sp.b0 + i);   // this sub-batch's geometry: read by the walks that follow
  for (int i = tid; i < nseg; i += PNT) prefix[i] = cnt[i];
  __syncthreads();
  const int per = (nseg + PNT - 1) / PNT;
  const int s0 = tid * per;
  int local = 0;
  for (int i = 0; i < per; ++i)
    if (s0 + i < nseg) local += prefix[s0 + i];
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  int run = base + incl - local;
  for (int i = 0; i < per; ++i)
    if (s0 + i < nseg) { const int cv = prefix[s0 + i]; prefix[s0 + i] = run; run += cv; }
  if (tid == PNT - 1) prefix[nseg] = run;
  __syncthreads();
  for (int i = tid; i <= nseg; i += PNT) sp.toff[i] = prefix[i];
  const int T = prefix[nseg];
  const int q = T > 0 ? (T + sp.nwaves - 1) / sp.nwaves : 0;
  for (int gw = tid; gw < sp.nwaves; gw += PNT) {
    int seg = -1, k = 0;
    const int tbeg = gw * q;
    if (q > 0 && tbeg < T) {
      int lo = 0, hi = nseg - 1;
      while (lo < hi) {  // last segment whose prefix is <= tbeg
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= tbeg) lo = mid; else hi = mid - 1;
      }
      seg = lo; k = tbeg - prefix[lo];
    }
    sp.wstart[2 * gw] = seg;
    sp.wstart[2 * gw + 1] = k;
  }
}

// ------------------------------------------------------------------------------------------
// the balanced walk shared by moments (PASS 0) and extents (PASS 1)
// ------------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(SNT, 4) void walk_kernel(const SplitParams sp) {
  const FitParams& p = sp.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
  const int gw = blockIdx.x * SNW + wave;
  const int nseg = sp.nb * sp.nband;
  const int* __restrict__ toff = sp.toff;
  const int T = toff[nseg];
  int fs = sp.wstart[2 * gw];
  if (T == 0 || fs < 0) return;
  int fk = sp.wstart[2 * gw + 1];
  const int q = (T + sp.nwaves - 1) / sp.nwaves;  // tiles per wave
  int ft = gw * q;
  const int tend = min(T, ft + q);

  const int W = p.W, H = p.H;
  const int r = lane >> 3, cq = lane & 7;
  double sv[6] = {0, 0, 0, 0, 0, 0};
  int n = 0;
  int cur = -1;       // instance (relative to b0) being accumulated
  bool live = false;  // PASS 1: instance has status OK
  double A0[3] = {0, 0, 0}, A1[3] = {0, 0, 0}, A2[3] = {0, 0, 0};
  double* part = (PASS == 0 ? sp.partA : sp.partB);

  auto flush = [&]() {
    if (cur < 0) return;
    double* o = part + (long long)(gw + cur) * PSTRIDE;
    if (PASS == 0) {
      const double r0 = wave_sum(sv[0]), r1 = wave_sum(sv[1]), r2 = wave_sum(sv[2]), r3 = wave_sum(sv[3]),
                   r4 = wave_sum(sv[4]);
      const int rc = wave_sum_i(n);
      if (lane == 0) { o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = (double)rc; }
    } else {
      const double r0 = wave_min(sv[0]), r1 = wave_max(sv[1]), r2 = wave_min(sv[2]), r3 = wave_max(sv[3]),
                   r4 = wave_min(sv[4]), r5 = wave_max(sv[5]);
      if (lane == 0) { o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; }
    }
  };

  // Two-stage software pipeline over batches of up to 4 consecutive entries of ONE segment:
  //   stage F: ids + mask words of batch i+2 requested (addresses are arithmetic: segment base + index),
  //   stage D: depth quads of batch i+1 requested (needs that batch's ids + words),
  //   stage C: batch i computed.
  // so neither the L2 round trip for the entries nor the HBM/MALL latency of the depth lines is exposed.
  const long long seg0 = (long long)sp.b0 * sp.nband;
  int seglen = toff[fs + 1] - toff[fs];
  auto plan = [&]() -> int {
    if (ft >= tend) return 0;
    while (fk >= seglen) { ++fs; fk = 0; seglen = toff[fs + 1] - toff[fs]; }
    return min(min(4, seglen - fk), tend - ft);
  };
  // stage F registers
  unsigned idF[4] = {0, 0, 0, 0}, wF[4] = {0, 0, 0, 0};
  int mF = 0, sF = 0;
  auto fetchF = [&]() {
    mF = plan();
    sF = fs;
    if (mF > 0) {
      const unsigned short* lst = sp.tlist + (seg0 + fs) * sp.tpb + fk;
      const unsigned* wb = sp.tbits + ((seg0 + fs) * sp.tpb + fk) * 8 + r;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        idF[g] = 0; wF[g] = 0;
        if (g < mF) { idF[g] = lst[g]; wF[g] = wb[g * 8]; }
      }
      fk += mF; ft += mF;
    }
  };
  // stage D registers
  unsigned nibD[4] = {0, 0, 0, 0};
  int txD[4] = {0, 0, 0, 0}, tyD[4] = {0, 0, 0, 0};
  uint4 dqD[4];
  int mD = 0, sD = 0;
  const float* dplD = nullptr;
  int instD = -1;
  auto fetchD = [&]() {  // consume stage F -> issue depth loads
    mD = mF; sD = sF;
    if (mD <= 0) return;
    const int irel = sD / sp.nband;
    if (irel != instD) {
      instD = irel;
      const int inst = sp.b0 + irel;
      const int img = p.image_index ? p.image_index[inst] : inst;
      dplD = p.depth + (long long)img * p.depth_plane_stride;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned tt = __builtin_amdgcn_readfirstlane(idF[g]);
      txD[g] = (int)(tt & 0xffu); tyD[g] = (int)(tt >> 8);
      nibD[g] = (g < mD && tyD[g] * 8 + r < H) ? (wF[g] >> (cq * 4)) & 0xFu : 0u;
      dqD[g] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (nibD[g]) dqD[g] = *reinterpret_cast<const uint4*>(dplD + (long long)(tyD[g] * 8 + r) * W + txD[g] * 32 + cq * 4);
  };

  fetchF();      // batch 0 entries
  fetchD();      // batch 0 depth
  fetchF();      // batch 1 entries
  while (mD > 0) {
    // move stage D to the compute registers
    const int m = mD, scur = sD;
    unsigned nib[4];
    int txs[4], tys[4];
    uint4 dq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { nib[g] = nibD[g]; txs[g] = txD[g]; tys[g] = tyD[g]; dq[g] = dqD[g]; }
    fetchD();    // next batch's depth loads in flight during this batch's math
    fetchF();    // and the entries of the one after

    const int irel = scur / sp.nband;
    if (irel != cur) {
      flush();
      cur = irel;
      const int inst = sp.b0 + irel;
      const double* geo = p.geo + (long long)inst * GEO_D;
      if (PASS == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { A0[j] = geo[j]; A2[j] = geo[6 + j]; }
        sv[0] = sv[1] = sv[2] = sv[3] = sv[4] = 0;
        n = 0;
        live = true;
      } else {
        const double* ax = sp.axis + (long long)inst * 4;
        const double cy = ax[0], sy = ax[1];
        live = ax[2] == 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // rows of rotate_y(yaw) @ M  (reference :154)
          A0[j] = cy * geo[j] + sy * geo[6 + j];
          A1[j] = geo[3 + j];
          A2[j] = -sy * geo[j] + cy * geo[6 + j];
        }
        sv[0] = sv[2] = sv[4] = INFINITY; sv[1] = sv[3] = sv[5] = -INFINITY;
      }
    }
    if (!live) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g >= m) break;
      const unsigned db[4] = {dq[g].x, dq[g].y, dq[g].z, dq[g].w};
      const double vd = (double)(tys[g] * 8 + r), ud = (double)(txs[g] * 32 + cq * 4);
      const double r0 = fma(A0[0], ud, fma(A0[1], vd, A0[2]));
      const double r2 = fma(A2[0], ud, fma(A2[1], vd, A2[2]));
      double r1 = 0;
      if (PASS == 1) r1 = fma(A1[0], ud, fma(A1[1], vd, A1[2]));
      quad_math<PASS>(nib[g], db, r0, r1, r2, A0[0], A1[0], A2[0], sv, &n);
    }
  }
  flush();
}

// slots of instance irel: waves floor(off/q) .. floor((off+cnt-1)/q), slot id = wave + irel
__device__ inline void slot_range(const SplitParams& sp, int irel, int* first, int* count, int* ntiles) {
  const int nseg = sp.nb * sp.nband;
  const int T = sp.toff[nseg];
  const int off = sp.toff[irel * sp.nband], cnt = sp.toff[(irel + 1) * sp.nband] - off;
  *ntiles = cnt;
  if (cnt == 0 || T == 0) { *first = 0; *count = 0; return; }
  const int q = (T + sp.nwaves - 1) / sp.nwaves;
  const int w0 = off / q, w1 = (off + cnt - 1) / q;
  *first = w0 + irel;
  *count = w1 - w0 + 1;
}

// one wave per instance: moments partials -> status / yaw axis / aux   (fixed summation order)
__global__ __launch_bounds__(SNT) void axis_kernel(const SplitParams sp) {
  const FitParams& p = sp.f;
  const int lane = threadIdx.x & 63;
  const int irel = blockIdx.x * SNW + (threadIdx.x >> 6);
  if (irel >= sp.nb) return;
  const int inst = sp.b0 + irel;
  int first, count, ntiles;
  slot_range(sp, irel, &first, &count, &ntiles);
  double s[5] = {0, 0, 0, 0, 0};
  double nd = 0;
  for (int j0 = 0; j0 < count; j0 += 64) {  // lanes take consecutive slots; butterfly sum per chunk, chunks in order
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (j0 + lane < count) {
      const double* o = sp.partA + (long long)(first + j0 + lane) * PSTRIDE;
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] += wave_sum(v[k]);
    nd += wave_sum(v[5]);
  }
  int nm = 0;
  for (int j = lane; j < sp.nband; j += 64) nm += sp.nmaskb[inst * sp.nband + j];
  nm = wave_sum_i(nm);
  // (every lane holds the sums: status and axis are computed by all of them, so that the rare second pass below is wave-uniform)
  const double* geo = p.geo + (long long)inst * GEO_D;
  const int n = (int)nd;
  int st = LA3D_BOX_OK;
  if (geo[18] != 0.0) st = LA3D_BOX_BAD_GROUND;
  else if (n == 0) st = LA3D_BOX_EMPTY;
  else if (n == 1) st = LA3D_BOX_TOO_FEW;
  double cy = NAN, sy = NAN, gap = NAN;
  if (st == LA3D_BOX_OK && axis_from_sums((double)n, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap)) {
    // ill-conditioned raw sums (axis_from_sums; round 6): the instance's tiles once more by this wave, the moments about the mean -
    // one tile per step, the checked pixel math (rare: a footprint far thinner than its distance)
    const double px0 = s[0] / (double)n, pz0 = s[1] / (double)n;
    const double A0[3] = {geo[0], geo[1], geo[2]}, A2[3] = {geo[6], geo[7], geo[8]};
    const int img = p.image_index ? p.image_index[inst] : inst;
    const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
    const int r = lane >> 3, cq = lane & 7;
    const long long seg0 = (long long)sp.b0 * sp.nband;
    double t[5] = {0, 0, 0, 0, 0};
    int cnt = 0;
    for (int b = 0; b < sp.nband; ++b) {
      const int seg = irel * sp.nband + b;
      const int len = sp.toff[seg + 1] - sp.toff[seg];
      const unsigned short* lst = sp.tlist + (seg0 + seg) * sp.tpb;
      const unsigned* wb = sp.tbits + (seg0 + seg) * sp.tpb * 8 + r;
#pragma clang loop unroll(disable)
      for (int k = 0; k < len; ++k) {
        const unsigned tt = __builtin_amdgcn_readfirstlane((unsigned)lst[k]);
        const int tx = (int)(tt & 0xffu), ty = (int)(tt >> 8);
        const int row = ty * 8 + r;
        const unsigned nib = row < p.H ? (wb[k * 8] >> (cq * 4)) & 0xFu : 0u;
        if (nib) {
          const uint4 dq = *reinterpret_cast<const uint4*>(dpl + (long long)row * p.W + tx * 32 + cq * 4);
          const unsigned db[4] = {dq.x, dq.y, dq.z, dq.w};
          const double vd = (double)row, ud = (double)(tx * 32 + cq * 4);
          quad_math<0, true, false, true>(nib, db, fma(A0[0], ud, fma(A0[1], vd, A0[2])), 0.0, fma(A2[0], ud, fma(A2[1], vd, A2[2])),
                                          A0[0], 0.0, A2[0], t, &cnt, px0, pz0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = wave_sum(t[k]);
    if (axis_from_sums((double)n, t[0], t[1], t[2], t[3], t[4], &cy, &sy, &gap)) gap = 0.0;   // no spread at all: unresolved
  }
  if (lane == 0) {
    double* ax = sp.axis + (long long)inst * 4;
    ax[0] = cy; ax[1] = sy; ax[2] = (double)st; ax[3] = 0;
    if (p.aux) {
      double* a = p.aux + (long long)inst * LA3D_AUX;
      a[0] = atan2(sy, cy); a[1] = (double)n; a[2] = (double)nm; a[3] = gap;
    }
    p.status[inst] = st;
    if (st != LA3D_BOX_OK) write_nan_box(p.out + (long long)inst * LA3D_REC);
  }
}

// one wave per instance: extent partials -> the 39-double record
__global__ __launch_bounds__(SNT) void final_kernel(const SplitParams sp) {
  const FitParams& p = sp.f;
  const int lane = threadIdx.x & 63;
  const int irel = blockIdx.x * SNW + (threadIdx.x >> 6);
  if (irel >= sp.nb) return;
  const int inst = sp.b0 + irel;
  const double* ax = sp.axis + (long long)inst * 4;
  if (ax[2] != 0.0) return;
  int first, count, ntiles;
  slot_range(sp, irel, &first, &count, &ntiles);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int j = lane; j < count; j += 64) {
    const double* o = sp.partB + (long long)(first + j) * PSTRIDE;
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = fmin(lo[k], o[2 * k]); hi[k] = fmax(hi[k], o[2 * k + 1]); }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] = wave_min(lo[k]); hi[k] = wave_max(hi[k]); }
  write_box_wave(p.out + (long long)inst * LA3D_REC, p.geo + (long long)inst * GEO_D + 9, ax[0], ax[1], lo[0], hi[0], lo[1],
                 hi[1], lo[2], hi[2], lane);
}

// ---- host side --------------------------------------------------------------------------------------
struct Layout {
  size_t geo, tbits, tcount, nmaskb, tlist, axis, sub, total;
  size_t sub_toff, sub_wstart, sub_partA, sub_partB, sub_size;
  int nband, tpb, ntx, nty, nwaves;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

int walk_waves(int nb) {
  // grid of the walking kernels: the walker holds two batches in flight (~105 VGPRs -> 4 waves/SIMD), so
  // 4 workgroups of 4 waves fill a CU; one resident round = 1024 workgroups
  int g = nb * 4;
  if (g < 256) g = 256;
  if (g > 1024) g = 1024;
  if (config().split_grid > 0) g = config().split_grid;
  return g * SNW;
}

Layout make_layout(int B, int H, int W) {
  Layout L;
  L.ntx = W / 32; L.nty = (H + 7) / 8;
  L.nband = (L.nty + BAND_TROWS - 1) / BAND_TROWS;
  L.tpb = BAND_TROWS * L.ntx;
  L.nwaves = walk_waves(B);
  size_t o = 0;
  L.geo = o; o += al((size_t)B * GEO_D * 8);
  L.tbits = o; o += al((size_t)B * L.nband * L.tpb * 8 * 4);
  L.tcount = o; o += al((size_t)B * L.nband * 4);
  L.nmaskb = o; o += al((size_t)B * L.nband * 4);
  L.tlist = o; o += al((size_t)B * L.nband * L.tpb * 2);
  L.axis = o; o += al((size_t)B * 4 * 8);
  L.sub_toff = 0;
  L.sub_wstart = al(((size_t)B * L.nband + 1) * 4);
  L.sub_partA = L.sub_wstart + al((size_t)L.nwaves * 2 * 4);
  L.sub_partB = L.sub_partA + al(((size_t)L.nwaves + B) * PSTRIDE * 8);
  L.sub_size = L.sub_partB + al(((size_t)L.nwaves + B) * PSTRIDE * 8);
  L.sub = o; o += L.sub_size * MAX_SUB;
  L.total = o;
  return L;
}

struct Streams {
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, scan_done[MAX_SUB] = {}, join[2] = {nullptr, nullptr};
  int device = -1;
  bool ok = false;
  hipError_t err = hipSuccess;
};

Streams& streams_for_current_device() {
  static thread_local Streams st[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  Streams& s = st[dev & 15];
  if (!s.ok || s.device != dev) {
    hipError_t first = hipSuccess;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    (void)hipGetLastError();   // a stale error of an earlier, unrelated call must not be mistaken for ours
    for (int i = 0; i < 2; ++i) chk(hipStreamCreateWithFlags(&s.aux[i], hipStreamNonBlocking));
    chk(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
    for (int i = 0; i < MAX_SUB; ++i) chk(hipEventCreateWithFlags(&s.scan_done[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) chk(hipEventCreateWithFlags(&s.join[i], hipEventDisableTiming));
    s.device = dev;
    s.ok = first == hipSuccess;
    s.err = first;
  }
  return s;
}

}  // namespace

size_t split_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || W % 32 != 0) return 0;
  return make_layout(B, H, W).total;
}

bool split_eligible(const FitParams& p, bool vec, bool ldsmask) {
  (void)ldsmask;
  if (!vec || p.sample_idx != nullptr || p.W % 32 != 0) return false;
  const int ntx = p.W / 32, nty = (p.H + 7) / 8;
  if (ntx > 64 || nty > 255) return false;  // scan keeps a band's bits in LDS: 8 tile rows x ntx x 32 B <= 16 KB
  const int nband = (nty + BAND_TROWS - 1) / BAND_TROWS;
  if (nband > 64) return false;
  // Measured on MI355X, round 3 (BASELINE config-2 inputs, us per call, split vs one workgroup per instance;
  // profiles/r03/r03_small_batches.txt): u8 planes B = 1 / 16 / 64 / 128 / 192 / 272 / 288 / 304 / 336: 34 / 37 / 45 / 50 / 58 / 70 / 74 / 76 /
  // 81 vs 40 / 52 / 57 / 58 / 62 / 75 / 72 / 71 / 76; run lengths B = 1 / 16 / 64 / 160 / 256 / 288 / 304: 36 / 40 / 46 / 53 / 59 / 65 / 66 vs
  // 47 / 57 / 59 / 66 / 64 / 66 / 65 (polygons ~1 us below both).  A lone 60-90 k-px instance keeps ONE CU's fp64 VALU busy for ~40 us
  // in the instance engine; the split engine spreads its tiles over the chip at the price of six dependent launches.
  const int e = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;   // per-call pin, else the process default
  if (e == LA3D_ENGINE_INSTANCE) return false;
  if (p.mask == nullptr) {   // run lengths / polygon parts (scan_bits_kernel): the bit image must fit the decode workgroup's LDS
    if (p.mask_lds_bytes <= 0 || (size_t)p.mask_lds_bytes + 64 + DEC_SCRATCH_BYTES > 160 * 1024 - 256 || p.filter_boundary >= 0) return false;
    if (e == LA3D_ENGINE_SPLIT) return true;
    return p.B <= LA3D_SPLIT_MAXB_NOMASK;
  }
  if (e == LA3D_ENGINE_SPLIT) return true;
  // round 6: u8 planes up to 160 instances only (where the band engine does not apply: it is the faster one there); above, one
  // workgroup per instance leads - grounded u8, split | instance: B = 192: 56.8 | 54.2, 256: 66.7 | 56.9 (profiles/r06/r06_engines_by_batch.txt)
  return p.B <= 160;
}

// One call = one batch.  Sub-batch j: scan on the scan stream (scans are bandwidth-bound, so they run
// back to back), then moments -> axis -> extents -> final on pass stream j % 2 once that scan is done.
int split_fit(const FitParams& pin, void* workspace, hipStream_t s) {
  Streams& st = streams_for_current_device();
  if (!st.ok) {
    snprintf(g_err, sizeof(g_err), "la3d split engine: could not create internal streams/events (%s)", hipGetErrorString(st.err));
    return LA3D_ERR_HIP;
  }
  const int B = pin.B;
  Layout L = make_layout(B, pin.H, pin.W);
  char* ws = static_cast<char*>(workspace);
  SplitParams sp;
  sp.f = pin;
  sp.f.ntx = L.ntx; sp.f.nty = L.nty;
  sp.f.geo = reinterpret_cast<double*>(ws + L.geo);
  sp.nband = L.nband; sp.tpb = L.tpb;
  sp.tbits = reinterpret_cast<unsigned*>(ws + L.tbits);
  sp.tcount = reinterpret_cast<int*>(ws + L.tcount);
  sp.nmaskb = reinterpret_cast<int*>(ws + L.nmaskb);
  sp.tlist = reinterpret_cast<unsigned short*>(ws + L.tlist);
  sp.axis = reinterpret_cast<double*>(ws + L.axis);

  // sub-batches: as many as keep >= 128 instances each and <= MAX_SEG segments, at most MAX_SUB
  int nsub = B / 256;
  if (nsub < 1) nsub = 1;
  if (nsub > 4) nsub = 4;
  if (config().split_sub > 0) nsub = config().split_sub;
  while ((B + nsub - 1) / nsub * L.nband > MAX_SEG && nsub < MAX_SUB) ++nsub;
  if (nsub > MAX_SUB) nsub = MAX_SUB;
  if ((B + nsub - 1) / nsub * L.nband > MAX_SEG) {
    set_err("la3d split engine: batch too large for one call (split the batch)");
    return LA3D_ERR_UNSUPPORTED;
  }
  const bool single = nsub == 1;
  hipStream_t scan_s = s;
  if (!single) {
    if (hipEventRecord(st.fork, s) != hipSuccess) return LA3D_ERR_HIP;
    for (int i = 0; i < 2; ++i)
      if (hipStreamWaitEvent(st.aux[i], st.fork, 0) != hipSuccess) return LA3D_ERR_HIP;
  }
  const int per = (B + nsub - 1) / nsub;
  for (int j = 0; j < nsub; ++j) {
    sp.b0 = j * per;
    sp.nb = (sp.b0 + per <= B) ? per : B - sp.b0;
    if (sp.nb <= 0) break;
    sp.nwaves = walk_waves(sp.nb);
    char* sub = ws + L.sub + (size_t)j * L.sub_size;
    sp.toff = reinterpret_cast<int*>(sub + L.sub_toff);
    sp.wstart = reinterpret_cast<int*>(sub + L.sub_wstart);
    sp.partA = reinterpret_cast<double*>(sub + L.sub_partA);
    sp.partB = reinterpret_cast<double*>(sub + L.sub_partB);
    if (sp.f.mask == nullptr) {   // run lengths / polygon parts: decode front end, one workgroup per instance
      const size_t dec_lds = (size_t)sp.f.mask_lds_bytes + 64 + DEC_SCRATCH_BYTES;
      if (dec_lds > 64 * 1024) {   // frames above ~690 k px: allow the large dynamic LDS (per kernel and device; cheap, rare)
        const void* fn = sp.f.poly_xy ? reinterpret_cast<const void*>(scan_bits_kernel<2>) : reinterpret_cast<const void*>(scan_bits_kernel<1>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
      }
      if (sp.f.poly_xy) hipLaunchKernelGGL(scan_bits_kernel<2>, dim3(sp.nb), dim3(DNT), dec_lds, scan_s, sp);
      else hipLaunchKernelGGL(scan_bits_kernel<1>, dim3(sp.nb), dim3(DNT), dec_lds, scan_s, sp);
      if (int rc = check_launch("scan_bits_kernel")) return rc;
    } else {
      const size_t scan_lds = (size_t)sp.tpb * 34 + 64 * 4 + SNW * 4 + 16;
      hipLaunchKernelGGL(scan_kernel, dim3(sp.nb * sp.nband), dim3(SNT), scan_lds, scan_s, sp);
      if (int rc = check_launch("scan_kernel")) return rc;
    }
    hipStream_t ps = scan_s;
    if (!single) {
      ps = st.aux[j & 1];
      if (hipEventRecord(st.scan_done[j], scan_s) != hipSuccess) return LA3D_ERR_HIP;
      if (hipStreamWaitEvent(ps, st.scan_done[j], 0) != hipSuccess) return LA3D_ERR_HIP;
    }
    const int grid = sp.nwaves / SNW;
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PNT), 0, ps, sp);
    hipLaunchKernelGGL(walk_kernel<0>, dim3(grid), dim3(SNT), 0, ps, sp);
    hipLaunchKernelGGL(axis_kernel, dim3((sp.nb + SNW - 1) / SNW), dim3(SNT), 0, ps, sp);
    hipLaunchKernelGGL(walk_kernel<1>, dim3(grid), dim3(SNT), 0, ps, sp);
    hipLaunchKernelGGL(final_kernel, dim3((sp.nb + SNW - 1) / SNW), dim3(SNT), 0, ps, sp);
    if (int rc = check_launch("split passes")) return rc;
  }
  if (!single)
    for (int i = 0; i < 2; ++i) {
      if (hipEventRecord(st.join[i], st.aux[i]) != hipSuccess) return LA3D_ERR_HIP;
      if (hipStreamWaitEvent(s, st.join[i], 0) != hipSuccess) return LA3D_ERR_HIP;
    }
  return LA3D_SUCCESS;
}

}  // namespace la3d
