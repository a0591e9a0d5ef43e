// la3d_walks.hpp - the walks over an instance's pixels shared by the fit engines (device code, included by la3d_instance.hip / la3d_band.hip /
// la3d_rows.hip): the generic row-linear walk, the tiled two-pass walk (pass-B culling, LDS-kept tiles) and the separable single pass.
#pragma once
#include "la3d_device.hpp"

using namespace la3d;

namespace {
// Generic walk (any W, unaligned planes, frames whose bit image does not fit LDS): row-linear chunks of
// 256 pixels per wave, 4 per lane.  PASS 0: count + moments of (x', z').  PASS 1: extents of all three
// axes in the yaw frame.  A0/A1/A2 are the rows mapping [u,v,1] to the ray components: PASS 0 uses rows 0
// and 2 of M; PASS 1 uses N0, M row 1, N2.
// PIV (pass 0): the moments about pivot[0 .. 2) - the re-run of an ill-conditioned instance (axis_from_sums).
template <bool VEC, bool LDSMASK, int PASS, bool PIV = false>
__device__ inline void sweep(const FitParams& p, const float* __restrict__ dpl, const unsigned char* __restrict__ mpl,
                             const unsigned* bits, const double* A0, const double* A1, const double* A2,
                             int wave, int lane, double* acc, int* cnt, int* nmask, const double* pivot = nullptr) {
  double px0 = 0, pz0 = 0;
  if constexpr (PIV) { px0 = pivot[0]; pz0 = pivot[1]; }
  const int HW = p.HW, W = p.W;
  const int nquads = (HW + 3) >> 2;
  const int nchunks = (nquads + 63) >> 6;
  const double a00 = A0[0], a01 = A0[1], a02 = A0[2];
  const double a20 = A2[0], a21 = A2[1], a22 = A2[2];
  double a10 = 0, a11 = 0, a12 = 0;
  if (PASS == 1) { a10 = A1[0]; a11 = A1[1]; a12 = A1[2]; }
  double s0 = acc[0], s1 = acc[1], s2 = acc[2], s3 = acc[3], s4 = acc[4];
  double xlo = acc[0], xhi = acc[1], ylo = acc[2], yhi = acc[3], zlo = acc[4], zhi = acc[5];
  int n = *cnt, nm = *nmask;
  for (int ch = wave; ch < nchunks; ch += NWAVE) {
    const int q = ch * 64 + lane;
    unsigned nib = 0;
    if (q < nquads) {
      if (LDSMASK) {
        nib = (bits[q >> 3] >> ((q & 7) * 4)) & 0xFu;
      } else {
        const int i0 = q * 4;
        if (VEC) {
          nib = nz4(*(const unsigned*)(mpl + i0));
        } else {
          for (int k = 0; k < 4; ++k)
            if (i0 + k < HW && mpl[i0 + k]) nib |= 1u << k;
        }
        if (PASS == 0 && !PIV) nm += __popc(nib);   // (the re-run counts the mask pixels no second time)
      }
    }
    if (__ballot(nib != 0) == 0) continue;  // wave-uniform skip: nothing of this 256-pixel chunk is masked
    if (nib) {
      const unsigned i0 = (unsigned)q * 4u;
      float dk[4];
      if (VEC) {
        const float4 t = *(const float4*)(dpl + i0);
        dk[0] = t.x; dk[1] = t.y; dk[2] = t.z; dk[3] = t.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) dk[k] = ((int)(i0 + k) < HW && ((nib >> k) & 1u)) ? dpl[i0 + k] : 0.f;
      }
      unsigned u0, v0;
      pix_uv(i0, W, p.rcpW, &u0, &v0);
      const double vd = (double)v0;
      const double b0 = fma(a01, vd, a02), b2 = fma(a21, vd, a22);
      double b1 = 0;
      if (PASS == 1) b1 = fma(a11, vd, a12);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = ((nib >> k) & 1u) && finite_f32(dk[k]);
        double r0, r1 = 0, r2;
        if (p.rows_aligned) {
          const double ud = (double)(u0 + k);
          r0 = fma(a00, ud, b0); r2 = fma(a20, ud, b2);
          if (PASS == 1) r1 = fma(a10, ud, b1);
        } else {
          unsigned uk, vk;
          pix_uv(i0 + k, W, p.rcpW, &uk, &vk);
          const double ud = (double)uk, vdk = (double)vk;
          r0 = fma(a00, ud, fma(a01, vdk, a02)); r2 = fma(a20, ud, fma(a21, vdk, a22));
          if (PASS == 1) r1 = fma(a10, ud, fma(a11, vdk, a12));
        }
        if (PASS == 0) {
          const double d = ok ? (double)dk[k] : 0.0;
          double x = d * r0, z = d * r2;
          if (PIV) { x = fma(d, r0, ok ? -px0 : 0.0); z = fma(d, r2, ok ? -pz0 : 0.0); }
          s0 += x; s1 += z;
          s2 = fma(x, x, s2); s3 = fma(x, z, s3); s4 = fma(z, z, s4);
          n += ok ? 1 : 0;
        } else {
          const double d = ok ? (double)dk[k] : (double)NAN;  // NaN is ignored by v_min/v_max_f64
          const double x = d * r0, y = d * r1, z = d * r2;
          xlo = dmin(xlo, x); xhi = dmax(xhi, x);
          ylo = dmin(ylo, y); yhi = dmax(yhi, y);
          zlo = dmin(zlo, z); zhi = dmax(zhi, z);
        }
      }
    }
  }
  if (PASS == 0) {
    acc[0] = s0; acc[1] = s1; acc[2] = s2; acc[3] = s3; acc[4] = s4;
    *cnt = n; *nmask = nm;
  } else {
    acc[0] = xlo; acc[1] = xhi; acc[2] = ylo; acc[3] = yhi; acc[4] = zlo; acc[5] = zhi;
  }
}

// TILED walk (W % 32 == 0): a wave owns one tile of 32 px x 8 rows per step — lane = (row r = lane>>3,
// quad cq = lane&7).  One bit-image word per tile row (broadcast to its 8 lanes), one full 128-B depth
// line per tile row, (u,v) from the tile coordinates without any division.  Only tiles on the
// compacted active list are visited.  Each wave takes TG consecutive list entries per step and issues
// all TG depth loads before computing (a single load per wave in flight leaves the walk bound by
// memory latency: ~2.5 us per tile under load).
// Branch-free pixel math: validity (mask bit AND finite depth) is a 0/-1 word; PASS 0 (moments) ANDs it
// into the depth bits (invalid -> +0.0 contributes nothing to the sums); PASS 1 (extents of all three
// axes) ORs its complement (invalid -> NaN, ignored by v_min/v_max_f64).
constexpr int TG = 4;   // tiles a wave takes per step: their depth loads are issued back to back

struct TileCtx {
  int W, H, ntx, r, cq;
  unsigned loff;   // byte offset of this lane's depth quad inside a tile: (r W + 4 cq) floats
  // compacted bit image (plain build): list entry e owns the eight row words of its tile at words [8e, 8e + 8) of the image
  // region, and the depth quads of list entries < keepn stay in the LDS that frees (1 KiB per tile) between the passes
  int compact, keepn;
  uint4* keep;
  // pass-B tile culling (plain build): pass A leaves the [min, max] of the valid depths of list entry e in rng[2e], rng[2e + 1]
  // (bit patterns: see tile_range); pass B then walks only the survivors, surv[j] = list entry
  unsigned* rng;
  const unsigned short* surv;
  double a00, a01, a02, a10, a11, a12, a20, a21, a22;
};

// list entry j of this walk -> tile coordinates (wave-uniform, in SGPRs)
template <bool SURV = false>
__device__ inline void tile_coords(const TileCtx& c, const unsigned short* list, bool dense, int j, int rev_base, int* tx, int* ty) {
  if (SURV) {
    const int e = __builtin_amdgcn_readfirstlane((int)c.surv[j]);
    const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[e]);
    *tx = (int)(t & 0xffu); *ty = (int)(t >> 8);
  }
  else if (dense) { *ty = j / c.ntx; *tx = j - *ty * c.ntx; }
  else {
    // rev_base >= 0: pass B walks the list backwards - the tiles pass A read last are re-read
    // first (L2 reuse; extents are order independent)
    const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[rev_base >= 0 ? rev_base - j : j]);
    *tx = (int)(t & 0xffu); *ty = (int)(t >> 8);
  }
}

// stage 1 of a step (TG consecutive list entries of one wave): bit-image nibbles, then all depth loads back to back.
// Returns the TG nibbles packed into one word.
// ZERO = false: dq is NOT cleared - lanes without a mask bit hold whatever their registers held (an empty asm statement
// "defines" the quad without an instruction).  That is harmless by construction: every consumer gates a quad through its
// nibble (pass A ANDs the validity word into the bits, pass B ORs its complement, tile_range does both), and saves four moves
// per tile.
template <int PASS, bool LK, bool SURV = false, bool ZERO = true, bool FULLC = false>
__device__ inline unsigned tile_fetch(const TileCtx& c, const float* __restrict__ dpl, const unsigned* bits,
                                      const unsigned short* list, int nsteps, bool dense, int j0, int rev_base, uint4* dq,
                                      int* tcs = nullptr) {   // tcs (SURV): the tiles' coordinates for tile_compute, which then need not look them up again
  unsigned nib[TG];
  int txs[TG], tys[TG], ent[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    const int j = j0 + g;
    nib[g] = 0; txs[g] = 0; tys[g] = 0; ent[g] = 0x7fffffff;
    if (ZERO) dq[g] = make_uint4(0u, 0u, 0u, 0u);
    else {
      u32x4 t;
      asm volatile("" : "=v"(t));
      dq[g] = make_uint4(t.x, t.y, t.z, t.w);
    }
    if (j < nsteps) {
      if (SURV) {   // survivor j -> list entry -> tile: two dependent LDS reads, done once per tile
        ent[g] = __builtin_amdgcn_readfirstlane((int)c.surv[j]);
        const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[ent[g]]);
        txs[g] = (int)(t & 0xffu); tys[g] = (int)(t >> 8);
        if (tcs) tcs[g] = (int)t;
      } else {
        tile_coords<false>(c, list, dense, j, rev_base, &txs[g], &tys[g]);
      }
      if (FULLC) {   // a tile of the list's first class: every pixel is a mask pixel - no row word to fetch (round 6)
        ent[g] = j;
        nib[g] = 0xFu;
      } else if (LK && c.compact) {   // uniform
        if (!SURV) ent[g] = rev_base >= 0 ? rev_base - j : j;
        nib[g] = (bits[ent[g] * 8 + c.r] >> (c.cq * 4)) & 0xFu;   // rows past the frame were stored as zeros
      } else {
        const int row = tys[g] * 8 + c.r;
        if (row < c.H) nib[g] = (bits[row * c.ntx + txs[g]] >> (c.cq * 4)) & 0xFu;
      }
    }
  }
  unsigned pk = 0;
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    if (LK && PASS == 1 && ent[g] < c.keepn) dq[g] = c.keep[ent[g] * 64 + (c.r * 8 + c.cq)];   // kept by pass A
    else if (nib[g]) {
      // uniform tile origin (scalar registers) + the lane's constant byte offset: the load takes its address as SGPR base + VGPR offset
      const float* tp = dpl + ((long long)(tys[g] * 8) * c.W + txs[g] * 32);
      dq[g] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(tp) + c.loff);
    }
    pk |= nib[g] << (4 * g);
  }
  return pk;
}

// Depth range of one tile for pass-B culling: [min, max] over the VALID pixels of the wave's tile, as bit patterns.  Non-negative
// floats order like unsigned integers, so the minimum is an unsigned min over (bits | ~valid) (invalid -> 0xffffffff) and the
// maximum an unsigned max over (bits & valid) (invalid -> 0).  A negative, infinite or NaN depth makes the maximum >= 0x7f800000,
// which cull_bound1's caller reads as "unbounded: never cull".  Six DPP steps per value leave the wave's result in lane 63, which stores it.
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
template <bool CHK, bool FULLC = false>
__device__ inline void tile_range(const TileCtx& c, int e, unsigned nib, const unsigned* db) {
  // per pixel: the validity word m (0 / -1) and db & m are the pixel math's own values (same expressions: shared after inlining);
  // db | ~m is one v_bfi_b32 (m ? db : ones).  The cross-lane steps carry the operation's identity as `old`, which lets the
  // compiler fold every move into its min / max (v_min_u32_dpp: one instruction per step instead of three).
  unsigned w[4], v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (FULLC && !CHK) { v[k] = db[k]; w[k] = db[k]; continue; }   // every pixel valid: nothing to gate
    int m = -(int)((nib >> k) & 1u);
    if (CHK) m = (((int)(db[k] & 0x7fffffffu) - 0x7f800000) >> 31) & m;
    v[k] = db[k] & (unsigned)m;
    asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(w[k]) : "v"(m), "v"(db[k]));
  }
  unsigned lo = min(min(w[0], w[1]), min(w[2], w[3])), hi = max(max(v[0], v[1]), max(v[2], v[3]));
  lo = min(lo, (unsigned)dpp_i32<DPP_XOR1>((int)lo)); hi = max(hi, (unsigned)dpp_i32<DPP_XOR1>((int)hi));
  lo = min(lo, (unsigned)dpp_i32<DPP_XOR2>((int)lo)); hi = max(hi, (unsigned)dpp_i32<DPP_XOR2>((int)hi));
  lo = min(lo, (unsigned)dpp_i32<DPP_HALF_MIRROR>((int)lo)); hi = max(hi, (unsigned)dpp_i32<DPP_HALF_MIRROR>((int)hi));
  lo = min(lo, (unsigned)dpp_i32<DPP_MIRROR>((int)lo)); hi = max(hi, (unsigned)dpp_i32<DPP_MIRROR>((int)hi));
  // rows 1 and 3 take in lane 15 of the row before them, then rows 2 and 3 lane 31: row 3 holds the wave's result
  lo = min(lo, (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lo, DPP_ROW_BCAST15, 0xa, 0xf, false));
  hi = max(hi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, DPP_ROW_BCAST15, 0xa, 0xf, false));
  lo = min(lo, (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lo, DPP_ROW_BCAST31, 0xc, 0xf, false));
  hi = max(hi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, DPP_ROW_BCAST31, 0xc, 0xf, false));
  if (c.r * 8 + c.cq == 63) *reinterpret_cast<uint2*>(c.rng + 2 * e) = make_uint2(lo, hi);
}

// stage 2: the pixel math of a step on quads dq / nibbles pk (all lanes; unmasked lanes carry zeros / NaNs)
template <int PASS, bool CHK, bool LK = false, bool SURV = false, bool RNG = false, bool SPEC = false, bool FULLC = false>
__device__ inline void tile_compute(const TileCtx& c, const unsigned short* list, int nsteps, bool dense, int j0, int rev_base,
                                    const uint4* dq, unsigned pk, double* sv, int* n, const int* tcs = nullptr) {
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    const int j = j0 + g;
    if (j >= nsteps) continue;   // wave-uniform
    if (LK && PASS == 0 && j < c.keepn) c.keep[j * 64 + (c.r * 8 + c.cq)] = dq[g];   // (pass A walks the list forwards: entry = j)
    const unsigned nib = FULLC ? 0xFu : (pk >> (4 * g)) & 0xFu;
    if (!FULLC && dense && __ballot(nib != 0) == 0) continue;
    int tx, ty;
    if (SURV && tcs) { tx = tcs[g] & 0xff; ty = tcs[g] >> 8; }
    else tile_coords<SURV>(c, list, dense, j, rev_base, &tx, &ty);
    const unsigned db[4] = {dq[g].x, dq[g].y, dq[g].z, dq[g].w};
    const double vd = (double)(ty * 8 + c.r), ud = (double)(tx * 32 + c.cq * 4);
    const double r0 = fma(c.a00, ud, fma(c.a01, vd, c.a02));
    const double r2 = fma(c.a20, ud, fma(c.a21, vd, c.a22));
    double r1 = 0;
    if (PASS == 1) r1 = fma(c.a10, ud, fma(c.a11, vd, c.a12));
    quad_math<PASS, CHK, SPEC>(nib, db, r0, r1, r2, c.a00, c.a10, c.a20, sv, n);
    if (RNG && PASS == 0) tile_range<CHK, FULLC>(c, j, nib, db);
  }
}

// words of the image region that pass-B culling takes behind the compacted entries: the depth ranges (two words per tile; the
// survivor list overwrites them later), 16-byte granules, then CULL_SCRATCH_WORDS for the champion search of cull_plan
// (kept out of `Shared`: every byte there comes off the tile list's capacity, i.e. off the mask size up to which the plain
// and the no-cull build group their partial sums alike)
constexpr int CULL_SCRATCH_WORDS = NWAVE * 6 * 2;   // per wave and direction: value (f32), list entry (u32)
__device__ inline int cull_rng_words(int nactive) { return ((2 * nactive + 3) & ~3) + CULL_SCRATCH_WORDS; }

// RNG (pass A, plain build, compact image): also leave every tile's depth range for pass-B culling (rng_words > 0 then).
// nsurv >= 0 (pass B): walk only the culling survivors surv[0 .. nsurv) (fit_instances_kernel builds the list).
// SPEC: the un-grounded, skew-free forms of the pixel math (quad_math): the caller checks M's row 2 == (0, 0, 1) for pass A,
// M[1][0] == 0 for pass B.
template <int PASS, bool CHK, bool RNG = false, bool SPEC = false>
__device__ inline void sweep_tiled(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits,
                                   const unsigned short* list, int nactive, const double* A0, const double* A1,
                                   const double* A2, int wave, int lane, double* acc, int* cnt,
                                   unsigned* qhead = nullptr, int compact = 0, int rng_words = 0, int nsurv = -1, int nfull = 0) {
  // nfull (pass A, compact image): list entries [0, nfull) are tiles completely inside the mask - walked by a loop of their own whose
  // body fetches no row words and gates no pixel (round 6: what the separable pass does, for the calls that need two passes)
  constexpr bool LK = true;   // (the compact image / LDS-kept tiles / survivor walk apply whenever the caller hands over `compact`)
  TileCtx c;
  c.W = p.W; c.H = p.H; c.ntx = p.ntx; c.r = lane >> 3; c.cq = lane & 7;
  c.loff = (unsigned)(c.r * p.W + c.cq * 4) * 4u;
  c.compact = LK ? compact : 0; c.keepn = 0; c.keep = nullptr;
  c.rng = nullptr; c.surv = nullptr;
  if (LK && compact) {   // uniform: the image region behind the compacted entries holds depth tiles between the passes
    const int k = (p.mask_lds_bytes - nactive * 32 - rng_words * 4) >> 10;
    c.keepn = k > 0 ? k : 0;
    c.keep = const_cast<uint4*>(reinterpret_cast<const uint4*>(bits + nactive * 8 + rng_words));
    c.rng = const_cast<unsigned*>(bits + nactive * 8);
    c.surv = reinterpret_cast<const unsigned short*>(bits + nactive * 8);   // (the survivors overwrite the ranges)
  }
  c.a00 = A0[0]; c.a01 = A0[1]; c.a02 = A0[2];
  c.a20 = A2[0]; c.a21 = A2[1]; c.a22 = A2[2];
  c.a10 = c.a11 = c.a12 = 0;
  if (PASS == 1) { c.a10 = A1[0]; c.a11 = A1[1]; c.a12 = A1[2]; }
  double sv[6];
#pragma unroll
  for (int i = 0; i < (PASS == 0 ? 5 : 6); ++i) sv[i] = acc[i];
  int n = *cnt;
  const bool dense = nactive < 0;                  // list overflow: walk every tile, skip empty ones
  const int nsteps = dense ? p.ntx * p.nty : nactive;
  const int jstart = wave * TG;
  const int rev_base = (PASS == 1 && !dense) ? nsteps - 1 : -1;
  if (LK && PASS == 1 && nsurv >= 0) {   // uniform: pass B of the plain build - the survivor list through an LDS work queue
    while (true) {
      unsigned off = 0;
      if (lane == 0) off = atomicAdd(qhead, (unsigned)TG);
      const int j0 = __builtin_amdgcn_readfirstlane((int)off);
      if (j0 >= nsurv) break;
      uint4 dq[TG];
      int tcs[TG];
      const unsigned pk = tile_fetch<PASS, LK, true, false>(c, dpl, bits, list, nsurv, false, j0, -1, dq, tcs);
      tile_compute<PASS, CHK, LK, true, false, SPEC>(c, list, nsurv, false, j0, -1, dq, pk, sv, &n, tcs);
    }
  } else {
    int jfirst = jstart;
    if (PASS == 0 && nfull > 0 && c.compact && !dense) {   // uniform
      for (int j0 = jstart; j0 < nfull; j0 += NWAVE * TG) {
        uint4 dq[TG];
        const unsigned pk = tile_fetch<PASS, LK, false, false, true>(c, dpl, bits, list, nfull, false, j0, -1, dq);
        tile_compute<PASS, CHK, LK, false, RNG && LK, SPEC, true>(c, list, nfull, false, j0, -1, dq, pk, sv, &n);
      }
      jfirst = nfull + jstart;
    }
    for (int j0 = jfirst; j0 < nsteps; j0 += NWAVE * TG) {
      uint4 dq[TG];
      const unsigned pk = tile_fetch<PASS, LK, false, false>(c, dpl, bits, list, nsteps, dense, j0, rev_base, dq);
      tile_compute<PASS, CHK, LK, false, RNG && LK, SPEC>(c, list, nsteps, dense, j0, rev_base, dq, pk, sv, &n);
    }
  }
#pragma unroll
  for (int i = 0; i < (PASS == 0 ? 5 : 6); ++i) acc[i] = sv[i];
  if (PASS == 0) *cnt = n;
}

// The moments of an ILL-CONDITIONED instance about a pivot (axis_from_sums; round 6): one tile per wave and step, one quad per lane,
// the checked pixel math - written for few registers, not for speed (the walk is rare and must not cost the common path a register:
// the tiled kernels sit at their 64-register budget).  The tile list / compact image as pass A left them; depth from memory (L2).
__device__ inline void pivot_pass(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits, const unsigned short* list,
                                  int nactive, const double* A0, const double* A2, int wave, int lane, int compact,
                                  const double* piv, double* acc, int* cnt) {
  const int r = lane >> 3, cq = lane & 7;
  const bool dense = nactive < 0;
  const int nsteps = dense ? p.ntx * p.nty : nactive;
  int n = 0;
#pragma clang loop unroll(disable)
  for (int j = wave; j < nsteps; j += NWAVE) {
    int tx, ty;
    if (dense) { ty = j / p.ntx; tx = j - ty * p.ntx; }
    else { const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[j]); tx = (int)(t & 0xffu); ty = (int)(t >> 8); }
    const int row = ty * 8 + r;
    unsigned nib = 0;
    if (compact) nib = (bits[j * 8 + r] >> (cq * 4)) & 0xFu;
    else if (row < p.H) nib = (bits[row * p.ntx + tx] >> (cq * 4)) & 0xFu;
    if (nib) {
      const uint4 dq = *reinterpret_cast<const uint4*>(dpl + (long long)row * p.W + tx * 32 + cq * 4);
      const unsigned db[4] = {dq.x, dq.y, dq.z, dq.w};
      const double vd = (double)row, ud = (double)(tx * 32 + cq * 4);
      quad_math<0, true, false, true>(nib, db, fma(A0[0], ud, fma(A0[1], vd, A0[2])), 0.0, fma(A2[0], ud, fma(A2[1], vd, A2[2])),
                                      A0[0], 0.0, A2[0], acc, &n, piv[0], piv[1]);
    }
  }
  *cnt = n;
}

// ------------------------------------------------------------------------------------------
// Separable single pass (round 5) - the un-grounded, skew-free camera, i.e. every call without a ground vector (BASELINE configs
// 2-5).  Then M = K^-1 = [[a00, 0, a02], [0, a11, a12], [0, 0, 1]] and a point is (x, y, z) = d * (r0(u), ry(v), 1): the x ray
// depends on the COLUMN only, the y ray on the ROW only, z is the depth itself.  Two consequences:
//   moments  Sx = sum_u r0(u) S1[u], Sxx = sum_u r0(u)^2 S2[u], Sxz = sum_u r0(u) S2[u], Sz = sum S1, Szz = sum S2 with the
//            per-column sums S1 = sum_v d, S2 = sum_v d^2: a lane that owns ONE column of a tile (4 rows) needs a conversion, an
//            add and an fma per pixel and nine operations per tile - 21 fp64 operations per tile and lane instead of 34;
//   extents  in the yaw frame x' = d * (cy r0(u) + sy), z' = d * (-sy r0(u) + cy): per column, a product of the depth with a
//            constant - monotone under rounding - so the column's extremes are attained at its smallest / largest depth.  Pass A
//            leaves [dmin, dmax] per column in LDS (non-negative floats order like unsigned integers: one ds_min_u32 + one
//            ds_max_u32 per lane and tile); after the axis, W columns x two products replace the whole of pass B.  The y extent
//            does not depend on the yaw at all and is taken per pixel in the same pass.
// So the depth is read ONCE (traffic = required bytes), there is no pass B, no depth range per tile, no cull plan (four barriers),
// no survivor list, no tile kept in LDS.  Lane = (half h = lane >> 5, column c = lane & 31) owns rows 4h .. 4h + 3 of column c of
// a 32 x 8 tile: four global_load_dword per tile (each instruction = two whole 128-byte lines), the tile's eight row words from
// the compacted bit image (one ds_read_b128 per lane).
// Optimistic like pass A: only the mask bit gates a pixel.  A NaN / inf depth turns the sums non-finite (stage_moments_to_axis
// sets sh->redo), a negative one would break the unsigned ordering (sh->sep_bad) - either way the workgroup re-runs the general
// two-pass path.  The rays are the canonical r0(u) = fma(a00, u, a02), ry(v) = fma(a11, v - v % 4, a12) + (v % 4) a11: pure
// functions of the column / row, used by every lane that meets them.  Sums are grouped per (lane, tile), so the records agree
// with the two-pass path to rounding (1e-12 relative), not bit for bit.
// ------------------------------------------------------------------------------------------
__device__ inline unsigned min3_u32(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ inline unsigned max3_u32(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// LDS words of the per-column depth range behind the compacted entries: colmin[W] | colmax[W], 16-byte granules
__device__ __host__ inline int sep_col_words(int W) { return (2 * W + 3) & ~3; }

// acc[0..4] += Sx, Sz, Sxx, Sxz, Szz of this wave's tiles; yext = [ymin, ymax]; *unsafe = max over the valid depth bit patterns
// (>= 0x7f800000: a NaN, an infinity or a negative depth under the mask).  col = colmin (colmax = col + W), initialised to
// 0xffffffff / 0 before the barrier in front of this call.
// EDGE: the frame's height is not a multiple of 8 - the last tile row sticks out of the frame (its own copy of the walk: frames of
// the common heights pay nothing for the test)
// One class of list entries [jbeg, jend) of the walk.  FULL (round 6): tiles whose 256 pixels all lie inside the mask - the lists
// hold them first (fit_instances_kernel sorts them to the front while compacting): no mask bits to fetch or extract, no per-pixel
// gating - 8 instead of 16 instructions per pixel on ~60 % of config 2's active tiles; the same values in the same operations, so a
// tile gives the same contributions whichever class walks it.  (A class per LOOP, not a branch per tile: two bodies inside one
// unrolled step cost 12-19 spilled vector registers.)
template <bool EDGE, bool FULL>
__device__ inline void sep_steps(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits, const unsigned short* list,
                                 int jbeg, int jend, double a00, double a02, double a11, double a12, unsigned* colq, int wave, int c, int h4,
                                 const unsigned* loff, int row0, double& s0, double& s1, double& s2, double& s3, double& s4, double& ylo,
                                 double& yhi, unsigned& bad) {
  for (int j0 = jbeg + wave * TG; j0 < jend; j0 += NWAVE * TG) {
    unsigned dq[TG][4];
    unsigned pk = 0;
    int tcs[TG];
    // stage 1: mask bits of this lane's column, then all depth loads back to back
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      const int e = j0 + g;
      tcs[g] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "=v"(dq[g][k]));   // (defined without an instruction: every use is gated by the mask bit)
      if (e < jend) {   // uniform
        const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[e]);
        tcs[g] = (int)t;
        // uniform tile origin in scalar registers + the lane's constant byte offsets
        const unsigned char* tp = reinterpret_cast<const unsigned char*>(dpl + ((long long)((t >> 8) * 8u) * p.W + (t & 0xffu) * 32u));
        if (FULL) {
#pragma unroll
          for (int k = 0; k < 4; ++k) dq[g][k] = *reinterpret_cast<const unsigned*>(tp + loff[k]);
          continue;
        }
        const uint4 w = *reinterpret_cast<const uint4*>(bits + e * 8 + h4);
        const unsigned nib = ((w.x >> c) & 1u) | (((w.y >> c) & 1u) << 1) | (((w.z >> c) & 1u) << 2) | (((w.w >> c) & 1u) << 3);
        pk |= nib << (4 * g);
        if (!EDGE || (t >> 8) * 8u + 8u <= (unsigned)p.H) {   // uniform: every row of the tile lies inside the frame
          if (nib) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dq[g][k] = *reinterpret_cast<const unsigned*>(tp + loff[k]);
          }
        } else {
          // the last tile row of a frame whose height is not a multiple of 8: a lane loads only the rows it holds a mask bit for
          // (rows past the frame carry none) - the block load above would read up to seven rows past the end of the depth plane
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((nib >> k) & 1u) dq[g][k] = *reinterpret_cast<const unsigned*>(tp + loff[k]);
        }
      }
    }
    // stage 2: the pixel math, tile after tile (the scheduling barriers keep the tiles' temporaries from overlapping: the kernel
    // lives in 64 registers)
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      if (j0 + g >= jend) continue;   // uniform
      const int tx = tcs[g] & 0xff, ty = tcs[g] >> 8;
      double ry = fma(a11, (double)(ty * 8 + h4 + row0), a12);
      double c1 = 0.0, c2 = 0.0;
      unsigned cmin = 0xffffffffu, cmax = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double d, ym;
        if (FULL) {
          cmin = min(cmin, dq[g][k]); cmax = max(cmax, dq[g][k]);
          d = (double)__uint_as_float(dq[g][k]);
          ym = d * ry;
        } else {
          const int m = __builtin_amdgcn_sbfe((int)pk, 4 * g + k, 1);   // the pixel's mask bit as 0 / -1: ONE v_bfe_i32
          const unsigned v = dq[g][k] & (unsigned)m;                 // invalid -> +0.0 (sums), 0 (unsigned max)
          unsigned w;
          asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(w) : "v"(m), "v"(dq[g][k]));   // invalid -> 0xffffffff (unsigned min)
          cmin = min(cmin, w); cmax = max(cmax, v);
          d = (double)__uint_as_float(v);
          // y extent: per pixel (the row ray), invalid pixels as NaN (ignored by v_min / v_max_f64): the high word through one v_bfi
          const double y = d * ry;
          int yh;
          asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(yh) : "v"(m), "v"(__double2hiint(y)));
          ym = __hiloint2double(yh, __double2loint(y));
        }
        if (k == 0) { c1 = d; c2 = d * d; }
        else { c1 += d; c2 = fma(d, d, c2); }
        ylo = dmin(ylo, ym); yhi = dmax(yhi, ym);
        ry += a11;
      }
      // the column's ray, once per tile
      const double r0 = fma(a00, (double)(tx * 32 + c), a02);
      const double t1 = r0 * c1, t2 = r0 * c2;
      s0 += t1; s1 += c1; s2 = fma(r0, t2, s2); s3 += t2; s4 += c2;
      // depth range of this lane's column in this tile -> the per-column arrays (a lane without a mask bit sends the identities)
      bad = max(bad, cmax);
      atomicMin(colq + tx * 32, cmin);
      atomicMax(colq + tx * 32 + p.W, cmax);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// nfull: list entries [0, nfull) are tiles that lie completely inside the mask (0: the list is not sorted by class)
template <bool EDGE = false>
__device__ inline void sweep_sep(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits,
                                 const unsigned short* list, int nactive, const double* Mg, unsigned* col, int wave, int lane,
                                 double* acc, double* yext, unsigned* unsafe, int row0 = 0, int nfull = 0) {
  // (row0: the frame row of tile row 0 - the row engine hands every workgroup a band of rows, dpl / bits / list band-local;
  // the instance engine passes the literal 0)
  const int c = lane & 31, h4 = (lane >> 5) * 4;
  const double a00 = Mg[0], a02 = Mg[2], a11 = Mg[4], a12 = Mg[5];
  unsigned loff[4];   // byte offsets of this lane's four pixels inside a tile (uniform tile origin + 32-bit vector offset: the saddr form)
#pragma unroll
  for (int k = 0; k < 4; ++k) loff[k] = (unsigned)((h4 + k) * p.W + c) * 4u;
  unsigned* colq = col + c;
  double s0 = acc[0], s1 = acc[1], s2 = acc[2], s3 = acc[3], s4 = acc[4];
  double ylo = yext[0], yhi = yext[1];
  unsigned bad = *unsafe;
  if (nfull > 0) sep_steps<EDGE, true>(p, dpl, bits, list, 0, nfull, a00, a02, a11, a12, colq, wave, c, h4, loff, row0, s0, s1, s2, s3, s4, ylo, yhi, bad);   // uniform
  sep_steps<EDGE, false>(p, dpl, bits, list, nfull, nactive, a00, a02, a11, a12, colq, wave, c, h4, loff, row0, s0, s1, s2, s3, s4, ylo, yhi, bad);
  acc[0] = s0; acc[1] = s1; acc[2] = s2; acc[3] = s3; acc[4] = s4;
  yext[0] = ylo; yext[1] = yhi;
  *unsafe = bad;
}

// x / z extents in the yaw frame from the per-column depth ranges: threads over the columns.  rho0(u) = fma(N0[0], u, N0[2]),
// rho2(u) = fma(N2[0], u, N2[2]) (rows 0 and 2 of rotate_y(yaw) @ M; their middle entries are zero here).
__device__ inline void sep_col_extents(const unsigned* col, int W, const double* N0, const double* N2, int tid, double* ext) {
  double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
  for (int u = tid; u < W; u += NT) {
    const unsigned lo = col[u], hi = col[W + u];
    if (lo <= hi) {   // the column holds a mask pixel
      const double dlo = (double)__uint_as_float(lo), dhi = (double)__uint_as_float(hi), ud = (double)u;
      const double q0 = fma(N0[0], ud, N0[2]), q2 = fma(N2[0], ud, N2[2]);
      const double xa = dlo * q0, xb = dhi * q0, za = dlo * q2, zb = dhi * q2;
      xlo = fmin(xlo, fmin(xa, xb)); xhi = fmax(xhi, fmax(xa, xb));
      zlo = fmin(zlo, fmin(za, zb)); zhi = fmax(zhi, fmax(za, zb));
    }
  }
  ext[0] = xlo; ext[1] = xhi; ext[4] = zlo; ext[5] = zhi;
}

}  // namespace
