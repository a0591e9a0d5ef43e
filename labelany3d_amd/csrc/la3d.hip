// la3d.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI for the LabelAny3D geometric hot path:
// pinhole back-projection of masked depth pixels -> per-object moments -> closed-form PCA yaw
// -> extents along the principal axes -> 39-double box record.
//
// Reference semantics (behaviour only; nothing is copied):
//   depth_to_points   /root/reference/src/util.py:52-75
//   estimate_bbox     /root/reference/src/util_3dbox.py:106-178   (+ helpers :20-103, PCA yaw :181-186)
//
// Memory-bound integer/byte + fp64 reduction work: no MFMA.  Layout, kernel design and measurements are in
// DESIGN.md.  This file holds the INSTANCE and BAND ENGINES of la3d_fit_instances (one workgroup per instance; used for
// B > 272 (u8 planes) / 288 (run lengths, polygon parts), for the fused instance filter, for reference-subsample mode and
// for frames the split engine does not take - la3d_split.hip is the other engine) and every other kernel of the C-ABI.  `fit_instances_kernel` in short:
//   one 512-thread workgroup (8 wave64) per instance, 64 VGPRs / 40 KB LDS -> 4 workgroups per CU (the opt-in "retaining" build:
//            128 VGPRs, 2 workgroups per CU, depth tiles kept in registers between the passes - the default of rounds 2-3 for u8 planes);
//   order    256 < B <= 3 resident sets: which instance a workgroup fits is decided in the kernel (order_select) from the sort
//            keys of ONE estimate kernel (or the caller's area hints: no helper launch) - size-balanced, speed only;
//   phase 0  streams the u8 mask plane once with 16-byte non-temporal loads (or decodes COCO run lengths / rasterises polygon
//            parts) into a 1-bit-per-pixel image in LDS (38.4 KB for 640x480); meanwhile one lane computes K^-1, Rg, M;
//   list     deterministic compaction of the 32 px x 8 row tiles that contain a set bit;
//   pass A   4 listed tiles per wave-step, their float4 depth loads issued back to back; branch-free fp64
//            accumulation of Sx, Sz, Sxx, Sxz, Szz -> DPP wave reduction -> LDS -> wave 0 (fixed order);
//   yaw      closed-form 2x2 principal axis with scikit-learn's sign rule, no trigonometry (one lane);
//   pass B   same walk, six extents in the yaw frame with NaN-ignoring raw v_min/v_max_f64;
//   epilog   wave 0 writes center / dims / R_cam / fp16-quantised vertices, one lane per output group.
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
    k.bands = (e && (atoi(e) == 4 || atoi(e) == 2)) ? atoi(e) : 0;
    e = getenv("LA3D_BAND_DEFAULT");      // 0: the band engine only when asked for (LA3D_ENGINE=band / opt_engine)
    k.band_default = !(e && e[0] == '0');
    e = getenv("LA3D_BAND_MAXB");
    k.band_maxb = (e && atoi(e) > 0) ? atoi(e) : 256;
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
    e = getenv("LA3D_RETAIN");
    k.retain = e ? (atoi(e) > 0 ? LA3D_BUILD_RETAINING : LA3D_BUILD_PLAIN) : LA3D_BUILD_DEFAULT;
    e = getenv("LA3D_RETAIN_MAXB");
    k.retain_maxb = e ? atoi(e) : -1;
    e = getenv("LA3D_RETAIN_NOMASK");
    k.retain_nomask = (e && e[0] == '1') ? 1 : 0;
    e = getenv("LA3D_LDSKEEP");
    k.ldskeep = !(e && e[0] == '0');
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
// split engine (la3d_split.hip)
bool split_eligible(const FitParams& p, bool vec, bool ldsmask);
int split_fit(const FitParams& p, void* workspace, hipStream_t s);
size_t split_workspace_bytes(int B, int H, int W);
}

using namespace la3d;

namespace {
// Generic walk (any W, unaligned planes, frames whose bit image does not fit LDS): row-linear chunks of
// 256 pixels per wave, 4 per lane.  PASS 0: count + moments of (x', z').  PASS 1: extents of all three
// axes in the yaw frame.  A0/A1/A2 are the rows mapping [u,v,1] to the ray components: PASS 0 uses rows 0
// and 2 of M; PASS 1 uses N0, M row 1, N2.
template <bool VEC, bool LDSMASK, int PASS>
__device__ inline void sweep(const FitParams& p, const float* __restrict__ dpl, const unsigned char* __restrict__ mpl,
                             const unsigned* bits, const double* A0, const double* A1, const double* A2,
                             int wave, int lane, double* acc, int* cnt, int* nmask) {
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
        if (PASS == 0) nm += __popc(nib);
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
          const double x = d * r0, z = d * r2;
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

// Depth tiles kept on chip between the two passes (RET > 0: the "retaining" build of the kernel, 128 VGPRs, two workgroups
// per CU).  The first RET steps of every wave (RET x TG tiles, i.e. RET x TG x NWAVE tiles per instance) keep their depth
// quads and mask nibbles in registers after pass A; pass B computes on them without touching memory.  Tiles beyond that
// are re-read in pass B exactly as in the RET = 0 build.  Register arrays need static indices, hence the unrolled steps.
template <int RET>
struct Keep {
  uint4 dq[RET > 0 ? RET : 1][TG];
  unsigned nib[RET > 0 ? RET : 1];   // TG nibbles per step
};

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
    // rev_base >= 0: pass B walks the not-retained part of the list backwards - the tiles pass A read last are re-read
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
template <int PASS, bool LK, bool SURV = false, bool ZERO = true>
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
      if (LK && c.compact) {   // uniform
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
template <bool CHK>
__device__ inline void tile_range(const TileCtx& c, int e, unsigned nib, const unsigned* db) {
  // per pixel: the validity word m (0 / -1) and db & m are the pixel math's own values (same expressions: shared after inlining);
  // db | ~m is one v_bfi_b32 (m ? db : ones).  The cross-lane steps carry the operation's identity as `old`, which lets the
  // compiler fold every move into its min / max (v_min_u32_dpp: one instruction per step instead of three).
  unsigned w[4], v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
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
template <int PASS, bool CHK, bool LK = false, bool SURV = false, bool RNG = false, bool SPEC = false>
__device__ inline void tile_compute(const TileCtx& c, const unsigned short* list, int nsteps, bool dense, int j0, int rev_base,
                                    const uint4* dq, unsigned pk, double* sv, int* n, const int* tcs = nullptr) {
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    const int j = j0 + g;
    if (j >= nsteps) continue;   // wave-uniform
    if (LK && PASS == 0 && j < c.keepn) c.keep[j * 64 + (c.r * 8 + c.cq)] = dq[g];   // (pass A walks the list forwards: entry = j)
    const unsigned nib = (pk >> (4 * g)) & 0xFu;
    if (dense && __ballot(nib != 0) == 0) continue;
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
    if (RNG && PASS == 0) tile_range<CHK>(c, j, nib, db);
  }
}

constexpr int LDS_KEEP_WAVE = TG * 1024 + 256;   // bytes of the LDS-kept step per wave
// lds_keep (RET > 0 builds with LDS to spare): one more step per wave kept in LDS (TG x 1 KiB per wave) instead of registers
// words of the image region that pass-B culling takes behind the compacted entries: the depth ranges (two words per tile; the
// survivor list overwrites them later), 16-byte granules, then CULL_SCRATCH_WORDS for the champion search of cull_plan
// (kept out of `Shared`: every byte there comes off the tile list's capacity, i.e. off the mask size up to which the plain
// and the retaining build group their partial sums alike)
constexpr int CULL_SCRATCH_WORDS = NWAVE * 6 * 2;   // per wave and direction: value (f32), list entry (u32)
__device__ inline int cull_rng_words(int nactive) { return ((2 * nactive + 3) & ~3) + CULL_SCRATCH_WORDS; }

// RNG (pass A, plain build, compact image): also leave every tile's depth range for pass-B culling (rng_words > 0 then).
// nsurv >= 0 (pass B): walk only the culling survivors surv[0 .. nsurv) (fit_instances_kernel builds the list).
// SPEC: the un-grounded, skew-free forms of the pixel math (quad_math): the caller checks M's row 2 == (0, 0, 1) for pass A,
// M[1][0] == 0 for pass B.
template <int PASS, bool CHK, int RET, bool RNG = false, bool SPEC = false>
__device__ inline void sweep_tiled(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits,
                                   const unsigned short* list, int nactive, const double* A0, const double* A1,
                                   const double* A2, int wave, int lane, double* acc, int* cnt, Keep<RET>& keep,
                                   uint4* lds_keep = nullptr, unsigned* qhead = nullptr, int compact = 0, int rng_words = 0,
                                   int nsurv = -1) {
  constexpr bool LK = RET == 0;
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
  int jstart = wave * TG, kept = 0;
  if (RET > 0 && !dense) {
    // the kept steps: pass A issues the loads of ALL of them before it computes (their destination registers are long-lived
    // anyway, so RET x TG tiles are in flight per wave at no register cost); pass B touches no memory
    if (PASS == 0) {
#pragma unroll
      for (int s = 0; s < RET; ++s) {
        const int j0 = (s * NWAVE + wave) * TG;
        keep.nib[s] = 0;
        if (j0 < nsteps) keep.nib[s] = tile_fetch<PASS, false>(c, dpl, bits, list, nsteps, false, j0, -1, keep.dq[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < RET; ++s) {
      const int j0 = (s * NWAVE + wave) * TG;
      if (j0 < nsteps) tile_compute<PASS, CHK, false, false, false, SPEC>(c, list, nsteps, false, j0, -1, keep.dq[s], keep.nib[s], sv, &n);
    }
    jstart = (RET * NWAVE + wave) * TG;
    kept = RET * NWAVE * TG;
    if (lds_keep != nullptr) {   // uniform
      if (jstart < nsteps) {
        // per wave: TG x 1 KiB of quads, then one word of nibbles per lane
        unsigned char* base = reinterpret_cast<unsigned char*>(lds_keep) + wave * LDS_KEEP_WAVE;
        uint4* slot = reinterpret_cast<uint4*>(base) + lane;
        unsigned* nslot = reinterpret_cast<unsigned*>(base + TG * 1024) + lane;
        uint4 dq[TG];
        unsigned pk;
        if (PASS == 0) {
          pk = tile_fetch<PASS, false>(c, dpl, bits, list, nsteps, false, jstart, -1, dq);
#pragma unroll
          for (int g = 0; g < TG; ++g) slot[g * 64] = dq[g];
          *nslot = pk;
        } else {
#pragma unroll
          for (int g = 0; g < TG; ++g) dq[g] = slot[g * 64];
          pk = *nslot;
        }
        tile_compute<PASS, CHK, false, false, false, SPEC>(c, list, nsteps, false, jstart, -1, dq, pk, sv, &n);
      }
      jstart += NWAVE * TG;
      kept += NWAVE * TG;
    }
  }
  const int rev_base = (PASS == 1 && !dense) ? kept + nsteps - 1 : -1;
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
  } else
  if (!LK && PASS == 1 && qhead != nullptr && !dense) {
    // pass B: the not-retained tiles are an LDS WORK QUEUE - a wave that is done pulls the next TG tiles (one ds_add_rtn per
    // step) instead of walking a fixed stride, so no wave waits for a slower neighbour at the end of the pass.  Extents are
    // min / max: exact whatever the order, so the records stay bit-identical (pass A, whose fp64 sums depend on the grouping,
    // keeps its static assignment).
    while (true) {
      unsigned off = 0;
      if (lane == 0) off = atomicAdd(qhead, (unsigned)TG);
      const int j0 = kept + __builtin_amdgcn_readfirstlane((int)off);
      if (j0 >= nsteps) break;
      uint4 dq[TG];
      const unsigned pk = tile_fetch<PASS, LK, false, false>(c, dpl, bits, list, nsteps, false, j0, rev_base, dq);
      tile_compute<PASS, CHK, LK, false, false, SPEC>(c, list, nsteps, false, j0, rev_base, dq, pk, sv, &n);
    }
  } else {
    for (int j0 = jstart; j0 < nsteps; j0 += NWAVE * TG) {
      uint4 dq[TG];
      const unsigned pk = tile_fetch<PASS, LK, false, false>(c, dpl, bits, list, nsteps, dense, j0, rev_base, dq);
      tile_compute<PASS, CHK, LK, false, RNG && LK, SPEC>(c, list, nsteps, dense, j0, rev_base, dq, pk, sv, &n);
    }
  }
#pragma unroll
  for (int i = 0; i < (PASS == 0 ? 5 : 6); ++i) acc[i] = sv[i];
  if (PASS == 0) *cnt = n;
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
template <bool EDGE = false>
__device__ inline void sweep_sep(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits,
                                 const unsigned short* list, int nactive, const double* Mg, unsigned* col, int wave, int lane,
                                 double* acc, double* yext, unsigned* unsafe, int row0 = 0) {
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
  for (int j0 = wave * TG; j0 < nactive; j0 += NWAVE * TG) {
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
      if (e < nactive) {   // uniform
        const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[e]);
        tcs[g] = (int)t;
        const uint4 w = *reinterpret_cast<const uint4*>(bits + e * 8 + h4);
        const unsigned nib = ((w.x >> c) & 1u) | (((w.y >> c) & 1u) << 1) | (((w.z >> c) & 1u) << 2) | (((w.w >> c) & 1u) << 3);
        pk |= nib << (4 * g);
        // uniform tile origin in scalar registers + the lane's constant byte offsets
        const unsigned char* tp = reinterpret_cast<const unsigned char*>(dpl + ((long long)((t >> 8) * 8u) * p.W + (t & 0xffu) * 32u));
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
      if (j0 + g >= nactive) continue;   // uniform
      const int tx = tcs[g] & 0xff, ty = tcs[g] >> 8;
      double ry = fma(a11, (double)(ty * 8 + h4 + row0), a12);
      double c1 = 0.0, c2 = 0.0;
      unsigned cmin = 0xffffffffu, cmax = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = __builtin_amdgcn_sbfe((int)pk, 4 * g + k, 1);   // the pixel's mask bit as 0 / -1: ONE v_bfe_i32
        const unsigned v = dq[g][k] & (unsigned)m;                 // invalid -> +0.0 (sums), 0 (unsigned max)
        unsigned w;
        asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(w) : "v"(m), "v"(dq[g][k]));   // invalid -> 0xffffffff (unsigned min)
        cmin = min(cmin, w); cmax = max(cmax, v);
        const double d = (double)__uint_as_float(v);
        if (k == 0) { c1 = d; c2 = d * d; }
        else { c1 += d; c2 = fma(d, d, c2); }
        // y extent: per pixel (the row ray), invalid pixels as NaN (ignored by v_min / v_max_f64): the high word through one v_bfi
        const double y = d * ry;
        int yh;
        asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(yh) : "v"(m), "v"(__double2hiint(y)));
        const double ym = __hiloint2double(yh, __double2loint(y));
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

// ------------------------------------------------------------------------------------------
// workgroup stages shared by the fit kernels (every thread of the workgroup must call them)
// ------------------------------------------------------------------------------------------
// moments of all waves -> wave 0 (fixed xor tree: bit-reproducible) -> status, yaw axis.
// On return sh->st / sh->cyaw / sh->syaw are valid for every thread.  The aux record (with its atan2) is
// written at the end of the kernel (stage_status_aux), off everybody's critical path.
// allow_redo: the sums come from the optimistic pass (quad_math<0, false>); if they are not finite, set sh->redo and return
// without deciding anything - the caller re-runs the checked pass and calls again with allow_redo = false.
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
    sh->redo = (allow_redo && !sh->bad_ground && !(fabs(chk) <= 1.79769313486231570815e308)) ? 1 : 0;
    double cy = NAN, sy = NAN;
    if (st == LA3D_BOX_OK) axis_from_sums((double)n, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap);
    sh->cyaw = cy; sh->syaw = sy;
    sh->qhead = 0u;   // pass B's work queue starts at the first not-retained tile
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
// CULL_MIN: active tiles below which the plan costs more than it saves (measured: profiles/r04/r04_cull.txt); the
                              // per-call value is FitParams::cull_min: 128 for u8 planes, whose launches are bandwidth-bound - after the
                              // cheaper tile range B = 1024 / 1536 / 2048 run 98.6 / 133.1 / 170.1 -> 96.8 / 130.3 / 164.6 us, config-5 masks
                              // unchanged, run lengths 68.0 -> 69.1 (hence 224 there); profiles/r04/r04_cull_threshold.txt
constexpr int CULL_MIN = 224;
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

// ------------------------------------------------------------------------------------------
// instance engine: one workgroup per instance
// ------------------------------------------------------------------------------------------
// SRC: where the mask comes from - 0 = u8 plane, 1 = COCO run lengths, 2 = polygon parts (both decoded into the LDS bit image)
template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED, int SRC, int RET>
__global__ __launch_bounds__(NT, RET > 0 ? NT / 128 : NT / 64) void fit_instances_kernel(const FitParams p) {
  constexpr bool RLE = SRC == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  Shared* sh = reinterpret_cast<Shared*>(smem + p.mask_lds_bytes);
  unsigned* prefix = reinterpret_cast<unsigned*>(smem + p.mask_lds_bytes + sizeof(Shared));  // SAMPLE only
  // TILED only: compacted list of active tile ids
  unsigned short* list = reinterpret_cast<unsigned short*>(smem + p.mask_lds_bytes + sizeof(Shared));

  // (builds that carry the separable pass take the lane from the execution mask, not from threadIdx.x - the workgroup's waves are
  // full -, and rebuild the thread index where it is used: neither then keeps the kernel's input register alive across the passes)
  constexpr bool REBUILD_TID = TILED && !SAMPLE && RET == 0;
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
  if (RET == 0 && !SAMPLE && p.order_self && (int)blockIdx.x < p.order_resident && !(p.order_self == 2 && blockIdx.x % 7 == 3)) {
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
  if (RET == 0 && !SAMPLE && p.stagger_ticks > 0 && p.order_nch > 0 && blockIdx.x < 1024) {
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
  if (RET > 0 && SRC == 0 && p.stagger_ticks > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    // retaining build, two workgroups per CU: the second-dispatched one (block b + 256 shares CU b % 256 with block b - measured
    // placement, speed only) holds back for about the time the first needs to stream its mask plane at full bandwidth, so that
    // the two run half a period apart: one streams while the other is in its passes (register-resident pass B moves no bytes)
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
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
      constexpr int P0U = RET > 0 ? 8 : 4;   // 16-byte loads in flight per lane
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
  int nactive = 0;
  // plain build: the bit image is compacted to the active tiles (eight row words per list entry) and the LDS that frees keeps
  // depth tiles between the passes (sweep_tiled)
  constexpr bool LK = TILED && !SAMPLE && RET == 0;
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
      int wcount = 0;
      if ((p.H & 7) == 0) {   // uniform: every tile row is complete (the common frame heights) - no row clamp, no select per word
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tbeg + k * 64 + lane;
          unsigned any = 0;
          if (t < tend) {
            const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;  // exact: see fit_dispatch
            const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const unsigned w = bw[rr * p.ntx];
              any |= w;
              if constexpr (LK) wrd[k][rr] = w;
            }
          }
          bal[k] = __ballot(any != 0);
          wcount += __popcll(bal[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tbeg + k * 64 + lane;
          unsigned any = 0;
          if (t < tend) {
            const int ty = (int)(((float)t + 0.5f) * p.rcp_ntx), tx = t - ty * p.ntx;  // exact: see fit_dispatch
            const int rmax = p.H - 1 - ty * 8;                                        // >= 0
            const unsigned* bw = bits + (ty * 8) * p.ntx + tx;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const unsigned w = bw[min(rr, rmax) * p.ntx];
              any |= w;
              if constexpr (LK) wrd[k][rr] = rr <= rmax ? w : 0u;
            }
          }
          bal[k] = __ballot(any != 0);
          wcount += __popcll(bal[k]);
        }
      }
      if (lane == 0) sh->scan[wave] = (unsigned)wcount;
      __syncthreads();
      LA3D_STAMP(14);
      for (int w = 0; w < NWAVE; ++w) {
        const int c = (int)sh->scan[w];
        if (w < wave) base += c;
        nactive += c;
      }
      if (nactive > p.list_cap) {
        nactive = -1;  // uniform: every thread sees the same total
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
        // (with pass-B culling the compact image also holds the survivor list / the depth ranges behind the entries)
        if constexpr (LK) if (nactive * 32 + cull_rng_words(nactive) * 4 <= p.mask_lds_bytes) {   // uniform
          // every wave read its row words before the barrier above: the image region can be overwritten in place
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
      if (p.H & 7) sweep_sep<true>(p, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe);   // uniform
      else sweep_sep<false>(p, dpl, bits, list, nactive, Mg, col, wave, lane, sacc, yx, &unsafe);
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
  Keep<RET> keep;   // RET > 0: depth quads of the first RET steps of this wave, kept in registers for pass B
  uint4* lds_keep = (RET > 0 && p.lds_keep_off > 0) ? reinterpret_cast<uint4*>(smem + p.lds_keep_off) : nullptr;
  if (!sampled) {
    if (TILED) {
      // (the un-grounded, skew-free forms of the pixel math where they apply: same records, fewer instructions - quad_math)
      // (not in the subsample build, which walks tiles only for its small masks and has no registers to spare, and not in the
      // retaining build, whose register allocation the extra bodies disturb: config 5 at B = 1024 +5 %, profiles/r04/r04_spec.txt)
      constexpr bool SP = !SAMPLE && RET == 0;
      const bool specA = SP && Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform
      if (LK && cull) {
        if (specA) sweep_tiled<0, false, RET, true, SP>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
        else sweep_tiled<0, false, RET, true>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
      } else {
        if (specA) sweep_tiled<0, false, RET, false, SP>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
        else sweep_tiled<0, false, RET>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
      }
      cnt = nmask;   // the optimistic pass does not count: with every masked depth finite, valid pixels = mask pixels
    }
    else sweep<VEC, LDSMASK, 0>(p, dpl, mpl, bits, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, &nmask);
  }

  LA3D_STAMP(3);
  stage_moments_to_axis(sh, p, inst_p, acc, cnt, nmask, tid, wave, lane, TILED && !sampled);
  if (TILED && sh->redo) {  // uniform
    __syncthreads();        // everyone has read sh->redo and the partials before they are rewritten
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = 0;
    cnt = 0;
    checked = true;
    if (LK && cull) sweep_tiled<0, true, RET, true>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
    else sweep_tiled<0, true, RET>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, lds_keep, nullptr, compact, rng_words);
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
      // (plain build only: in the retaining build the queue covers just the not-retained remainder and measured 3 us SLOWER at
      // B = 1024; plain build: run-length input 74.8 -> 71.3 us, B = 512 88.7 -> 85.5, config 5 at 16 k 945 -> 927;
      // profiles/r03/r03_pass_b_queue.txt)
      unsigned* qh = RET == 0 ? &sh->qhead : nullptr;
      int nsurv = -1;
      if constexpr (LK) {
        if (compact) nsurv = nactive;   // the identity list written before the axis stage
        if (cull) {   // uniform
          nsurv = checked ? cull_plan<true>(sh, p, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext)
                          : cull_plan<false>(sh, p, dpl, bits, list, nactive, rng_words, N0, Mg + 3, N2, tid, wave, lane, ext);
        }
      }
      if (checked) sweep_tiled<1, true, RET>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, lds_keep, qh, compact, rng_words, nsurv);
      else if (!SAMPLE && RET == 0 && Mg[3] == 0.0) sweep_tiled<1, false, RET, false, !SAMPLE && RET == 0>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, lds_keep, qh, compact, rng_words, nsurv);
      else sweep_tiled<1, false, RET>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, lds_keep, qh, compact, rng_words, nsurv);
    }
    else sweep<VEC, LDSMASK, 1>(p, dpl, mpl, bits, N0, Mg + 3, N2, wave, lane, ext, &d0, &d1);
  }
  LA3D_STAMP(5);
  stage_extents_to_box(sh, p, inst_p, ext, tid, wave, lane);
  stage_status_aux(sh, p, inst_p, tid);
  LA3D_STAMP(6);
}
#undef tid

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
    if (lane == 0) {
      double* x = band_xch<NB>(p, inst);
      double* mine = x + (h * 2 + round) * BAND_XD;
#pragma unroll
      for (int k = 0; k < 5; ++k) st_agent(mine + k, s[k]);
      st_agent(mine + 5, (double)n); st_agent(mine + 6, (double)nm);
      band_release();                                      // the record's stores are acknowledged before the arrival is issued
      unsigned long long* arrive = p.band_arrive + (long long)inst * 4 + round;
      const unsigned spin_max = p.band_test == 1 ? (1u << 10) : BAND_SPIN_MAX;
      unsigned spins = 0;
      if (tagged_arrive(arrive, p.band_tag) < (unsigned)NB)
      while (tagged_count(arrive, p.band_tag) < (unsigned)NB && spins < spin_max) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
      }
      const bool timeout = spins >= spin_max;
      band_acquire();                                      // the other bands' records are read after their arrivals were seen
      double t[5] = {0, 0, 0, 0, 0}, tn = 0, tm = 0;
#pragma unroll 1   // (unrolled, the compiler keeps all NB records in flight: 56 registers at NB = 4 -> spills)
      for (int hb = 0; hb < NB; ++hb) {                    // band order: the same sum in every band of the instance
        const double* r = x + (hb * 2 + round) * BAND_XD;
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] += ld_agent(r + k);
        tn += ld_agent(r + 5); tm += ld_agent(r + 6);
      }
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
      sh->redo = (allow_redo && !timeout && !sh->bad_ground && !(fabs(chk) <= 1.79769313486231570815e308)) ? 1 : 0;
      double cy = NAN, sy = NAN;
      if (st == LA3D_BOX_OK) axis_from_sums((double)nt, t[0], t[1], t[2], t[3], t[4], &cy, &sy, &gap);
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
  Keep<0> keep;
  bool checked = false;
  const bool specA = Mg[6] == 0.0 && Mg[7] == 0.0 && Mg[8] == 1.0;   // uniform (quad_math: the un-grounded forms, same records)
  if (cull) {
    if (specA) sweep_tiled<0, false, 0, true, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
    else sweep_tiled<0, false, 0, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
  } else {
    if (specA) sweep_tiled<0, false, 0, false, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
    else sweep_tiled<0, false, 0>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
  }
  cnt = nmask;   // optimistic pass: valid pixels = mask pixels
  band_moments_to_axis<NB>(sh, p, inst_p, h, 0, acc, cnt, nmask, tid, wave, lane, true);
  if (sh->st >= BAND_ST_TAKEOVER) {   // uniform: watchdog timeout
    if (sh->st == BAND_ST_TAKEOVER) band_takeover(sh, p, inst_p, tid, wave, lane);
    return;
  }
  if (sh->redo) {   // uniform, and the same in every band of the instance: the summed moments decide
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = 0;
    cnt = 0;
    checked = true;
    if (cull) sweep_tiled<0, true, 0, true>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
    else sweep_tiled<0, true, 0>(pb, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, keep, nullptr, nullptr, compact, rng_words);
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
    if (checked) sweep_tiled<1, true, 0>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, nullptr, &sh->qhead, compact, rng_words, nsurv);
    else if (Mg[3] == 0.0) sweep_tiled<1, false, 0, false, true>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, nullptr, &sh->qhead, compact, rng_words, nsurv);
    else sweep_tiled<1, false, 0>(pb, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0, keep, nullptr, &sh->qhead, compact, rng_words, nsurv);
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
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { st_agent(xe + h * 6 + 2 * k, lo[k]); st_agent(xe + h * 6 + 2 * k + 1, hi[k]); }
    band_release();
    last = tagged_arrive(p.band_arrive + (long long)inst_p * 4 + 2, p.band_tag) == (unsigned)NB ? 1 : 0;
    band_acquire();
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // min / max over the bands (uniform loads: every lane reads the same words)
    double l = ld_agent(xe + 2 * k), u = ld_agent(xe + 2 * k + 1);
#pragma unroll 1
    for (int hb = 1; hb < NB; ++hb) { l = fmin(l, ld_agent(xe + hb * 6 + 2 * k)); u = fmax(u, ld_agent(xe + hb * 6 + 2 * k + 1)); }
    lo[k] = uniform_f64(l); hi[k] = uniform_f64(u);
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

template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED, int SRC, int RET = 0>
int launch_fit_inst(const FitParams& p_in, size_t lds, hipStream_t s, void* workspace) {
  auto kern = fit_instances_kernel<VEC, LDSMASK, SAMPLE, TILED, SRC, RET>;
  allow_big_lds(reinterpret_cast<const void*>(kern));
  FitParams p = p_in;
  p.order_nch = 0; p.order_keys = nullptr; p.order_resident = 0; p.order_shift = 0;
  p.order_self = 0; p.order_flags = nullptr; p.order_nonce = 0; p.est_step = 1;
  // size-balanced launch order: needs the 16-byte mask groups (VEC), more than one workgroup per CU, and a batch
  // the O(B^2) ranking is cheap for
  if (workspace && VEC && !SAMPLE && p.B > 256 && p.B <= ORDER_MAX_B && balance_enabled(p)) {
    const int max_rounds = balance_max_rounds();
    int wg_per_cu = (RET > 0 ? 1024 : 2048) / NT;  // wave slots: 32 per CU at 64 VGPRs, 16 at 128 (the retaining build)
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
        bool self = RET == 0 && config().order_self != 0 && wg_per_cu * 256 >= 256;
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

// ---- band engine (fit_bands_kernel) ----
constexpr int BAND_NB_MAX = 4;
constexpr int BAND_MINB = 16;   // smallest batch the band engine takes by default (48 while a memset preceded the launch)
constexpr size_t band_xch_doubles(int nb) { return (size_t)nb * (2 * BAND_XD + 6); }

// workspace of the band engine: [B] u32 sort keys | [B][4] u64 tagged arrival words | [B][NB_MAX * 22] f64 exchange records
inline size_t band_keys_bytes(int B) { return ((size_t)B * 4 + 255) & ~(size_t)255; }
inline size_t band_workspace_bytes(int B) { return band_keys_bytes(B) + (size_t)B * 32 + (size_t)B * band_xch_doubles(BAND_NB_MAX) * 8 + 256; }

inline bool band_frame_ok(int H, int W, int nb) {
  if (W % 32 != 0 || (long long)H * W % 16 != 0) return false;
  const int ntx = W / 32, nty = (H + 7) / 8;
  if (ntx > 255 || nty > 255 || nty < nb) return false;
  const int tb = nty / nb, tmax = nty - (nb - 1) * tb;   // the last band takes the remainder
  return (long long)ntx * tmax <= 256 * NWAVE;          // one-pass tile list: <= 256 tiles per wave
}

// Bands per instance: LA3D_BANDS pins 2 or 4; otherwise four up to 288 instances, two beyond (measured, us per call, u8 planes,
// split | instance | two bands | four bands - profiles/r04/r04_band.txt: B = 4: 34 | 37 | 44 | 31; 64: 43 | 56 | 51 | 38;
// 256: 67 | 67 | 64 | 63; 320: 80 | 76 | 68 | 69; 384: - | 75 | 73 | 77; 512: - | 80 | 85 | 95; 1024: - | 107 | 134 | 168).
inline int band_count(const FitParams& p) {
  int nb = config().bands ? config().bands : (p.B <= 288 ? 4 : 2);
  if (nb == 4 && !band_frame_ok(p.H, p.W, 4)) nb = 2;
  return nb;
}

// u8 planes, 16-byte aligned, full-mask mode.  By default the band engine takes 16 <= B <= 256: below, it ties with the split
// engine (32-34 us per call either way) and the split engine stays (us per call, split | four bands, once the band launch lost its
// memset: B = 1: 32.2 | 32.3; 4: 32.9 | 33.3; 16: 34.9 | 33.9; 32: 36.0 | 34.0; 48: 39.5 | 35.9; 64: 42.1 | 36.6); above, the instance engine - since the
// end of round 4 with the staggered start and without a helper launch - is as fast or faster (us per call, instance | two bands | four
// bands: B = 256: 59.7 | 58.6 | 60.0; 288: 61.4 | 62.9 | 66.2; 320: 64.5 | 62.4 | 69.9; 384: 63.4 | 66.2 | 78.1; 448: 63.4 | 70.6 |
// 83.3; until then the bands held up to 400).  LA3D_ENGINE=band / opt_engine pins it for any batch, LA3D_BAND_MAXB moves the limit.
inline bool band_eligible(const FitParams& p, bool vec, bool sample) {
  const int e = p.opt_engine != LA3D_ENGINE_DEFAULT ? p.opt_engine : config().engine;
  if (e == LA3D_ENGINE_INSTANCE || e == LA3D_ENGINE_SPLIT) return false;
  if (!vec || sample || p.mask == nullptr || !band_frame_ok(p.H, p.W, band_count(p))) return false;
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

// run-length input is its own instantiation (it needs the LDS bit image), so the u8 kernels carry no decode code
template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED = false, int RET = 0>
int launch_fit(const FitParams& p, size_t lds, hipStream_t s, void* workspace = nullptr) {
  if (LDSMASK && p.rle_counts != nullptr) return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, LDSMASK ? 1 : 0, RET>(p, lds, s, workspace);
  if (LDSMASK && p.poly_xy != nullptr) return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, LDSMASK ? 2 : 0, RET>(p, lds, s, workspace);
  return launch_fit_inst<VEC, LDSMASK, SAMPLE, TILED, 0, RET>(p, lds, s, workspace);
}

constexpr int RETAIN_STEPS = 4;   // steps per wave the retaining build keeps in registers (x TG tiles x NWAVE waves = 128 tiles)
constexpr int RETAIN_MAXB_DEFAULT = 0;   // the retaining build is opt-in since round 4 (see below)
inline int retain_steps(const FitParams& p) {
  // The 128-VGPR build keeps up to 160 depth tiles per instance on chip between the passes: two
  // workgroups per CU instead of four, the second one of every CU staggered by the time a mask plane takes to stream.
  // Rounds 2-3 it was the default for u8 planes up to 1280 instances (108 vs 111 us at B = 1024).  Round 4's plain build - pass-B
  // culling, the LDS-kept tiles, the shorter pixel and reduction code - has overtaken it (u8 planes, us per call, retaining vs
  // plain, profiles/r04/r04_plain_vs_retaining.txt): config-2 masks B = 448 / 512 / 768 / 1024 / 1280 / 1536: 73.6 / 78.7 / 90.1 /
  // 106.5 / 128.5 / 143.8 vs 71.6 / 76.3 / 85.8 / 104.0 / 125.3 / 139.4; config-5 masks: 80.1 / 81.6 / 83.8 / 88.9 / 104.6 / 120.4 vs
  // 73.9 / 75.6 / 83.1 / 90.2 / 100.2 / 115.1.  So the plain build is the default everywhere; opt_build / LA3D_RETAIN=1 pin the
  // retaining one, LA3D_RETAIN_MAXB=n makes it the default again up to n instances.
  const Config& c = config();
  const int pin = p.opt_build != LA3D_BUILD_DEFAULT ? p.opt_build : c.retain;
  if (pin == LA3D_BUILD_PLAIN) return 0;
  if (pin == LA3D_BUILD_RETAINING) return RETAIN_STEPS;
  // a caller that has switched the launch order off is pipelining batches on several streams: that regime behaves like one
  // large batch, where the plain build wins by more (81.6 vs 87.0 us per 1024-instance call)
  if (!balance_enabled(p)) return 0;
  return p.B <= (c.retain_maxb >= 0 ? c.retain_maxb : RETAIN_MAXB_DEFAULT) ? RETAIN_STEPS : 0;
}

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
inline size_t rows_workspace_bytes(int B, int H, int W) {
  RowsArgs ra;
  if (!rows_plan(B, H, W, &ra)) return 0;
  return (((size_t)B * ra.nb * ROWS_PART_D * 8 + 255) & ~(size_t)255) + (((size_t)B * ra.nb * 2 * W * 4 + 255) & ~(size_t)255) + (size_t)B * 8 + 256;
}

// The plainest walk over one instance: thread t visits pixels t, t + NT, ... of the u8 plane, one at a time.  PASS 0: count and
// moments of (x', z'); PASS 1: the six extents (A0 / A1 / A2 as in `sweep`).  Used where a path is rare and registers are scarce.
template <int PASS>
__device__ inline void sweep_plain(const FitParams& p, const float* __restrict__ dpl, const unsigned char* __restrict__ mpl,
                                   const double* A0, const double* A1, const double* A2, int tid, double* acc, int* cnt, int* nmask) {
#pragma clang loop unroll(disable) vectorize(disable)
  for (int i = tid; i < p.HW; i += NT) {
    if (!mpl[i]) continue;
    if (PASS == 0) *nmask += 1;
    const float df = dpl[i];
    if (!finite_f32(df)) continue;
    unsigned u, v;
    pix_uv((unsigned)i, p.W, p.rcpW, &u, &v);
    const double ud = (double)u, vd = (double)v, d = (double)df;
    const double x = d * fma(A0[0], ud, fma(A0[1], vd, A0[2])), z = d * fma(A2[0], ud, fma(A2[1], vd, A2[2]));
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
  if (tid == 18) { sh->bad_ground = 0; sh->order_inst = inst; sh->sep_bad = 0; }
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
    if (sh->redo) { generic = true; __syncthreads(); }   // uniform: non-finite sums
  }
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
  const unsigned char* mpl = p.mask + (long long)inst * p.HW;
  if (generic) {
    double gacc[5] = {0, 0, 0, 0, 0};
    int cnt = 0, nmask = 0;
    sweep_plain<0>(p, dpl, mpl, sh->M, sh->M + 3, sh->M + 6, tid, gacc, &cnt, &nmask);
    stage_moments_to_axis(sh, p, inst, gacc, cnt, nmask, tid, wave, lane, false);
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
  if ((p.opt_build != LA3D_BUILD_DEFAULT ? p.opt_build : config().retain) == LA3D_BUILD_RETAINING) return false;
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
  p.lds_keep_off = 0;
  p.stagger_ticks = 0;
  p.band_test = config().band_test;
  p.sep_off = (config().sep == 0 || (opts && opts->build == LA3D_BUILD_PLAIN)) ? 1 : 0;
  p.band_trows = 0; p.band_arrive = nullptr; p.band_tag = 0; p.band_xch = nullptr;
  p.cull_min = config().cull_min > 0 ? config().cull_min : (mask != nullptr ? config().cull_min_u8 : CULL_MIN);
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
                                (eng == LA3D_ENGINE_DEFAULT || eng == LA3D_ENGINE_ROWS || eng == LA3D_ENGINE_ROWS2) &&   // (rows pinned but not applicable: as by default)
                                (p.opt_build != LA3D_BUILD_DEFAULT ? p.opt_build : config().retain) != LA3D_BUILD_RETAINING;
  {
    RowsArgs ra;
    if (rows_eligible(p, vec, sample, &ra)) return launch_fit_rows(p, ra, s, workspace);
  }
  if (!single_pass_call && band_eligible(p, vec, sample)) {   // u8 planes, 16 <= B <= 256 (or pinned): two / four workgroups per instance, ONE launch
    return band_count(p) == 4 ? launch_fit_bands<4>(p, s, workspace) : launch_fit_bands<2>(p, s, workspace);
  }
  if (!single_pass_call && !sample && p.frame_w == W && split_eligible(p, vec, ldsmask)) {   // (the split engine's decoders know no padded rows)
    const int rc = split_fit(p, workspace, s);   // (the split engine's final kernel does not project: one small follow-up launch)
    if (rc != LA3D_SUCCESS || !p.proj) return rc;
    return la3d_project_boxes(out, K, k_stride, image_index, B, p.proj_w, p.proj_h, p.proj, stream);   // (la3d_aux.hip)
  }
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
    // u8 planes only: run-length / polygon input has no mask stream to overlap, and with nothing to stream the plain build's four
    // workgroups per CU hide the passes' latency better (LA3D_RETAIN_NOMASK=1 forces the retaining build for measurements:
    // profiles/r03/r03_rle_poly.txt)
    const int ret = (mask != nullptr || config().retain_nomask || p.opt_build == LA3D_BUILD_RETAINING) ? retain_steps(p) : 0;
    for (int wg_per_cu = ret > 0 ? 2 : 4; wg_per_cu >= 1 && cap < want; --wg_per_cu) {
      const long budget = (160 * 1024 / wg_per_cu) & ~15L;
      cap = (budget - (long)fixed) / 2;
    }
    if (cap > ntiles) cap = ntiles;
    if (cap >= 64) {
      p.list_cap = (int)cap;
      if (ret > 0) {
        // one more kept step per wave in LDS when two workgroups per CU leave the room (NWAVE x LDS_KEEP_WAVE bytes)
        size_t tot = (fixed + ((size_t)cap * 2 > poly_stage ? (size_t)cap * 2 : poly_stage) + 15) & ~(size_t)15;
        const size_t keep_bytes = (size_t)NWAVE * LDS_KEEP_WAVE;
        if (config().ldskeep && tot + keep_bytes <= 80 * 1024) { p.lds_keep_off = (int)tot; tot += keep_bytes; }
        if (mask != nullptr && B > 256) {   // u8 planes: 256 workgroups stream 256 x H*W bytes at ~6 TB/s
          double us = 0.9 * 256.0 * (double)p.HW / 6.0e6;
          if (config().stagger_us >= 0) us = config().stagger_us;
          p.stagger_ticks = (int)(us * 100.0);
        }
        return launch_fit<true, true, false, true, RETAIN_STEPS>(p, tot, s, workspace);
      }
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
      a.opt_build < 0 || a.opt_build > LA3D_BUILD_RETAINING) {
    set_err("la3d_fit_instances_ex: bad opt_engine / opt_launch_order / opt_build");
    return LA3D_ERR_ARG;
  }
  const CallOpts co{a.opt_engine, a.opt_launch_order, a.opt_build, a.frame_width};
  return fit_dispatch(a.depth, a.depth_plane_stride, a.image_index, a.mask, a.rle_counts, a.rle_offsets, a.K, a.k_stride, a.ground,
                      a.sample_idx, a.B, a.H, a.W, a.out, a.status, a.aux, a.workspace, a.stream, "la3d_fit_instances_ex",
                      a.poly_xy ? &pa : nullptr, filter_on ? &fa : nullptr, a.proj ? &pr : nullptr, a.area_hint, &co);
}

}  // extern "C"
