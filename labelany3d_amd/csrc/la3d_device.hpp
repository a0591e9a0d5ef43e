// la3d_device.hpp — device-side building blocks shared by the fit engines (la3d.hip: one workgroup per
// instance; la3d_split.hip: band scan + tile-range-balanced passes).  Reference semantics cited per
// function (paths relative to /root/reference).
#pragma once
#ifndef LA3D_NT
#define LA3D_NT 512
#endif
#include <hip/hip_runtime.h>
#include <mutex>
#include <utility>
#include <vector>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "la3d.h"

namespace la3d {


constexpr int NT = LA3D_NT;      // threads per workgroup (fit_instances)
constexpr int NWAVE = NT / 64;   // wave64
constexpr int NTP = 256;         // threads per workgroup (fit_points)
constexpr int NWAVEP = NTP / 64;
constexpr double PI_2 = 1.57079632679489661923;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native vector: usable with nontemporal builtins

extern thread_local char g_err[256];
inline void set_err(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

// Process defaults of the scheduling decisions, read from the environment ONCE (config(), la3d.hip) and immutable afterwards:
// the hot path of the C-ABI never calls getenv.  Measurement scripts set the variables before the first call; a caller that
// wants a different decision for one call uses la3d_fit_args::opt_*.  Speed only - records never depend on any of it.
struct Config {
  int engine;          // LA3D_ENGINE=instance|split|band -> LA3D_ENGINE_*
  int bands;           // LA3D_BANDS=2|4 pins the workgroups per instance of the band engine (0: by batch size)
  int band_default;    // LA3D_BAND_DEFAULT=0: the band engine only when pinned
  int band_maxb;       // LA3D_BAND_MAXB: largest batch the band engine takes by default
  int rows_maxb;       // LA3D_ROWS_MAXB: largest batch the row engine takes by default (u8 planes, no ground array)
  int rows_fused;      // LA3D_ROWS_FUSED=0: the row engine in its two-launch form (a merge launch behind the band launch)
  int rows_wgs;        // LA3D_ROWS_WGS: workgroups the row engine spreads a batch over, at most (default 640)
  int balance;         // LA3D_BALANCE=0 -> launch order off by default
  int balance_rounds;  // LA3D_BALANCE_ROUNDS: batches up to this many resident sets are ordered (default 3)
  int build;           // LA3D_BUILD=plain|nocull (LA3D_RETAIN=0|1: the old spelling) -> LA3D_BUILD_PLAIN / LA3D_BUILD_NOCULL for every call
  int cull_min, cull_min_u8;   // LA3D_CULL_MIN (all inputs; 0: defaults), LA3D_CULL_MIN_U8 (u8 planes, default 128)
  int order_self;      // LA3D_ORDER_SELF=0: keep the estimate kernel in front of ordered launches of up to one resident set
  double stagger_us;   // LA3D_STAGGER_US (< 0: the computed default)
  double stagger_nomask_us;   // LA3D_STAGGER_NOMASK_US: the same for run-length / polygon input (default 0: off)
  int split_grid;      // LA3D_SPLIT_GRID (0: by batch size)
  int split_sub;       // LA3D_SPLIT_SUB (0: by batch size)
  int sep;             // LA3D_SEP=0: the separable single pass off (two passes for every camera)
  int band_test;       // LA3D_BAND_TEST (tests only): 1 = band 1 of every third instance never arrives and the watchdog is short (the
                       // takeover path runs); 2 = blocks permuted so that the bands of an instance sit on different XCDs
};
const Config& config();

// ------------------------------------------------------------------------------------------
// float64 -> float16 (round to nearest even, overflow to inf, gradual underflow) -> float64.
// Mirrors numpy's astype(float16) applied to the 8 corners at reference src/util_3dbox.py:165.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline double f16_round(double x) {
  if (x != x) return x;
  const double ax = fabs(x);
  if (ax >= 65520.0) return x > 0 ? INFINITY : -INFINITY;  // halfway to 65536 rounds to even = overflow
  double q;
  if (ax < 6.103515625e-05) {  // below 2^-14: half subnormals, fixed quantum 2^-24
    q = 5.9604644775390625e-08;
  } else {
    int e;
    (void)frexp(ax, &e);       // ax = m * 2^e, m in [0.5, 1)  ->  floor(log2 ax) = e - 1
    q = ldexp(1.0, e - 11);    // 10 explicit mantissa bits
  }
  return rint(x / q) * q;      // both scalings are exact powers of two; rint is RNE
}

// ------------------------------------------------------------------------------------------
// small fp64 algebra, done by one thread per box
// ------------------------------------------------------------------------------------------
// 3x3 inverse by Gaussian elimination with partial pivoting on [A | I] (np.linalg.inv is LAPACK
// gesv: same elimination order, same pivot rule - the first row of largest magnitude; reference src/util.py:56).
// Straight-line code (round 4): the pivot row is swapped in with selects and every index is a compile-time constant, so the
// rows stay in registers - the dynamically indexed form cost plan_kernel and unproject_batch_kernel 160 B of scratch per lane.
__device__ inline void inv3(const double* A, double* X) {
  double a[3][6];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      a[i][j] = A[i * 3 + j];
      a[i][3 + j] = (i == j) ? 1.0 : 0.0;
    }
  {   // column 0: pivot among rows 0, 1, 2 (strict > keeps the first maximum, like idamax)
    const bool p1 = fabs(a[1][0]) > fabs(a[0][0]);
    const bool p2 = fabs(a[2][0]) > (p1 ? fabs(a[1][0]) : fabs(a[0][0]));
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double r0 = a[0][j], r1 = a[1][j], r2 = a[2][j];
      a[0][j] = p2 ? r2 : (p1 ? r1 : r0);
      a[1][j] = (p1 && !p2) ? r0 : r1;
      a[2][j] = p2 ? r0 : r2;
    }
    const double inv = 1.0 / a[0][0];
#pragma unroll
    for (int r = 1; r < 3; ++r) {
      const double f = a[r][0] * inv;
#pragma unroll
      for (int j = 0; j < 6; ++j) a[r][j] -= f * a[0][j];
    }
  }
  {   // column 1: pivot among rows 1, 2
    const bool q = fabs(a[2][1]) > fabs(a[1][1]);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double r1 = a[1][j], r2 = a[2][j];
      a[1][j] = q ? r2 : r1;
      a[2][j] = q ? r1 : r2;
    }
    const double inv = 1.0 / a[1][1];
    const double f = a[2][1] * inv;
#pragma unroll
    for (int j = 1; j < 6; ++j) a[2][j] -= f * a[1][j];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // back substitution per right-hand side
#pragma unroll
    for (int r = 2; r >= 0; --r) {
      double s = a[r][3 + j];
#pragma unroll
      for (int k = r + 1; k < 3; ++k) s -= a[r][k] * X[k * 3 + j];
      X[r * 3 + j] = s / a[r][r];
    }
  }
}

// Same inverse by cofactors: straight-line code without pivot bookkeeping (no dynamically indexed arrays, so
// it can live inside the streaming kernel without costing it scratch memory).  Camera intrinsics are
// well conditioned; it agrees with the elimination above to a few ulp.
__device__ inline void inv3_cofactor(const double* A, double* X) {
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
  const double inv = 1.0 / (a * c00 + b * c01 + c * c02);
  X[0] = c00 * inv; X[1] = (c * h - b * i) * inv; X[2] = (b * f - c * e) * inv;
  X[3] = c01 * inv; X[4] = (a * i - c * g) * inv; X[5] = (c * d - a * f) * inv;
  X[6] = c02 * inv; X[7] = (b * g - a * h) * inv; X[8] = (a * e - b * d) * inv;
}

// Camera intrinsics: the cofactor inverse, with the last row exact for an affine camera (K's last row (0, 0, 1): the true inverse
// has exactly (0, 0, 1) there, and so has LAPACK's - back substitution divides 1 by 1 -, while (a e) * (1 / (a e)) may round to
// 1 - ulp).  z = depth EXACTLY for an un-grounded call is what the separable single pass of la3d.hip keys on.
__device__ inline void inv3_camera(const double* A, double* X) {
  inv3_cofactor(A, X);
  if (A[6] == 0.0 && A[7] == 0.0 && A[8] == 1.0) { X[6] = 0.0; X[7] = 0.0; X[8] = 1.0; }
}

// Rg of reference src/util_3dbox.py:128-134 (+ :20-25, :37-55).  ground == nullptr or a NaN
// first component selects the identity ("ground_equ is None").  Returns 1 when the matrix
// is not finite (parallel / antiparallel / zero ground vector -> 0/0).
__device__ inline int ground_rotation(const double* ground, double* Rg) {
  for (int i = 0; i < 9; ++i) Rg[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (ground == nullptr) return 0;
  double g0 = ground[0], g1 = ground[1], g2 = ground[2];
  if (g0 != g0) return 0;
  // dot([0,-1,0], g) = 0*g0 + (-1)*g1 + 0*g2  <= 0  -> negate           (:129-131)
  const double dotp = 0.0 * g0 + (-1.0) * g1 + 0.0 * g2;
  if (dotp <= 0) { g0 = -g0; g1 = -g1; g2 = -g2; }
  const double nrm = sqrt(g0 * g0 + g1 * g1 + g2 * g2);  // normalize(): unchanged when 0 (:20-25)
  if (nrm != 0) { g0 /= nrm; g1 /= nrm; g2 /= nrm; }
  // vec1 = [0,-1,0];  axis = cross(vec1, vec2);  cos = dot(vec1, vec2)   (:43-44)
  const double ax = (-1.0) * g2 - 0.0 * g1;
  const double ay = 0.0 * g0 - 0.0 * g2;
  const double az = 0.0 * g1 - (-1.0) * g0;
  const double cs = 0.0 * g0 + (-1.0) * g1 + 0.0 * g2;
  const double k[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
  const double an = sqrt(ax * ax + ay * ay + az * az);
  const double f = (1.0 - cs) / (an * an);
  int bad = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double kk = 0;
      for (int m = 0; m < 3; ++m) kk += k[i * 3 + m] * k[m * 3 + j];
      const double r = ((i == j) ? 1.0 : 0.0) + k[i * 3 + j] + kk * f;
      Rg[i * 3 + j] = r;
      if (!(fabs(r) <= 1.79769313486231570815e308)) bad = 1;
    }
  return bad;
}

// scikit-learn PCA(2) first axis in closed form + svd_flip(u_based_decision=False)
// (reference src/util_3dbox.py:181-186).  Raw sums -> (cos yaw, sin yaw), eigen-gap.
// The reference goes eigenvector -> atan2 -> cos/sin; here the unit eigenvector (vx, vz) IS
// (cos yaw, sin yaw), obtained without trigonometry from cos 2t = (a-c)/2r, sin 2t = b/r by the
// stable half-angle form (agrees with the trig route to ~1 ulp; keeps fp64 libm range reduction out
// of the streaming kernel's register budget and off the per-workgroup serial path).
//
// Returns true when the sums are ILL-CONDITIONED for this (round 6, found by profiles/r06/fuzz_engines.py): the covariance entries
// a, b, c are differences of raw second moments and carry an absolute rounding error of a few 2^-52 (sxx + szz), so their relative
// error is ~2^-50 kappa with kappa = (sxx + szz) / l1 = (mean square distance from the ORIGIN of the sums) / (variance along the
// axis).  The reference centres the points before its SVD / eigh and has no such term.  kappa > 2^17 (a cloud whose spread in the
// x'z' plane is below ~1/360 of its distance: a one-pixel column at constant depth, a sliver 0.5 mm wide 48 m away) would put the
// axis error above ~1e-10 / gap: the caller then re-runs the moments about a pivot next to the cloud (the mean of the first pass;
// the sums handed over afterwards are translated and this function is translation invariant), or - where a path has no second
// pass, and for clouds that stay unresolved about their own mean (no spread at all: the reference's axis is rounding noise too) -
// reports gap = 0, the documented "don't care" value.
constexpr double ILL_KAPPA = 131072.0;   // 2^17
__device__ inline bool axis_from_sums(double n, double sx, double sz, double sxx, double sxz, double szz,
                                      double* cyaw, double* syaw, double* gap) {
  const double a = sxx - sx * sx / n;
  const double c = szz - sz * sz / n;
  const double b = sxz - sx * sz / n;
  const double half = 0.5 * (a - c);
  const double rad = sqrt(half * half + b * b);
  const double l1 = 0.5 * (a + c) + rad;
  *gap = (l1 > 0) ? 2.0 * rad / l1 : 0.0;
  const bool ill = !((sxx + szz) <= ILL_KAPPA * l1);   // (also true for NaN sums and for l1 <= 0 < sxx + szz)
  if (b == 0 && a == c) {  // exact isotropy: eigh branch (n >= 20) -> yaw = pi/2; SVD branch recorded as 0
    if (n >= 20) { *cyaw = 6.123233995736766e-17; *syaw = 1.0; }  // np.cos(pi/2), np.sin(pi/2)
    else { *cyaw = 1.0; *syaw = 0.0; }
    return ill;
  }
  const double c2 = half / rad, s2 = b / rad;
  double vx, vz;  // (cos t, sin t), t in [-pi/2, pi/2]
  if (c2 >= 0) { vx = sqrt(0.5 * (1.0 + c2)); vz = 0.5 * s2 / vx; }
  else { vz = copysign(sqrt(0.5 * (1.0 - c2)), s2); vx = 0.5 * s2 / vz; }
  if (fabs(vx) >= fabs(vz)) {   // svd_flip: the larger-|.| entry becomes positive, first index on ties
    if (vx < 0) { vx = -vx; vz = -vz; }
  } else if (vz < 0) {
    vx = -vx; vz = -vz;
  }
  *cyaw = vx; *syaw = vz;
  return ill;
}

// Steps (6)-(12) of estimate_bbox (reference src/util_3dbox.py:157-176) from the extents.
__device__ inline void write_box(double* out, const double* Rg, double cyaw, double syaw,
                                 double xmin, double xmax, double ymin, double ymax, double zmin, double zmax) {
  const double dx = xmax - xmin, dy = ymax - ymin, dz = zmax - zmin;
  const double c[3] = {(xmin + xmax) / 2, (ymin + ymax) / 2, (zmin + zmax) / 2};
  const double h[3] = {dx / 2, dy / 2, dz / 2};
  // rotate_y(-yaw): cos(-y) = cos y, sin(-y) = -sin y                     (:28-34)
  const double Ry[9] = {cyaw, 0, -syaw, 0, 1, 0, syaw, 0, cyaw};
  // center_cam = Rg^T @ (rotate_y(-yaw) @ c)                              (:172-173)
  double w[3];
  for (int i = 0; i < 3; ++i) w[i] = Ry[i * 3] * c[0] + Ry[i * 3 + 1] * c[1] + Ry[i * 3 + 2] * c[2];
  for (int i = 0; i < 3; ++i) out[i] = Rg[i] * w[0] + Rg[3 + i] * w[1] + Rg[6 + i] * w[2];
  out[3] = dz; out[4] = dy; out[5] = dx;                                // dimension = [dz, dy, dx]  (:175)
  // R_cam = Rg^T @ rotate_y(-yaw)                                         (:176)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      out[6 + i * 3 + j] = Rg[i] * Ry[j] + Rg[3 + i] * Ry[3 + j] + Rg[6 + i] * Ry[6 + j];
  // 8 corners, fixed sign order (:83-92), fp16 cast (:165), un-rotate with rotate_y(-yaw) then Rg (:168-169)
  const int sg[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
  for (int v = 0; v < 8; ++v) {
    double g[3], r[3];
    for (int i = 0; i < 3; ++i) g[i] = f16_round(sg[v][i] * h[i] + c[i]);
    for (int i = 0; i < 3; ++i) r[i] = Ry[i * 3] * g[0] + Ry[i * 3 + 1] * g[1] + Ry[i * 3 + 2] * g[2];
    for (int i = 0; i < 3; ++i) out[15 + v * 3 + i] = r[0] * Rg[i * 3] + r[1] * Rg[i * 3 + 1] + r[2] * Rg[i * 3 + 2];
  }
}

// Same record, written by one wave: lanes 0..7 take one corner each, lane 8 center + dims, lanes 9..11 one
// R_cam row each (the single-thread version is ~3000 dependent fp64 instructions, i.e. ~10 us of pure
// latency when a whole kernel consists of it).  Bit-identical to write_box.
// DPP cross-lane moves (see the wave reductions below)
// (full row / bank masks and a control whose source lanes all exist: written as a move without an `old` value and with
// bound_ctrl, the compiler folds it into the consuming 32-bit operation - v_add_u32_dpp, v_min_u32_dpp, ...: one instruction
// per reduction step instead of three; fp64 steps become two moves + the operation instead of four + it)
template <int CTRL>
__device__ inline int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
__device__ inline double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

// one corner through the image's K: (K @ P)[:2] / (K @ P)[2]  (project_to_2d, reference src/tools/combine_results.py:105-108)
__device__ inline void project_corner(const double* k, double x, double y, double z, double* px, double* py) {
  const double hx = k[0] * x + k[1] * y + k[2] * z, hy = k[3] * x + k[4] * y + k[5] * z, hz = k[6] * x + k[7] * y + k[8] * z;
  *px = hx / hz; *py = hy / hz;
}

// proj (optional, with the image's K and size): the record's 2-D boxes as la3d_project_boxes writes them - bbox2D_proj (4) and its
// clamp to the frame (4) - from the corners this call has in registers (lanes 0..7; an 8-lane DPP min / max)
__device__ inline void write_box_wave(double* out, const double* Rg, double cyaw, double syaw, double xmin, double xmax,
                                      double ymin, double ymax, double zmin, double zmax, int lane,
                                      double* proj = nullptr, const double* Kp = nullptr, double Wd = 0, double Hd = 0) {
  double cpx = INFINITY, cpy = INFINITY, cqx = -INFINITY, cqy = -INFINITY;   // this lane's corner, projected
  int cbad = 0;
  const double dx = xmax - xmin, dy = ymax - ymin, dz = zmax - zmin;
  const double c[3] = {(xmin + xmax) / 2, (ymin + ymax) / 2, (zmin + zmax) / 2};
  const double h[3] = {dx / 2, dy / 2, dz / 2};
  const double Ry[9] = {cyaw, 0, -syaw, 0, 1, 0, syaw, 0, cyaw};
  if (lane < 8) {
    const int sx = (lane == 1 || lane == 2 || lane == 5 || lane == 6) ? 1 : -1;
    const int sy = (lane == 2 || lane == 3 || lane == 6 || lane == 7) ? 1 : -1;
    const int sz = (lane >= 4) ? 1 : -1;
    double g[3], r[3];
    g[0] = f16_round(sx * h[0] + c[0]); g[1] = f16_round(sy * h[1] + c[1]); g[2] = f16_round(sz * h[2] + c[2]);
    for (int i = 0; i < 3; ++i) r[i] = Ry[i * 3] * g[0] + Ry[i * 3 + 1] * g[1] + Ry[i * 3 + 2] * g[2];
    double cv[3];
    for (int i = 0; i < 3; ++i) { cv[i] = r[0] * Rg[i * 3] + r[1] * Rg[i * 3 + 1] + r[2] * Rg[i * 3 + 2]; out[15 + lane * 3 + i] = cv[i]; }
    if (proj) {
      double px, py;
      project_corner(Kp, cv[0], cv[1], cv[2], &px, &py);
      cbad = (px != px || py != py) ? 1 : 0;
      cpx = cqx = px; cpy = cqy = py;
    }
  } else if (lane == 8) {
    double w[3];
    for (int i = 0; i < 3; ++i) w[i] = Ry[i * 3] * c[0] + Ry[i * 3 + 1] * c[1] + Ry[i * 3 + 2] * c[2];
    for (int i = 0; i < 3; ++i) out[i] = Rg[i] * w[0] + Rg[3 + i] * w[1] + Rg[6 + i] * w[2];
    out[3] = dz; out[4] = dy; out[5] = dx;
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i)   // lanes 9, 10, 11: one row of R_cam each (static indices: no scratch)
      if (lane == 9 + i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) out[6 + i * 3 + j] = Rg[i] * Ry[j] + Rg[3 + i] * Ry[3 + j] + Rg[6 + i] * Ry[6 + j];
      }
  }
  if (proj) {   // uniform
    cpx = fmin(cpx, dpp_f64<DPP_XOR1>(cpx)); cpx = fmin(cpx, dpp_f64<DPP_XOR2>(cpx)); cpx = fmin(cpx, dpp_f64<DPP_HALF_MIRROR>(cpx));
    cpy = fmin(cpy, dpp_f64<DPP_XOR1>(cpy)); cpy = fmin(cpy, dpp_f64<DPP_XOR2>(cpy)); cpy = fmin(cpy, dpp_f64<DPP_HALF_MIRROR>(cpy));
    cqx = fmax(cqx, dpp_f64<DPP_XOR1>(cqx)); cqx = fmax(cqx, dpp_f64<DPP_XOR2>(cqx)); cqx = fmax(cqx, dpp_f64<DPP_HALF_MIRROR>(cqx));
    cqy = fmax(cqy, dpp_f64<DPP_XOR1>(cqy)); cqy = fmax(cqy, dpp_f64<DPP_XOR2>(cqy)); cqy = fmax(cqy, dpp_f64<DPP_HALF_MIRROR>(cqy));
    cbad |= dpp_i32<DPP_XOR1>(cbad); cbad |= dpp_i32<DPP_XOR2>(cbad); cbad |= dpp_i32<DPP_HALF_MIRROR>(cbad);
    if (lane == 0) {
      if (cbad) { for (int j = 0; j < 8; ++j) proj[j] = NAN; }   // Python's min() / max() over NaN are order dependent: report NaN
      else {
        proj[0] = cpx; proj[1] = cpy; proj[2] = cqx; proj[3] = cqy;
        proj[4] = fmax(0.0, cpx); proj[5] = fmax(0.0, cpy); proj[6] = fmin(Wd, cqx); proj[7] = fmin(Hd, cqy);
      }
    }
  }
}

__device__ inline void write_nan_box(double* out) {
  for (int i = 0; i < LA3D_REC; ++i) out[i] = NAN;
}

// ------------------------------------------------------------------------------------------
// wave64 reductions (fixed butterfly order -> deterministic)
// ------------------------------------------------------------------------------------------
// Cross-lane steps are DPP moves (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: after them every lane of a 16-lane row holds
// the row's result), the four rows are combined through v_readlane.  The same reductions written with __shfl_xor compile to
// ds_bpermute_b32 chains the scheduler barely overlaps: 72 LDS round trips = 1.7 us per reduction stage of the fit kernel,
// twice per instance (profiles/timeline.py, round 2).  All 64 lanes must be active.

__device__ inline double wave_sum(double v) {
  v += dpp_f64<DPP_XOR1>(v);
  v += dpp_f64<DPP_XOR2>(v);
  v += dpp_f64<DPP_HALF_MIRROR>(v);
  v += dpp_f64<DPP_MIRROR>(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ inline double wave_min(double v) {
  v = fmin(v, dpp_f64<DPP_XOR1>(v));
  v = fmin(v, dpp_f64<DPP_XOR2>(v));
  v = fmin(v, dpp_f64<DPP_HALF_MIRROR>(v));
  v = fmin(v, dpp_f64<DPP_MIRROR>(v));
  return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ inline double wave_max(double v) {
  v = fmax(v, dpp_f64<DPP_XOR1>(v));
  v = fmax(v, dpp_f64<DPP_XOR2>(v));
  v = fmax(v, dpp_f64<DPP_HALF_MIRROR>(v));
  v = fmax(v, dpp_f64<DPP_MIRROR>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ inline int wave_sum_i(int v) {
  v += dpp_i32<DPP_XOR1>(v);
  v += dpp_i32<DPP_XOR2>(v);
  v += dpp_i32<DPP_HALF_MIRROR>(v);
  v += dpp_i32<DPP_MIRROR>(v);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__device__ inline int wave_min_i(int v) {
  v = min(v, dpp_i32<DPP_XOR1>(v)); v = min(v, dpp_i32<DPP_XOR2>(v)); v = min(v, dpp_i32<DPP_HALF_MIRROR>(v)); v = min(v, dpp_i32<DPP_MIRROR>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ inline int wave_max_i(int v) {
  v = max(v, dpp_i32<DPP_XOR1>(v)); v = max(v, dpp_i32<DPP_XOR2>(v)); v = max(v, dpp_i32<DPP_HALF_MIRROR>(v)); v = max(v, dpp_i32<DPP_MIRROR>(v));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// The four quantities of the reference's instance filter (analyze_mask / get_maximum_height, src/util.py:291-335) from a
// row-major bit image in LDS, by all NTH threads of the workgroup (threads over rows): out4 = area, rows holding a pixel,
// last - first + 1, pixels inside the `boundary`-px border strips (a pixel in a corner counts twice, as the reference's
// four slice sums do).  red: LDS, 5 * NTH/64 ints.  The result is valid in every thread; ends with a barrier.
template <int NTH>
__device__ inline void bits_filter_stats(const unsigned* bits, int H, int W, int boundary, int* red, int tid, int* out4, int row_bits = 0) {
  // (row_bits: bits per row of the image in memory when the rows are padded - la3d_fit_args::frame_width -, 0 = W)
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NTH / 64;
  const int bc = min(boundary, W), br = min(boundary, H);
  int area = 0, edge = 0, rows = 0, first = H, last = -1;
  for (int r = tid; r < H; r += NTH) {
    int cnt = 0, e = 0;
    const unsigned base = (unsigned)r * (unsigned)(row_bits > 0 ? row_bits : W);
    for (int c0 = 0; c0 < W; c0 += 32) {       // 32 columns at a time (unaligned rows: the word is assembled from two)
      const unsigned i = base + c0, wi = i >> 5, sh = i & 31;
      unsigned w = bits[wi] >> sh;
      if (sh && c0 + (32 - (int)sh) < W) w |= bits[wi + 1] << (32 - sh);
      const int valid = min(32, W - c0);
      if (valid < 32) w &= (1u << valid) - 1u;
      cnt += __popc(w);
      const int nlo = min(max(bc - c0, 0), 32), fhi = min(max(W - bc - c0, 0), 32);
      e += __popc(nlo >= 32 ? w : (w & ((1u << nlo) - 1u))) + __popc(fhi >= 32 ? 0u : (w & ~((1u << fhi) - 1u)));
    }
    area += cnt;
    edge += e + cnt * ((r < br ? 1 : 0) + (r >= H - br ? 1 : 0));
    if (cnt != 0) { rows += 1; first = min(first, r); last = max(last, r); }
  }
  area = wave_sum_i(area); edge = wave_sum_i(edge); rows = wave_sum_i(rows);
  first = wave_min_i(first); last = wave_max_i(last);
  if (lane == 0) { red[wave] = area; red[NW + wave] = edge; red[2 * NW + wave] = rows; red[3 * NW + wave] = first; red[4 * NW + wave] = last; }
  __syncthreads();
  int a = 0, ed = 0, rw = 0, f = H, l = -1;
#pragma unroll
  for (int w = 0; w < NW; ++w) { a += red[w]; ed += red[NW + w]; rw += red[2 * NW + w]; f = min(f, red[3 * NW + w]); l = max(l, red[4 * NW + w]); }
  out4[0] = a; out4[1] = rw; out4[2] = (l >= f) ? l - f + 1 : 0; out4[3] = ed;
  __syncthreads();
}

// wave-uniform double -> SGPR pair (the value is identical in every lane by construction)
__device__ inline double uniform_f64(double v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}

// XCD-aware block -> work-item map: the dispatcher is observed to place block b on XCD b % 8
// (speed only, never correctness), so consecutive instances — which share an image's depth
// plane in the shared-depth layout — land on one XCD's L2.  Bijective for any nb.
__device__ inline int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (b >> 3);
}

// 4 mask bytes -> 4 bits (bit k = byte k non-zero)
// 16 pixels (four words of four bytes) -> 16-bit pattern, bit 4*k + j = byte j of word k is non-zero.  The per-byte flags
// (0x80 or 0) are gathered with two dot products per half: sum flag_j * 2^(4k+j) = 128 * pattern.
__device__ inline unsigned nz16(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  const unsigned f0 = (((w0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w0) & 0x80808080u;
  const unsigned f1 = (((w1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w1) & 0x80808080u;
  const unsigned f2 = (((w2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w2) & 0x80808080u;
  const unsigned f3 = (((w3 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w3) & 0x80808080u;
  const unsigned lo = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
  const unsigned hi = __builtin_amdgcn_udot4(f3, 0x80402010u, __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}

__device__ inline unsigned nz4(unsigned w) {
  const unsigned t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;  // high bit of each non-zero byte
  return ((t >> 7) * 0x01020408u) >> 24;                                       // gather bits 0,8,16,24 -> 0..3
}
__device__ inline bool finite_f32(float d) { return (__float_as_uint(d) & 0x7f800000u) != 0x7f800000u; }

// Raw fp64 min/max.  fmin()/fmax() on a loop-carried accumulator make hipcc emit a canonicalising
// v_max_f64 x,x before every use (it cannot prove the accumulator is not a signalling NaN): +1 DP
// instruction per min/max.  The hardware instructions already implement IEEE minNum/maxNum — a quiet
// NaN operand returns the OTHER operand — which is exactly what the masked walk relies on: pixels
// that are unmasked or non-finite carry a NaN depth and are ignored by the extents.
__device__ inline double dmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline double dmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct FitParams {
  const float* depth;
  long long depth_plane_stride;
  const int* image_index;
  const unsigned char* mask;
  const double* K;
  int k_stride;
  const double* ground;
  const int* sample_idx;
  int B, H, W, HW;
  int nwords;          // ceil(HW / 32) bit-image words
  int mask_lds_bytes;  // bit-image bytes in LDS (16-aligned), 0 when the image does not fit
  int rows_aligned;    // W % 4 == 0: a 4-pixel quad never straddles a row
  float rcpW;
  double* geo;         // workspace: [B][GEO_D]
  const int* rle_counts;        // masks given as COCO run lengths (column-major, zeros first) instead of u8 planes
  const long long* rle_offsets; // [B+1] into rle_counts
  const int* poly_xy;           // masks given as polygon parts: int32 (x, y) pairs ...
  const long long* poly_ring_off;   // ... [R+1] point offsets of the parts ...
  const long long* poly_inst_rings; // ... [B+1] part offsets of the instances
  int ntx, nty;        // TILED: tiles of 32 px x 8 rows (ntx = W/32, nty = ceil(H/8))
  int tiles_per_wave;  // TILED: ceil(ntx*nty / NWAVE)
  int list_cap;        // TILED: entries of the active-tile list that fit the LDS budget
  float rcp_ntx;       // TILED: 1 / ntx
  // size-balanced launch order (order_nch > 0; otherwise workgroup b fits instance xcd_remap(b)): sort keys per instance from
  // the estimate kernel, or built on the fly from area_hint; every workgroup ranks the <= ORDER_CHUNK keys of its chunk itself
  const unsigned* order_keys;
  // self-estimating launch (round 4; order_self != 0): no helper kernel - workgroup b estimates the key of instance b in its prologue
  // and publishes it with a per-call nonce; order_select waits for the nonces of its chunk (la3d.hip: estimate_publish)
  unsigned long long* order_flags;
  unsigned long long order_nonce;
  int order_self, est_step;
  int order_nch;       // chunks of consecutive instances (ceil(B / ORDER_CHUNK)), 0 = launch order off
  int order_resident;  // workgroups of the grid that are resident at once
  int order_shift;     // area_hint >> order_shift fits 18 bits
  int cull_min;        // pass-B culling: instances with at least this many active tiles plan (cull_plan)
  int stagger_ticks;   // u8 planes: resident groups of 256 workgroups start this many 100 MHz ticks apart (0: off; see fit_instances_kernel)
  // instance filter fused into the fit (run-length / polygon input): boundary < 0 = off
  int filter_boundary, filter_min_area, filter_max_edge;
  int* filter_stats;   // [B][4] area, rows, span, edge (may be null)
  // optional epilogue: bbox2D_proj | bbox2D_trunc of every record ([B][8], la3d_project_boxes' layout), frame size proj_w x proj_h
  double* proj;
  double proj_w, proj_h;
  const int* area_hint;   // [B] mask areas known to the caller (launch order without the estimate pass), or null
  int opt_engine, opt_order, opt_build;   // per-call overrides (la3d_fit_args::opt_*; host side only), 0 = the library's choice
  // band engine (fit_bands_kernel): tile rows per band (the last band takes the remainder), arrival counters [B][4] (zeroed before
  // the launch) and the exchange area [B][NB * 22] doubles, both in the workspace
  int band_trows;
  unsigned long long* band_arrive;   // [B][4] arrival words: 48-bit per-call tag | 16-bit count (tagged_arrive in la3d.hip): never cleared
  unsigned long long band_tag;
  double* band_xch;
  int band_test;       // Config::band_test
  int sep_off;         // 1: never take the separable single pass (LA3D_SEP=0, opt_build = LA3D_BUILD_PLAIN)
  double* out;
  int* status;
  double* aux;
  int frame_w;         // la3d_fit_args::frame_width (run-length / polygon input): image columns of the W-wide planes; == W when not given
};

// per-instance geometry in the workspace (20 doubles = 160 B), written by the split engine's plan_kernel (geo_one)
// (the instance engine keeps it in LDS)
constexpr int GEO_D = 20;  // M[9] (= Rg^T Kinv : p' = d * (M @ [u,v,1])), Rg[9], bad_ground, pad

struct alignas(16) Shared {
  double part[NWAVE][7];
  double M[9];     // Rg^T * Kinv, computed in-kernel by one lane while the others stream the mask
  double Rg[9];
  double cyaw, syaw;
  int cnt[NWAVE];
  int nmask[NWAVE];
  unsigned scan[NWAVE];
  int n_valid;
  int st;
  int bad_ground;
  int redo;        // optimistic pass A met a non-finite masked depth: run the checked passes
  double gap;      // relative eigenvalue gap (aux[3]), kept for the deferred aux write
  int nm;          // mask pixels (aux[2])
  int order_inst;  // the instance this workgroup fits (size-balanced launch order)
  unsigned qhead;  // pass B: head of the LDS work queue over the survivor list
  int sep_bad;     // separable single pass: a NaN / inf / negative depth under the mask - re-run the general path
  unsigned qpad[2];
#ifdef LA3D_TIMELINE
  double* tl;      // measurement build: this workgroup's stamp row (profiles/timeline.py)
#endif
};
#ifdef LA3D_TIMELINE
#define LA3D_SUBSTAMP(sh, k) do { if (threadIdx.x == 0 && (sh)->tl) (sh)->tl[k] = (double)wall_clock64(); } while (0)
#else
#define LA3D_SUBSTAMP(sh, k) do { } while (0)
#endif

__device__ inline void pix_uv(unsigned i, int W, float rcpW, unsigned* u, unsigned* v) {
  unsigned vv = (unsigned)((float)i * rcpW);
  int r = (int)i - (int)(vv * (unsigned)W);
  if (r < 0) { vv -= 1; r += W; }
  else if (r >= W) { vv += 1; r -= W; }
  *u = (unsigned)r;
  *v = vv;
}

// one pixel quad of a tile: PASS 0 accumulates count + moments, PASS 1 the six extents.
// r0/r1/r2: ray components at the quad's first pixel; a00/a10/a20: their per-pixel (u+1) increments.
// CHK = false is the optimistic form: only the mask bit gates a pixel.  Its results are identical to CHK = true as long as
// every masked depth is finite; a non-finite one turns the fp64 sums of pass 0 into inf/NaN for good (inf and NaN are
// sticky under + and fma), which the caller detects after the reduce and answers by re-running the checked form.
// SPEC (round 4, the "VALU diet"): the form for a camera without ground rotation and without skew - by far the common call (config
// 2, every un-grounded call).  Bit-identical to the general form, just fewer instructions:
//   pass 0: row 2 of M is exactly (0, 0, 1), so z = d * 1 = d: no z product, no ray increment for it (8 instead of 10 fp64 ops);
//   pass 1: M[1][0] == 0, so the y ray r1 is constant along the quad: multiplication by a constant is monotone under rounding,
//           hence min_k fl(d_k r1) is fl(dmin r1) or fl(dmax r1) - the quad's y extent costs two fp32 min3 / max3 pairs on the raw
//           depths plus six fp64 operations per QUAD instead of four fp64 operations per PIXEL.
__device__ inline float min3_f32(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ inline float max3_f32(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ inline float min_f32_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline float max_f32_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// PIV (pass 0, checked form only): the moments about the pivot (px0, pz0) - the re-run of an ill-conditioned instance (axis_from_sums)
template <int PASS, bool CHK = true, bool SPEC = false, bool PIV = false>
__device__ inline void quad_math(unsigned nib, const unsigned* db, double r0, double r1, double r2, double a00,
                                 double a10, double a20, double* s, int* n, double px0 = 0.0, double pz0 = 0.0) {
  static_assert(!PIV || (PASS == 0 && CHK && !SPEC), "the pivot exists in the checked general pass A only");
  float fk[4];   // SPEC pass 1: the quad's depths with NaN for invalid pixels (ignored by min / max)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int m;  // 0 / -1 validity word
    if (CHK) {  // mask bit k set AND exponent field != 0xff
      const int fin = ((int)(db[k] & 0x7fffffffu) - 0x7f800000) >> 31;
      m = fin & -(int)((nib >> k) & 1u);
    } else {
      m = -(int)((nib >> k) & 1u);
    }
    if (PASS == 0) {
      const double d = (double)__uint_as_float(db[k] & (unsigned)m);   // invalid -> +0.0
      if (SPEC) {
        const double x = d * r0;
        s[0] += x; s[1] += d;
        s[2] = fma(x, x, s[2]); s[3] = fma(x, d, s[3]); s[4] = fma(d, d, s[4]);
      } else {
        double x = d * r0, z = d * r2;
        if (PIV) { x = fma(d, r0, m ? -px0 : 0.0); z = fma(d, r2, m ? -pz0 : 0.0); }   // (an invalid pixel still contributes nothing)
        s[0] += x; s[1] += z;
        s[2] = fma(x, x, s[2]); s[3] = fma(x, z, s[3]); s[4] = fma(z, z, s[4]);
      }
      if (CHK) *n -= m;   // the optimistic form does not count: its caller takes the mask popcount
    } else {
      const unsigned bits = db[k] | ~(unsigned)m;                      // invalid -> NaN, ignored by min/max
      const double d = (double)__uint_as_float(bits);
      const double x = d * r0, z = d * r2;
      s[0] = dmin(s[0], x); s[1] = dmax(s[1], x);
      s[4] = dmin(s[4], z); s[5] = dmax(s[5], z);
      if (SPEC) {
        fk[k] = __uint_as_float(bits);
      } else {
        const double y = d * r1;
        s[2] = dmin(s[2], y); s[3] = dmax(s[3], y);
        r1 += a10;
      }
    }
    r0 += a00;
    if (!(SPEC && PASS == 0)) r2 += a20;   // next pixel of the row: u + 1
  }
  if (SPEC && PASS == 1) {
    const double dlo = (double)min_f32_raw(min3_f32(fk[0], fk[1], fk[2]), fk[3]);
    const double dhi = (double)max_f32_raw(max3_f32(fk[0], fk[1], fk[2]), fk[3]);
    const double ya = dlo * r1, yb = dhi * r1;
    s[2] = dmin(s[2], dmin(ya, yb)); s[3] = dmax(s[3], dmax(ya, yb));
  }
}


// COCO run lengths -> row-major 1-bit-per-pixel image in LDS (the decode step of pycocotools' rleDecode,
// called by the reference at src/util.py:367,401-402).  Runs are over the (H, W) mask in COLUMN-major order,
// alternating zeros / ones, zeros first.  NTH threads take NTH runs per step: a workgroup scan of the run
// lengths gives every run its start.  Word-aligned rows (W % 32 == 0, scratch given): toggle form - two ds_xor per run, then
// one column-wise XOR scan of the image (column_xor_scan).  Otherwise each wave paints the ones-runs of its 64 lanes one
// after the other, pixel by pixel (lanes = consecutive pixels of the run = consecutive rows).  wtot: LDS, NTH/64 words.
// Returns (per thread) the number of mask pixels it accounted for; bits must hold ceil(H*W/32) words and is zeroed here.
// Column-wise inclusive XOR scan down the rows of a word-aligned bit image (ntx words per row): after it, bit (r, c) is the
// parity of the toggles at rows <= r of column c.  NTH threads split every word column into RB row blocks (block totals go
// through `scratch`, RB * ntx words); each word is read twice and written once, whatever the number of runs.
template <int NTH>
__device__ inline void column_xor_scan(unsigned* bits, int H, int ntx, unsigned* scratch, int scratch_words, int tid) {
  int RB = NTH / ntx;
  if (RB > scratch_words / ntx) RB = scratch_words / ntx;
  if (RB < 1) RB = 1;
  const int RPB = (H + RB - 1) / RB;
  const int items = ntx * RB;
  if (RB > 1) {
    for (int it = tid; it < items; it += NTH) {
      const int blk = it / ntx, cw = it - blk * ntx;
      const int r1 = min(blk * RPB + RPB, H);
      unsigned x = 0;
      for (int r = blk * RPB; r < r1; ++r) x ^= bits[r * ntx + cw];
      scratch[it] = x;
    }
    __syncthreads();
  }
  for (int it = tid; it < items; it += NTH) {
    const int blk = it / ntx, cw = it - blk * ntx;
    const int r1 = min(blk * RPB + RPB, H);
    unsigned carry = 0;
    for (int b2 = 0; b2 < blk; ++b2) carry ^= scratch[b2 * ntx + cw];
    for (int r = blk * RPB; r < r1; ++r) {
      carry ^= bits[r * ntx + cw];
      bits[r * ntx + cw] = carry;
    }
  }
}

// scratch / scratch_words: LDS for the block totals of the column scan (word-aligned rows); with less than 2 * (W / 32) words
// the runs are painted directly instead (the slower route, also taken when W % 32 != 0).
template <int NTH>
__device__ inline int rle_to_bits(const int* __restrict__ counts, int nr, unsigned* bits, int nwords, int H, int W,
                                  unsigned* wtot, int tid, unsigned* scratch = nullptr, int scratch_words = 0, int frame_w = 0) {
  // (frame_w: la3d_fit_args::frame_width - the image is frame_w columns wide, its rows are stored W bits apart; the runs are
  // column-major, so the image ends at pixel H * frame_w of their order and nothing else changes)
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NTH / 64;
  const int HW = H * (frame_w > 0 && frame_w < W ? frame_w : W);
  const float rcpH = 1.0f / (float)H;
  for (int i = tid; i < nwords; i += NTH) bits[i] = 0;
  unsigned carry = 0;
  int nm = 0;
  const bool toggles = (W & 31) == 0 && scratch != nullptr && scratch_words >= 2 * (W >> 5);   // uniform
  for (int c0 = 0; c0 < nr; c0 += NTH) {
    const int j = c0 + tid;
    unsigned len = 0;
    if (j < nr) { const int v = counts[j]; len = v > 0 ? (unsigned)v : 0u; }
    unsigned incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();                       // previous step's readers of wtot are done; bits zeroing is ordered
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned base = carry, total = 0;
    for (int w = 0; w < NW; ++w) { if (w < wave) base += wtot[w]; total += wtot[w]; }
    const unsigned start = base + incl - len;
    unsigned L = 0;
    if ((j & 1) && len > 0 && start < (unsigned)HW) L = min(len, (unsigned)HW - start);
    nm += (int)L;
    carry += total;
    unsigned long long todo = __ballot(L != 0);
    if (toggles) {
      // Word-aligned rows, toggle form: a run is the set of pixels between two toggles of the column-major order.  Every lane
      // marks the start of its own run and the pixel after its end (two ds_xor per run, all runs of the wave at once); a run
      // that crosses into further columns also toggles row 0 of each of them.  column_xor_scan() below turns the toggles
      // into the filled image.
      const int ntx = W >> 5;
      unsigned long long wrap = 0;
      unsigned c_first = 0, c_last = 0;
      if (L != 0) {
        unsigned col0, row0, cole, rowe;
        pix_uv(start, H, rcpH, &row0, &col0);
        pix_uv(start + L, H, rcpH, &rowe, &cole);   // one past the run
        atomicXor(&bits[row0 * ntx + (col0 >> 5)], 1u << (col0 & 31));
        if (rowe != 0) atomicXor(&bits[rowe * ntx + (cole >> 5)], 1u << (cole & 31));
        c_first = col0 + 1;
        c_last = rowe != 0 ? cole : cole - 1;        // column of the run's last pixel
      }
      wrap = __ballot(L != 0 && c_last >= c_first);
      while (wrap) {                                 // wave-uniform, rare: runs longer than the rest of their column
        const int src = __ffsll((long long)wrap) - 1;
        wrap &= wrap - 1;
        const unsigned cf = (unsigned)__builtin_amdgcn_readlane((int)c_first, src);
        const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)c_last, src);
        for (unsigned cw = (cf >> 5) + lane; cw <= (cl >> 5); cw += 64) {   // row 0 of columns cf..cl, one word per lane
          const unsigned lo = cw == (cf >> 5) ? (cf & 31u) : 0u, hi = cw == (cl >> 5) ? (cl & 31u) : 31u;
          atomicXor(&bits[cw], (0xffffffffu >> (31u - hi)) & (0xffffffffu << lo));
        }
      }
      todo = 0;
    }
    while (todo) {                         // wave-uniform loop over the remaining ones-runs (they wrap columns, or W % 32 != 0)
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const unsigned S = __shfl(start, src), Lr = __shfl(L, src);
      unsigned col0, row0;
      pix_uv(S, H, rcpH, &row0, &col0);     // run start: position = col0 * H + row0 (same in every lane)
      if (row0 + Lr <= (unsigned)H) {       // the run stays inside one column: row0+q, col0
        const unsigned base = row0 * (unsigned)W + col0;
        for (unsigned q = lane; q < Lr; q += 64) {
          const unsigned idx = base + q * (unsigned)W;
          atomicOr(&bits[idx >> 5], 1u << (idx & 31));
        }
      } else {                              // wraps into following columns
        for (unsigned q = lane; q < Lr; q += 64) {
          unsigned col, row;
          pix_uv(S + q, H, rcpH, &row, &col);
          const unsigned idx = row * (unsigned)W + col;
          atomicOr(&bits[idx >> 5], 1u << (idx & 31));
        }
      }
    }
    if (carry >= (unsigned)HW) break;      // the frame is full: later runs fall outside it (uniform)
  }
  __syncthreads();
  if (toggles) {
    column_xor_scan<NTH>(bits, H, W >> 5, scratch, scratch_words, tid);
    __syncthreads();
  }
  return nm;
}

inline int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return LA3D_ERR_HIP;
  }
  return LA3D_SUCCESS;
}

// Dynamic LDS above the 64 KiB default has to be allowed per kernel AND per device (a process may drive several GPUs): one
// hipFuncSetAttribute per (kernel, device), remembered in a small table.
inline void allow_big_lds(const void* fn, int bytes = 160 * 1024) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  std::lock_guard<std::mutex> g(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dev) return;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) (void)hipGetLastError();
  done.emplace_back(fn, dev);
}

constexpr int CULL_MIN = 224;   // pass-B culling: active tiles from which an instance plans (run lengths / polygons; la3d_stages.hpp)
constexpr int MAX_MASK_LDS = 128 * 1024;  // bit image budget; larger frames re-read the u8 mask instead

}  // namespace la3d
