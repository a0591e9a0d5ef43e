// la3d.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI for the LabelAny3D geometric hot path:
// pinhole back-projection of masked depth pixels -> per-object moments -> closed-form PCA yaw
// -> extents along the principal axes -> 39-double box record.
//
// Reference semantics (behaviour only; nothing is copied):
//   depth_to_points   /root/reference/src/util.py:52-75
//   estimate_bbox     /root/reference/src/util_3dbox.py:106-178   (+ helpers :20-103, PCA yaw :181-186)
//
// Memory-bound integer/byte + fp64 reduction work: no MFMA.  Layout and kernel design are
// described in DESIGN.md; the short version for the fused kernel `fit_instances_kernel`:
//   one 512-thread workgroup (8 wave64) per instance;
//   phase 0  streams the u8 mask plane once with 16-byte non-temporal loads and packs it to a
//            1-bit-per-pixel image in LDS (38.4 KB for 640x480);
//   pass A   walks the bit image, 4 pixels per lane; only quads with a set bit load their
//            float4 of depth (coalesced 1 KB per wave), unproject in fp64 and accumulate
//            n, Sx, Sz, Sxx, Sxz, Szz, ymin, ymax per lane -> wave shuffle reduce -> LDS -> thread 0;
//   yaw      closed-form 2x2 principal axis with scikit-learn's sign rule (thread 0);
//   pass B   same walk (depth now L2/Infinity-Cache resident), min/max of the yaw-rotated x,z;
//   epilog   thread 0 writes center / dims / R_cam / fp16-quantised vertices.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "la3d.h"

namespace {

constexpr int NT = 512;          // threads per workgroup (fit_instances)
constexpr int NWAVE = NT / 64;   // wave64
constexpr int NTP = 256;         // threads per workgroup (fit_points)
constexpr int NWAVEP = NTP / 64;
constexpr double PI_2 = 1.57079632679489661923;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native vector: usable with nontemporal builtins

thread_local char g_err[256] = "";

void set_err(const char* fmt, const char* a = "") { snprintf(g_err, sizeof(g_err), fmt, a); }

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
// gesv: same elimination order; reference src/util.py:56).
__device__ inline void inv3(const double* A, double* X) {
  double a[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = A[i * 3 + j];
      a[i][3 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 6; ++j) {
        double t = a[c][j];
        a[c][j] = a[piv][j];
        a[piv][j] = t;
      }
    const double inv = 1.0 / a[c][c];
    for (int r = c + 1; r < 3; ++r) {
      const double f = a[r][c] * inv;
      for (int j = c; j < 6; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int j = 0; j < 3; ++j) {  // back substitution per right-hand side
    for (int r = 2; r >= 0; --r) {
      double s = a[r][3 + j];
      for (int k = r + 1; k < 3; ++k) s -= a[r][k] * X[k * 3 + j];
      X[r * 3 + j] = s / a[r][r];
    }
  }
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
__device__ inline void axis_from_sums(double n, double sx, double sz, double sxx, double sxz, double szz,
                                      double* cyaw, double* syaw, double* gap) {
  const double a = sxx - sx * sx / n;
  const double c = szz - sz * sz / n;
  const double b = sxz - sx * sz / n;
  const double half = 0.5 * (a - c);
  const double rad = sqrt(half * half + b * b);
  const double l1 = 0.5 * (a + c) + rad;
  *gap = (l1 > 0) ? 2.0 * rad / l1 : 0.0;
  if (b == 0 && a == c) {  // exact isotropy: eigh branch (n >= 20) -> yaw = pi/2; SVD branch recorded as 0
    if (n >= 20) { *cyaw = 6.123233995736766e-17; *syaw = 1.0; }  // np.cos(pi/2), np.sin(pi/2)
    else { *cyaw = 1.0; *syaw = 0.0; }
    return;
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

__device__ inline void write_nan_box(double* out) {
  for (int i = 0; i < LA3D_REC; ++i) out[i] = NAN;
}

// ------------------------------------------------------------------------------------------
// wave64 reductions (fixed butterfly order -> deterministic)
// ------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
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
  int ntx, nty;        // TILED: tiles of 32 px x 8 rows (ntx = W/32, nty = ceil(H/8))
  int tiles_per_wave;  // TILED: ceil(ntx*nty / NWAVE)
  int list_cap;        // TILED: entries of the active-tile list that fit the LDS budget
  double* out;
  int* status;
  double* aux;
};

// per-instance geometry written by prep_kernel into the workspace (20 doubles = 160 B)
constexpr int GEO_D = 20;  // M[9] (= Rg^T Kinv : p' = d * (M @ [u,v,1])), Rg[9], bad_ground, pad

struct alignas(16) Shared {
  double part[NWAVE][7];
  double cyaw, syaw;
  int cnt[NWAVE];
  int nmask[NWAVE];
  unsigned scan[NWAVE];
  int n_valid;
  int st;
};

__device__ inline void pix_uv(unsigned i, int W, float rcpW, unsigned* u, unsigned* v) {
  unsigned vv = (unsigned)((float)i * rcpW);
  int r = (int)i - (int)(vv * (unsigned)W);
  if (r < 0) { vv -= 1; r += W; }
  else if (r >= W) { vv += 1; r -= W; }
  *u = (unsigned)r;
  *v = vv;
}

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

// one pixel quad of a tile: PASS 0 accumulates count + moments, PASS 1 the six extents.
// r0/r1/r2: ray components at the quad's first pixel; a00/a10/a20: their per-pixel (u+1) increments.
template <int PASS>
__device__ inline void quad_math(unsigned nib, const unsigned* db, double r0, double r1, double r2, double a00,
                                 double a10, double a20, double* s, int* n) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // 0 / -1 validity word: mask bit k set AND exponent field != 0xff
    const int fin = ((int)(db[k] & 0x7fffffffu) - 0x7f800000) >> 31;
    const int m = fin & -(int)((nib >> k) & 1u);
    if (PASS == 0) {
      const double d = (double)__uint_as_float(db[k] & (unsigned)m);   // invalid -> +0.0
      const double x = d * r0, z = d * r2;
      s[0] += x; s[1] += z;
      s[2] = fma(x, x, s[2]); s[3] = fma(x, z, s[3]); s[4] = fma(z, z, s[4]);
      *n -= m;
    } else {
      const double d = (double)__uint_as_float(db[k] | ~(unsigned)m);  // invalid -> NaN, ignored by min/max
      const double x = d * r0, y = d * r1, z = d * r2;
      s[0] = dmin(s[0], x); s[1] = dmax(s[1], x);
      s[2] = dmin(s[2], y); s[3] = dmax(s[3], y);
      s[4] = dmin(s[4], z); s[5] = dmax(s[5], z);
      r1 += a10;
    }
    r0 += a00; r2 += a20;   // next pixel of the row: u + 1
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
constexpr int TG = 4;

template <int PASS>
__device__ inline void sweep_tiled(const FitParams& p, const float* __restrict__ dpl, const unsigned* bits,
                                   const unsigned short* list, int nactive, const double* A0, const double* A1,
                                   const double* A2, int wave, int lane, double* acc, int* cnt) {
  const int W = p.W, H = p.H, ntx = p.ntx;
  const int r = lane >> 3, cq = lane & 7;
  const double a00 = A0[0], a01 = A0[1], a02 = A0[2];
  const double a20 = A2[0], a21 = A2[1], a22 = A2[2];
  double a10 = 0, a11 = 0, a12 = 0;
  if (PASS == 1) { a10 = A1[0]; a11 = A1[1]; a12 = A1[2]; }
  double sv[6];
#pragma unroll
  for (int i = 0; i < (PASS == 0 ? 5 : 6); ++i) sv[i] = acc[i];
  int n = *cnt;
  const bool dense = nactive < 0;                  // list overflow: walk every tile, skip empty ones
  const int nsteps = dense ? ntx * p.nty : nactive;
  for (int j0 = wave * TG; j0 < nsteps; j0 += NWAVE * TG) {
    unsigned nib[TG];
    int txs[TG], tys[TG];
    uint4 dq[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) {   // stage 1: bit-image nibbles, then all depth loads back to back
      const int j = j0 + g;
      nib[g] = 0; txs[g] = 0; tys[g] = 0;
      dq[g] = make_uint4(0u, 0u, 0u, 0u);
      if (j < nsteps) {
        int tx, ty;
        if (dense) { ty = j / ntx; tx = j - ty * ntx; }
        else {
          const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)list[j]);  // wave-uniform -> SGPR
          tx = (int)(t & 0xffu); ty = (int)(t >> 8);
        }
        txs[g] = tx; tys[g] = ty;
        const int row = ty * 8 + r;
        if (row < H) nib[g] = (bits[row * ntx + tx] >> (cq * 4)) & 0xFu;
      }
    }
#pragma unroll
    for (int g = 0; g < TG; ++g)
      if (nib[g]) dq[g] = *reinterpret_cast<const uint4*>(dpl + (long long)(tys[g] * 8 + r) * W + txs[g] * 32 + cq * 4);
#pragma unroll
    for (int g = 0; g < TG; ++g) {   // stage 2: compute (all lanes; unmasked lanes carry zeros / NaNs)
      if (dense && __ballot(nib[g] != 0) == 0) continue;
      const unsigned db[4] = {dq[g].x, dq[g].y, dq[g].z, dq[g].w};
      const double vd = (double)(tys[g] * 8 + r), ud = (double)(txs[g] * 32 + cq * 4);
      const double r0 = fma(a00, ud, fma(a01, vd, a02));
      const double r2 = fma(a20, ud, fma(a21, vd, a22));
      double r1 = 0;
      if (PASS == 1) r1 = fma(a10, ud, fma(a11, vd, a12));
      quad_math<PASS>(nib[g], db, r0, r1, r2, a00, a10, a20, sv, &n);
    }
  }
#pragma unroll
  for (int i = 0; i < (PASS == 0 ? 5 : 6); ++i) acc[i] = sv[i];
  if (PASS == 0) *cnt = n;
}

// ------------------------------------------------------------------------------------------
// workgroup stages shared by the fit kernels (every thread of the workgroup must call them)
// ------------------------------------------------------------------------------------------
// moments of all waves -> thread 0 (fixed order: bit-reproducible) -> status, yaw axis, aux.
// On return sh->st / sh->cyaw / sh->syaw are valid for every thread.
__device__ inline void stage_moments_to_axis(Shared* sh, const FitParams& p, int inst, const double* geo,
                                             const double* acc, int cnt, int nmask, int tid, int wave, int lane,
                                             int nw = NWAVE) {
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
  if (tid == 0) {
    double s[5] = {0, 0, 0, 0, 0};
    int n = 0, nm = 0;
#pragma unroll 1
    for (int w = 0; w < nw; ++w) {  // fixed order (not unrolled: keeps thread 0's live set small)
      for (int k = 0; k < 5; ++k) s[k] += sh->part[w][k];
      n += sh->cnt[w];
      nm += sh->nmask[w];
    }
    int st = LA3D_BOX_OK;
    if (geo[18] != 0.0) st = LA3D_BOX_BAD_GROUND;
    else if (n == 0) st = LA3D_BOX_EMPTY;
    else if (n == 1) st = LA3D_BOX_TOO_FEW;
    double cy = NAN, sy = NAN, gap = NAN;
    if (st == LA3D_BOX_OK) axis_from_sums((double)n, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap);
    sh->cyaw = cy; sh->syaw = sy;
    sh->st = st;
    sh->n_valid = n;
    if (p.aux) {
      double* a = p.aux + (long long)inst * LA3D_AUX;
      a[0] = atan2(sy, cy); a[1] = (double)n; a[2] = (double)nm; a[3] = gap;
    }
    p.status[inst] = st;
    if (st != LA3D_BOX_OK) write_nan_box(p.out + (long long)inst * LA3D_REC);
  }
  __syncthreads();
}

// extents (x,y,z : lo,hi) of all waves -> thread 0 -> the 39-double record
__device__ inline void stage_extents_to_box(Shared* sh, const FitParams& p, int inst, const double* Rgg,
                                            const double* ext, int tid, int wave, int lane, int nw = NWAVE) {
  {
    const double r0 = wave_min(ext[0]), r1 = wave_max(ext[1]), r2 = wave_min(ext[2]), r3 = wave_max(ext[3]),
                 r4 = wave_min(ext[4]), r5 = wave_max(ext[5]);
    if (lane == 0) {
      double* pp = sh->part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; pp[5] = r5;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 1
    for (int w = 0; w < nw; ++w)
      for (int k = 0; k < 3; ++k) {
        lo[k] = fmin(lo[k], sh->part[w][2 * k]);
        hi[k] = fmax(hi[k], sh->part[w][2 * k + 1]);
      }
    write_box(p.out + (long long)inst * LA3D_REC, Rgg, sh->cyaw, sh->syaw, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
  }
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
// per-instance geometry: Kinv, Rg, M = Rg^T Kinv  (one thread per instance; keeps the 3x3
// elimination and Rodrigues algebra out of the streaming kernel's register budget)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void prep_kernel(const FitParams p) {
  const int inst = blockIdx.x * 64 + threadIdx.x;
  if (inst >= p.B) return;
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
// fused kernel: one workgroup per instance
// ------------------------------------------------------------------------------------------
template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED>
__global__ __launch_bounds__(NT, 8) void fit_instances_kernel(const FitParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  Shared* sh = reinterpret_cast<Shared*>(smem + p.mask_lds_bytes);
  unsigned* prefix = reinterpret_cast<unsigned*>(smem + p.mask_lds_bytes + sizeof(Shared));  // SAMPLE only
  // TILED only: compacted list of active tile ids
  unsigned short* list = reinterpret_cast<unsigned short*>(smem + p.mask_lds_bytes + sizeof(Shared));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inst = xcd_remap(blockIdx.x, p.B);
  const int img = p.image_index ? p.image_index[inst] : inst;
  const int HW = p.HW;
  const float* dpl = p.depth + (long long)img * p.depth_plane_stride;
  const unsigned char* mpl = p.mask + (long long)inst * HW;

  const double* geo = p.geo + (long long)inst * GEO_D;  // uniform address -> scalar loads
  const double* Rgg = geo + 9;
  // read before the first barrier: the compiler only keeps uniform loads on the scalar path (SGPRs)
  // while nothing in the kernel can have clobbered them
  double Mg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Mg[i] = geo[i];

  // ---- phase 0: u8 mask plane -> bit image in LDS --------------------------------------
  int nmask = 0;
  if (LDSMASK) {
    unsigned short* b16 = reinterpret_cast<unsigned short*>(bits);
    const int ngroups = (HW + 15) >> 4;
    if (VEC) {
      const u32x4* m4 = reinterpret_cast<const u32x4*>(mpl);
#pragma unroll 4
      for (int g = tid; g < ngroups; g += NT) {
        const u32x4 w = __builtin_nontemporal_load(m4 + g);
        const unsigned pat = nz4(w.x) | (nz4(w.y) << 4) | (nz4(w.z) << 8) | (nz4(w.w) << 12);
        b16[g] = (unsigned short)pat;
        nmask += __popc(pat);
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

  // ---- active-tile list (deterministic two-pass compaction: count, prefix, write) ----------------
  int nactive = 0;
  if (TILED) {
    const int ntiles = p.ntx * p.nty, per = p.tiles_per_wave;
    const int tbeg = wave * per, tend = min(tbeg + per, ntiles);
    int base = 0;
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
    __syncthreads();
  }

  // ---- pass A: moments ------------------------------------------------------------------
  double acc[5] = {0, 0, 0, 0, 0};
  int cnt = 0;
  // sampled-point state (SAMPLE only): the point of this thread in the ground-aligned frame
  double px = 0, py = 0, pz = 0;
  bool pok = false;
  bool sampled = false;

  if (SAMPLE) {
    // the reference subsamples when in_pc.shape[0] > 500 (src/util_3dbox.py:123): needs N first
    const int wsum = wave_sum_i(nmask);
    if (lane == 0) sh->nmask[wave] = wsum;
    __syncthreads();
    int ntot = 0;
    for (int w = 0; w < NWAVE; ++w) ntot += sh->nmask[w];
    sampled = ntot > LA3D_NSAMPLE;
    if (sampled) {
      // exclusive prefix of per-word popcounts: thread t owns words [t*per, t*per+per)
      const int per = (p.nwords + NT - 1) / NT;
      const int w0 = tid * per;
      unsigned local = 0;
      for (int i = 0; i < per; ++i)
        if (w0 + i < p.nwords) local += __popc(bits[w0 + i]);
      unsigned incl = local;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane == 63) sh->scan[wave] = incl;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; ++w) base += sh->scan[w];
      unsigned run = base + incl - local;
      for (int i = 0; i < per; ++i)
        if (w0 + i < p.nwords) { prefix[w0 + i] = run; run += __popc(bits[w0 + i]); }
      __syncthreads();
      if (tid < LA3D_NSAMPLE) {
        int r = p.sample_idx[(long long)inst * LA3D_NSAMPLE + tid];
        r = r < 0 ? 0 : (r >= ntot ? ntot - 1 : r);
        int lo = 0, hi = p.nwords - 1;
        while (lo < hi) {  // last word whose exclusive prefix is <= r
          const int mid = (lo + hi + 1) >> 1;
          if (prefix[mid] <= (unsigned)r) lo = mid; else hi = mid - 1;
        }
        unsigned w = bits[lo];
        for (int k = r - (int)prefix[lo]; k > 0; --k) w &= w - 1;  // drop k lowest set bits
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
#ifndef LA3D_ABL_NO_PASSA
  if (!sampled) {
    if (TILED) sweep_tiled<0>(p, dpl, bits, list, nactive, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt);
    else sweep<VEC, LDSMASK, 0>(p, dpl, mpl, bits, Mg, Mg + 3, Mg + 6, wave, lane, acc, &cnt, &nmask);
  }
#else
  if (tid == 0) { cnt = 2; acc[0] = 1; acc[1] = 2; acc[2] = 3; acc[3] = 1; acc[4] = 5; }
#endif

  stage_moments_to_axis(sh, p, inst, geo, acc, cnt, nmask, tid, wave, lane);
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
#ifndef LA3D_ABL_NO_PASSB
    double N0[3], N2[3];
    yaw_rows(sh, Mg, N0, N2);
    int d0 = 0, d1 = 0;
    if (TILED) sweep_tiled<1>(p, dpl, bits, list, nactive, N0, Mg + 3, N2, wave, lane, ext, &d0);
    else sweep<VEC, LDSMASK, 1>(p, dpl, mpl, bits, N0, Mg + 3, N2, wave, lane, ext, &d0, &d1);
#else
    ext[0] = 0; ext[1] = 1; ext[2] = 0; ext[3] = 1; ext[4] = 0; ext[5] = 1;
#endif
  }
  stage_extents_to_box(sh, p, inst, Rgg, ext, tid, wave, lane);
}

// ------------------------------------------------------------------------------------------
// point-cloud fit: one workgroup per cloud  (estimate_bbox on explicit (N,3) float64 input)
// ------------------------------------------------------------------------------------------
struct PtsParams {
  const double* points;
  const long long* offsets;
  const double* ground;
  const int* sample_idx;
  int B;
  int method;
  double* out;
  int* status;
  double* aux;
};

constexpr int HULL_MAX = 512;  // points the convex-hull method holds in LDS (the reference feeds it <= 500, :123)

struct alignas(16) SharedP {
  double part[NWAVEP][8];
  double Rg[9];
  double cyaw, syaw;
  int cnt[NWAVEP];
  int inf[NWAVEP];
  int bad_ground;
  int st;
  int nvalid;
  int hull_n;       // number of hull vertices found (0 = method not run)
  int fill;
  int pad;
};

// LDS of the convex-hull method (separate struct: only the hull instantiation pays for it)
struct alignas(16) SharedHull {
  double x[HULL_MAX], z[HULL_MAX];      // valid (x', z') footprint, sorted lexicographically
  double area[HULL_MAX];                // enclosing-rectangle area per hull edge
  double yaw[HULL_MAX];
  unsigned short hull[2 * HULL_MAX + 2];
};

// Minimum-area enclosing rectangle over hull-edge directions — reference src/util_3dbox.py:189-224
// (SciPy/Qhull there; here: bitonic sort in LDS, Andrew's monotone chain, one thread per hull edge).
// Reproduces the reference's conventions: yaw = atan2(edge_z, edge_x); points rotated by
// [[cos,-sin],[sin,cos]] (:204-208); area of the axis-aligned extent; the FIRST strict minimum wins
// (:216) in counter-clockwise vertex order.  Returns false when there is no 2-D hull (fewer than 3
// vertices: Qhull raises there and the reference falls back to PCA, :222-224).
__device__ inline bool hull_yaw(SharedHull* hs, SharedP* sh, int tid, double* yaw_out) {
  const int n = sh->nvalid;
  // pad to a power of two for the bitonic network
  for (int i = n + tid; i < HULL_MAX; i += NTP) { hs->x[i] = INFINITY; hs->z[i] = INFINITY; }
  __syncthreads();
  for (int k = 2; k <= HULL_MAX; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < HULL_MAX; i += NTP) {
        const int l = i ^ j;
        if (l > i) {
          const double xi = hs->x[i], zi = hs->z[i], xl = hs->x[l], zl = hs->z[l];
          const bool gt = (xi > xl) || (xi == xl && zi > zl);
          if (((i & k) == 0) ? gt : !gt) { hs->x[i] = xl; hs->z[i] = zl; hs->x[l] = xi; hs->z[l] = zi; }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {  // monotone chain: lower hull left->right, then upper hull right->left (counter-clockwise)
    unsigned short* H = hs->hull;
    int k = 0;
    auto cross = [&](int o, int a, int b) {
      return (hs->x[a] - hs->x[o]) * (hs->z[b] - hs->z[o]) - (hs->z[a] - hs->z[o]) * (hs->x[b] - hs->x[o]);
    };
    for (int i = 0; i < n; ++i) {
      while (k >= 2 && cross(H[k - 2], H[k - 1], i) <= 0) --k;
      H[k++] = (unsigned short)i;
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
      while (k >= t && cross(H[k - 2], H[k - 1], i) <= 0) --k;
      H[k++] = (unsigned short)i;
    }
    sh->hull_n = k - 1;  // last vertex repeats the first
  }
  __syncthreads();
  const int h = sh->hull_n;
  if (h < 3) return false;
  for (int e = tid; e < h; e += NTP) {
    const int i0 = hs->hull[e], i1 = hs->hull[(e + 1 == h) ? 0 : e + 1];
    const double yaw = atan2(hs->z[i1] - hs->z[i0], hs->x[i1] - hs->x[i0]);
    const double cs = cos(yaw), sn = sin(yaw);
    double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
    for (int j = 0; j < n; ++j) {
      const double px = hs->x[j], pz = hs->z[j];
      const double rx = cs * px - sn * pz, rz = sn * px + cs * pz;
      xlo = fmin(xlo, rx); xhi = fmax(xhi, rx); zlo = fmin(zlo, rz); zhi = fmax(zhi, rz);
    }
    hs->area[e] = (xhi - xlo) * (zhi - zlo);
    hs->yaw[e] = yaw;
  }
  __syncthreads();
  if (tid == 0) {
    double best = INFINITY, by = 0.0;
    for (int e = 0; e < h; ++e)
      if (hs->area[e] < best) { best = hs->area[e]; by = hs->yaw[e]; }
    hs->yaw[0] = by;
  }
  __syncthreads();
  *yaw_out = hs->yaw[0];
  return true;
}

template <bool HULL> struct HullStore {};
template <> struct HullStore<true> { SharedHull h; };

template <bool HULL>
__global__ __launch_bounds__(NTP) void fit_points_kernel(const PtsParams p) {
  __shared__ SharedP sh;
  __shared__ HullStore<HULL> hstore;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x;
  const long long off = p.offsets[c];
  const long long n_in = p.offsets[c + 1] - off;
  const bool sampled = p.sample_idx != nullptr && n_in > LA3D_NSAMPLE;  // reference :123
  const long long m = sampled ? LA3D_NSAMPLE : n_in;
  const int* sidx = sampled ? p.sample_idx + (long long)c * LA3D_NSAMPLE : nullptr;
  if (tid == 0) {
    sh.bad_ground = ground_rotation(p.ground ? p.ground + (long long)c * 4 : nullptr, sh.Rg);
    sh.fill = 0;
    sh.hull_n = 0;
  }
  __syncthreads();
  const double R00 = sh.Rg[0], R01 = sh.Rg[1], R02 = sh.Rg[2], R10 = sh.Rg[3], R11 = sh.Rg[4], R12 = sh.Rg[5],
               R20 = sh.Rg[6], R21 = sh.Rg[7], R22 = sh.Rg[8];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, ylo = INFINITY, yhi = -INFINITY;
  int n = 0, ninf = 0;
  for (long long i = tid; i < m; i += NTP) {
    long long row = i;
    if (sampled) {
      long long r = sidx[i];
      row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
    }
    const double* q = p.points + (off + row) * 3;
    const double a = q[0], b = q[1], cc = q[2];
    // rotated = in_pc @ Rg                                                (:136)
    const double x = a * R00 + b * R10 + cc * R20;
    const double y = a * R01 + b * R11 + cc * R21;
    const double z = a * R02 + b * R12 + cc * R22;
    const bool ok = !(x != x || y != y || z != z);                      // drop rows with any NaN (:139-140)
    if (ok) {
      if (isinf(x) || isinf(z)) ninf += 1;                              // scikit-learn rejects inf in X
      s0 += x; s1 += z; s2 = fma(x, x, s2); s3 = fma(x, z, s3); s4 = fma(z, z, s4);
      ylo = fmin(ylo, y); yhi = fmax(yhi, y);
      n += 1;
      if constexpr (HULL) {  // footprint for the hull method (order is irrelevant: it is sorted next)
        const int slot = atomicAdd(&sh.fill, 1);
        if (slot < HULL_MAX) { hstore.h.x[slot] = x; hstore.h.z[slot] = z; }
      }
    }
  }
  {
    const double r0 = wave_sum(s0), r1 = wave_sum(s1), r2 = wave_sum(s2), r3 = wave_sum(s3), r4 = wave_sum(s4),
                 r5 = wave_min(ylo), r6 = wave_max(yhi);
    const int rc = wave_sum_i(n), ri = wave_sum_i(ninf);
    if (lane == 0) {
      double* pp = sh.part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; pp[5] = r5; pp[6] = r6;
      sh.cnt[wave] = rc; sh.inf[wave] = ri;
    }
  }
  __syncthreads();
  double ymin = 0, ymax = 0;
  if (tid == 0) {
    double s[5] = {0, 0, 0, 0, 0};
    ymin = INFINITY; ymax = -INFINITY;
    int nn = 0, ni = 0;
    for (int w = 0; w < NWAVEP; ++w) {
      for (int k = 0; k < 5; ++k) s[k] += sh.part[w][k];
      ymin = fmin(ymin, sh.part[w][5]); ymax = fmax(ymax, sh.part[w][6]);
      nn += sh.cnt[w]; ni += sh.inf[w];
    }
    int st = LA3D_BOX_OK;
    if (sh.bad_ground) st = LA3D_BOX_BAD_GROUND;
    else if (nn == 0) st = LA3D_BOX_EMPTY;
    else if (ni > 0) st = LA3D_BOX_NONFINITE;
    else if (nn == 1) st = LA3D_BOX_TOO_FEW;
    double cy = NAN, sy = NAN, gap = NAN;
    if (st == LA3D_BOX_OK) axis_from_sums((double)nn, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap);
    if (HULL && st == LA3D_BOX_OK && nn > HULL_MAX) st = LA3D_BOX_UNSUPPORTED;
    sh.cyaw = cy; sh.syaw = sy; sh.st = st; sh.nvalid = nn;
    if (p.aux) {
      double* a = p.aux + (long long)c * LA3D_AUX;
      a[0] = atan2(sy, cy); a[1] = (double)nn; a[2] = (double)n_in; a[3] = gap;
    }
    p.status[c] = st;
    if (st != LA3D_BOX_OK) write_nan_box(p.out + (long long)c * LA3D_REC);
  }
  __syncthreads();
  if (sh.st != LA3D_BOX_OK) return;
  if constexpr (HULL) {
    double yaw;
    if (hull_yaw(&hstore.h, &sh, tid, &yaw)) {   // else: degenerate hull -> the PCA axis stands (reference :222-224)
      if (tid == 0) {
        double sy_, cy_;
        sincos(yaw, &sy_, &cy_);
        sh.cyaw = cy_; sh.syaw = sy_;
        if (p.aux) {
          double* a = p.aux + (long long)c * LA3D_AUX;
          a[0] = yaw;
          a[3] = -(double)sh.hull_n;  // negative: the hull decided the yaw (value = number of hull vertices)
        }
      }
      __syncthreads();
    }
  }
  const double cy = sh.cyaw, sy = sh.syaw;
  double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
  for (long long i = tid; i < m; i += NTP) {
    long long row = i;
    if (sampled) {
      long long r = sidx[i];
      row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
    }
    const double* q = p.points + (off + row) * 3;
    const double a = q[0], b = q[1], cc = q[2];
    const double x = a * R00 + b * R10 + cc * R20;
    const double y = a * R01 + b * R11 + cc * R21;
    const double z = a * R02 + b * R12 + cc * R22;
    if (!(x != x || y != y || z != z)) {
      const double x2 = cy * x + sy * z, z2 = -sy * x + cy * z;  // rotate_y(yaw) @ rotated^T  (:154)
      xlo = fmin(xlo, x2); xhi = fmax(xhi, x2); zlo = fmin(zlo, z2); zhi = fmax(zhi, z2);
    }
  }
  {
    const double r0 = wave_min(xlo), r1 = wave_max(xhi), r2 = wave_min(zlo), r3 = wave_max(zhi);
    if (lane == 0) {
      double* pp = sh.part[wave];
      pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
    for (int w = 0; w < NWAVEP; ++w) {
      xmin = fmin(xmin, sh.part[w][0]); xmax = fmax(xmax, sh.part[w][1]);
      zmin = fmin(zmin, sh.part[w][2]); zmax = fmax(zmax, sh.part[w][3]);
    }
    write_box(p.out + (long long)c * LA3D_REC, sh.Rg, cy, sy, xmin, xmax, ymin, ymax, zmin, zmax);
  }
}

// ------------------------------------------------------------------------------------------
// depth_to_points for a whole frame (write-bound: 4 B in, 24 B out per pixel)
// ------------------------------------------------------------------------------------------
struct UnprojParams {
  double Kinv[9];
  double R[9];
  double t[3];
  int has_rt;
  int H, W, HW;
  float rcpW;
};

template <typename OutT>
__global__ __launch_bounds__(256) void unproject_kernel(const float* __restrict__ depth, OutT* __restrict__ out,
                                                        const UnprojParams p) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.HW; i += stride) {
    unsigned u, v;
    pix_uv((unsigned)i, p.W, p.rcpW, &u, &v);
    const double d = (double)depth[i], ud = (double)u, vd = (double)v;
    // (D * Kinv) @ [u, v, 1]   — precedence as in the reference, src/util.py:71-72
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = (d * p.Kinv[r * 3]) * ud + (d * p.Kinv[r * 3 + 1]) * vd + (d * p.Kinv[r * 3 + 2]);
    if (p.has_rt) {  // R @ p + t  (:74)
      double w[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) w[r] = p.R[r * 3] * q[0] + p.R[r * 3 + 1] * q[1] + p.R[r * 3 + 2] * q[2] + p.t[r];
      q[0] = w[0]; q[1] = w[1]; q[2] = w[2];
    } else {
      // R = I, t = 0 in the reference still multiplies: 1*x + 0*y + 0*z + 0 — a NaN/inf component
      // poisons its neighbours exactly as there
      const double w0 = 1.0 * q[0] + 0.0 * q[1] + 0.0 * q[2] + 0.0;
      const double w1 = 0.0 * q[0] + 1.0 * q[1] + 0.0 * q[2] + 0.0;
      const double w2 = 0.0 * q[0] + 0.0 * q[1] + 1.0 * q[2] + 0.0;
      q[0] = w0; q[1] = w1; q[2] = w2;
    }
    OutT* o = out + (long long)i * 3;
    o[0] = (OutT)q[0]; o[1] = (OutT)q[1]; o[2] = (OutT)q[2];
  }
}

__global__ __launch_bounds__(256) void mask_counts_kernel(const unsigned char* __restrict__ mask, int HW, int vec,
                                                          int* __restrict__ counts) {
  __shared__ int part[4];
  const unsigned char* m = mask + (long long)blockIdx.x * HW;
  int n = 0;
  if (vec) {
    const uint4* m4 = reinterpret_cast<const uint4*>(m);
    for (int g = threadIdx.x; g < HW / 16; g += 256) {
      const uint4 w = m4[g];
      n += __popc(nz4(w.x)) + __popc(nz4(w.y)) + __popc(nz4(w.z)) + __popc(nz4(w.w));
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) n += m[i] ? 1 : 0;
  }
  n = wave_sum_i(n);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// host-side 3x3 inverse (same elimination as inv3 above)
void inv3_host(const double* A, double* X) {
  double a[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { a[i][j] = A[i * 3 + j]; a[i][3 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 6; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    const double inv = 1.0 / a[c][c];
    for (int r = c + 1; r < 3; ++r) { const double f = a[r][c] * inv; for (int j = c; j < 6; ++j) a[r][j] -= f * a[c][j]; }
  }
  for (int j = 0; j < 3; ++j)
    for (int r = 2; r >= 0; --r) {
      double s = a[r][3 + j];
      for (int k = r + 1; k < 3; ++k) s -= a[r][k] * X[k * 3 + j];
      X[r * 3 + j] = s / a[r][r];
    }
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return LA3D_ERR_HIP;
  }
  return LA3D_SUCCESS;
}

constexpr int MAX_MASK_LDS = 128 * 1024;  // bit image budget; larger frames re-read the u8 mask instead

template <bool VEC, bool LDSMASK, bool SAMPLE, bool TILED = false>
int launch_fit(const FitParams& p, size_t lds, hipStream_t s) {
  auto kern = fit_instances_kernel<VEC, LDSMASK, SAMPLE, TILED>;
  static bool attr_done = false;  // one flag per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.B), dim3(NT), lds, s, p);
  return check_launch("fit_instances_kernel");
}

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int la3d_version(void) { return LA3D_ABI_VERSION; }

const char* la3d_last_error(void) { return g_err; }

double la3d_f16_round_host(double x) { return f16_round(x); }

size_t la3d_workspace_bytes(int B, int H, int W) {
  (void)H; (void)W;
  return B > 0 ? (size_t)B * GEO_D * sizeof(double) : 0;  // per-instance geometry (prep_kernel)
}

int la3d_unproject(const float* depth, const double* K9, const double* Rt12, int H, int W, void* out,
                   int out_is_f64, void* stream) {
  if (!depth || !K9 || !out || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 4) {
    set_err("la3d_unproject: bad argument");
    return LA3D_ERR_ARG;
  }
  UnprojParams p;
  inv3_host(K9, p.Kinv);
  p.has_rt = Rt12 != nullptr;
  for (int i = 0; i < 9; ++i) p.R[i] = Rt12 ? Rt12[i] : ((i % 4 == 0) ? 1.0 : 0.0);
  for (int i = 0; i < 3; ++i) p.t[i] = Rt12 ? Rt12[9 + i] : 0.0;
  p.H = H; p.W = W; p.HW = H * W; p.rcpW = 1.0f / (float)W;
  const int blocks = (p.HW + 255) / 256 < 2048 ? (p.HW + 255) / 256 : 2048;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_is_f64) hipLaunchKernelGGL(unproject_kernel<double>, dim3(blocks), dim3(256), 0, s, depth, static_cast<double*>(out), p);
  else hipLaunchKernelGGL(unproject_kernel<float>, dim3(blocks), dim3(256), 0, s, depth, static_cast<float*>(out), p);
  return check_launch("unproject_kernel");
}

int la3d_mask_counts(const uint8_t* mask, int B, int H, int W, int32_t* counts, void* stream) {
  if (!mask || !counts || B < 0 || H <= 0 || W <= 0) {
    set_err("la3d_mask_counts: bad argument");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  const int HW = H * W;
  const int vec = (HW % 16 == 0) && ((reinterpret_cast<uintptr_t>(mask) & 15) == 0);
  hipLaunchKernelGGL(mask_counts_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), mask, HW, vec, counts);
  return check_launch("mask_counts_kernel");
}

int la3d_fit_instances(const float* depth, int64_t depth_plane_stride, const int32_t* image_index,
                       const uint8_t* mask, const double* K, int32_t k_stride, const double* ground,
                       const int32_t* sample_idx, int B, int H, int W, double* out, int32_t* status, double* aux,
                       void* workspace, void* stream) {
  if (!depth || !mask || !K || !out || !status || B < 0 || H <= 0 || W <= 0 || depth_plane_stride < 0 ||
      (k_stride != 0 && k_stride < 9) || (long long)H * W > (1LL << 28)) {
    set_err("la3d_fit_instances: bad argument");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
    set_err("la3d_fit_instances: workspace of la3d_workspace_bytes() bytes (8-aligned) required");
    return LA3D_ERR_ARG;
  }
  FitParams p;
  p.geo = static_cast<double*>(workspace);
  p.depth = depth; p.depth_plane_stride = depth_plane_stride; p.image_index = image_index;
  p.mask = mask; p.K = K; p.k_stride = k_stride; p.ground = ground; p.sample_idx = sample_idx;
  p.B = B; p.H = H; p.W = W; p.HW = H * W;
  p.nwords = (p.HW + 31) / 32;
  p.rows_aligned = (W % 4 == 0);
  p.rcpW = 1.0f / (float)W;
  p.out = out; p.status = status; p.aux = aux;
  p.ntx = p.nty = p.tiles_per_wave = p.list_cap = 0;
  const int bit_bytes = ((((p.HW + 15) / 16 + 1) / 2) * 4 + 15) & ~15;  // u16 per 16 px, padded to u32, 16-aligned
  const bool ldsmask = bit_bytes <= MAX_MASK_LDS;
  p.mask_lds_bytes = ldsmask ? bit_bytes : 0;
  // 16-byte vector path: every plane base 16-aligned
  const bool vec = (p.HW % 16 == 0) && ((reinterpret_cast<uintptr_t>(mask) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(depth) & 15) == 0) && (depth_plane_stride % 4 == 0);
  const bool sample = sample_idx != nullptr;
  size_t lds = (size_t)p.mask_lds_bytes + sizeof(Shared);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(prep_kernel, dim3((B + 63) / 64), dim3(64), 0, s, p);
  if (int rc = check_launch("prep_kernel")) return rc;
  if (sample) {
    if (!ldsmask) {
      set_err("la3d_fit_instances: reference-subsample mode needs the bit image in LDS (H*W <= 1048576)");
      return LA3D_ERR_UNSUPPORTED;
    }
    lds += (size_t)p.nwords * 4 + 16;
    if (lds > 160 * 1024 - 256) {
      set_err("la3d_fit_instances: reference-subsample mode: frame too large for LDS");
      return LA3D_ERR_UNSUPPORTED;
    }
    return vec ? launch_fit<true, true, true>(p, lds, s) : launch_fit<false, true, true>(p, lds, s);
  }
  // tiled fast path: 32-px-wide tiles map to exactly one bit-image word / one 128-B depth line per row
  p.ntx = W / 32; p.nty = (H + 7) / 8;
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
      return launch_fit<true, true, false, true>(p, fixed + (size_t)cap * 2, s);
    }
  }
  if (ldsmask) return vec ? launch_fit<true, true, false>(p, lds, s) : launch_fit<false, true, false>(p, lds, s);
  return vec ? launch_fit<true, false, false>(p, lds, s) : launch_fit<false, false, false>(p, lds, s);
}

int la3d_fit_points(const double* points, const int64_t* offsets, const double* ground, const int32_t* sample_idx,
                    int method, int B, double* out, int32_t* status, double* aux, void* stream) {
  if (!offsets || !out || !status || B < 0 || (B > 0 && !points && false)) {
    set_err("la3d_fit_points: bad argument");
    return LA3D_ERR_ARG;
  }
  if (method != LA3D_METHOD_PCA && method != LA3D_METHOD_CONVEX_HULL) {
    set_err("la3d_fit_points: unknown method");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  PtsParams p;
  p.points = points; p.offsets = reinterpret_cast<const long long*>(offsets); p.ground = ground;
  p.sample_idx = sample_idx; p.B = B; p.method = method; p.out = out; p.status = status; p.aux = aux;
  if (method == LA3D_METHOD_CONVEX_HULL)
    hipLaunchKernelGGL(fit_points_kernel<true>, dim3(B), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(fit_points_kernel<false>, dim3(B), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("fit_points_kernel");
}

}  // extern "C"
