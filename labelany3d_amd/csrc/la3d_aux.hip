// la3d_aux.hip - every kernel of the C-ABI that is not a fit engine, and its entry points: explicit point clouds (estimate_bbox as the
// reference calls it: 500 mesh samples per object, src/util_3dbox.py:273-278; the host-pointer single call), whole-frame
// depth_to_points (src/util.py:52-75), mask decode / statistics for run lengths and polygon parts (src/util.py:291-415), the box
// consumers (src/tools/combine_results.py:105-124, :238-252), the masked depth-ratio median and the depth-alignment selection
// (src/util.py:464-494, src/batch_scripts/depth.py:52-92), the matcher's unprojection (src/matching/matcher.py:70-91).
// The fit engines (la3d_fit_instances*) live in la3d.hip / la3d_split.hip; shared device code in la3d_device.hpp / la3d_poly.hpp.
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

using namespace la3d;

namespace {

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

// One pass of Andrew's monotone chain: visits cnt entries of the candidate list cl starting at position q0 in direction dq, pushes
// point indices on the stack S (k0 entries on entry; a pop needs at least t), returns the stack size.  The coordinates of the two
// stack tops are carried in registers, so a step that pops nothing waits for no dependent LDS read.  The turn test is the textbook
// cross(o, a, b) = (xa - xo)(zb - zo) - (za - zo)(xb - xo) <= 0 -> pop.
__device__ inline int chain_pass(const SharedHull* hs, const unsigned short* cl, int q0, int dq, int cnt, unsigned short* S, int k0, int t) {
  int k = k0;
  double ox = 0, oz = 0, ax = 0, az = 0;
  if (k >= 1) { const int a = S[k - 1]; ax = hs->x[a]; az = hs->z[a]; }
  if (k >= 2) { const int o = S[k - 2]; ox = hs->x[o]; oz = hs->z[o]; }
  for (int c = 0, q = q0; c < cnt; ++c, q += dq) {
    const int i = cl[q];
    const double px = hs->x[i], pz = hs->z[i];
    while (k >= t) {
      const double cr = (ax - ox) * (pz - oz) - (az - oz) * (px - ox);
      if (!(cr <= 0)) break;
      --k;
      ax = ox; az = oz;
      if (k >= 2) { const int o = S[k - 2]; ox = hs->x[o]; oz = hs->z[o]; }
    }
    S[k++] = (unsigned short)i;
    ox = ax; oz = az; ax = px; az = pz;
  }
  return k;
}

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
  // Andrew's monotone chain is serial (every step depends on the stack the previous one left) and each of its cross products is a
  // chain of dependent LDS reads - one lane needed ~150 us for 500 points.  Round 3: (1) sixteen lanes each run the chain over a
  // sixteenth of the sorted points and mark what survives in their chunk (a point inside its chunk's hull cannot be a vertex of the
  // whole hull; collinear points drop out either way), the survivors are compacted in sorted order; (2) four lanes do the same
  // over quarters of the survivors; (3) one lane runs the SAME chain over what is left.  With exact orientation predicates the
  // vertex sequence - hence every edge, area and the winning yaw - is the one the chain over all points gives; the fp64 cross
  // products are rounded, so in NEARLY collinear configurations (or with duplicate points straddling a chunk boundary) a point
  // may be kept by one form and dropped by the other: the hulls then differ by a vertex that moves no edge beyond rounding, and the
  // minimum-area yaw can only move between edges whose areas tie to rounding (the documented don't-care; profiles/r03/stress_hull.py
  // holds both forms to the oracle with a yaw / area tolerance).  The two stack tops live in registers (chain_pass).
  unsigned short* cl = reinterpret_cast<unsigned short*>(hs->yaw);   // current candidates in sorted order (yaw[] is written after the chain)
  for (int i = tid; i < n; i += NTP) cl[i] = (unsigned short)i;
  int m = n;
  for (int level = 0; level < 2; ++level) {
    const int nch = level == 0 ? 16 : 4;
    if (m <= 4 * nch) continue;                                      // uniform
    for (int i = tid; i < n; i += NTP) hs->area[i] = 0.0;            // survivor flags by point (area[] is written after the chain)
    __syncthreads();
    if (tid < nch) {
      const int lo = (int)((long long)m * tid / nch), hi = (int)((long long)m * (tid + 1) / nch);
      unsigned short* S = hs->hull + lo;                             // this lane's stack: as many slots as its chunk has entries
      for (int pass = 0; pass < 2; ++pass) {                         // lower hull left -> right, then upper hull right -> left
        const int k = chain_pass(hs, cl, pass == 0 ? lo : hi - 1, pass == 0 ? 1 : -1, hi - lo, S, 0, 2);
        for (int q = 0; q < k; ++q) hs->area[S[q]] = 1.0;
      }
    }
    __syncthreads();
    if (tid < 64) {                                                  // in-place compaction of the survivors, ascending (one wave:
      int base = 0;                                                  // a block's reads precede its writes, and it writes behind itself)
      for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + tid;
        const unsigned short id = i < m ? cl[i] : (unsigned short)0;
        const bool on = i < m && hs->area[id] != 0.0;
        const unsigned long long bal = __ballot(on);
        if (on) cl[base + __popcll(bal & ((1ull << tid) - 1ull))] = id;
        base += __popcll(bal);
      }
      if (tid == 0) sh->hull_n = base;                               // (number of survivors, until the chain below replaces it)
    }
    __syncthreads();
    m = sh->hull_n;
    __syncthreads();
  }
  if (tid == 0) {  // monotone chain over the survivors: lower hull left->right, then upper hull right->left (counter-clockwise)
    unsigned short* H = hs->hull;
    int k = chain_pass(hs, cl, 0, 1, m, H, 0, 2);
    k = chain_pass(hs, cl, m - 2, -1, m - 1, H, k, k + 1);
    sh->hull_n = k - 1;  // last vertex repeats the first
  }
  __syncthreads();
  const int h = sh->hull_n;
  if (h < 3) return false;
  // one hull edge per wave at a time, lanes over the points (min / max are order independent: the areas are those of a serial sweep)
  const int lane = tid & 63, wave = tid >> 6;
  for (int e = wave; e < h; e += NTP / 64) {
    const int i0 = hs->hull[e], i1 = hs->hull[(e + 1 == h) ? 0 : e + 1];
    const double yaw = atan2(hs->z[i1] - hs->z[i0], hs->x[i1] - hs->x[i0]);
    const double cs = cos(yaw), sn = sin(yaw);
    double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
    for (int j = lane; j < n; j += 64) {
      const double px = hs->x[j], pz = hs->z[j];
      const double rx = cs * px - sn * pz, rz = sn * px + cs * pz;
      xlo = fmin(xlo, rx); xhi = fmax(xhi, rx); zlo = fmin(zlo, rz); zhi = fmax(zhi, rz);
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); zlo = wave_min(zlo); zhi = wave_max(zhi);
    if (lane == 0) {
      // (area[] / yaw[] slots below h: the survivor flags and the survivor list are dead by now - the barrier above)
      hs->area[e] = (xhi - xlo) * (zhi - zlo);
      hs->yaw[e] = yaw;
    }
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
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

// PCA method, small clouds (LA3D_HINT_SMALL_CLOUDS): one wave per cloud, four clouds per workgroup.  Everything a cloud needs
// lives in its wave: the ground rotation and the axis are computed redundantly by all lanes (their inputs are wave-uniform), the
// reductions are DPP wave reductions, the box is written lane-parallel - no LDS, no barrier.  The second walk re-reads the points
// (12 KB per 500-point cloud: cache hits).  Same arithmetic per point as fit_points_kernel; the sums associate differently.
// one cloud by one wave (all 64 lanes): `pts` = the cloud's rows (global memory, or LDS for la3d_estimate_bbox_host - after inlining
// the address space is static)
__device__ __forceinline__ void fit_cloud_wave(const double* pts, long long n_in, const int* sidx, const double* ground, double* out,
                                      int* status, double* aux, int lane) {
  const bool sampled = sidx != nullptr;
  const long long m = sampled ? LA3D_NSAMPLE : n_in;
  double Rg[9];
  const int bad_ground = ground_rotation(ground, Rg);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, ylo = INFINITY, yhi = -INFINITY;
  int n = 0, ninf = 0;
  for (long long i = lane; i < m; i += 64) {
    long long row = i;
    if (sampled) {
      const long long r = sidx[i];
      row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
    }
    const double* q = pts + row * 3;
    const double a = q[0], b = q[1], cc = q[2];
    const double x = a * Rg[0] + b * Rg[3] + cc * Rg[6];                 // rotated = in_pc @ Rg   (:136)
    const double y = a * Rg[1] + b * Rg[4] + cc * Rg[7];
    const double z = a * Rg[2] + b * Rg[5] + cc * Rg[8];
    if (!(x != x || y != y || z != z)) {                                  // drop rows with any NaN (:139-140)
      if (isinf(x) || isinf(z)) ninf += 1;
      s0 += x; s1 += z; s2 = fma(x, x, s2); s3 = fma(x, z, s3); s4 = fma(z, z, s4);
      ylo = fmin(ylo, y); yhi = fmax(yhi, y);
      n += 1;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
  const double ymin = wave_min(ylo), ymax = wave_max(yhi);
  const int nn = wave_sum_i(n), ni = wave_sum_i(ninf);
  int st = LA3D_BOX_OK;
  if (bad_ground) st = LA3D_BOX_BAD_GROUND;
  else if (nn == 0) st = LA3D_BOX_EMPTY;
  else if (ni > 0) st = LA3D_BOX_NONFINITE;
  else if (nn == 1) st = LA3D_BOX_TOO_FEW;
  double cy = NAN, sy = NAN, gap = NAN;
  if (st == LA3D_BOX_OK) axis_from_sums((double)nn, s0, s1, s2, s3, s4, &cy, &sy, &gap);
  if (lane == 0) {
    if (aux) { aux[0] = atan2(sy, cy); aux[1] = (double)nn; aux[2] = (double)n_in; aux[3] = gap; }
    *status = st;
    if (st != LA3D_BOX_OK) write_nan_box(out);
  }
  if (st != LA3D_BOX_OK) return;   // wave-uniform
  double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY;
  for (long long i = lane; i < m; i += 64) {
    long long row = i;
    if (sampled) {
      const long long r = sidx[i];
      row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
    }
    const double* q = pts + row * 3;
    const double a = q[0], b = q[1], cc = q[2];
    const double x = a * Rg[0] + b * Rg[3] + cc * Rg[6];
    const double y = a * Rg[1] + b * Rg[4] + cc * Rg[7];
    const double z = a * Rg[2] + b * Rg[5] + cc * Rg[8];
    if (!(x != x || y != y || z != z)) {
      const double x2 = cy * x + sy * z, z2 = -sy * x + cy * z;          // rotate_y(yaw) @ rotated^T  (:154)
      xlo = fmin(xlo, x2); xhi = fmax(xhi, x2); zlo = fmin(zlo, z2); zhi = fmax(zhi, z2);
    }
  }
  const double xmin = wave_min(xlo), xmax = wave_max(xhi), zmin = wave_min(zlo), zmax = wave_max(zhi);
  write_box_wave(out, Rg, cy, sy, xmin, xmax, ymin, ymax, zmin, zmax, lane);
}

__global__ __launch_bounds__(NTP) void fit_points_wave_kernel(const PtsParams p) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (NTP / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (c >= p.B) return;   // wave-uniform
  const long long off = p.offsets[c];
  const long long n_in = p.offsets[c + 1] - off;
  const bool sampled = p.sample_idx != nullptr && n_in > LA3D_NSAMPLE;  // reference :123
  fit_cloud_wave(p.points + off * 3, n_in, sampled ? p.sample_idx + (long long)c * LA3D_NSAMPLE : nullptr,
                 p.ground ? p.ground + (long long)c * 4 : nullptr, p.out + (long long)c * LA3D_REC, p.status + c,
                 p.aux ? p.aux + (long long)c * LA3D_AUX : nullptr, lane);
}

// ------------------------------------------------------------------------------------------
// la3d_estimate_bbox_host (round 5): ONE cloud that lives in HOST memory - the reference's own calling pattern, estimate_bbox once
// per object on a NumPy array (src/util_3dbox.py:273-278).  The block is the library's pinned, device-mapped staging buffer:
// [0] offsets (unused) | [32] ground 4 f64 | [64] record 39 f64 | [376] aux 4 f64 | [408] status i32 | [416] done u32 |
// [512] points n x 3 f64.  The kernel pulls the cloud over the host link into LDS with 16-byte loads (one round trip for a
// 500-point cloud), fits it there - PCA: the arithmetic of fit_points_wave_kernel, bit for bit - writes the record straight back
// into the block and stores the call's sequence number into `done` with a system-scope release: the host polls that word.
// ------------------------------------------------------------------------------------------
constexpr int HOSTFIT_MAXN = 1024;          // rows staged through LDS (24 KiB); larger clouds are read in place
constexpr size_t HOSTFIT_HDR = 512;
__global__ __launch_bounds__(NTP) void fit_points_host_kernel(unsigned char* blk, long long n, int has_ground, unsigned seq) {
  __shared__ __attribute__((aligned(16))) double stage[HOSTFIT_MAXN * 3];
  const int tid = threadIdx.x, lane = tid & 63;
  const double* pts = reinterpret_cast<const double*>(blk + HOSTFIT_HDR);
  const bool staged = n <= HOSTFIT_MAXN;
  if (staged) {
    const int n16 = (int)((n * 24 + 15) / 16);   // (the staging buffer is padded: reading the last partial 16 bytes is safe)
    const u32x4* src = reinterpret_cast<const u32x4*>(pts);
    u32x4* dst = reinterpret_cast<u32x4*>(stage);
    for (int i = tid; i < n16; i += NTP) dst[i] = src[i];
  }
  __syncthreads();
  if (tid < 64) {
    const double* ground = has_ground ? reinterpret_cast<const double*>(blk + 32) : nullptr;
    double* out = reinterpret_cast<double*>(blk + 64);
    double* aux = reinterpret_cast<double*>(blk + 376);
    int* status = reinterpret_cast<int*>(blk + 408);
    if (staged) fit_cloud_wave(stage, n, nullptr, ground, out, status, aux, lane);
    else fit_cloud_wave(pts, n, nullptr, ground, out, status, aux, lane);
    // every lane's stores are complete and visible to the host before the flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(blk + 416), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

// One wave turns 64 consecutive pixels into 64 points per step.  The points go through a per-wave LDS stage so that the wave
// writes its 1536 (f64) / 768 (f32) contiguous output bytes as whole 16-byte non-temporal stores (the output is written once and
// read by somebody else: measured 64 / 256 / 1024 frames of 640x480 -> f64: 148 / 541 / 1921 us with plain per-lane stores,
// 89 / 477 / 1656 us this way = 6.1 / 4.6 / 5.3 TB/s; a device copy of the same size moves 5.3 / 4.4 / 4.7 TB/s, a pure fill
// 6.4 / 6.8 / 6.8 TB/s: profiles/r03/r03_unproject.txt).  vec16: every frame's output base is
// 16-byte aligned.  kinv: the frame's inverse intrinsics in LDS.
template <typename OutT>
__device__ inline void unproject_frame(const float* __restrict__ dp, OutT* __restrict__ op, const double* kinv, OutT* sl,
                                       const UnprojParams& p, int first, int stride, int lane, bool vec16) {
  constexpr int N16 = 64 * 3 * (int)sizeof(OutT) / 16;   // 16-byte pieces per 64 points
  for (int i0 = first; i0 < p.HW; i0 += stride) {   // wave-uniform trip count
    const int i = i0 + lane;
    double w[3] = {0, 0, 0};
    if (i < p.HW) {
      unsigned u, v;
      pix_uv((unsigned)i, p.W, p.rcpW, &u, &v);
      // (plain load: a depth plane that a previous kernel left in the cache should be found there)
      const double d = (double)dp[i], ud = (double)u, vd = (double)v;
      // (D * Kinv) @ [u, v, 1]   - precedence as in the reference, src/util.py:71-72
      double q[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) q[r] = (d * kinv[r * 3]) * ud + (d * kinv[r * 3 + 1]) * vd + (d * kinv[r * 3 + 2]);
      if (p.has_rt) {   // R @ p + t  (:74)
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = p.R[r * 3] * q[0] + p.R[r * 3 + 1] * q[1] + p.R[r * 3 + 2] * q[2] + p.t[r];
      } else {
        // R = I, t = 0 in the reference still multiplies: 1*x + 0*y + 0*z + 0 - a NaN / inf component poisons its
        // neighbours exactly as there
        w[0] = 1.0 * q[0] + 0.0 * q[1] + 0.0 * q[2] + 0.0;
        w[1] = 0.0 * q[0] + 1.0 * q[1] + 0.0 * q[2] + 0.0;
        w[2] = 0.0 * q[0] + 0.0 * q[1] + 1.0 * q[2] + 0.0;
      }
    }
    sl[lane * 3] = (OutT)w[0]; sl[lane * 3 + 1] = (OutT)w[1]; sl[lane * 3 + 2] = (OutT)w[2];
    // lanes exchange through LDS: the hardware completes a wave's LDS operations in order, but the compiler must be told that the
    // reads below depend on OTHER lanes' writes (it can prove that 3 lane + 1 never equals 64 + lane and would hoist that read)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long base = (long long)i0 * 3, lim = (long long)p.HW * 3;
    if (vec16 && i0 + 64 <= p.HW) {   // uniform
      const u32x4* s16 = reinterpret_cast<const u32x4*>(sl);
      u32x4* o16 = reinterpret_cast<u32x4*>(op + base);
#pragma unroll
      for (int k = 0; k < (N16 + 63) / 64; ++k)
        if (k * 64 + lane < N16) __builtin_nontemporal_store(s16[k * 64 + lane], o16 + k * 64 + lane);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (base + k * 64 + lane < lim) __builtin_nontemporal_store(sl[k * 64 + lane], op + base + k * 64 + lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void unproject_kernel(const float* __restrict__ depth, OutT* __restrict__ out,
                                                        const UnprojParams p, int vec16) {
  __shared__ double kinv[9];
  __shared__ __attribute__((aligned(16))) OutT stage[4][192];
  if (threadIdx.x < 9) kinv[threadIdx.x] = p.Kinv[threadIdx.x];   // (inverted on the host: la3d_unproject)
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unproject_frame<OutT>(depth, out, kinv, stage[wave], p, blockIdx.x * blockDim.x + wave * 64, gridDim.x * blockDim.x, lane, vec16 != 0);
}

// P frames in one launch: blockIdx.y = frame; the frame's K is inverted by one thread (device inv3 = the host routine's elimination)
template <typename OutT>
__global__ __launch_bounds__(256) void unproject_batch_kernel(const float* __restrict__ depth, const double* __restrict__ K,
                                                              int k_stride, OutT* __restrict__ out, const UnprojParams p, int vec16) {
  __shared__ double kinv[9];
  __shared__ __attribute__((aligned(16))) OutT stage[4][192];
  if (threadIdx.x == 0) inv3(K + (long long)blockIdx.y * k_stride, kinv);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unproject_frame<OutT>(depth + (long long)blockIdx.y * p.HW, out + (long long)blockIdx.y * p.HW * 3, kinv, stage[wave], p,
                        blockIdx.x * blockDim.x + wave * 64, gridDim.x * blockDim.x, lane, vec16 != 0);
}

// Depth rows padded on the right with zeros: [rows][W] f32 -> [rows][Wp] f32, Wp % 4 == 0 (la3d_fit_args::frame_width: frames whose
// width is not a multiple of 32).  One 16-byte store per thread and step; the loads are 4-byte (a row of odd width starts anywhere),
// consecutive lanes read consecutive floats.
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, long long rows, int W, int Wp, float* __restrict__ dst) {
  const int qpr = Wp >> 2;                                   // 16-byte groups per padded row
  const long long total = rows * qpr;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const long long r = g / qpr;
    const int c = (int)(g - r * qpr) * 4;
    const float* s = src + r * W + c;
    u32x4 v;   // (bit patterns: the store is a plain 16-byte move)
    v.x = c < W ? __float_as_uint(s[0]) : 0u; v.y = c + 1 < W ? __float_as_uint(s[1]) : 0u;
    v.z = c + 2 < W ? __float_as_uint(s[2]) : 0u; v.w = c + 3 < W ? __float_as_uint(s[3]) : 0u;
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst) + g);
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

constexpr int NT_DEC = 512;   // decode kernels: 8 waves per workgroup (four workgroups per CU by LDS: 32 waves keep the stores coming)

// bit image in LDS -> u8 plane (0/1), coalesced 16-byte non-temporal stores where the plane allows; NTH threads.  Four bits
// become four bytes with one multiply: bit i of the nibble lands at 8 i through the partial product shifted by 7 i (the 16 partial
// products hit 16 different bit positions: no carries).
template <int NTH>
__device__ inline void bits_to_plane(const unsigned* bits, int HW, unsigned char* o, int tid) {
  const unsigned short* b16 = reinterpret_cast<const unsigned short*>(bits);
  if (HW % 16 == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
#pragma unroll 4
    for (int g = tid; g < HW / 16; g += NTH) {
      const unsigned pat = b16[g];
      u32x4 v;
      v.x = ((pat & 0xFu) * 0x00204081u) & 0x01010101u;
      v.y = (((pat >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
      v.z = (((pat >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
      v.w = (((pat >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o + g * 16));
    }
  } else {
    for (int i = tid; i < HW; i += NTH) o[i] = (bits[i >> 5] >> (i & 31)) & 1u;
  }
}

// mask_utils.decode for a batch (reference src/util.py:367,401-402): run lengths -> u8 planes.  The runs are
// decoded into an LDS bit image (rle_to_bits) and expanded with coalesced 16-byte stores.
__global__ __launch_bounds__(NT_DEC) void rle_decode_kernel(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                            int H, int W, int nwords, int scan_words, unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  unsigned* wtot = bits + nwords;
  const int tid = threadIdx.x;
  const long long o0 = offsets[blockIdx.x];
  (void)rle_to_bits<NT_DEC>(counts + o0, (int)(offsets[blockIdx.x + 1] - o0), bits, nwords, H, W, wtot, tid, wtot + 16, scan_words);
  bits_to_plane<NT_DEC>(bits, H * W, out + (long long)blockIdx.x * H * W, tid);
}

// create_boolean_mask_from_polygon for a batch (reference src/util.py:386-400): polygon parts -> u8 planes.  Dynamic LDS:
// bit image (16-aligned), side stage, flags.
__global__ __launch_bounds__(NT_DEC) void poly_decode_kernel(const int* __restrict__ xy, const long long* __restrict__ ring_off,
                                                          const long long* __restrict__ inst_rings, int H, int W, int nwords,
                                                          unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  const size_t bit_bytes = ((size_t)nwords * 4 + 15) & ~(size_t)15;
  PolySide* stage = reinterpret_cast<PolySide*>(smem + bit_bytes);
  unsigned* flags = reinterpret_cast<unsigned*>(smem + bit_bytes + POLY_STAGE_BYTES);
  const int tid = threadIdx.x;
  (void)poly_to_bits<NT_DEC>(xy, ring_off, inst_rings[blockIdx.x], inst_rings[blockIdx.x + 1], stage, flags, bits, nwords, H, W, tid);
  bits_to_plane<NT_DEC>(bits, H * W, out + (long long)blockIdx.x * H * W, tid);
}

// The reference's filter quantities (mask_stats) for polygon annotations without materialising a plane: rasterise into
// LDS, count there.  Dynamic LDS: bit image, side stage, flags (64 B), per-row counts (H ints), 20 ints.
__global__ __launch_bounds__(256) void mask_stats_poly_kernel(const int* __restrict__ xy, const long long* __restrict__ ring_off,
                                                              const long long* __restrict__ inst_rings, int H, int W, int nwords,
                                                              int boundary, int* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  const size_t bit_bytes = ((size_t)nwords * 4 + 15) & ~(size_t)15;
  PolySide* stage = reinterpret_cast<PolySide*>(smem + bit_bytes);
  unsigned* flags = reinterpret_cast<unsigned*>(smem + bit_bytes + POLY_STAGE_BYTES);
  int* rowcnt = reinterpret_cast<int*>(smem + bit_bytes + POLY_STAGE_BYTES + 64);
  int* red = rowcnt + H;
  const int tid = threadIdx.x;
  (void)poly_to_bits<256>(xy, ring_off, inst_rings[blockIdx.x], inst_rings[blockIdx.x + 1], stage, flags, bits, nwords, H, W, tid);
  int o4[4];
  bits_stats_256(bits, H, W, boundary, rowcnt, red, tid, o4);
  if (tid == 0) {
    int* o = stats + (long long)blockIdx.x * 4;
    o[0] = o4[0]; o[1] = o4[1]; o[2] = o4[2]; o[3] = o4[3];
  }
}

// The quantities of the reference's instance filter (src/util.py:291-335, :367-376) per mask plane:
// stats[0] = area, [1] = rows holding a pixel, [2] = last row - first row + 1, [3] = pixels inside the four
// boundary strips of `boundary` px (corners counted twice, as analyze_mask does).
__global__ __launch_bounds__(256) void mask_stats_kernel(const unsigned char* __restrict__ mask, int H, int W, int boundary,
                                                         int* __restrict__ stats) {
  __shared__ int red[4][4];
  const unsigned char* m = mask + (long long)blockIdx.x * H * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int area = 0, rows = 0, first = H, last = -1, trunc = 0;
  for (int r = wave; r < H; r += 4) {  // one wave per row
    int cnt = 0, edge = 0;
    for (int c = lane; c < W; c += 64) {
      const int on = m[(long long)r * W + c] ? 1 : 0;
      cnt += on;
      if (on) edge += (c < boundary ? 1 : 0) + (c >= W - boundary ? 1 : 0);
    }
    cnt = wave_sum_i(cnt);
    edge = wave_sum_i(edge);
    area += cnt;
    trunc += edge;
    if (r < boundary || r >= H - boundary) trunc += (r < boundary && r >= H - boundary) ? 2 * cnt : cnt;
    if (cnt) { rows += 1; first = min(first, r); last = max(last, r); }
  }
  if (lane == 0) { red[wave][0] = area; red[wave][1] = rows; red[wave][2] = first; red[wave][3] = last; }
  __shared__ int tr[4];
  if (lane == 0) tr[wave] = trunc;
  __syncthreads();
  if (tid == 0) {
    int a = 0, rw = 0, f = H, l = -1, t = 0;
    for (int w = 0; w < 4; ++w) { a += red[w][0]; rw += red[w][1]; f = min(f, red[w][2]); l = max(l, red[w][3]); t += tr[w]; }
    int* o = stats + (long long)blockIdx.x * 4;
    o[0] = a; o[1] = rw; o[2] = (l >= f) ? l - f + 1 : 0; o[3] = t;
  }
}

// ---- the same four quantities, wide loads and run-length input -------------------------------------------------
// Shared tail: rowv[r] != 0 <=> row r holds a pixel.  Returns rows / first / last over the workgroup (256 threads);
// red: LDS, 3 x 4 ints.
__device__ inline void rows_summary(const int* rowv, int H, int* red, int tid, int* rows_out, int* span_out) {
  const int lane = tid & 63, wave = tid >> 6;
  int rows = 0, first = H, last = -1;
  for (int r = tid; r < H; r += 256)
    if (rowv[r] != 0) { rows += 1; first = min(first, r); last = max(last, r); }
  rows = wave_sum_i(rows);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { first = min(first, __shfl_xor(first, o)); last = max(last, __shfl_xor(last, o)); }
  if (lane == 0) { red[wave] = rows; red[4 + wave] = first; red[8 + wave] = last; }
  __syncthreads();
  int rw = 0, f = H, l = -1;
  for (int w = 0; w < 4; ++w) { rw += red[w]; f = min(f, red[4 + w]); l = max(l, red[8 + w]); }
  *rows_out = rw;
  *span_out = (l >= f) ? l - f + 1 : 0;
}

// u8 planes with W % 16 == 0 and 16-byte aligned planes: 16 pixels per load, four loads in flight per lane, per-row
// pixel counts accumulated in LDS (one atomic per non-empty group).  Dynamic LDS: H ints.
__global__ __launch_bounds__(256) void mask_stats_vec_kernel(const unsigned char* __restrict__ mask, int H, int W, int boundary,
                                                             int* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* rowcnt = reinterpret_cast<int*>(smem);
  __shared__ int red[12];
  __shared__ int tot[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = tid; r < H; r += 256) rowcnt[r] = 0;
  __syncthreads();
  const u32x4* m4 = reinterpret_cast<const u32x4*>(mask + (long long)blockIdx.x * H * W);
  const int gpr = W >> 4, ngroups = H * gpr;
  const int step_row = 256 / gpr, step_col = 256 % gpr;
  const int bc = min(boundary, W);
  int row = tid / gpr, col = tid - row * gpr;
  int area = 0, edge = 0;
#pragma unroll 4
  for (int g = tid; g < ngroups; g += 256) {
    const u32x4 w = __builtin_nontemporal_load(m4 + g);
    const unsigned pat = nz4(w.x) | (nz4(w.y) << 4) | (nz4(w.z) << 8) | (nz4(w.w) << 12);
    if (pat) {
      const int c0 = col << 4, n = __popc(pat);
      area += n;
      atomicAdd(&rowcnt[row], n);
      const int nlo = min(max(bc - c0, 0), 16), fhi = min(max(W - bc - c0, 0), 16);
      edge += __popc(pat & ((1u << nlo) - 1u)) + __popc(pat & (0xffffu & ~((1u << fhi) - 1u)));
    }
    col += step_col; row += step_row;
    if (col >= gpr) { col -= gpr; ++row; }
  }
  __syncthreads();
  // top / bottom strips from the row counts (a row inside both strips counts twice, as m[:b].sum() + m[-b:].sum() does)
  const int br = min(boundary, H);
  for (int r = tid; r < H; r += 256) {
    const int k = (r < br ? 1 : 0) + (r >= H - br ? 1 : 0);
    if (k) edge += k * rowcnt[r];
  }
  area = wave_sum_i(area);
  edge = wave_sum_i(edge);
  if (lane == 0) { tot[wave][0] = area; tot[wave][1] = edge; }
  int rows, span;
  rows_summary(rowcnt, H, red, tid, &rows, &span);  // has the barrier that publishes tot
  if (tid == 0) {
    int* o = stats + (long long)blockIdx.x * 4;
    o[0] = tot[0][0] + tot[1][0] + tot[2][0] + tot[3][0];
    o[1] = rows; o[2] = span;
    o[3] = tot[0][1] + tot[1][1] + tot[2][1] + tot[3][1];
  }
}

// COCO run lengths (column-major, zeros first): no plane is decoded.  A ones-run is the interval [s, e) of the
// column-major pixel index i = col * H + row, so every quantity is interval arithmetic: area = sum of lengths; the
// left / right strips are the index ranges [0, b*H) and [(W-b)*H, W*H); the top / bottom strips are the residues
// i mod H in [0, b) and [H-b, H), counted in closed form; row presence goes through a difference array over rows
// (two LDS atomics per run) and one prefix scan.  Dynamic LDS: H + 1 ints.
__device__ inline long long strip_rows_below(long long x, int H, int br) {  // pixels i < x with i mod H in the two row strips
  const long long q = x / H;
  const int r = (int)(x - q * H);
  return q * 2 * br + min(r, br) + max(0, r - (H - br));
}

__global__ __launch_bounds__(256) void mask_stats_rle_kernel(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                             int H, int W, int boundary, int* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* diff = reinterpret_cast<int*>(smem);  // [H + 1]
  __shared__ int red[12];
  __shared__ unsigned wtot[4];
  __shared__ long long tot[4][2];
  __shared__ int full;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long o0 = offsets[blockIdx.x];
  const int nr = (int)(offsets[blockIdx.x + 1] - o0);
  const int* cnt = counts + o0;
  const long long HW = (long long)H * W;
  for (int r = tid; r <= H; r += 256) diff[r] = 0;
  if (tid == 0) full = 0;
  const int bc = min(boundary, W), br = min(boundary, H);
  const long long left_end = (long long)bc * H, right_beg = (long long)(W - bc) * H;
  long long area = 0, edge = 0;
  unsigned long long carry = 0;
  for (int c0 = 0; c0 < nr; c0 += 256) {
    const int j = c0 + tid;
    unsigned len = 0;
    if (j < nr) { const int v = cnt[j]; len = v > 0 ? (unsigned)v : 0u; }
    unsigned incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();  // previous step's readers of wtot are done; the zeroing of diff is ordered
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long base = carry, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wtot[w]; total += wtot[w]; }
    const long long s = (long long)(base + incl - len);
    carry += total;
    if ((j & 1) && len > 0 && s < HW) {
      const long long e = min(s + (long long)len, HW);
      area += e - s;
      edge += max(0LL, min(e, left_end) - s) + max(0LL, e - max(s, right_beg));
      edge += strip_rows_below(e, H, br) - strip_rows_below(s, H, br);
      if (e - s >= H) {
        full = 1;  // every row holds a pixel
      } else {
        const int r0 = (int)(s % H), r1 = (int)((e - 1) % H);
        if (r0 <= r1) { atomicAdd(&diff[r0], 1); atomicAdd(&diff[r1 + 1], -1); }
        else { atomicAdd(&diff[r0], 1); atomicAdd(&diff[H], -1); atomicAdd(&diff[0], 1); atomicAdd(&diff[r1 + 1], -1); }
      }
    }
    if (carry >= (unsigned long long)HW) break;  // uniform: later runs fall outside the frame
  }
  __syncthreads();
  // prefix scan of the difference array in place (chunk per thread, chunk sums scanned through LDS)
  const int per = (H + 255) / 256, rb = min(tid * per, H), re = min(rb + per, H);
  int csum = 0;
  for (int r = rb; r < re; ++r) csum += diff[r];
  int incl = csum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = (unsigned)incl;
  __syncthreads();
  int run = incl - csum;
  for (int w = 0; w < wave; ++w) run += (int)wtot[w];
  const int all_rows = full;
  for (int r = rb; r < re; ++r) { run += diff[r]; diff[r] = (run > 0 || all_rows) ? 1 : 0; }
  // publish the sums, then rows / span (rows_summary's barrier orders the diff writes and tot)
  for (int o = 32; o > 0; o >>= 1) { area += __shfl_xor(area, o); edge += __shfl_xor(edge, o); }
  if (lane == 0) { tot[wave][0] = area; tot[wave][1] = edge; }
  __syncthreads();
  int rows, span;
  rows_summary(diff, H, red, tid, &rows, &span);
  if (tid == 0) {
    int* o = stats + (long long)blockIdx.x * 4;
    o[0] = (int)(tot[0][0] + tot[1][0] + tot[2][0] + tot[3][0]);
    o[1] = rows; o[2] = span;
    o[3] = (int)(tot[0][1] + tot[1][1] + tot[2][1] + tot[3][1]);
  }
}

// Box consumers (reference src/tools/combine_results.py:105-108, :238-252): project the 8 corners of every
// record with its image's K, 2-D AABB and its clamp to the frame.  One thread per box.
__global__ __launch_bounds__(128) void project_boxes_kernel(const double* __restrict__ rec, const double* __restrict__ K,
                                                            int k_stride, const int* __restrict__ image_index, int B,
                                                            double Wd, double Hd, double* __restrict__ out) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= B) return;
  const double* k = K + (long long)(image_index ? image_index[i] : i) * k_stride;
  const double* c = rec + (long long)i * LA3D_REC + 15;
  double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
  bool bad = false;
  for (int v = 0; v < 8; ++v) {
    double px, py;
    project_corner(k, c[v * 3], c[v * 3 + 1], c[v * 3 + 2], &px, &py);   // (K @ P)[:2] / (K @ P)[2]
    if (px != px || py != py) bad = true;              // Python's min()/max() over NaN are order dependent: report NaN
    lo[0] = fmin(lo[0], px); hi[0] = fmax(hi[0], px);
    lo[1] = fmin(lo[1], py); hi[1] = fmax(hi[1], py);
  }
  double* o = out + (long long)i * 8;
  if (bad) { for (int j = 0; j < 8; ++j) o[j] = NAN; return; }
  o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
  o[4] = fmax(0.0, lo[0]); o[5] = fmax(0.0, lo[1]); o[6] = fmin(Wd, hi[0]); o[7] = fmin(Hd, hi[1]);
}

// IoU of every pair of xyxy boxes (iou2D, reference src/tools/combine_results.py:111-124): the negated matrix is
// the Hungarian cost matrix of :131-135.
__global__ __launch_bounds__(256) void iou_matrix_kernel(const double* __restrict__ a, int na, const double* __restrict__ b,
                                                         int nb, double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)na * nb) return;
  const double* p = a + (t / nb) * 4;
  const double* q = b + (t % nb) * 4;
  const double x1 = fmax(p[0], q[0]), y1 = fmax(p[1], q[1]), x2 = fmin(p[2], q[2]), y2 = fmin(p[3], q[3]);
  const double inter = fmax(0.0, x2 - x1) * fmax(0.0, y2 - y1);
  out[t] = inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter + 1e-6);
}

// Masked depth-ratio median — reference src/util.py:476-486 (align_to_depth_match): overlap = mask_a & mask_b,
// scale = np.median(num[overlap] / den[overlap]) in float32.  One 512-thread workgroup per instance:
//   phase 1  the two u8 masks are read once with 16-byte loads into an overlap bit image in LDS;
//   fast     (round 3) sample -> bracket -> one counting sweep -> one collecting sweep -> exact select in LDS: see the block
//            marked "fast path" in the kernel; 650 -> 360 us per 1024 VGA instances, identical results; falls through to the
//            rounds below whenever a count does not confirm it
//   rounds   the k-th smallest ratio is found exactly by a most-significant-first radix select on an order-preserving key,
//            four rounds of 8 bits; a wave takes 64 consecutive pixels per step (coalesced 256-byte loads of num and den,
//            chunks without an overlap pixel are skipped), so the ratios are re-derived from memory once per round instead of
//            being kept.  Depth ratios share their leading bits, so a plain LDS histogram would serialise on a handful of
//            bins: every bin has 16 copies (one per lane & 15, laid out [bin][copy] so that equal bins fall on different
//            banks) - at most four lanes of a wave ever meet on one word.
//   even n   np.median averages the two middle values (in float32).  The upper one equals the lower one when the lower
//            key occurs often enough; otherwise it is the smallest key above it (one more sweep with an LDS atomicMin).
// Any NaN ratio makes the result NaN, as np.median does; an empty overlap gives count 0, NaN.
__device__ inline unsigned f32_key(float v) {
  const unsigned b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ inline float f32_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

constexpr int RM_COPIES = 16;

constexpr int RM_NT = 512;   // threads per workgroup
constexpr int RM_CAP = 6144; // keys the LDS buffer of the fast path holds (sample, then the candidates of the median's bin)

// Keys at ranks ra <= rb (0-based, ascending) among the m keys in LDS buf: most-significant-first radix select, four rounds of
// 8 bits, both ranks at once (wave 0 follows ra, wave 1 follows rb).  h2: LDS [2][256]; st: LDS [4] = prefix a, rank a, prefix b,
// rank b (initialised here).  Every thread of the workgroup calls it; results in st[0], st[2] after the final barrier.
__device__ inline void lds_select2(const unsigned* buf, int m, unsigned ra, unsigned rb, unsigned* h2, unsigned* st, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { st[0] = 0u; st[1] = ra; st[2] = 0u; st[3] = rb; }
  unsigned pmask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    h2[tid] = 0u;                        // RM_NT == 512 == 2 * 256
    __syncthreads();
    const unsigned pa = st[0], pb = st[2];
    for (int i = tid; i < m; i += RM_NT) {
      const unsigned k = buf[i], d = (k >> shift) & 0xffu;
      if ((k & pmask) == pa) atomicAdd(&h2[d], 1u);
      if ((k & pmask) == pb) atomicAdd(&h2[256 + d], 1u);
    }
    __syncthreads();
    if (wave < 2) {                      // four bins per lane, exclusive scan over the lanes
      const unsigned* h = h2 + 256 * wave;
      const unsigned b0 = h[4 * lane], b1 = h[4 * lane + 1], b2 = h[4 * lane + 2], b3 = h[4 * lane + 3];
      const unsigned mine = b0 + b1 + b2 + b3;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const unsigned rank = st[2 * wave + 1], excl = incl - mine;
      if (excl <= rank && rank < incl) { // exactly one lane
        unsigned acc = excl, bsel = 0;
        if (rank >= acc + b0) { acc += b0; bsel = 1;
          if (rank >= acc + b1) { acc += b1; bsel = 2;
            if (rank >= acc + b2) { acc += b2; bsel = 3; } } }
        st[2 * wave] = st[2 * wave] | ((4u * (unsigned)lane + bsel) << shift);
        st[2 * wave + 1] = rank - acc;
      }
    }
    pmask |= 0xffu << shift;
    __syncthreads();
  }
}

// One sweep of the fast path over the chunks that hold an overlap pixel (every cstep-th one).  MODE 0: append every key to buf
// (the sample).  MODE 1: count the keys below klo, histogram those in [klo, khi] by (key - klo) >> sh (256 bins x 4 copies),
// note NaN ratios.  MODE 2: append the keys in [klo, khi] to buf.  cnt: LDS counter of appended keys (entries beyond RM_CAP are
// dropped but counted); lt: LDS counter; nanflag: LDS.
template <int MODE>
__device__ inline void rm_sweep(const float* __restrict__ np_, const float* __restrict__ dp, const unsigned* bits, int nwords,
                                const unsigned short* clist, int nact, int cstep, unsigned klo, unsigned khi, int sh, unsigned* buf,
                                unsigned* cnt, unsigned* h1, unsigned* lt, unsigned* nanflag, int wave, int lane) {
  constexpr int U = 8;     // chunks in flight per wave
  unsigned lt_local = 0, nan_local = 0;
  for (int j0 = wave * U * cstep; j0 < nact; j0 += (RM_NT / 64) * U * cstep) {
    float a[U], d[U];
    unsigned on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      on[u] = 0; a[u] = 0.f; d[u] = 1.f;
      const int j = j0 + u * cstep;
      if (j < nact) {
        const int c = clist[j];
        const unsigned w0 = bits[2 * c], w1 = (2 * c + 1 < nwords) ? bits[2 * c + 1] : 0u;
        on[u] = lane < 32 ? (w0 >> lane) & 1u : (w1 >> (lane - 32)) & 1u;
        if (on[u]) { a[u] = np_[c * 64 + lane]; d[u] = dp[c * 64 + lane]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u * cstep >= nact) continue;   // uniform
      const float r = a[u] / d[u];
      const unsigned key = f32_key(r);
      bool take = on[u] != 0;
      if (MODE == 1) {
        if (take) {
          nan_local |= (r != r) ? 1u : 0u;
          lt_local += key < klo ? 1u : 0u;
          if (key >= klo && key <= khi) atomicAdd(&h1[((key - klo) >> sh) * 4 + (lane & 3)], 1u);
        }
        continue;
      }
      if (MODE == 2) take = take && key >= klo && key <= khi;
      const unsigned long long bal = __ballot(take);
      if (bal == 0) continue;
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(cnt, (unsigned)__popcll(bal));
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
      if (take) {
        const unsigned pos = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (pos < (unsigned)RM_CAP) buf[pos] = key;
      }
    }
  }
  if (MODE == 1) {
    lt_local = (unsigned)wave_sum_i((int)lt_local);
    if (lane == 0 && lt_local) atomicAdd(lt, lt_local);
    if (__ballot(nan_local != 0) != 0 && lane == 0) *nanflag = 1u;
  }
}

__global__ __launch_bounds__(RM_NT) void ratio_median_kernel(const float* __restrict__ num, long long num_stride,
                                                           const int* __restrict__ image_index, const float* __restrict__ den,
                                                           const unsigned char* __restrict__ mask_a,
                                                           const unsigned char* __restrict__ mask_b, int HW, int nwords,
                                                           float* __restrict__ median, int* __restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* bits = reinterpret_cast<unsigned*>(smem);
  unsigned* hist = bits + ((nwords + 3) & ~3);   // fallback rounds: [256][RM_COPIES]; fast path: the key buffer, RM_CAP words
  static_assert(RM_CAP >= 256 * RM_COPIES, "the key buffer also holds the fallback's histogram");
  unsigned* h1 = hist + RM_CAP;                  // fast path: [256][4] bins of the bracket; lds_select2: [2][256]
  unsigned* bsum = h1 + 1024;                    // [256] bin totals
  unsigned* misc = bsum + 256;                   // [0] n, [1] nan flag, [2] prefix, [3] rank, [4] count of the selected bin, [5] min key above,
                                                 // [6] number of active chunks, [7] appended keys, [8] keys below the bracket,
                                                 // [9] fast-path verdict, [10..13] lds_select2 state, [14] lo2, [15] hi2
  unsigned short* clist = reinterpret_cast<unsigned short*>(misc + 16);   // ids of the 64-pixel chunks holding an overlap pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, inst = blockIdx.x;
  const float* np_ = num + (long long)(image_index ? image_index[inst] : inst) * num_stride;
  const float* dp = den + (long long)inst * HW;
  const unsigned char* ma = mask_a + (long long)inst * HW;
  const unsigned char* mb = mask_b ? mask_b + (long long)inst * HW : nullptr;
  if (tid < 16) misc[tid] = tid == 5 ? 0xffffffffu : 0u;
  // ---- phase 1: overlap bit image ----
  unsigned n_local = 0;
  const bool vec = (HW % 16 == 0) && ((reinterpret_cast<uintptr_t>(ma) & 15) == 0) && (!mb || (reinterpret_cast<uintptr_t>(mb) & 15) == 0);
  if (vec) {
    unsigned short* b16 = reinterpret_cast<unsigned short*>(bits);
    const u32x4* a4 = reinterpret_cast<const u32x4*>(ma);
    const u32x4* b4 = reinterpret_cast<const u32x4*>(mb);
    const int ngroups = HW >> 4;
#pragma unroll 4
    for (int g = tid; g < ngroups; g += RM_NT) {
      const u32x4 wa = __builtin_nontemporal_load(a4 + g);
      unsigned pat = nz16(wa.x, wa.y, wa.z, wa.w);
      if (mb) {
        const u32x4 wb = __builtin_nontemporal_load(b4 + g);
        pat &= nz16(wb.x, wb.y, wb.z, wb.w);
      }
      b16[g] = (unsigned short)pat;
      n_local += __popc(pat);
    }
    if ((ngroups & 1) && tid == 0) b16[ngroups] = 0;
  } else {
    for (int w = tid; w < nwords; w += RM_NT) {
      unsigned word = 0;
      const int i0 = w * 32;
      for (int k = 0; k < 32; ++k) {
        const int i = i0 + k;
        if (i < HW && ma[i] && (!mb || mb[i])) word |= 1u << k;
      }
      bits[w] = word;
      n_local += __popc(word);
    }
  }
  n_local = (unsigned)wave_sum_i((int)n_local);
  __syncthreads();                       // misc is initialised
  if (lane == 0) atomicAdd(&misc[0], n_local);
  __syncthreads();
  const unsigned n = misc[0];
  if (n == 0) {
    if (tid == 0) { median[inst] = NAN; count[inst] = 0; }
    return;
  }
  if (tid == 0) misc[3] = (n & 1u) ? n / 2 : n / 2 - 1;   // 0-based rank of the (lower) middle value
  const int nchunks = (HW + 63) >> 6;
  // active-chunk list (order is irrelevant): the rounds visit only chunks with an overlap pixel
  for (int c0 = 0; c0 < nchunks; c0 += RM_NT) {
    const int c = c0 + tid;
    const bool act = c < nchunks && ((bits[2 * c] | ((2 * c + 1 < nwords) ? bits[2 * c + 1] : 0u)) != 0);
    const unsigned long long bal = __ballot(act);
    unsigned base = 0;
    if (lane == 0 && bal) base = atomicAdd(&misc[6], (unsigned)__popcll(bal));
    base = __shfl(base, 0);
    if (act) clist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)c;
  }
  __syncthreads();
  const int nact = (int)misc[6];
  // ---- fast path (round 3): sample -> bracket -> one counting sweep -> one collecting sweep -> exact select in LDS --------
  // The four radix rounds below visit every overlap pixel four (five) times, each time re-deriving the ratio from two loads and
  // a division, and they are latency-bound.  Here: (0) the keys of every cstep-th active chunk (~2-4 k keys) go to LDS and two
  // of their order statistics, 50 % -/+ 1/12, bracket the median; (1) one sweep counts the keys below the bracket and
  // histograms the keys inside it in <= 256 power-of-two bins; the bin(s) holding the middle rank(s) hold n / 1000 keys or so;
  // (2) one sweep collects exactly those keys; (3) an in-LDS radix select gives the exact middle value(s).  Every step is
  // verified by counts: if the bracket misses the median, a bin overflows the buffer, or the sample was too small, the
  // verdict stays 0 and the radix rounds below run as before.  An overlap of <= RM_CAP pixels is selected from step (0) alone.
  {
    unsigned* st = misc + 10;
    const unsigned rlo = (n & 1u) ? n / 2 : n / 2 - 1, rhi = n / 2;       // 0-based ranks of the middle value(s)
    constexpr unsigned RM_SAMPLE = 2048u;   // keys the bracket is estimated from
    const int cstep = (int)(n <= (unsigned)RM_CAP ? 1u : (n + RM_SAMPLE - 1u) / RM_SAMPLE);
    rm_sweep<0>(np_, dp, bits, nwords, clist, nact, cstep, 0u, 0xffffffffu, 0, hist, &misc[7], h1, &misc[8], &misc[1], wave, lane);
    __syncthreads();
    const unsigned ns_all = misc[7];
    const int ns = (int)(ns_all < (unsigned)RM_CAP ? ns_all : (unsigned)RM_CAP);
    if (cstep == 1 && ns_all == n) {     // uniform: every key is in LDS - select directly (NaN keys sort last: check them here)
      unsigned nanl = 0;
      for (int i = tid; i < ns; i += RM_NT) nanl |= (hist[i] > 0xff800000u || (hist[i] < 0x007fffffu)) ? 1u : 0u;   // NaN keys
      if (__ballot(nanl != 0) != 0 && lane == 0) misc[1] = 1u;
      lds_select2(hist, ns, rlo, rhi, h1, st, tid);
      if (tid == 0) {
        const float v0 = f32_unkey(st[0]), v1 = f32_unkey(st[2]);
        median[inst] = misc[1] ? NAN : ((n & 1u) ? v0 : (v0 + v1) / 2.0f);
        count[inst] = (int)n;
      }
      return;
    }
    if (ns >= 512) {                     // uniform: enough of a sample to bracket with
      const unsigned w = (unsigned)ns / 12u;
      lds_select2(hist, ns, (unsigned)ns / 2u - w, (unsigned)ns / 2u + w, h1, st, tid);
      const unsigned klo = st[0], khi = st[2];
      const unsigned width = khi - klo;
      const int sh = width < 256u ? 0 : (32 - __clz((int)width)) - 8;    // (width >> sh) < 256
      __syncthreads();                   // everyone has read st before the counters are reused
      for (int i = tid; i < 1024; i += RM_NT) h1[i] = 0u;
      if (tid == 0) { misc[7] = 0u; misc[8] = 0u; }
      __syncthreads();
      rm_sweep<1>(np_, dp, bits, nwords, clist, nact, 1, klo, khi, sh, hist, &misc[7], h1, &misc[8], &misc[1], wave, lane);
      __syncthreads();
      if (misc[1] != 0) {                // uniform: a NaN ratio
        if (tid == 0) { median[inst] = NAN; count[inst] = (int)n; }
        return;
      }
      if (tid < 256) bsum[tid] = h1[4 * tid] + h1[4 * tid + 1] + h1[4 * tid + 2] + h1[4 * tid + 3];
      __syncthreads();
      if (tid == 0) {                    // (256 bins, one thread: ~1 us, once per instance)
        const unsigned below = misc[8];
        unsigned acc = below, b0 = 256, b1 = 256, before = 0;
        for (unsigned b = 0; b < 256; ++b) {
          const unsigned c = bsum[b];
          if (b0 == 256 && rlo >= acc && rlo < acc + c) { b0 = b; before = acc; }
          if (b1 == 256 && rhi >= acc && rhi < acc + c) b1 = b;
          acc += c;
        }
        unsigned ok = (rlo >= below && b0 < 256 && b1 < 256) ? 1u : 0u;
        unsigned tot = 0;
        if (ok) {
          for (unsigned b = b0; b <= b1; ++b) tot += bsum[b];
          if (tot > (unsigned)RM_CAP) ok = 0;
        }
        misc[9] = ok;
        if (ok) {
          misc[14] = klo + (b0 << sh);
          const unsigned long long top = (unsigned long long)klo + ((unsigned long long)(b1 + 1) << sh) - 1ull;
          misc[15] = top > (unsigned long long)khi ? khi : (unsigned)top;
          misc[4] = rlo - before; misc[3] = rhi - before; misc[2] = tot;
        }
      }
      __syncthreads();
      if (misc[9] != 0) {                // uniform
        rm_sweep<2>(np_, dp, bits, nwords, clist, nact, 1, misc[14], misc[15], 0, hist, &misc[7], h1, &misc[8], &misc[1], wave, lane);
        __syncthreads();
        if (misc[7] == misc[2]) {        // uniform: exactly the keys the histogram promised
          lds_select2(hist, (int)misc[7], misc[4], misc[3], h1, st, tid);
          if (tid == 0) {
            const float v0 = f32_unkey(st[0]), v1 = f32_unkey(st[2]);
            median[inst] = (n & 1u) ? v0 : (v0 + v1) / 2.0f;
            count[inst] = (int)n;
          }
          return;
        }
      }
    }
    __syncthreads();
    if (tid < 16 && tid != 0 && tid != 6) misc[tid] = tid == 5 ? 0xffffffffu : 0u;   // back to the state the radix rounds expect
    if (tid == 0) misc[3] = (n & 1u) ? n / 2 : n / 2 - 1;
    __syncthreads();
  }
  const int copy = lane & (RM_COPIES - 1);
  unsigned pmask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256 * RM_COPIES; i += RM_NT) hist[i] = 0;
    __syncthreads();                     // also publishes misc[2..3] of the previous round
    const unsigned prefix = misc[2];
    unsigned nan_local = 0;
    constexpr int RM_U = 4;      // chunks in flight per wave: 2 x RM_U coalesced loads issued before any is used
    for (int j0 = wave * RM_U; j0 < nact; j0 += (RM_NT / 64) * RM_U) {
      float a[RM_U], d[RM_U];
      unsigned on[RM_U];
#pragma unroll
      for (int u = 0; u < RM_U; ++u) {
        on[u] = 0; a[u] = 0.f; d[u] = 1.f;
        if (j0 + u < nact) {
          const int c = clist[j0 + u];
          const unsigned w0 = bits[2 * c], w1 = (2 * c + 1 < nwords) ? bits[2 * c + 1] : 0u;
          on[u] = lane < 32 ? (w0 >> lane) & 1u : (w1 >> (lane - 32)) & 1u;
          if (on[u]) { a[u] = np_[c * 64 + lane]; d[u] = dp[c * 64 + lane]; }
        }
      }
#pragma unroll
      for (int u = 0; u < RM_U; ++u) {
        if (on[u]) {
          const float r = a[u] / d[u];
          if (shift == 24) nan_local |= (r != r) ? 1u : 0u;
          const unsigned key = f32_key(r);
          if ((key & pmask) == prefix) atomicAdd(&hist[((key >> shift) & 0xffu) * RM_COPIES + copy], 1u);
        }
      }
    }
    if (shift == 24 && __ballot(nan_local != 0) != 0 && lane == 0) misc[1] = 1u;
    __syncthreads();
    if (shift == 24 && misc[1] != 0) {   // uniform
      if (tid == 0) { median[inst] = NAN; count[inst] = (int)n; }
      return;
    }
    if (tid < 256) {   // bin totals, then the bin holding the wanted rank (wave 0: four bins per lane, exclusive scan over the lanes)
      unsigned t = 0;
#pragma unroll
      for (int k = 0; k < RM_COPIES; ++k) t += hist[tid * RM_COPIES + k];
      bsum[tid] = t;
    }
    __syncthreads();
    if (wave == 0) {
      const unsigned b0 = bsum[4 * lane], b1 = bsum[4 * lane + 1], b2 = bsum[4 * lane + 2], b3 = bsum[4 * lane + 3];
      const unsigned mine = b0 + b1 + b2 + b3;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const unsigned rank = misc[3], excl = incl - mine;
      if (excl <= rank && rank < incl) {   // exactly one lane
        unsigned acc = excl, bsel = 0, cnt = b0;
        if (rank >= acc + b0) { acc += b0; bsel = 1; cnt = b1;
          if (rank >= acc + b1) { acc += b1; bsel = 2; cnt = b2;
            if (rank >= acc + b2) { acc += b2; bsel = 3; cnt = b3; } } }
        misc[2] = prefix | ((4u * (unsigned)lane + bsel) << shift);
        misc[3] = rank - acc;
        misc[4] = cnt;
      }
    }
    pmask |= 0xffu << shift;
    __syncthreads();
  }
  const unsigned key0 = misc[2];
  unsigned key1 = key0;
  if (!(n & 1u) && misc[3] + 1 >= misc[4]) {   // uniform: the upper middle value is the smallest key above key0
    for (int j = wave; j < nact; j += RM_NT / 64) {
      const int c = clist[j];
      const unsigned w0 = bits[2 * c], w1 = (2 * c + 1 < nwords) ? bits[2 * c + 1] : 0u;
      const unsigned on = lane < 32 ? (w0 >> lane) & 1u : (w1 >> (lane - 32)) & 1u;
      unsigned k = 0xffffffffu;
      if (on) {
        const int i = c * 64 + lane;
        const unsigned key = f32_key(np_[i] / dp[i]);
        if (key > key0) k = key;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) k = min(k, (unsigned)__shfl_xor((int)k, o));
      if (lane == 0 && k != 0xffffffffu) atomicMin(&misc[5], k);
    }
    __syncthreads();
    key1 = misc[5];
  }
  if (tid == 0) {
    const float v0 = f32_unkey(key0);
    median[inst] = (n & 1u) ? v0 : (v0 + f32_unkey(key1)) / 2.0f;   // float32 mean of the two middle values
    count[inst] = (int)n;
  }
}

// ---- align_depth support (reference src/batch_scripts/depth.py:52-92) -----------------------------------------------
// valid = ~isinf(relative) & (metric < max_valid) [& mask]; the regressor (scikit-learn RANSAC, third party, random) is fed
// relative[valid], metric[valid] in row-major order, and its prediction is scattered back over a 10000.0-filled frame.
// Order-preserving stream compaction in three small kernels: per-tile counts, scan of the counts, scatter.
constexpr int AL_TILE = 4096;   // elements per 256-thread workgroup (16 per thread, four float4)

__device__ inline bool align_valid(float rel, float met, unsigned char m, bool has_mask, float max_valid) {
  const bool isinf_rel = (__float_as_uint(rel) & 0x7fffffffu) == 0x7f800000u;   // np.isinf: NaN is NOT excluded
  return !isinf_rel && (met < max_valid) && (!has_mask || m != 0);
}

__global__ __launch_bounds__(256) void align_count_kernel(const float* __restrict__ rel, const float* __restrict__ met,
                                                          const unsigned char* __restrict__ mask, long long n, float max_valid,
                                                          long long* __restrict__ counts) {
  __shared__ int part[4];
  // blockIdx.y = frame of a batch (la3d_align_select_batch): planes n apart, (gridDim.x + 1) count slots per frame
  rel += (long long)blockIdx.y * n; met += (long long)blockIdx.y * n;
  if (mask) mask += (long long)blockIdx.y * n;
  counts += (long long)blockIdx.y * (gridDim.x + 1);
  const long long base = (long long)blockIdx.x * AL_TILE;
  int c = 0;
  for (int k = 0; k < AL_TILE / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) c += align_valid(rel[i], met[i], mask ? mask[i] : 1, mask != nullptr, max_valid) ? 1 : 0;
  }
  c = wave_sum_i(c);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the tile counts in place, total appended at counts[nb]; one workgroup
__global__ __launch_bounds__(256) void align_scan_kernel(long long* __restrict__ counts, int nb, long long* __restrict__ total) {
  __shared__ long long carry;
  __shared__ long long wsum[4];
  counts += (long long)blockIdx.x * (nb + 1);   // one workgroup per frame
  total += blockIdx.x;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const long long v = b < nb ? counts[b] : 0;
    long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long t = __shfl_up(incl, o);
      if ((threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    long long off = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    if (b < nb) counts[b] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { counts[nb] = carry; *total = carry; }
}

__global__ __launch_bounds__(256) void align_scatter_kernel(const float* __restrict__ rel, const float* __restrict__ met,
                                                            const unsigned char* __restrict__ mask, long long n, float max_valid,
                                                            const long long* __restrict__ offsets, float* __restrict__ rel_out,
                                                            float* __restrict__ met_out) {
  __shared__ int wtot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  rel += (long long)blockIdx.y * n; met += (long long)blockIdx.y * n;       // frame of a batch: outputs have capacity n per frame
  if (mask) mask += (long long)blockIdx.y * n;
  rel_out += (long long)blockIdx.y * n; met_out += (long long)blockIdx.y * n;
  offsets += (long long)blockIdx.y * (gridDim.x + 1);
  const long long base = (long long)blockIdx.x * AL_TILE;
  long long out = offsets[blockIdx.x];
  for (int k = 0; k < AL_TILE / 256; ++k) {   // 256 consecutive elements per step: row-major order is kept
    const long long i = base + k * 256 + threadIdx.x;
    float r = 0.f, m = 0.f;
    bool v = false;
    if (i < n) { r = rel[i]; m = met[i]; v = align_valid(r, m, mask ? mask[i] : 1, mask != nullptr, max_valid); }
    const unsigned long long bal = __ballot(v);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    long long pos = out;
    for (int w = 0; w < wave; ++w) pos += wtot[w];
    const int step_total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (v) {
      pos += __popcll(bal & ((1ull << lane) - 1ull));
      rel_out[pos] = r;
      met_out[pos] = m;
    }
    out += step_total;
    __syncthreads();
  }
}

// depth = full(fill); depth[sel] = relative[sel] * coef + intercept, sel = mask (if given) else ~isinf(relative)  (:82-90)
__global__ __launch_bounds__(256) void align_apply_kernel(const float* __restrict__ rel, const unsigned char* __restrict__ mask,
                                                          long long n, float coef, float intercept, float fill,
                                                          float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float r = rel[i];
  const bool sel = mask ? mask[i] != 0 : (__float_as_uint(r) & 0x7fffffffu) != 0x7f800000u;
  // LinearRegression.predict on float32: X @ coef_.T (one float32 product) + intercept_
  out[i] = sel ? __fadd_rn(__fmul_rn(r, coef), intercept) : fill;
}

// Sparse unprojection at match points — reference src/matching/matcher.py:70-91: depth looked up at
// (int(v), int(u)), points with depth == -1 dropped, u' = flip - u, v' = flip - v (flip = 512 there),
// p = ((u'-cx) d / fx, (v'-cy) d / fy, d), world = R (p - T).  One thread per match.
struct MatchParams {
  double fx, fy, cx, cy, flip;
  double R[9], T[3];
  int has_rt, use_flip;
  int H, W, N;
};
__global__ __launch_bounds__(128) void unproject_matches_kernel(const float* __restrict__ depth, const double* __restrict__ uv,
                                                                const MatchParams p, double* __restrict__ out,
                                                                int* __restrict__ valid) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= p.N) return;
  const double mu = uv[2 * i], mv = uv[2 * i + 1];
  const long long cu = (long long)mu, cv = (long long)mv;    // astype(int): truncation toward zero
  double* o = out + (long long)i * 3;
  bool ok = cu >= 0 && cu < p.W && cv >= 0 && cv < p.H;
  float df = -1.f;
  if (ok) df = depth[cv * p.W + cu];
  ok = ok && (df != -1.f);
  valid[i] = ok ? 1 : 0;
  if (!ok) { o[0] = o[1] = o[2] = NAN; return; }
  const double d = (double)df;
  const double u = p.use_flip ? p.flip - mu : mu, v = p.use_flip ? p.flip - mv : mv;
  double q[3] = {(u - p.cx) * d / p.fx, (v - p.cy) * d / p.fy, d};
  if (p.has_rt) {
    const double a = q[0] - p.T[0], b = q[1] - p.T[1], c = q[2] - p.T[2];
    q[0] = p.R[0] * a + p.R[1] * b + p.R[2] * c;
    q[1] = p.R[3] * a + p.R[4] * b + p.R[5] * c;
    q[2] = p.R[6] * a + p.R[7] * b + p.R[8] * c;
  }
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
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

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

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
  const int vec16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (out_is_f64) hipLaunchKernelGGL(unproject_kernel<double>, dim3(blocks), dim3(256), 0, s, depth, static_cast<double*>(out), p, vec16);
  else hipLaunchKernelGGL(unproject_kernel<float>, dim3(blocks), dim3(256), 0, s, depth, static_cast<float*>(out), p, vec16);
  return check_launch("unproject_kernel");
}

int la3d_unproject_batch(const float* depth, const double* K, int32_t k_stride, const double* Rt12, int P, int H, int W, void* out,
                         int out_is_f64, void* stream) {
  if (!depth || !K || !out || P < 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 4 || (k_stride != 0 && k_stride < 9) ||
      P > 65535) {
    set_err("la3d_unproject_batch: bad argument (P <= 65535)");
    return LA3D_ERR_ARG;
  }
  if (P == 0) return LA3D_SUCCESS;
  UnprojParams p;
  for (int i = 0; i < 9; ++i) p.Kinv[i] = 0.0;
  p.has_rt = Rt12 != nullptr;
  for (int i = 0; i < 9; ++i) p.R[i] = Rt12 ? Rt12[i] : ((i % 4 == 0) ? 1.0 : 0.0);
  for (int i = 0; i < 3; ++i) p.t[i] = Rt12 ? Rt12[9 + i] : 0.0;
  p.H = H; p.W = W; p.HW = H * W; p.rcpW = 1.0f / (float)W;
  int bx = (p.HW + 255) / 256;
  // enough workgroups over all frames to fill the chip several times - but never fewer than 256 per frame: the ~2000 resident
  // workgroups then write into ~8 frames at a time instead of 64 (profiles/r05/r05_unproject_sweep.txt: 256 / 1024 frames of 640x480,
  // f64 out: 4.66 / 5.24 TB/s with 32 workgroups per frame, 5.41 / 5.78 with 256)
  int want = (8192 + P - 1) / P;
  if (want < 256) want = 256;
  if (bx > want) bx = want;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 16-byte stores need every frame's output base 16-aligned: HW * 3 * sizeof(OutT) a multiple of 16
  const int vec16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ((long long)p.HW * 3 * (out_is_f64 ? 8 : 4)) % 16 == 0;
  if (out_is_f64) hipLaunchKernelGGL(unproject_batch_kernel<double>, dim3(bx, P), dim3(256), 0, s, depth, K, k_stride, static_cast<double*>(out), p, vec16);
  else hipLaunchKernelGGL(unproject_batch_kernel<float>, dim3(bx, P), dim3(256), 0, s, depth, K, k_stride, static_cast<float*>(out), p, vec16);
  return check_launch("unproject_batch_kernel");
}

int la3d_pad_rows(const float* src, int64_t rows, int W, int Wp, float* dst, void* stream) {
  if (rows < 0 || W <= 0 || Wp < W || Wp % 4 != 0 || (rows > 0 && (!src || !dst)) || (reinterpret_cast<uintptr_t>(dst) & 15)) {
    set_err("la3d_pad_rows: bad argument (Wp >= W, Wp % 4 == 0, dst 16-byte aligned)");
    return LA3D_ERR_ARG;
  }
  if (rows == 0) return LA3D_SUCCESS;
  const long long total = (long long)rows * (Wp / 4);
  const long long want = (total + 255) / 256;
  const int blocks = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, (long long)rows, W, Wp, dst);
  return check_launch("pad_rows_kernel");
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

int la3d_rle_from_string_host(const char* s, int64_t len, int32_t* counts, int cap) {
  // pycocotools rleFrString (maskApi.c): 5-bit groups, char - 48, bit 5 = continuation, bit 4 of the last
  // group = sign; counts beyond the third are stored as a difference to the count two places earlier
  if (!s || len < 0 || (!counts && cap > 0)) return -1;
  int m = 0;
  int64_t pz = 0;
  while (pz < len && s[pz]) {
    long x = 0;
    int k = 0, more = 1;
    while (more) {
      if (pz >= len) return -1;
      const int c = s[pz] - 48;
      x |= (long)(c & 0x1f) << (5 * k);
      more = c & 0x20;
      ++pz; ++k;
      if (!more && (c & 0x10)) x |= -1L << (5 * k);
    }
    if (m > 2) x += counts[m - 2];
    if (m >= cap) return -1;
    counts[m++] = (int32_t)x;
  }
  return m;
}

int la3d_rle_decode(const int32_t* counts, const int64_t* offsets, int B, int H, int W, uint8_t* mask_out, void* stream) {
  if ((!counts && B > 0) || !offsets || !mask_out || B < 0 || H <= 0 || W <= 0 || (long long)H * W > (1LL << 20)) {
    set_err("la3d_rle_decode: bad argument (H*W <= 1048576)");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  const int nwords = (H * W + 31) / 32;
  // behind the bit image: 16 words of wave totals, then the block totals of the column scan (word-aligned rows)
  const int scan_words = (W % 32 == 0) ? ((NT_DEC / (W / 32) > 2 ? NT_DEC / (W / 32) : 2) * (W / 32)) : 0;
  const size_t lds = (size_t)nwords * 4 + 64 + (size_t)scan_words * 4;
  if (lds > 160 * 1024 - 256) {
    set_err("la3d_rle_decode: frame too large for LDS");
    return LA3D_ERR_UNSUPPORTED;
  }
  allow_big_lds(reinterpret_cast<const void*>(rle_decode_kernel));
  hipLaunchKernelGGL(rle_decode_kernel, dim3(B), dim3(NT_DEC), lds, static_cast<hipStream_t>(stream), counts,
                     reinterpret_cast<const long long*>(offsets), H, W, nwords, scan_words, mask_out);
  return check_launch("rle_decode_kernel");
}

int la3d_poly_decode(const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings, int B, int H, int W,
                     uint8_t* mask_out, void* stream) {
  if (((!poly_xy || !ring_offsets || !inst_rings || !mask_out) && B > 0) || B < 0 || H <= 0 || W <= 0 ||
      (long long)H * W > (1LL << 20)) {
    set_err("la3d_poly_decode: bad argument (H*W <= 1048576)");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  const int nwords = (H * W + 31) / 32;
  const size_t lds = (((size_t)nwords * 4 + 15) & ~(size_t)15) + POLY_STAGE_BYTES + 64;
  allow_big_lds(reinterpret_cast<const void*>(poly_decode_kernel));
  hipLaunchKernelGGL(poly_decode_kernel, dim3(B), dim3(NT_DEC), lds, static_cast<hipStream_t>(stream), poly_xy,
                     reinterpret_cast<const long long*>(ring_offsets), reinterpret_cast<const long long*>(inst_rings), H, W, nwords,
                     mask_out);
  return check_launch("poly_decode_kernel");
}

int la3d_mask_stats_poly(const int32_t* poly_xy, const int64_t* ring_offsets, const int64_t* inst_rings, int B, int H, int W,
                         int boundary, int32_t* stats, void* stream) {
  if (((!poly_xy || !ring_offsets || !inst_rings || !stats) && B > 0) || B < 0 || H <= 0 || W <= 0 || boundary < 0 ||
      (long long)H * W > (1LL << 20)) {
    set_err("la3d_mask_stats_poly: bad argument (H*W <= 1048576)");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  const int nwords = (H * W + 31) / 32;
  const size_t lds = (((size_t)nwords * 4 + 15) & ~(size_t)15) + POLY_STAGE_BYTES + 64 + (size_t)H * 4 + 128;
  if (lds > 160 * 1024 - 256) {
    set_err("la3d_mask_stats_poly: frame too large for LDS");
    return LA3D_ERR_UNSUPPORTED;
  }
  allow_big_lds(reinterpret_cast<const void*>(mask_stats_poly_kernel));
  hipLaunchKernelGGL(mask_stats_poly_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), poly_xy,
                     reinterpret_cast<const long long*>(ring_offsets), reinterpret_cast<const long long*>(inst_rings), H, W, nwords,
                     boundary, stats);
  return check_launch("mask_stats_poly_kernel");
}

static void stats_lds_attr() {  // rows beyond 16 K need more than the default 64 KiB of dynamic LDS
  for (const void* k : {reinterpret_cast<const void*>(mask_stats_rle_kernel), reinterpret_cast<const void*>(mask_stats_vec_kernel)})
    allow_big_lds(k, 160 * 1024 - 1024);
}

int la3d_mask_stats(const uint8_t* mask, int B, int H, int W, int boundary, int32_t* stats, void* stream) {
  if (!mask || !stats || B < 0 || H <= 0 || W <= 0 || boundary < 0) {
    set_err("la3d_mask_stats: bad argument");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  if (W % 16 == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0 && H <= 32768) {
    stats_lds_attr();
    hipLaunchKernelGGL(mask_stats_vec_kernel, dim3(B), dim3(256), (size_t)H * 4, static_cast<hipStream_t>(stream), mask, H, W,
                       boundary, stats);
    return check_launch("mask_stats_vec_kernel");
  }
  hipLaunchKernelGGL(mask_stats_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), mask, H, W, boundary, stats);
  return check_launch("mask_stats_kernel");
}

int la3d_mask_stats_rle(const int32_t* counts, const int64_t* offsets, int B, int H, int W, int boundary, int32_t* stats,
                        void* stream) {
  if ((!counts && B > 0) || !offsets || !stats || B < 0 || H <= 0 || W <= 0 || boundary < 0 || H > 32768 ||
      (long long)H * W > (1LL << 30)) {
    set_err("la3d_mask_stats_rle: bad argument (H <= 32768, H*W <= 2^30)");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  stats_lds_attr();
  hipLaunchKernelGGL(mask_stats_rle_kernel, dim3(B), dim3(256), (size_t)(H + 1) * 4, static_cast<hipStream_t>(stream), counts,
                     reinterpret_cast<const long long*>(offsets), H, W, boundary, stats);
  return check_launch("mask_stats_rle_kernel");
}

int la3d_masked_ratio_median(const float* num, int64_t num_plane_stride, const int32_t* image_index, const float* den,
                             const uint8_t* mask_a, const uint8_t* mask_b, int B, int H, int W, float* median,
                             int32_t* count, void* stream) {
  if (!num || !den || !mask_a || !median || !count || B < 0 || H <= 0 || W <= 0 || num_plane_stride < 0) {
    set_err("la3d_masked_ratio_median: bad argument (null pointer, negative size or stride)");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  // ONE size limit, from the LDS the kernel needs: bit image + chunk list + key buffer in one CU's 160 KiB (about 819 k pixels)
  const long long HWl = (long long)H * W;
  const long long nwl = (HWl + 31) / 32;
  const long long ldsl = ((nwl + 3) & ~3LL) * 4 + (long long)(RM_CAP + 1024 + 256 + 16) * 4 + ((HWl + 63) / 64) * 2 + 16;
  if (ldsl > 160 * 1024) {
    set_err("la3d_masked_ratio_median: frame too large for the LDS bit image (H*W up to about 819200)");
    return LA3D_ERR_UNSUPPORTED;
  }
  const int HW = (int)HWl, nwords = (int)nwl;
  const size_t lds = (size_t)ldsl;
  allow_big_lds(reinterpret_cast<const void*>(ratio_median_kernel));
  hipLaunchKernelGGL(ratio_median_kernel, dim3(B), dim3(RM_NT), lds, static_cast<hipStream_t>(stream), num,
                     (long long)num_plane_stride, image_index, den, mask_a, mask_b, HW, nwords, median, count);
  return check_launch("ratio_median_kernel");
}

size_t la3d_align_workspace_bytes(int64_t n) {
  if (n <= 0) return 8;
  return (size_t)((n + AL_TILE - 1) / AL_TILE + 2) * 8;   // per frame: la3d_align_select_batch needs P times this
}

int la3d_align_select_batch(const float* relative, const float* metric, const uint8_t* mask, int P, int64_t n,
                            float max_valid_depth, float* relative_out, float* metric_out, int64_t* counts, void* workspace,
                            void* stream) {
  if (P < 0 || P > 65535 || n < 0 || (P > 0 && (!counts || !workspace)) ||
      (P > 0 && n > 0 && (!relative || !metric || !relative_out || !metric_out))) {
    set_err("la3d_align_select_batch: bad argument (P <= 65535)");
    return LA3D_ERR_ARG;
  }
  if (P == 0) return LA3D_SUCCESS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  long long* tile_counts = static_cast<long long*>(workspace);   // [P][nb + 1]
  const int nb = (int)((n + AL_TILE - 1) / AL_TILE);
  if (nb > 0)
    hipLaunchKernelGGL(align_count_kernel, dim3(nb, P), dim3(256), 0, s, relative, metric, mask, (long long)n, max_valid_depth,
                       tile_counts);
  hipLaunchKernelGGL(align_scan_kernel, dim3(P), dim3(256), 0, s, tile_counts, nb, reinterpret_cast<long long*>(counts));
  if (nb > 0)
    hipLaunchKernelGGL(align_scatter_kernel, dim3(nb, P), dim3(256), 0, s, relative, metric, mask, (long long)n, max_valid_depth,
                       tile_counts, relative_out, metric_out);
  return check_launch("align_select_batch");
}

int la3d_align_select(const float* relative, const float* metric, const uint8_t* mask, int64_t n, float max_valid_depth,
                      float* relative_out, float* metric_out, int64_t* count, void* workspace, void* stream) {
  if (n < 0 || !count || !workspace || (n > 0 && (!relative || !metric || !relative_out || !metric_out))) {
    set_err("la3d_align_select: bad argument");
    return LA3D_ERR_ARG;
  }
  return la3d_align_select_batch(relative, metric, mask, 1, n, max_valid_depth, relative_out, metric_out, count, workspace, stream);
}

int la3d_align_apply(const float* relative, const uint8_t* mask, int64_t n, float coef, float intercept, float fill,
                     float* out, void* stream) {
  if (n < 0 || (n > 0 && (!relative || !out))) {
    set_err("la3d_align_apply: bad argument");
    return LA3D_ERR_ARG;
  }
  if (n == 0) return LA3D_SUCCESS;
  hipLaunchKernelGGL(align_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     relative, mask, (long long)n, coef, intercept, fill, out);
  return check_launch("align_apply_kernel");
}

int la3d_unproject_matches(const float* depth, int H, int W, const double* uv, int N, double fx, double fy, double cx,
                           double cy, int use_flip, double flip, const double* R9, const double* T3, double* out,
                           int32_t* valid, void* stream) {
  if (!depth || (!uv && N > 0) || !out || !valid || N < 0 || H <= 0 || W <= 0 || (R9 == nullptr) != (T3 == nullptr)) {
    set_err("la3d_unproject_matches: bad argument");
    return LA3D_ERR_ARG;
  }
  if (N == 0) return LA3D_SUCCESS;
  MatchParams p;
  p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.flip = flip; p.use_flip = use_flip; p.H = H; p.W = W; p.N = N;
  p.has_rt = R9 != nullptr;
  for (int i = 0; i < 9; ++i) p.R[i] = R9 ? R9[i] : 0.0;
  for (int i = 0; i < 3; ++i) p.T[i] = T3 ? T3[i] : 0.0;
  hipLaunchKernelGGL(unproject_matches_kernel, dim3((N + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), depth, uv,
                     p, out, valid);
  return check_launch("unproject_matches_kernel");
}

int la3d_project_boxes(const double* records, const double* K, int32_t k_stride, const int32_t* image_index, int B,
                       double width, double height, double* out, void* stream) {
  if ((!records && B > 0) || !K || !out || B < 0 || (k_stride != 0 && k_stride < 9)) {
    set_err("la3d_project_boxes: bad argument");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  hipLaunchKernelGGL(project_boxes_kernel, dim3((B + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), records, K,
                     k_stride, image_index, B, width, height, out);
  return check_launch("project_boxes_kernel");
}

int la3d_iou_matrix(const double* boxes_a, int na, const double* boxes_b, int nb, double* out, void* stream) {
  if (na < 0 || nb < 0 || ((!boxes_a || !boxes_b || !out) && na > 0 && nb > 0)) {
    set_err("la3d_iou_matrix: bad argument");
    return LA3D_ERR_ARG;
  }
  if (na == 0 || nb == 0) return LA3D_SUCCESS;
  const long long n = (long long)na * nb;
  hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     boxes_a, na, boxes_b, nb, out);
  return check_launch("iou_matrix_kernel");
}

int la3d_fit_points(const double* points, const int64_t* offsets, const double* ground, const int32_t* sample_idx,
                    int method, int B, double* out, int32_t* status, double* aux, void* stream) {
  if (B < 0 || (B > 0 && (!offsets || !out || !status || !points))) {
    set_err("la3d_fit_points: bad argument");
    return LA3D_ERR_ARG;
  }
  const bool small = (method & LA3D_HINT_SMALL_CLOUDS) != 0;
  method &= ~LA3D_HINT_SMALL_CLOUDS;
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
  else if (small)
    hipLaunchKernelGGL(fit_points_wave_kernel, dim3((B + NTP / 64 - 1) / (NTP / 64)), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(fit_points_kernel<false>, dim3(B), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("fit_points_kernel");
}


// ------------------------------------------------------------------------------------------
// Host-pointer single calls (round 5): the reference calls estimate_bbox once per object on a NumPy cloud
// (src/util_3dbox.py:273-278) and depth_to_points once per image on a NumPy frame (src/batch_scripts/depth.py:154).  One C call =
// upload + kernel + download on a private stream of the calling thread; the staging memory (pinned + device-mapped for the cloud,
// device scratch for the frame) belongs to the library, is per thread and per device, grows on demand and is kept.
// ------------------------------------------------------------------------------------------
namespace {
struct HostCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  unsigned char* pin = nullptr;      // pinned host block, mapped into the device's address space
  unsigned char* pin_dev = nullptr;  // ... its device address
  size_t pin_bytes = 0;
  unsigned char* dev = nullptr;      // device scratch
  size_t dev_bytes = 0;
  unsigned seq = 0;
};
thread_local HostCtx t_host;

void host_ctx_release(HostCtx& c) {
  if (c.stream) { (void)hipStreamSynchronize(c.stream); (void)hipStreamDestroy(c.stream); }
  if (c.pin) (void)hipHostFree(c.pin);
  if (c.dev) (void)hipFree(c.dev);
  c = HostCtx();
  (void)hipGetLastError();
}

int host_ctx(HostCtx** out, size_t pin_need, size_t dev_need, const char* who) {
  HostCtx& c = t_host;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: no device", who); (void)hipGetLastError(); return LA3D_ERR_HIP; }
  if (c.device != dev) {
    if (c.device >= 0) {   // the thread moved to another GPU: the old context's memory belongs to the old device
      int cur = dev;
      (void)hipSetDevice(c.device);
      host_ctx_release(c);
      (void)hipSetDevice(cur);
    }
    c.device = dev;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
      snprintf(g_err, sizeof(g_err), "%s: hipStreamCreate failed", who); (void)hipGetLastError(); c = HostCtx(); return LA3D_ERR_HIP;
    }
  }
  auto grow = [](size_t need) { size_t n = 64 * 1024; while (n < need) n *= 2; return n; };
  if (pin_need > c.pin_bytes) {
    (void)hipStreamSynchronize(c.stream);
    if (c.pin) (void)hipHostFree(c.pin);
    c.pin = nullptr; c.pin_bytes = 0;
    const size_t n = grow(pin_need);
    void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, n, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      if (h) (void)hipHostFree(h);
      snprintf(g_err, sizeof(g_err), "%s: cannot allocate %zu bytes of pinned staging memory", who, n); (void)hipGetLastError(); return LA3D_ERR_HIP;
    }
    memset(h, 0, HOSTFIT_HDR);
    c.pin = static_cast<unsigned char*>(h); c.pin_dev = static_cast<unsigned char*>(d); c.pin_bytes = n;
  }
  if (dev_need > c.dev_bytes) {
    (void)hipStreamSynchronize(c.stream);
    if (c.dev) (void)hipFree(c.dev);
    c.dev = nullptr; c.dev_bytes = 0;
    const size_t n = grow(dev_need);
    void* d = nullptr;
    if (hipMalloc(&d, n) != hipSuccess) {
      snprintf(g_err, sizeof(g_err), "%s: cannot allocate %zu bytes of device scratch", who, n); (void)hipGetLastError(); return LA3D_ERR_HIP;
    }
    c.dev = static_cast<unsigned char*>(d); c.dev_bytes = n;
  }
  *out = &c;
  return LA3D_SUCCESS;
}
}  // namespace

int la3d_estimate_bbox_host(const double* points, int64_t n, const double* ground4, int method, double* out39, double* aux4,
                            int32_t* status) {
  if (n < 0 || (n > 0 && !points) || !out39 || !status || n > (int64_t)1 << 31) {
    set_err("la3d_estimate_bbox_host: bad argument");
    return LA3D_ERR_ARG;
  }
  if (method != LA3D_METHOD_PCA && method != LA3D_METHOD_CONVEX_HULL) {
    set_err("la3d_estimate_bbox_host: unknown method");
    return LA3D_ERR_ARG;
  }
  HostCtx* c = nullptr;
  const int rc = host_ctx(&c, HOSTFIT_HDR + (size_t)n * 24 + 16, 0, "la3d_estimate_bbox_host");
  if (rc != LA3D_SUCCESS) return rc;
  if (n > 0) memcpy(c->pin + HOSTFIT_HDR, points, (size_t)n * 24);
  const int has_ground = ground4 != nullptr && ground4[0] == ground4[0];   // NULL or a NaN first entry: "ground_equ is None"
  if (has_ground) memcpy(c->pin + 32, ground4, 32);
  *reinterpret_cast<int32_t*>(c->pin + 408) = -1;
  if (method == LA3D_METHOD_PCA) {
    if (++c->seq == 0) c->seq = 1;
    volatile unsigned* done = reinterpret_cast<volatile unsigned*>(c->pin + 416);
    // la3d_fit_annotations_host shares this block and copies caller data over byte 416: a stale word there could equal this call's
    // sequence number and end the poll before the kernel has run.  (Nothing is in flight on the private stream here.)
    *done = 0;
    hipLaunchKernelGGL(fit_points_host_kernel, dim3(1), dim3(NTP), 0, c->stream, c->pin_dev, (long long)n, has_ground, c->seq);
    const int lrc = check_launch("fit_points_host_kernel");
    if (lrc != LA3D_SUCCESS) return lrc;
    // the kernel stores the sequence number last (system-scope release): poll it for a while, then fall back to the runtime's wait
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; !seen; ++spins) {
      seen = *done == c->seq;
      if (!seen && (spins & 255u) == 255u &&
          std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen && hipStreamSynchronize(c->stream) != hipSuccess) return check_launch("la3d_estimate_bbox_host");
  } else {
    long long* offs = reinterpret_cast<long long*>(c->pin);
    offs[0] = 0; offs[1] = n;
    PtsParams p;
    p.points = reinterpret_cast<const double*>(c->pin_dev + HOSTFIT_HDR); p.offsets = reinterpret_cast<const long long*>(c->pin_dev);
    p.ground = has_ground ? reinterpret_cast<const double*>(c->pin_dev + 32) : nullptr; p.sample_idx = nullptr; p.B = 1; p.method = method;
    p.out = reinterpret_cast<double*>(c->pin_dev + 64); p.status = reinterpret_cast<int*>(c->pin_dev + 408);
    p.aux = reinterpret_cast<double*>(c->pin_dev + 376);
    hipLaunchKernelGGL(fit_points_kernel<true>, dim3(1), dim3(NTP), 0, c->stream, p);
    const int lrc = check_launch("fit_points_kernel");
    if (lrc != LA3D_SUCCESS) return lrc;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return check_launch("la3d_estimate_bbox_host");
  }
  memcpy(out39, c->pin + 64, LA3D_REC * sizeof(double));
  if (aux4) memcpy(aux4, c->pin + 376, LA3D_AUX * sizeof(double));
  *status = *reinterpret_cast<const int32_t*>(c->pin + 408);
  return LA3D_SUCCESS;
}

int la3d_unproject_host(const float* depth, const double* K9, const double* Rt12, int H, int W, void* out, int out_is_f64) {
  if (!depth || !K9 || !out || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 4) {
    set_err("la3d_unproject_host: bad argument");
    return LA3D_ERR_ARG;
  }
  const size_t in_bytes = (size_t)H * W * 4, out_bytes = (size_t)H * W * 3 * (out_is_f64 ? 8 : 4);
  const size_t out_off = (in_bytes + 255) & ~(size_t)255;
  HostCtx* c = nullptr;
  int rc = host_ctx(&c, 0, out_off + out_bytes, "la3d_unproject_host");
  if (rc != LA3D_SUCCESS) return rc;
  if (hipMemcpyAsync(c->dev, depth, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return check_launch("la3d_unproject_host: upload");
  rc = la3d_unproject(reinterpret_cast<const float*>(c->dev), K9, Rt12, H, W, c->dev + out_off, out_is_f64, c->stream);
  if (rc != LA3D_SUCCESS) return rc;
  if (hipMemcpyAsync(out, c->dev + out_off, out_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return check_launch("la3d_unproject_host: download");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return check_launch("la3d_unproject_host");
  return LA3D_SUCCESS;
}

void la3d_host_release(void) { host_ctx_release(t_host); }

// ------------------------------------------------------------------------------------------
// la3d_fit_annotations_host (round 5): the reference's per-IMAGE pattern - the annotations of one image (or a few), depth plane(s)
// already resident - as ONE foreign call: every small array (run lengths / polygon parts, offsets, K, ground, area hints, image
// index) is a HOST pointer, the records come back into HOST arrays.  Inside: one copy of the inputs into the calling thread's
// pinned block, one asynchronous upload, la3d_fit_instances_ex on the private stream with the outputs pointing INTO the pinned,
// device-mapped block, a one-lane kernel that raises a flag behind it, and a poll of that flag.  (The convenience wrappers of the
// Python layer spent ~150 us per image around ~40 us of GPU work: fit_annotations 187-222 us per 8-annotation image.)
// ------------------------------------------------------------------------------------------
namespace {
__global__ void host_flag_kernel(unsigned* flag, unsigned seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
inline size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }
}  // namespace

int la3d_fit_annotations_host(const la3d_fit_args* args) {
  constexpr int32_t V1_SIZE = (int32_t)offsetof(la3d_fit_args, area_hint);
  if (!args || args->struct_size < V1_SIZE) {
    set_err("la3d_fit_annotations_host: bad struct_size");
    return LA3D_ERR_ARG;
  }
  la3d_fit_args a;
  memset(&a, 0, sizeof(a));
  memcpy(&a, args, (size_t)args->struct_size < sizeof(a) ? (size_t)args->struct_size : sizeof(a));
  const bool rle = a.rle_counts != nullptr, poly = a.poly_xy != nullptr;
  if (a.B < 0 || a.H <= 0 || a.W <= 0 || a.mask || rle == poly || (rle && !a.rle_offsets) || (poly && (!a.ring_offsets || !a.inst_rings)) ||
      !a.depth || !a.K || !a.out || !a.status || a.sample_idx || a.proj || (a.k_stride != 0 && a.k_stride < 9)) {
    set_err("la3d_fit_annotations_host: bad argument (run lengths or polygon parts, host arrays; depth on the device; no u8 planes, "
            "sample_idx or proj)");
    return LA3D_ERR_ARG;
  }
  const int B = a.B;
  if (B == 0) return LA3D_SUCCESS;
  int64_t P = 1;
  if (a.image_index) for (int i = 0; i < B; ++i) { if (a.image_index[i] < 0) { set_err("la3d_fit_annotations_host: negative image_index"); return LA3D_ERR_ARG; } if (a.image_index[i] + 1 > P) P = a.image_index[i] + 1; }
  else if (a.depth_plane_stride != 0 || a.k_stride != 0) P = B;
  const int64_t R = poly ? a.inst_rings[B] : 0;
  const int64_t T = rle ? a.rle_offsets[B] : a.ring_offsets[R];
  if (T < 0 || R < 0) { set_err("la3d_fit_annotations_host: bad offsets"); return LA3D_ERR_ARG; }
  // block layout: [0] flag | inputs (uploaded) | outputs (written by the kernels through the mapping)
  size_t off = 64;
  const size_t o_idx = off;    off += up64(a.image_index ? (size_t)B * 4 : 0);
  const size_t o_data = off;   off += up64(rle ? (size_t)(T > 0 ? T : 1) * 4 : (size_t)(T > 0 ? T : 1) * 8);
  const size_t o_off1 = off;   off += up64(rle ? (size_t)(B + 1) * 8 : (size_t)(R + 1) * 8);
  const size_t o_off2 = off;   off += up64(poly ? (size_t)(B + 1) * 8 : 0);
  const size_t o_K = off;      off += up64((size_t)(a.k_stride ? P * a.k_stride : 9) * 8);
  const size_t o_ground = off; off += up64(a.ground ? (size_t)B * 32 : 0);
  const size_t o_hint = off;   off += up64(a.area_hint ? (size_t)B * 4 : 0);
  const size_t in_end = off;
  const size_t o_out = off;    off += up64((size_t)B * LA3D_REC * 8);
  const size_t o_aux = off;    off += up64((size_t)B * LA3D_AUX * 8);
  const size_t o_status = off; off += up64((size_t)B * 4);
  const size_t o_stats = off;  off += up64(a.stats ? (size_t)B * 16 : 0);
  const size_t ws_bytes = la3d_workspace_bytes(B, a.H, a.W);
  const size_t d_ws = (in_end + 255) & ~(size_t)255;
  HostCtx* c = nullptr;
  const int rc = host_ctx(&c, off, d_ws + ws_bytes + 256, "la3d_fit_annotations_host");
  if (rc != LA3D_SUCCESS) return rc;
  unsigned char* h = c->pin;
  if (a.image_index) memcpy(h + o_idx, a.image_index, (size_t)B * 4);
  if (T > 0) memcpy(h + o_data, rle ? (const void*)a.rle_counts : (const void*)a.poly_xy, rle ? (size_t)T * 4 : (size_t)T * 8);
  memcpy(h + o_off1, rle ? (const void*)a.rle_offsets : (const void*)a.ring_offsets, rle ? (size_t)(B + 1) * 8 : (size_t)(R + 1) * 8);
  if (poly) memcpy(h + o_off2, a.inst_rings, (size_t)(B + 1) * 8);
  memcpy(h + o_K, a.K, (size_t)(a.k_stride ? P * a.k_stride : 9) * 8);
  if (a.ground) memcpy(h + o_ground, a.ground, (size_t)B * 32);
  if (a.area_hint) memcpy(h + o_hint, a.area_hint, (size_t)B * 4);
  *reinterpret_cast<volatile unsigned*>(h) = 0;   // the completion flag: la3d_estimate_bbox_host (hull) writes offsets over it
  // The depth plane(s) were produced on the CALLER's stream (a depth model's output, an upload, la3d_pad_rows ...): the whole call -
  // upload, fit, flag - is enqueued on THAT stream (args->stream; NULL = the legacy default stream), behind everything it holds, so
  // no cross-stream ordering is needed (an event record + a stream wait on the thread's private stream cost 8 us per call).  The
  // call is synchronous either way: it returns when the flag behind the fit has been raised.
  const hipStream_t ws = static_cast<hipStream_t>(a.stream);
  if (hipMemcpyAsync(c->dev + 64, h + 64, in_end - 64, hipMemcpyHostToDevice, ws) != hipSuccess) return check_launch("la3d_fit_annotations_host: upload");
  la3d_fit_args d = a;
  d.struct_size = (int32_t)sizeof(la3d_fit_args);
  d.image_index = a.image_index ? reinterpret_cast<const int32_t*>(c->dev + o_idx) : nullptr;
  if (rle) { d.rle_counts = reinterpret_cast<const int32_t*>(c->dev + o_data); d.rle_offsets = reinterpret_cast<const int64_t*>(c->dev + o_off1); }
  else { d.poly_xy = reinterpret_cast<const int32_t*>(c->dev + o_data); d.ring_offsets = reinterpret_cast<const int64_t*>(c->dev + o_off1);
         d.inst_rings = reinterpret_cast<const int64_t*>(c->dev + o_off2); }
  d.K = reinterpret_cast<const double*>(c->dev + o_K);
  d.ground = a.ground ? reinterpret_cast<const double*>(c->dev + o_ground) : nullptr;
  d.area_hint = a.area_hint ? reinterpret_cast<const int32_t*>(c->dev + o_hint) : nullptr;
  d.out = reinterpret_cast<double*>(c->pin_dev + o_out);
  d.aux = reinterpret_cast<double*>(c->pin_dev + o_aux);
  d.status = reinterpret_cast<int32_t*>(c->pin_dev + o_status);
  d.stats = a.stats ? reinterpret_cast<int32_t*>(c->pin_dev + o_stats) : nullptr;
  d.workspace = c->dev + d_ws;
  d.stream = ws;
  const int frc = la3d_fit_instances_ex(&d);
  if (frc != LA3D_SUCCESS) return frc;
  if (++c->seq == 0) c->seq = 1;
  volatile unsigned* done = reinterpret_cast<volatile unsigned*>(c->pin);
  hipLaunchKernelGGL(host_flag_kernel, dim3(1), dim3(1), 0, ws, reinterpret_cast<unsigned*>(c->pin_dev), c->seq);
  const int lrc = check_launch("host_flag_kernel");
  if (lrc != LA3D_SUCCESS) return lrc;
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; !seen; ++spins) {
    seen = *done == c->seq;
    if (!seen && (spins & 255u) == 255u &&
        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 5000) break;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (!seen && hipStreamSynchronize(ws) != hipSuccess) return check_launch("la3d_fit_annotations_host");
  memcpy(a.out, h + o_out, (size_t)B * LA3D_REC * 8);
  if (a.aux) memcpy(a.aux, h + o_aux, (size_t)B * LA3D_AUX * 8);
  memcpy(a.status, h + o_status, (size_t)B * 4);
  if (a.stats) memcpy(a.stats, h + o_stats, (size_t)B * 16);
  return LA3D_SUCCESS;
}

}  // extern "C"
