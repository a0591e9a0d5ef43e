// la3d_points.hip - explicit point clouds: estimate_bbox as the reference calls it (500 mesh samples per object, src/util_3dbox.py:273-278;
// PCA and convex-hull yaw, :181-224) for batches of clouds (la3d_fit_points), and the HOST-pointer single calls of the reference's own calling
// pattern: la3d_estimate_bbox_host, la3d_unproject_host, la3d_fit_annotations_host (one image's annotations -> records, depth resident).
// Split out of la3d_aux.hip in round 6; shared device code in la3d_device.hpp / la3d_poly.hpp.
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

constexpr int HULL_MAX = 2048;  // points the convex-hull method holds in LDS (the reference feeds it <= 500, :123; round 6: 512 -> 2048,
                                // 24 bytes of LDS per point: coordinates, candidate list, survivor flags, chain stacks)
constexpr int HULL_SMALL = 512; // ... and the form for calls that promise at most 512 valid rows per cloud (LA3D_HINT_HULL_512, a sample_idx
                                // array, the scalar drop-in's <= 500 points): 12 KB of LDS instead of 48 - eight workgroups per CU instead
                                // of three (the chain is one lane's serial work: 7.4 M vs 5.7 M clouds/s on 500-point batches)

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
  int redo;         // the raw sums are ill-conditioned (axis_from_sums): a second moments pass about pivot[]
  double pivot[2];
};

// LDS of the convex-hull method (separate struct: only the hull instantiation pays for it)
template <int HCAP>
struct alignas(16) SharedHullT {
  double x[HCAP], z[HCAP];              // valid (x', z') footprint, sorted lexicographically
  unsigned short cand[HCAP];            // current candidates of the chain, in sorted order
  unsigned short flag[HCAP];            // survivor flags by point
  unsigned short hull[2 * HCAP + 2];
  double best_area[NTP / 64], best_yaw[NTP / 64];   // per wave: the first strict minimum among its edges ...
  int best_edge[NTP / 64];                          // ... and that edge's index (ties across waves go to the smaller index)
};

// One pass of Andrew's monotone chain: visits cnt entries of the candidate list cl starting at position q0 in direction dq, pushes
// point indices on the stack S (k0 entries on entry; a pop needs at least t), returns the stack size.  The coordinates of the two
// stack tops are carried in registers, so a step that pops nothing waits for no dependent LDS read.  The turn test is the textbook
// cross(o, a, b) = (xa - xo)(zb - zo) - (za - zo)(xb - xo) <= 0 -> pop.
template <typename SharedHull>
__device__ inline int chain_pass(const SharedHull* hs, const unsigned short* cl, int q0, int dq, int cnt, unsigned short* S, int k0, int t) {
  int k = k0;
  double ox = 0, oz = 0, ax = 0, az = 0;
  if (k >= 1) { const int a = S[k - 1]; ax = hs->x[a]; az = hs->z[a]; }
  if (k >= 2) { const int o = S[k - 2]; ox = hs->x[o]; oz = hs->z[o]; }
  for (int c = 0, q = q0; c < cnt; ++c, q += dq) {
    const int i = cl[q];
    const double px = hs->x[i], pz = hs->z[i];
    while (k >= t) {
      // both products ROUNDED (no contraction into an fma, which keeps one product exact: for a point that repeats the stack top -
      // the reference's subsample draws with replacement, src/util_3dbox.py:124 - the two products are the same two factors and
      // the difference must be exactly zero, so that the repeat is popped; fused, the difference was the rounding error of one
      // product, of either sign, and a repeated hull vertex could stay: a zero-length edge, i.e. a candidate yaw of 0 the
      // reference never tries.  Found by profiles/r06/fuzz_points.py (round 6): every batched call with sample_idx and
      // method = convex_hull was exposed, the scalar drop-in too)
      double cr;
      {
#pragma clang fp contract(off)
        const double t0 = (ax - ox) * (pz - oz), t1 = (az - oz) * (px - ox);
        cr = t0 - t1;
      }
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
template <typename SharedHull>
__device__ inline bool hull_yaw(SharedHull* hs, SharedP* sh, int tid, double* yaw_out) {
  const int n = sh->nvalid;
  // pad to a power of two for the bitonic network: the smallest one that holds the cloud (512 for the reference's 500 points)
  int P2 = 64;
  while (P2 < n) P2 <<= 1;                                             // uniform; n <= HULL_MAX
  for (int i = n + tid; i < P2; i += NTP) { hs->x[i] = INFINITY; hs->z[i] = INFINITY; }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += NTP) {
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
  unsigned short* cl = hs->cand;   // current candidates in sorted order
  for (int i = tid; i < n; i += NTP) cl[i] = (unsigned short)i;
  int m = n;
  for (int level = 0; level < 2; ++level) {
    const int nch = level == 0 ? 16 : 4;
    if (m <= 4 * nch) continue;                                      // uniform
    for (int i = tid; i < n; i += NTP) hs->flag[i] = 0;              // survivor flags by point
    __syncthreads();
    if (tid < nch) {
      const int lo = (int)((long long)m * tid / nch), hi = (int)((long long)m * (tid + 1) / nch);
      unsigned short* S = hs->hull + lo;                             // this lane's stack: as many slots as its chunk has entries
      for (int pass = 0; pass < 2; ++pass) {                         // lower hull left -> right, then upper hull right -> left
        const int k = chain_pass(hs, cl, pass == 0 ? lo : hi - 1, pass == 0 ? 1 : -1, hi - lo, S, 0, 2);
        for (int q = 0; q < k; ++q) hs->flag[S[q]] = 1;
      }
    }
    __syncthreads();
    if (tid < 64) {                                                  // in-place compaction of the survivors, ascending (one wave:
      int base = 0;                                                  // a block's reads precede its writes, and it writes behind itself)
      for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + tid;
        const unsigned short id = i < m ? cl[i] : (unsigned short)0;
        const bool on = i < m && hs->flag[id] != 0;
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
  // one hull edge per wave at a time, lanes over the points (min / max are order independent: the areas are those of a serial sweep);
  // every wave keeps the FIRST strict minimum among its edges (e = wave, wave + 8, ... ascending), thread 0 then takes the smallest
  // area over the waves, ties to the smaller edge index: the first strict minimum of the serial sweep (:216), with no per-edge array
  const int lane = tid & 63, wave = tid >> 6;
  double wbest = INFINITY, wyaw = 0.0;
  int wedge = 0x7fffffff;
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
    const double area = (xhi - xlo) * (zhi - zlo);
    if (area < wbest) { wbest = area; wyaw = yaw; wedge = e; }   // (every lane holds the wave's values)
  }
  if (lane == 0) { hs->best_area[wave] = wbest; hs->best_yaw[wave] = wyaw; hs->best_edge[wave] = wedge; }
  __syncthreads();
  if (tid == 0) {
    double best = INFINITY, by = 0.0;
    int be = 0x7fffffff;
    for (int w = 0; w < NTP / 64; ++w) {
      const double a = hs->best_area[w];
      if (a < best || (a == best && hs->best_edge[w] < be)) { best = a; by = hs->best_yaw[w]; be = hs->best_edge[w]; }
    }
    hs->best_yaw[0] = by;
  }
  __syncthreads();
  *yaw_out = hs->best_yaw[0];
  return true;
}

template <bool HULL, int HCAP> struct HullStore {};
template <int HCAP> struct HullStore<true, HCAP> { SharedHullT<HCAP> h; };

template <bool HULL, int HCAP = HULL_SMALL>
__global__ __launch_bounds__(NTP) void fit_points_kernel(const PtsParams p) {
  __shared__ SharedP sh;
  __shared__ HullStore<HULL, HCAP> hstore;
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
        if (slot < HCAP) { hstore.h.x[slot] = x; hstore.h.z[slot] = z; }
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
    // ill-conditioned raw sums (axis_from_sums: a cloud whose footprint is far thinner than its distance from the origin of its
    // frame): the moments once more about the mean, below
    const bool ill = st == LA3D_BOX_OK && axis_from_sums((double)nn, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap);
    if (HULL && st == LA3D_BOX_OK && nn > HCAP) st = LA3D_BOX_UNSUPPORTED;
    sh.redo = (ill && st == LA3D_BOX_OK) ? 1 : 0;
    sh.pivot[0] = s[0] / (double)nn; sh.pivot[1] = s[1] / (double)nn;
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
  if (sh.redo) {   // uniform, rare
    const double px0 = sh.pivot[0], pz0 = sh.pivot[1];
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (long long i = tid; i < m; i += NTP) {
      long long row = i;
      if (sampled) {
        long long r = sidx[i];
        row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
      }
      const double* q = p.points + (off + row) * 3;
      const double a = q[0], b = q[1], cc = q[2];
      const double x = a * R00 + b * R10 + cc * R20, y = a * R01 + b * R11 + cc * R21, z = a * R02 + b * R12 + cc * R22;
      if (!(x != x || y != y || z != z)) {
        const double dx = x - px0, dz = z - pz0;
        t0 += dx; t1 += dz; t2 = fma(dx, dx, t2); t3 = fma(dx, dz, t3); t4 = fma(dz, dz, t4);
      }
    }
    __syncthreads();   // (thread 0 has read part[] of the first pass)
    {
      const double r0 = wave_sum(t0), r1 = wave_sum(t1), r2 = wave_sum(t2), r3 = wave_sum(t3), r4 = wave_sum(t4);
      if (lane == 0) { double* pp = sh.part[wave]; pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3; pp[4] = r4; }
    }
    __syncthreads();
    if (tid == 0) {
      double s[5] = {0, 0, 0, 0, 0};
      for (int w = 0; w < NWAVEP; ++w)
        for (int k = 0; k < 5; ++k) s[k] += sh.part[w][k];
      double cy = NAN, sy = NAN, gap = NAN;
      if (axis_from_sums((double)sh.nvalid, s[0], s[1], s[2], s[3], s[4], &cy, &sy, &gap)) gap = 0.0;   // no spread at all: unresolved
      sh.cyaw = cy; sh.syaw = sy;
      if (p.aux) {
        double* a = p.aux + (long long)c * LA3D_AUX;
        a[0] = atan2(sy, cy); a[3] = gap;
      }
    }
    __syncthreads();
  }
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
  if (st == LA3D_BOX_OK && axis_from_sums((double)nn, s0, s1, s2, s3, s4, &cy, &sy, &gap)) {   // wave-uniform, rare (see fit_points_kernel):
    const double px0 = s0 / (double)nn, pz0 = s1 / (double)nn;                                 // the moments once more about the mean
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (long long i = lane; i < m; i += 64) {
      long long row = i;
      if (sampled) {
        const long long r = sidx[i];
        row = r < 0 ? 0 : (r >= n_in ? n_in - 1 : r);
      }
      const double* q = pts + row * 3;
      const double a = q[0], b = q[1], cc = q[2];
      const double x = a * Rg[0] + b * Rg[3] + cc * Rg[6], y = a * Rg[1] + b * Rg[4] + cc * Rg[7], z = a * Rg[2] + b * Rg[5] + cc * Rg[8];
      if (!(x != x || y != y || z != z)) {
        const double dx = x - px0, dz = z - pz0;
        t0 += dx; t1 += dz; t2 = fma(dx, dx, t2); t3 = fma(dx, dz, t3); t4 = fma(dz, dz, t4);
      }
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3); t4 = wave_sum(t4);
    if (axis_from_sums((double)nn, t0, t1, t2, t3, t4, &cy, &sy, &gap)) gap = 0.0;   // no spread at all: unresolved
  }
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

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int la3d_fit_points(const double* points, const int64_t* offsets, const double* ground, const int32_t* sample_idx,
                    int method, int B, double* out, int32_t* status, double* aux, void* stream) {
  if (B < 0 || (B > 0 && (!offsets || !out || !status || !points))) {
    set_err("la3d_fit_points: bad argument");
    return LA3D_ERR_ARG;
  }
  const bool small = (method & LA3D_HINT_SMALL_CLOUDS) != 0;
  const bool hull512 = (method & LA3D_HINT_HULL_512) != 0 || sample_idx != nullptr;   // (sampled clouds hold 500 rows)
  method &= ~(LA3D_HINT_SMALL_CLOUDS | LA3D_HINT_HULL_512);
  if (method != LA3D_METHOD_PCA && method != LA3D_METHOD_CONVEX_HULL) {
    set_err("la3d_fit_points: unknown method");
    return LA3D_ERR_ARG;
  }
  if (B == 0) return LA3D_SUCCESS;
  PtsParams p;
  p.points = points; p.offsets = reinterpret_cast<const long long*>(offsets); p.ground = ground;
  p.sample_idx = sample_idx; p.B = B; p.method = method; p.out = out; p.status = status; p.aux = aux;
  if (method == LA3D_METHOD_CONVEX_HULL && hull512)
    hipLaunchKernelGGL((fit_points_kernel<true, HULL_SMALL>), dim3(B), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
  else if (method == LA3D_METHOD_CONVEX_HULL)
    hipLaunchKernelGGL((fit_points_kernel<true, HULL_MAX>), dim3(B), dim3(NTP), 0, static_cast<hipStream_t>(stream), p);
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
    if (n <= HULL_SMALL) hipLaunchKernelGGL((fit_points_kernel<true, HULL_SMALL>), dim3(1), dim3(NTP), 0, c->stream, p);
    else hipLaunchKernelGGL((fit_points_kernel<true, HULL_MAX>), dim3(1), dim3(NTP), 0, c->stream, p);
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
