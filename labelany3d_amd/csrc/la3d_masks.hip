// la3d_masks.hip - whole-frame depth_to_points (src/util.py:52-75), row padding, and the mask side of the path: decode and the
// filter statistics of u8 planes, COCO run lengths and polygon parts (src/util.py:291-415) as kernels of their own (the fit engines decode
// inside the fit launch).  Split out of la3d_aux.hip in round 6.
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

// host-side 3x3 inverse (same elimination as the device-side inv3)
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

}  // extern "C"
