// la3d_consumers.hip - the steps around the path: box consumers (src/tools/combine_results.py:105-124, :238-252), the masked
// depth-ratio median and the depth-alignment selection / scatter (src/util.py:464-494, src/batch_scripts/depth.py:52-92), the matcher's
// unprojection (src/matching/matcher.py:70-91).  Split out of la3d_aux.hip in round 6.
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
  // LinearRegression.predict on float32: X @ coef_.T (one float32 product, rounded) + intercept_ (rounded again).  The two roundings
  // must survive the compiler: `__fadd_rn(__fmul_rn(..))` are plain operations for hipcc and were contracted into one fma - a last-bit
  // difference for a non-zero intercept (the reference fits without one, depth.py:66, so its own calls never saw it; found by
  // profiles/r06/fuzz_aux.py, round 6)
  float y;
  {
#pragma clang fp contract(off)
    const float prod = r * coef;
    y = prod + intercept;
  }
  out[i] = sel ? y : fill;
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

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

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

}  // extern "C"
