// la3d_poly.hpp — polygon segmentations -> 1-bit-per-pixel image in LDS, with the semantics of the reference's
// create_boolean_mask_from_polygon (/root/reference/src/util.py:386-400): every part (ring) of an instance is truncated to
// int32 vertices on the host and filled on its own by cv2.fillPoly(mask, [points], 1) — OpenCV's drawing.cpp for an 8-bit
// image, LINE_8, shift 0: CollectPolyEdges (each side drawn with the 8-connected LineIterator, clipped by clipLine, and turned
// into a 16.16 fixed-point scan edge) + FillEdgeCollection (even-odd pairing of the x-sorted crossings of every scanline
// y0 <= y < y1, columns x_left >> 16 .. x_right >> 16 inclusive).  The CPU restatement is oracle/poly_oracle.py (PARITY UNPINNED
// against OpenCV itself: cv2 is not installed in the build container; see that file's header for what is pinned).
//
// GPU form (poly_to_bits): the sides of a ring go through LDS 32 at a time — one thread computes a side (clipLine, scan edge),
// one wave paints its Bresenham pixels lane-parallel from the closed form  minor(i) = floor((2 i dmin + dmaj - 1) / (2 dmaj));
// every scanline (one thread each) keeps the POLY_KEEP smallest crossings in (x, side) order in registers while the sides stream by
// and then ORs the paired spans into the bit image word by word; a scanline with more crossings takes further sweeps.
// No global scratch, any number of sides / crossings.
#pragma once
#include "la3d_device.hpp"

namespace la3d {

struct alignas(8) PolyEdge {
  int y0, y1;        // y0 < y1; y0 == y1: not a scan edge (horizontal side)
  long long x, dx;   // 16.16 fixed point: x at scanline y0, increment per scanline
};

// one polygon side staged in LDS: its scan edge and the (clipped) segment Line() draws
struct alignas(8) PolySide {
  PolyEdge e;
  int sx, sy, ex, ey;   // left-to-right end points of the drawn segment; sx > ex: nothing to draw
};

constexpr int POLY_XY_SHIFT = 16;
constexpr int POLY_CHUNK = 32;                                   // sides staged per step
constexpr int POLY_MAX_JOINT = 8;                                // parts of one instance that may share the fast pass (see poly_to_bits)
constexpr int POLY_META_INTS = 48;                               // ring bounds (POLY_MAX_JOINT + 1), bounding boxes (4 each), verdict
constexpr int POLY_STAGE_BYTES = POLY_CHUNK * (int)sizeof(PolySide) + POLY_META_INTS * 4;
#ifndef LA3D_POLY_KEEP
#define LA3D_POLY_KEEP 4
#endif
// crossings a scanline keeps per sweep of the general form (more: another sweep).  Measured on second / third parts of 1024 instances
// (profiles/r02_poly_parts.txt): convex parts 8 -> 4 kept: -7...-27 % per launch (the 64-VGPR kernel spills 72 -> 31 registers in
// these loops), 20-40 % more with 2; spiky 40 / 80-vertex parts: +4...9 % with 4, +25...40 % with 2.
constexpr int POLY_KEEP = LA3D_POLY_KEEP;

// cv::clipLine(Size2l, Point2l&, Point2l&): end points are updated in place even when the line misses the image.
// OpenCV works on int64 points; with shift 0 every coordinate (int32 vertices, and every clipped value, which lies between the two
// end points) stays inside int32, and (double)(p - q) of two int32 values equals (double)p - (double)q exactly - so the same
// double products / quotients / truncations are formed from 32-bit registers (the int64 form kept ~30 more VGPRs alive inside
// the scanline loops of poly_ring_general: the fit kernel spilled).
// p + (int64)((double)(a - q) * (s1 - s0) / (r1 - r0)), the update clipLine applies to a coordinate p.  The truncated term can
// exceed int32 for end points 2^31 apart, the sum cannot (it lies between the two end points): add in double, convert once.
__device__ inline int poly_clip_add(int p, int a, int q, int s0, int s1, int r0, int r1) {
  const double t = ((double)a - (double)q) * ((double)s1 - (double)s0) / ((double)r1 - (double)r0);
  return (int)((double)p + trunc(t));
}
__device__ inline bool poly_clip_line(int width, int height, int& x1, int& y1, int& x2, int& y2) {
  const int right = width - 1, bottom = height - 1;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 = poly_clip_add(x1, a, y1, x1, x2, y1, y2);
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 = poly_clip_add(x2, a, y2, x1, x2, y1, y2);
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 = poly_clip_add(y1, a, x1, y1, y2, x1, x2);
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 = poly_clip_add(y2, a, x2, y1, y2, x1, x2);
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

// trunc(num / den) for |num| < 2^52, den != 0 (C++ truncating division): the double quotient is within one of the exact one, one
// multiply-subtract puts it right - a fraction of the registers and instructions of the inlined 64-bit integer division
__device__ inline long long poly_div_trunc(long long num, long long den) {
  const unsigned long long a = (unsigned long long)(num < 0 ? -num : num), b = (unsigned long long)(den < 0 ? -den : den);
  unsigned long long q = (unsigned long long)((double)a / (double)b);
  const long long r = (long long)(a - q * b);
  if (r < 0) --q;
  else if ((unsigned long long)r >= b) ++q;
  return ((num < 0) != (den < 0)) ? -(long long)q : (long long)q;
}

// One polygon side p0 -> p1 (one thread): CollectPolyEdges for line_type 8, shift 0, offset 0 — the segment Line() draws
// (LineIterator: 8-connected, left to right, clipped) and the scan edge.
__device__ inline PolySide poly_side(int x0i, int y0i, int x1i, int y1i, int W, int H) {
  int ax = x0i, ay = y0i, bx = x1i, by = y1i;
  const bool inside = (unsigned)x0i < (unsigned)W && (unsigned)x1i < (unsigned)W && (unsigned)y0i < (unsigned)H && (unsigned)y1i < (unsigned)H;
  bool draw = true;
  if (!inside) draw = poly_clip_line(W, H, ax, ay, bx, by);
  // (ax, ay, bx, by) are also what CollectPolyEdges sees after its own clipLine call (same function, same inputs)
  PolySide s;
  if (draw) {
    if (bx < ax) { s.sx = bx; s.sy = by; s.ex = ax; s.ey = ay; }    // start from the left end point
    else { s.sx = ax; s.sy = ay; s.ex = bx; s.ey = by; }
  } else {
    s.sx = 1; s.ex = 0; s.sy = s.ey = 0;
  }
  PolyEdge& e = s.e;
  e.y0 = e.y1 = 0; e.x = e.dx = 0;
  if (y0i == y1i) return s;                                       // horizontal sides are not swept
  long long c0x = ((long long)x0i << POLY_XY_SHIFT), c1x = ((long long)x1i << POLY_XY_SHIFT);
  int c0y = y0i, c1y = y1i;
  if (inside) {
    c0x += 1 << (POLY_XY_SHIFT - 1);
    c1x += 1 << (POLY_XY_SHIFT - 1);
  } else if (ay != by) {                                          // clipped end points, without the half
    c0x = (long long)ax << POLY_XY_SHIFT; c0y = ay;
    c1x = (long long)bx << POLY_XY_SHIFT; c1y = by;
  }
  e.dx = poly_div_trunc(c1x - c0x, (long long)c1y - (long long)c0y);   // C++ truncating division (|numerator| < 2^49)
  if (y0i < y1i) { e.y0 = y0i; e.y1 = y1i; e.x = c0x + ((long long)y0i - c0y) * e.dx; }
  else { e.y0 = y1i; e.y1 = y0i; e.x = c1x + ((long long)y1i - c1y) * e.dx; }
  return s;
}

__device__ inline void poly_set_bit(unsigned* bits, int W, int x, int y) {
  const unsigned i = (unsigned)y * (unsigned)W + (unsigned)x;
  atomicOr(&bits[i >> 5], 1u << (i & 31));
}

// the pixels of one drawn segment, lanes over the major axis: minor(i) = floor((2 i dmin + dmaj - 1) / (2 dmaj)) is the
// closed form of LineIterator's err < 0 stepping (exact halves stay on the start side); everything fits 32 bits inside a frame
__device__ inline void poly_draw(const PolySide& s, int W, unsigned* bits, int lane) {
  if (s.sx > s.ex) return;
  const int dx = s.ex - s.sx, dy = s.ey - s.sy;
  const int ady = dy < 0 ? -dy : dy, sgn = dy < 0 ? -1 : 1;
  if (ady > dx) {   // y is the major axis
    for (int i = lane; i <= ady; i += 64) {
      const int k = (int)((2u * (unsigned)i * (unsigned)dx + (unsigned)ady - 1u) / (2u * (unsigned)ady));
      poly_set_bit(bits, W, s.sx + k, s.sy + sgn * i);
    }
  } else {
    for (int i = lane; i <= dx; i += 64) {
      const int k = dx ? (int)((2u * (unsigned)i * (unsigned)ady + (unsigned)dx - 1u) / (2u * (unsigned)dx)) : 0;
      poly_set_bit(bits, W, s.sx + i, s.sy + sgn * k);
    }
  }
}

// OR the columns x1..x2 (inclusive, already clipped to the row) of row y into the bit image
__device__ inline void poly_fill_span(unsigned* bits, int W, int y, int x1, int x2) {
  const unsigned i1 = (unsigned)y * (unsigned)W + (unsigned)x1, i2 = (unsigned)y * (unsigned)W + (unsigned)x2;
  const unsigned w1 = i1 >> 5, w2 = i2 >> 5;
  const unsigned m1 = 0xffffffffu << (i1 & 31), m2 = 0xffffffffu >> (31 - (i2 & 31));
  if (w1 == w2) { atomicOr(&bits[w1], m1 & m2); return; }
  atomicOr(&bits[w1], m1);
  for (unsigned w = w1 + 1; w < w2; ++w) atomicOr(&bits[w], 0xffffffffu);
  atomicOr(&bits[w2], m2);
}

// FAST form for the first part of an instance when rows are word aligned (W % 32 == 0) and the image is still empty: no
// sorting, no pairing.  With the sorted crossings x_0 <= x_1 <= ... of a scanline and f = x >> 16, FillEdgeCollection's spans
// [f(x_2k), f(x_2k+1)] cover column c  <=>  (#crossings with f < c) is odd  OR  some crossing has f == c.  So every (side,
// scanline) crossing is handled on its own: pass 0 XORs a toggle at column f + 1 (clamped to >= 0; >= W falls outside), an
// inclusive prefix-XOR along every row turns the toggles into the parity term, pass 1 ORs in the f == c term and the
// Bresenham outline.  Work: O(crossings + H*W/32) instead of O(sides x scanlines).
// rb / nrb: the part boundaries inside the n points starting at p0 (rb[0] = 0 < rb[1] < ... rb[nrb] = n, in LDS): several parts
// whose bounding boxes are pairwise disjoint go through ONE pass - on every scanline their crossings then lie in disjoint column
// ranges, each with even parity, so the parity fill of all crossings together is the union of the parts' fills.
template <int NTH>
__device__ __forceinline__ void poly_ring_fast(const int* __restrict__ xy, long long p0, int n, const int* rb, int nrb, PolySide* stage,
                                      int* box, unsigned* bits, int H, int W, int tid, int clipW) {
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NTH / 64;
  const int ntxw = W >> 5;
  if (tid == 0) { box[0] = H; box[1] = 0; box[2] = ntxw; box[3] = 0; }   // rows [ymin, ymax) and words [wlo, whi] that hold toggles
  for (int pass = 0; pass < 2; ++pass) {
    for (int c0 = 0; c0 < n; c0 += POLY_CHUNK) {
      const int m = min(POLY_CHUNK, n - c0);
      if (pass == 0 || n > POLY_CHUNK) {                          // a single chunk stays in the stage for the second pass
        __syncthreads();                                          // the stage is free; the previous phase is complete
        if (tid < m) {
          const int i = c0 + tid;
          int lo = 0, hi = n;                                     // the part this point belongs to: the side closes inside it
          for (int j = 0; j < nrb; ++j)
            if (i >= rb[j] && i < rb[j + 1]) { lo = rb[j]; hi = rb[j + 1]; }
          const int ip = (i == lo) ? hi - 1 : i - 1;
          const PolySide sd = poly_side(xy[2 * (p0 + ip)], xy[2 * (p0 + ip) + 1], xy[2 * (p0 + i)], xy[2 * (p0 + i) + 1], clipW, H);
          stage[tid] = sd;
          if (pass == 0 && sd.e.y0 < sd.e.y1) {                   // bounding rows / words of the toggles of this side
            const int ya = max(sd.e.y0, 0), yb = min(sd.e.y1, H);
            if (ya < yb) {
              const long long fa = (sd.e.x + (long long)(ya - sd.e.y0) * sd.e.dx) >> POLY_XY_SHIFT;
              const long long fb = (sd.e.x + (long long)(yb - 1 - sd.e.y0) * sd.e.dx) >> POLY_XY_SHIFT;
              const long long lo = (fa < fb ? fa : fb) + 1, hi = (fa < fb ? fb : fa) + 1;
              atomicMin(&box[0], ya); atomicMax(&box[1], yb);
              if (lo < W) atomicMin(&box[2], (int)((lo < 0 ? 0 : lo) >> 5));
              // a toggle that falls off the right border (pos >= W) leaves the parity odd to the end of the row
              atomicMax(&box[3], (int)((hi >= W ? W - 1 : (hi < 0 ? 0 : hi)) >> 5));
            }
          }
        }
      }
      __syncthreads();
      for (int j = wave; j < m; j += NW) {                        // one wave per side, lanes over its scanlines
        if (pass == 1) poly_draw(stage[j], W, bits, lane);
        const int y0 = stage[j].e.y0, y1 = stage[j].e.y1;
        if (y0 >= y1) continue;
        const long long ex = stage[j].e.x, edx = stage[j].e.dx;
        const int yb = min(y1, H);
        for (int y = max(y0, 0) + lane; y < yb; y += 64) {
          const long long f = (ex + (long long)(y - y0) * edx) >> POLY_XY_SHIFT;   // arithmetic shift = floor
          if (pass == 0) {
            const long long pos = f < -1 ? 0 : f + 1;
            if (pos < W) atomicXor(&bits[y * ntxw + (int)(pos >> 5)], 1u << ((int)pos & 31));
          } else if (f >= 0 && f < W) {
            atomicOr(&bits[y * ntxw + (int)(f >> 5)], 1u << ((int)f & 31));
          }
        }
      }
    }
    if (pass == 0) {
      __syncthreads();
      const int ymin = box[0], ymax = box[1], wlo = box[2], whi = box[3];
      for (int y = ymin + tid; y < ymax; y += NTH) {              // inclusive prefix-XOR along the row, inside the toggle box
        unsigned carry = 0;
        unsigned* row = bits + y * ntxw;
        for (int w = wlo; w <= whi; ++w) {
          unsigned t = row[w];
          t ^= t << 1; t ^= t << 2; t ^= t << 4; t ^= t << 8; t ^= t << 16;
          t ^= carry;
          row[w] = t;
          carry = (unsigned)((int)t >> 31);                       // all ones when the parity entering the next word is odd
        }
      }
    }
  }
  __syncthreads();
}

// General form (any width, image may already hold earlier parts): every scanline (one thread each) keeps the POLY_KEEP smallest
// crossings in (x, side) order in registers while the sides stream by (compare-swap insertion, static indices; sides outside the
// wave's 64 scanlines are skipped by a scalar branch), then ORs the paired spans into the image; a scanline with more crossings
// takes further sweeps over the ring until all are consumed.
template <int NTH>
__device__ __forceinline__ void poly_ring_general(const int* __restrict__ xy, long long p0, int n, PolySide* stage, unsigned* flags,
                                         unsigned* bits, int H, int W, int tid, int clipW) {
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NTH / 64;
  constexpr long long XINF = 0x7fffffffffffffffLL;
  for (int ybase = 0; ybase < H; ybase += NTH) {
    const int y = ybase + tid;
    long long lastx = -XINF - 1;
    int lasti = -1;
    bool first = ybase == 0;    // the outline is drawn once per ring
    const int wy0 = ybase + (tid & ~63), wy1 = wy0 + 64;   // the scanlines of this wave
    while (true) {              // sweeps over the ring (uniform); one unless a scanline has more than POLY_KEEP crossings
      long long cx[POLY_KEEP];
      int ci[POLY_KEEP];
#pragma unroll
      for (int k = 0; k < POLY_KEEP; ++k) { cx[k] = XINF; ci[k] = 0x7fffffff; }
      int found = 0;
      for (int c0 = 0; c0 < n; c0 += POLY_CHUNK) {
        const int m = min(POLY_CHUNK, n - c0);
        __syncthreads();                                          // the stage is free (and the zeroed image is published)
        if (tid < m) {
          const int i = c0 + tid, ip = (i == 0) ? n - 1 : i - 1;
          stage[tid] = poly_side(xy[2 * (p0 + ip)], xy[2 * (p0 + ip) + 1], xy[2 * (p0 + i)], xy[2 * (p0 + i) + 1], clipW, H);
        }
        __syncthreads();
        if (first)
          for (int j = wave; j < m; j += NW) poly_draw(stage[j], W, bits, lane);
        if (n >= 2) {
          for (int j = 0; j < m; ++j) {
            // sides that do not reach this wave's 64 scanlines are skipped by a scalar branch
            const int y0 = __builtin_amdgcn_readfirstlane(stage[j].e.y0), y1 = __builtin_amdgcn_readfirstlane(stage[j].e.y1);
            if (y1 <= wy0 || y0 >= wy1) continue;
            if (y < H && y0 <= y && y < y1) {
              long long x = stage[j].e.x + (long long)(y - y0) * stage[j].e.dx;
              int i = c0 + j;
              if (x > lastx || (x == lastx && i > lasti)) {       // not consumed by an earlier sweep
                ++found;
#pragma unroll
                for (int k = 0; k < POLY_KEEP; ++k) {             // insertion: keeps the POLY_KEEP smallest (x, i)
                  const bool lt = x < cx[k] || (x == cx[k] && i < ci[k]);
                  const long long tx = lt ? cx[k] : x;
                  const int ti = lt ? ci[k] : i;
                  cx[k] = lt ? x : cx[k];
                  ci[k] = lt ? i : ci[k];
                  x = tx; i = ti;
                }
              }
            }
          }
        }
      }
      const int npairs = min(found, POLY_KEEP) >> 1;
#pragma unroll
      for (int k = 0; k < POLY_KEEP / 2; ++k) {
        if (k < npairs) {
          const long long c1 = cx[2 * k] >> POLY_XY_SHIFT, c2 = cx[2 * k + 1] >> POLY_XY_SHIFT;   // arithmetic shift = floor
          if (c1 < W && c2 >= 0) poly_fill_span(bits, W, y, (int)(c1 < 0 ? 0 : c1), (int)(c2 >= W ? W - 1 : c2));
        }
      }
      const bool more = found > POLY_KEEP;
      if (more) { lastx = cx[POLY_KEEP - 1]; lasti = ci[POLY_KEEP - 1]; }
      first = false;
      const unsigned long long any = __ballot(more);
      __syncthreads();                                            // every reader of flags from the previous sweep is done
      if (lane == 0) flags[wave] = any != 0 ? 1u : 0u;
      __syncthreads();
      unsigned again = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) again |= flags[w];
      if (!again) break;                                          // uniform
    }
  }
}

// All rings [r0, r1) of one instance -> bit image (zeroed here).  xy: int32 pairs; ring_off: point offsets of the rings
// (ring r = points ring_off[r] .. ring_off[r+1]).  stage: LDS, POLY_STAGE_BYTES; flags: LDS, max(NTH/64, 4) words.
// Returns this thread's share of the pixel count.
// wave-uniform 64-bit value -> SGPR pair (the ring bookkeeping below is uniform, but it is loaded through vector memory
// instructions: left alone it occupies ~10 VGPRs across the rasteriser's loops, which is what made the 64-VGPR fit kernel spill)
__device__ inline long long poly_uniform(long long v) {
  return ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

template <int NTH>
__device__ __forceinline__ int poly_to_bits(const int* __restrict__ xy, const long long* __restrict__ ring_off, long long r0, long long r1,
                                   PolySide* stage, unsigned* flags, unsigned* bits, int nwords, int H, int W, int tid, int clipW = 0) {
  // clipW (la3d_fit_args::frame_width): 0 < clipW < W, W % 32 == 0 - the image is clipW columns wide, its rows are stored W bits
  // apart.  The polygon sides are clipped to the IMAGE (clipLine changes the geometry of a side that leaves the frame); everything
  // else works on the padded rows - a span or a parity run may reach into the columns [clipW, W) - and those columns are cleared
  // before the count.
  if (clipW <= 0 || clipW > W) clipW = W;
  r0 = poly_uniform(r0); r1 = poly_uniform(r1);
  for (int i = tid; i < nwords; i += NTH) bits[i] = 0;
  int* meta = reinterpret_cast<int*>(stage + POLY_CHUNK);   // [0..8] part bounds, [9 + 4 j ..] bounding box of part j, [41] verdict
  const int nr = (int)(r1 - r0);
  long long r = r0;
  if (nr >= 1 && (W & 31) == 0) {
    // The fast form wants an empty image, so it can take only the first part - unless the parts cannot interact: up to
    // POLY_MAX_JOINT parts with pairwise disjoint vertex bounding boxes (the usual case of an object seen in several pieces) are
    // filled in one pass.  Measured, 1024 instances of 2 / 3 / 5 parts: 117 / 148 / 225 us per launch with the extra parts in the
    // general form (profiles/exp_poly_rings.py).
    int joint = 1;
    if (nr >= 2 && nr <= POLY_MAX_JOINT) {
      const int lane = tid & 63, wave = tid >> 6;
      const long long pf = ring_off[r0];
      for (int j = wave; j < nr; j += NTH / 64) {             // one wave per part: bounding box of its vertices
        const long long a = ring_off[r0 + j], b = ring_off[r0 + j + 1];
        int x0 = 0x7fffffff, x1 = -0x7fffffff - 1, y0 = 0x7fffffff, y1 = -0x7fffffff - 1;
        for (long long q = a + lane; q < b; q += 64) {
          const int x = xy[2 * q], y = xy[2 * q + 1];
          x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
        }
        x0 = wave_min_i(x0); x1 = wave_max_i(x1); y0 = wave_min_i(y0); y1 = wave_max_i(y1);
        if (lane == 0) { meta[9 + 4 * j] = x0; meta[10 + 4 * j] = x1; meta[11 + 4 * j] = y0; meta[12 + 4 * j] = y1; meta[j] = (int)(a - pf); }
      }
      if (tid == 0) meta[nr] = (int)(ring_off[r1] - pf);
      __syncthreads();
      if (tid == 0) {
        int ok = 1;
        for (int i = 0; i < nr; ++i)
          for (int j = i + 1; j < nr; ++j) {
            const bool apart = meta[10 + 4 * i] < meta[9 + 4 * j] || meta[10 + 4 * j] < meta[9 + 4 * i] ||
                               meta[12 + 4 * i] < meta[11 + 4 * j] || meta[12 + 4 * j] < meta[11 + 4 * i];   // (an empty part is apart from all)
            if (!apart) ok = 0;
          }
        meta[41] = ok;
      }
      __syncthreads();
      if (meta[41]) joint = nr;   // uniform
    }
    const long long p0 = poly_uniform(ring_off[r0]);
    const int n = __builtin_amdgcn_readfirstlane((int)(ring_off[r0 + joint] - p0));
    if (joint == 1) {
      if (tid == 0) { meta[0] = 0; meta[1] = n; }
      __syncthreads();
    }
    // (cost of the fast form: independent of the number of crossings, no per-scanline state: measured on 1024 instances of
    // 640x480 with 60 / 120-vertex non-convex parts 142 / 273 us -> 95 / 119 us per launch; convex parts of 4..31 vertices 3-6 us
    // slower than the general form)
    poly_ring_fast<NTH>(xy, p0, n, meta, joint, stage, reinterpret_cast<int*>(flags), bits, H, W, tid, clipW);
    r = r0 + joint;
  }
  for (; r < r1; ++r) {   // whatever is left is OR-ed in by the general form
    const long long p0 = poly_uniform(ring_off[r]);
    const int n = __builtin_amdgcn_readfirstlane((int)(ring_off[r + 1] - p0));
    poly_ring_general<NTH>(xy, p0, n, stage, flags, bits, H, W, tid, clipW);
  }
  __syncthreads();
  if (clipW < W) {   // uniform: clear the padding columns of every row (word-aligned rows)
    const int ntxw = W >> 5, wq = clipW >> 5, per = ntxw - wq;
    const unsigned keep = (1u << (clipW & 31)) - 1u;   // bits of word wq that belong to the image (0: none)
    for (int i = tid; i < H * per; i += NTH) {
      const int row = i / per, k = i - row * per;
      unsigned* w = bits + row * ntxw + wq + k;
      *w = k == 0 ? (*w & keep) : 0u;
    }
    __syncthreads();
  }
  int nm = 0;
  for (int i = tid; i < nwords; i += NTH) nm += __popc(bits[i]);
  return nm;
}

// The four quantities of the reference's instance filter (src/util.py:291-335) from a bit image in LDS; 256 threads.
// red: LDS, 16 ints.  Results valid in thread 0.
__device__ inline void bits_stats_256(const unsigned* bits, int H, int W, int boundary, int* rowcnt, int* red, int tid, int* out4) {
  const int lane = tid & 63, wave = tid >> 6;
  const int bc = min(boundary, W), br = min(boundary, H);
  int area = 0, edge = 0;
  for (int r = tid; r < H; r += 256) {
    int cnt = 0, e = 0;
    const unsigned base = (unsigned)r * (unsigned)W;
    for (int c0 = 0; c0 < W; c0 += 32) {       // 32 columns at a time (unaligned rows: assemble the word from two)
      const unsigned i = base + c0, wi = i >> 5, sh = i & 31;
      unsigned w = bits[wi] >> sh;
      if (sh && c0 + (32 - (int)sh) < W) w |= bits[wi + 1] << (32 - sh);
      const int valid = min(32, W - c0);
      if (valid < 32) w &= (1u << valid) - 1u;
      cnt += __popc(w);
      // columns c0..c0+31 inside the left strip [0, bc) or the right strip [W-bc, W)
      const int nlo = min(max(bc - c0, 0), 32), fhi = min(max(W - bc - c0, 0), 32);
      e += __popc(nlo >= 32 ? w : (w & ((1u << nlo) - 1u))) + __popc(fhi >= 32 ? 0u : (w & ~((1u << fhi) - 1u)));
    }
    rowcnt[r] = cnt;
    area += cnt;
    edge += e + cnt * ((r < br ? 1 : 0) + (r >= H - br ? 1 : 0));
  }
  area = wave_sum_i(area);
  edge = wave_sum_i(edge);
  if (lane == 0) { red[12 + wave] = area; red[16 + wave] = edge; }
  __syncthreads();
  int rows = 0, first = H, last = -1;
  for (int r = tid; r < H; r += 256)
    if (rowcnt[r] != 0) { rows += 1; first = min(first, r); last = max(last, r); }
  rows = wave_sum_i(rows);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { first = min(first, __shfl_xor(first, o)); last = max(last, __shfl_xor(last, o)); }
  if (lane == 0) { red[wave] = rows; red[4 + wave] = first; red[8 + wave] = last; }
  __syncthreads();
  if (tid == 0) {
    int rw = 0, f = H, l = -1, a = 0, ed = 0;
    for (int w = 0; w < 4; ++w) { rw += red[w]; f = min(f, red[4 + w]); l = max(l, red[8 + w]); a += red[12 + w]; ed += red[16 + w]; }
    out4[0] = a; out4[1] = rw; out4[2] = (l >= f) ? l - f + 1 : 0; out4[3] = ed;
  }
}

}  // namespace la3d
