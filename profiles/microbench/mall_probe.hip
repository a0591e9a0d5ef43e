// Does the 256 MB Infinity Cache (MALL) keep a 131 MB working set (the depth tiles of a 1024-instance launch) across a
// 315 MB stream (the u8 mask planes)?  And how fast is a MALL-served re-read?  Decides whether pass B of the plain build
// could be served on-die.   hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 non-temporal
__global__ __launch_bounds__(512) void reader(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 a, b, c, d;
    if (MODE == 1) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride); c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride); }
    else { a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride]; }
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n16; i += stride) acc ^= p[i].x;
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

static float run(int mode, const void* p, size_t bytes, unsigned* sink, hipStream_t s) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  if (mode == 1) hipLaunchKernelGGL(reader<1>, dim3(2048), dim3(512), 0, s, (const u32x4*)p, bytes / 16, sink);
  else hipLaunchKernelGGL(reader<0>, dim3(2048), dim3(512), 0, s, (const u32x4*)p, bytes / 16, sink);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  const size_t MB = 1 << 20;
  const size_t szA = 128 * MB, szB = 320 * MB, szC = 512 * MB;
  char *A, *B, *C; unsigned* sink;
  hipMalloc(&A, szA); hipMalloc(&B, szB); hipMalloc(&C, szC); hipMalloc(&sink, 4096 * 4);
  hipMemset(A, 1, szA); hipMemset(B, 2, szB); hipMemset(C, 3, szC);
  hipStream_t s; hipStreamCreate(&s);
  auto gbs = [](size_t b, float ms) { return b / (ms * 1e-3) / 1e9; };
  for (int rep = 0; rep < 3; ++rep) {
    run(0, C, szC, sink, s);                                   // flush: 512 MB of something else
    float cold = run(0, A, szA, sink, s);
    float hot = run(0, A, szA, sink, s);
    float hot2 = run(0, A, szA, sink, s);
    printf("rep %d: A 128 MB cold %.1f GB/s | re-read %.1f | again %.1f\n", rep, gbs(szA, cold), gbs(szA, hot), gbs(szA, hot2));
    for (int modeB = 0; modeB < 2; ++modeB) {
      run(0, C, szC, sink, s);
      run(0, A, szA, sink, s);
      float tb = run(modeB, B, szB, sink, s);
      float after = run(0, A, szA, sink, s);
      printf("        A, then B 320 MB %s (%.1f GB/s), then A again: %.1f GB/s\n", modeB ? "non-temporal" : "plain", gbs(szB, tb), gbs(szA, after));
    }
    for (size_t small : {16 * MB, 32 * MB, 64 * MB}) {        // smaller sets: L2 (32 MB aggregate) vs MALL
      run(0, C, szC, sink, s);
      run(0, A, small, sink, s);
      float h = run(0, A, small, sink, s);
      printf("        %zu MB re-read: %.1f GB/s\n", small / MB, gbs(small, h));
    }
  }
  return 0;
}
