// Where do the workgroups of a 1024 x 512-thread launch with 40 KB LDS land?  (speed-only knowledge: used to
// decide whether a size-sorted launch order can balance the per-CU load of the instance engine.)
//   hipcc --offload-arch=gfx950 -O3 wg_census.hip -o wg_census && ./wg_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void census(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = xcc;
  }
  // keep the workgroup alive so that all of them are resident at once
  volatile unsigned* p = (volatile unsigned*)smem;
  for (int i = 0; i < spin; ++i) p[threadIdx.x] = i;
}

int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 8);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(census, dim3(B), dim3(512), 40 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 2);
    hipMemcpy(h.data(), d, B * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu2blocks;
    for (int b = 0; b < B; ++b) {
      const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      cu2blocks[key].push_back(b);
    }
    printf("rep %d: %zu distinct CUs; blocks per CU:", rep, cu2blocks.size());
    std::map<size_t, int> hist;
    for (auto& kv : cu2blocks) hist[kv.second.size()]++;
    for (auto& kv : hist) printf(" %zu blocks x %d CUs;", kv.first, kv.second);
    printf("\n");
    int shown = 0;
    for (auto& kv : cu2blocks) {
      if (shown++ >= 6) break;
      printf("  cu key %05x:", kv.first);
      for (int b : kv.second) printf(" %d", b);
      printf("\n");
    }
    // is block b -> XCC b % 8 ?
    int okx = 0;
    for (int b = 0; b < B; ++b) okx += ((h[2 * b + 1] & 0xf) == (unsigned)(b % 8));
    printf("  blocks with xcc == b %% 8: %d / %d\n", okx, B);
  }
  return 0;
}
