// Issue rate of fp64 VALU instructions as a function of waves per SIMD and independent dependency chains per wave (gfx950):
// how much of the SIMD a wave can use when its accumulators form few dependent chains (the box-fit passes: 5 sums / 6 extents).
//   hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency && ./valu_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 512
template <int CHAINS, int OP>
__global__ void k(double* out, int iters) {
  double a[CHAINS];
  double b = threadIdx.x * 1e-9 + 1.0, cc = 0.999;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(cc));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        if (OP == 2) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
      }
    }
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123.456) out[0] = s;
}

template <int CHAINS, int OP>
void run(const char* name, int waves_per_simd, double* out) {
  const int iters = 200;
  const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;   // one workgroup per CU
  const int wgs = 256 * (64 * 4 * waves_per_simd / threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(wgs), dim3(threads), 0, 0, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(wgs), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * REP;
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("%-10s waves/SIMD %d chains %d: %7.2f cycles per instruction per WAVE, %6.2f per SIMD-slot\n", name, waves_per_simd, CHAINS,
         cyc / instr_per_wave, cyc / instr_per_wave / waves_per_simd);
}

int main() {
  double* out;
  hipMalloc(&out, 8);
  for (int w : {1, 2, 4, 8}) {
    run<1, 0>("fma_f64", w, out); run<2, 0>("fma_f64", w, out); run<4, 0>("fma_f64", w, out); run<8, 0>("fma_f64", w, out);
  }
  for (int w : {1, 2, 4}) { run<1, 1>("add_f64", w, out); run<4, 1>("add_f64", w, out); run<1, 2>("min_f64", w, out); run<6, 2>("min_f64", w, out); }
  return 0;
}
