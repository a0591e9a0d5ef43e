// Throughput of the fp64 / int VALU instructions the box-fit kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
// Each kernel issues N independent-chain instructions per wave; 8 waves/SIMD resident so the pipe is
// saturated; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define REP 256
#define CHAINS 8

#define KERNEL(name, DECL, BODY)                                                          \
  __global__ __launch_bounds__(512) void name(double* out, int iters) {                   \
    DECL;                                                                                 \
    for (int it = 0; it < iters; ++it) {                                                  \
      _Pragma("unroll") for (int r = 0; r < REP / CHAINS; ++r) { BODY; }                  \
    }                                                                                     \
    double s = 0;                                                                         \
    for (int c = 0; c < CHAINS; ++c) s += (double)a[c];                                   \
    if (s == 123.456) out[0] = s;                                                         \
  }

#define D8 double a[CHAINS]; double b = threadIdx.x * 1e-9 + 1.0, cc = 0.999; for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x
#define F8 float a[CHAINS]; float b = threadIdx.x * 1e-6f + 1.0f, cc = 0.999f; for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x
#define I8 unsigned a[CHAINS]; unsigned b = threadIdx.x | 1; for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x

#define ASM2(op, T) _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(op " %0, %0, %1" : "+v"(a[c]) : "v"(b))
#define ASM3I(op) _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(op " %0, %0, 3, 1" : "+v"(a[c]))
#define ASM3(op) _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(op " %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(cc))

KERNEL(k_fma_f64, D8, ASM3("v_fma_f64"))
KERNEL(k_add_f64, D8, ASM2("v_add_f64", double))
KERNEL(k_mul_f64, D8, ASM2("v_mul_f64", double))
KERNEL(k_min_f64, D8, ASM2("v_min_f64", double))
KERNEL(k_max_f64, D8, ASM2("v_max_f64", double))
KERNEL(k_fma_f32, F8, ASM3("v_fma_f32"))
KERNEL(k_add_u32, I8, ASM2("v_add_u32", unsigned))
KERNEL(k_and_b32, I8, ASM2("v_and_b32", unsigned))
KERNEL(k_mul_lo_u32, I8, ASM2("v_mul_lo_u32", unsigned))

__global__ __launch_bounds__(512) void k_cvt_f64_f32(double* out, int iters) {
  double a[CHAINS]; float f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { f[c] = c + threadIdx.x; a[c] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
    }
  }
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(512) void k_cvt_f64_i32(double* out, int iters) {
  double a[CHAINS]; int f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { f[c] = c + threadIdx.x; a[c] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
    }
  }
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(512) void k_cndmask(double* out, int iters) {
  unsigned a[CHAINS]; unsigned b = threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b));
    }
  }
  unsigned s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123456789u) out[0] = s;
}
__global__ __launch_bounds__(512) void k_cmp_class(double* out, int iters) {
  float a[CHAINS]; unsigned acc = 0;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[c]), "v"(0x1f8) : "vcc");
    }
  }
  if (acc == 123456789u) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_cmp_lt_f64(double* out, int iters) {
  double a[CHAINS]; double b = threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc");
    }
  }
  if (b == 123456789.0) out[0] = b;
}

__global__ __launch_bounds__(512) void k_cndmask_e64(double* out, int iters) {
  unsigned a[CHAINS]; unsigned b = threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[c]) : "v"(b) : "s10", "s11");
    }
  }
  unsigned s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123456789u) out[0] = s;
}
__global__ __launch_bounds__(512) void k_cmp_cndmask_pair(double* out, int iters) {
  unsigned a[CHAINS]; unsigned b = threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS / 2; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b) : "vcc");
    }
  }
  unsigned s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123456789u) out[0] = s;
}
__global__ __launch_bounds__(512) void k_hip_select(double* out, int iters) {
  unsigned a[CHAINS]; unsigned b = threadIdx.x * 2654435761u;
  for (int c = 0; c < CHAINS; ++c) a[c] = c * 7919u + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS / 2; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) { a[c] = (a[c] < b) ? a[c] + 12345u : a[c] ^ 777u; asm volatile("" : "+v"(a[c])); }
    }
  }
  unsigned s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 123456789u) out[0] = s;
}
KERNEL(k_bfe_i32, I8, ASM3I("v_bfe_i32"))
KERNEL(k_ashr_i32, I8, ASM2("v_ashrrev_i32", unsigned))
KERNEL(k_or_b32, I8, ASM2("v_or_b32", unsigned))
KERNEL(k_pk_fma_f32, D8, ASM3("v_pk_fma_f32"))
KERNEL(k_pk_add_f32, D8, ASM2("v_pk_add_f32", double))

typedef void (*kern_t)(double*, int);

int main() {
  double* out; hipMalloc(&out, 64);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  struct { const char* name; kern_t k; } ks[] = {
    {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_min_f64", k_min_f64},
    {"v_max_f64", k_max_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_i32", k_cvt_f64_i32},
    {"v_cmp_lt_f64", k_cmp_lt_f64}, {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
    {"v_mul_lo_u32", k_mul_lo_u32}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_class_f32", k_cmp_class},
    {"v_cndmask_b32_e64 sgpr", k_cndmask_e64}, {"v_cmp+v_cndmask (2 instr)", k_cmp_cndmask_pair}, {"hip select (cmp+add+xor+cnd)", k_hip_select},
    {"v_bfe_i32", k_bfe_i32}, {"v_ashrrev_i32", k_ashr_i32}, {"v_or_b32", k_or_b32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_add_f32", k_pk_add_f32},
  };
  const int iters = 200;
  const int blocks = cus * 4;  // 4 x 512 threads per CU = 8 waves/SIMD
  printf("%d CUs, clock %d kHz\n", cus, prop.clockRate);
  for (auto& e : ks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(512), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves x iters x REP instructions
    const double insts_per_simd = 8.0 * iters * REP;
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
    printf("%-30s %8.3f ms  -> %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", e.name, ms, cyc);
  }
  return 0;
}
