// Driving the C-ABI of libla3d.so from a plain host program: no Python, no torch.
//   hipcc -O2 -I include examples/fit_from_c.cpp -L labelany3d_amd/lib -lla3d -Wl,-rpath,'$ORIGIN' -o labelany3d_amd/lib/fit_from_c
//   labelany3d_amd/lib/fit_from_c <B> <H> <W> <seed>
// Builds B synthetic instances (private depth planes, one rectangle each, a ground plane per instance) with a small
// LCG, runs la3d_fit_instances on HIP buffers and prints status + the 39 doubles of every record in hex-exact form
// ("%a"), then repeats the fit through la3d_fit_instances_ex (C struct argument block) with the 2-D boxes of the records ("P"
// lines).  tests/test_gpu_cabi.py regenerates the same inputs in NumPy and checks the printed records against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "la3d.h"

static uint32_t lcg_state;
static uint32_t lcg() { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 8; }   // 24 random bits
static double unit() { return (double)lcg() / 16777216.0; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s B H W seed\n", argv[0]); return 1; }
  const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]);
  lcg_state = (uint32_t)atoi(argv[4]);
  const size_t HW = (size_t)H * W;
  std::vector<float> depth(B * HW);
  std::vector<uint8_t> mask(B * HW, 0);
  std::vector<double> ground(B * 4), K = {0.8 * W, 0, 0.5 * W, 0, 0.8 * W, 0.5 * H, 0, 0, 1};
  for (int i = 0; i < B; ++i) {
    for (size_t p = 0; p < HW; ++p) depth[i * HW + p] = (float)(0.5 + 9.5 * unit());
    const int h = 1 + (int)(lcg() % (uint32_t)H), w = 1 + (int)(lcg() % (uint32_t)W);
    const int r0 = (int)(lcg() % (uint32_t)(H - h + 1)), c0 = (int)(lcg() % (uint32_t)(W - w + 1));
    for (int r = r0; r < r0 + h; ++r)
      for (int c = c0; c < c0 + w; ++c) mask[i * HW + (size_t)r * W + c] = (uint8_t)(1 + i % 250);
    ground[i * 4 + 0] = 0.02 + 0.1 * (unit() - 0.5); ground[i * 4 + 1] = -0.98 + 0.1 * (unit() - 0.5);
    ground[i * 4 + 2] = 0.1 + 0.1 * (unit() - 0.5);  ground[i * 4 + 3] = 1.5;
  }
  if (la3d_version() != LA3D_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 3; }
  float* d_depth; uint8_t* d_mask; double *d_K, *d_ground, *d_out, *d_aux; int32_t* d_status; void* d_ws;
  const size_t ws = la3d_workspace_bytes(B, H, W);
  HIPCHK(hipMalloc(&d_depth, depth.size() * 4)); HIPCHK(hipMalloc(&d_mask, mask.size()));
  HIPCHK(hipMalloc(&d_K, 72)); HIPCHK(hipMalloc(&d_ground, ground.size() * 8));
  HIPCHK(hipMalloc(&d_out, (size_t)B * LA3D_REC * 8)); HIPCHK(hipMalloc(&d_aux, (size_t)B * LA3D_AUX * 8));
  HIPCHK(hipMalloc(&d_status, (size_t)B * 4)); HIPCHK(hipMalloc(&d_ws, ws ? ws : 8));
  HIPCHK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_K, K.data(), 72, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_ground, ground.data(), ground.size() * 8, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIPCHK(hipStreamCreate(&stream));
  const int rc = la3d_fit_instances(d_depth, (int64_t)HW, nullptr, d_mask, d_K, 0, d_ground, nullptr, B, H, W, d_out, d_status,
                                    d_aux, d_ws, stream);
  if (rc != LA3D_SUCCESS) { fprintf(stderr, "la3d_fit_instances: %d %s\n", rc, la3d_last_error()); return 4; }
  HIPCHK(hipStreamSynchronize(stream));
  std::vector<double> out((size_t)B * LA3D_REC);
  std::vector<int32_t> status(B);
  HIPCHK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(status.data(), d_status, (size_t)B * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < B; ++i) {
    printf("%d", status[i]);
    for (int k = 0; k < LA3D_REC; ++k) printf(" %a", out[(size_t)i * LA3D_REC + k]);
    printf("\n");
  }
  // the same fit through the extensible entry point, with the records' 2-D boxes from the same epilogue: the records must be
  // identical, and the boxes those of la3d_project_boxes on the finished records ("P" lines)
  {
    double *d_out2, *d_proj, *d_proj2;
    HIPCHK(hipMalloc(&d_out2, (size_t)B * LA3D_REC * 8)); HIPCHK(hipMalloc(&d_proj, (size_t)B * 64)); HIPCHK(hipMalloc(&d_proj2, (size_t)B * 64));
    la3d_fit_args a = {};
    a.struct_size = (int32_t)sizeof(a);
    a.B = B; a.H = H; a.W = W;
    a.depth = d_depth; a.depth_plane_stride = (int64_t)HW; a.mask = d_mask; a.K = d_K; a.k_stride = 0; a.ground = d_ground;
    // (a zero-initialised block means: no fused instance filter, no area hints, no 2-D boxes unless proj is set)
    a.proj = d_proj; a.image_width = W; a.image_height = H;
    a.out = d_out2; a.status = d_status; a.aux = d_aux; a.workspace = d_ws; a.stream = stream;
    if (la3d_fit_instances_ex(&a) != LA3D_SUCCESS) { fprintf(stderr, "la3d_fit_instances_ex: %s\n", la3d_last_error()); return 6; }
    if (la3d_project_boxes(d_out2, d_K, 0, nullptr, B, (double)W, (double)H, d_proj2, stream) != LA3D_SUCCESS) return 7;
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<double> out2(out.size()), pr((size_t)B * 8), pr2((size_t)B * 8);
    HIPCHK(hipMemcpy(out2.data(), d_out2, out2.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pr.data(), d_proj, pr.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pr2.data(), d_proj2, pr2.size() * 8, hipMemcpyDeviceToHost));
    if (memcmp(out.data(), out2.data(), out.size() * 8) != 0) { fprintf(stderr, "la3d_fit_instances_ex: records differ\n"); return 8; }
    if (memcmp(pr.data(), pr2.data(), pr.size() * 8) != 0) { fprintf(stderr, "la3d_fit_instances_ex: 2-D boxes differ from la3d_project_boxes\n"); return 9; }
    for (int i = 0; i < B; ++i) {
      printf("P");
      for (int k = 0; k < 8; ++k) printf(" %a", pr[(size_t)i * 8 + k]);
      printf("\n");
    }
    a.struct_size = 8;   // a truncated argument block is an error, not a crash
    if (la3d_fit_instances_ex(&a) != LA3D_ERR_ARG) { fprintf(stderr, "expected LA3D_ERR_ARG for a short struct\n"); return 10; }
  }
  // a bad call must come back as an error code, not a crash
  if (la3d_fit_instances(nullptr, (int64_t)HW, nullptr, d_mask, d_K, 0, nullptr, nullptr, B, H, W, d_out, d_status, nullptr,
                         d_ws, stream) != LA3D_ERR_ARG) { fprintf(stderr, "expected LA3D_ERR_ARG\n"); return 5; }
  return 0;
}
