// la3d_engines.hpp - what the fit engines (one translation unit each) offer la3d_fit_instances' dispatcher (la3d.hip).  Host side only.
#pragma once
#include "la3d_device.hpp"

namespace la3d {
// instance engine (la3d_instance.hip): one workgroup per instance - every call the other engines do not take.  `lds` = bit image +
// Shared, `poly_stage` = the polygon side stage; picks the instantiation of fit_instances_kernel for the frame and launches it.
int instance_fit(FitParams p, bool vec, bool ldsmask, bool sample, size_t lds, size_t poly_stage, hipStream_t s, void* workspace,
                 const char* who);
// band engine (la3d_band.hip): two / four / eight workgroups per instance that meet through the workspace (grounded u8 batches of 1..160)
bool band_eligible(const FitParams& p, bool vec, bool sample);
bool band_frame_ok(int H, int W, int nb);
size_t band_workspace_bytes(int B);
int band_fit(const FitParams& p, hipStream_t s, void* workspace);
// row engine (la3d_rows.hip): up to sixteen workgroups per instance, one per band of rows (un-grounded u8 batches up to 160)
bool rows_fit_if_eligible(const FitParams& p, bool vec, bool sample, hipStream_t s, void* workspace, int* rc);
size_t rows_workspace_bytes(int B, int H, int W);
// split engine (la3d_split.hip)
bool split_eligible(const FitParams& p, bool vec, bool ldsmask);
int split_fit(const FitParams& p, void* workspace, hipStream_t s);
size_t split_workspace_bytes(int B, int H, int W);
}  // namespace la3d
