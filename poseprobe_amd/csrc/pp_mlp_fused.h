// Internal interface between pp_mlp.hip (C-ABI entry points) and pp_mlp_fused.hip (layer-fused kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// parameter block of the warp net: W0[128x3] b0 | W1..W3[128x128] b | W4[4x128] b4   (poseprobe_amd/engine.py FlatParams)
#define WPF_W0 0
#define WPF_B0 (128 * 3)
#define WPF_W1 (WPF_B0 + 128)
#define WPF_B1 (WPF_W1 + 128 * 128)
#define WPF_W2 (WPF_B1 + 128)
#define WPF_B2 (WPF_W2 + 128 * 128)
#define WPF_W3 (WPF_B2 + 128)
#define WPF_B3 (WPF_W3 + 128 * 128)
#define WPF_W4 (WPF_B3 + 128)
#define WPF_B4 (WPF_W4 + 4 * 128)

// persistent grid of the fused kernels: one work-group per CU (weights stationary in ~200 registers per lane)
#define PP_FUSED_WGS 256

int pp_launch_warp_fused_fwd(const float* params, const float* pts, const int32_t* count, int capacity, float out_range,
                             float* acts, float* out, hipStream_t st);
int pp_launch_warp_fused_bwd(const float* params, const float* pts, const float* acts, const float* out_grad,
                             const int32_t* count, int capacity, float out_range, float* ybar, float* params_grad,
                             float* pts_grad, hipStream_t st);
