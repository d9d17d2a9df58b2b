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

// persistent grid of the fused kernels: one work-group per CU (weights stationary in ~200 registers per lane, LDS 70-156 KB);
// the CU count of the current device is queried once (256 on an MI355X in SPX mode)
int pp_fused_wgs();
#define PP_FUSED_WGS pp_fused_wgs()

int pp_launch_warp_fused_fwd(const float* params, const float* pts, const int32_t* count, int capacity, float out_range,
                             float* acts, float* out, hipStream_t st);
int pp_launch_warp_fused_bwd(const float* params, const float* pts, const float* acts, const float* out_grad,
                             const int32_t* count, int capacity, float out_range, float* ybar, float* params_grad,
                             float* pts_grad, hipStream_t st);
// operands of one layer of the weight-gradient chain (split-precision kernel, pp_mlp_split.hip)
struct WgradOperands {
  const float* Y;      // [R][128]  gradient w.r.t. the layer's pre-activation (already gated)
  const float* X;      // [R][KX]   input activations of the layer
  float* Wbar;         // [128][KX]
  float* bbar;         // [128] or nullptr: += column sums of Y (every row, or the primal rows of the 4-row form)
};
int pp_launch_wgrad_chain_s(const float* YA, const float* XA, float* WA, const float* YB, const float* XB, float* WB,
                            const float* YC, const float* XC, float* WC, int kxc, const int32_t* count, int rmul, int rcap,
                            hipStream_t st, float* bA, float* bB, float* bC, int wgs = 0);
// split-precision variants (pp_mlp_split.hip, option "mlp_split"): same contracts
int pp_launch_warp_fused_fwd_s(const float* params, const float* pts, const int32_t* count, int capacity, float out_range,
                               float* acts, float* out, hipStream_t st);
int pp_launch_warp_fused_bwd_s(const float* params, const float* pts, const float* acts, const float* out_grad,
                               const int32_t* count, int capacity, float out_range, float* ybar, float* params_grad,
                               float* pts_grad, hipStream_t st);
// weight gradients of three layers (Y_l^T X_l accumulated into W_l) in one persistent kernel; kxc = width of X of layer C;
// bA / bB / bC: also accumulate the bias gradients = column sums of Y over the primal rows (kxc == 128: the warp net's 4-row
// form, every fourth row) or over all rows (kxc == 64: rgbnet)
int pp_launch_wgrad_chain(const float* YA, const float* XA, float* WA, const float* YB, const float* XB, float* WB,
                          const float* YC, const float* XC, float* WC, int kxc, const int32_t* count, int rmul, int rcap,
                          hipStream_t st, float* bA = nullptr, float* bB = nullptr, float* bC = nullptr, int wgs = 0 /* 0: one per CU */);

// parameter block of rgbnet (64-wide padded input): W0[128x64] b0 | W1[128x128] b1 | W2[128x128] b2 | W3[3x128] b3
#define RGF_W0 0
#define RGF_B0 (128 * 64)
#define RGF_W1 (RGF_B0 + 128)
#define RGF_B1 (RGF_W1 + 128 * 128)
#define RGF_W2 (RGF_B1 + 128)
#define RGF_B2 (RGF_W2 + 128 * 128)
#define RGF_W3 (RGF_B2 + 128)
#define RGF_B3 (RGF_W3 + 3 * 128)

int pp_launch_rgb_fused_fwd(const float* params, const float* feat, const int32_t* count, int capacity,
                            const float* logit_add, int add_ld, float* acts, float* rgb, hipStream_t st);
int pp_launch_rgb_fused_fwd_s(const float* params, const float* feat, const int32_t* count, int capacity,
                              const float* logit_add, int add_ld, float* acts, float* rgb, hipStream_t st);
int pp_launch_rgb_fused_bwd_s(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                              const int32_t* count, int capacity, float* ybar, float* params_grad, float* feat_grad,
                              float* logit_grad, int lg_ld, hipStream_t st);
int pp_launch_rgb_fused_bwd(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                            const int32_t* count, int capacity, float* ybar, float* params_grad, float* feat_grad,
                            float* logit_grad, int lg_ld, hipStream_t st);
