// The two shallow MLPs of the object branch on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
//  * rgbnet  64(57)->128->128->128->3 + sigmoid            (lib/voxurf_coarse.py:208-216, :1032-1033)
//  * warp    3->128->128->128->128->4, ReLU                 (lib/deformation/deform_net.py:12-31, modules.py:43-124)
//
// The warp net is evaluated in "4-row" form: for every sample row 0 is the primal activation and rows 1-3 are the
// forward-mode tangents d/dp_i.  A hidden layer is then ONE GEMM over 4M rows whose epilogue adds the bias to row 0
// only and applies row 0's ReLU mask to all four rows; the backward of (value + Jacobian) is the same GEMM chain
// run in reverse with identical masking.  This replaces the reference's four autograd.grad(create_graph=True)
// passes and their double backward (lib/voxurf_coarse.py:968-984) by plain matrix products.
//
// This file holds the C-ABI entry points of both MLPs and the LAYER-BY-LAYER kernels: a persistent NT GEMM (work-group =
// 4 wavefronts in 2x2, 64-row x 128-feature tile, K-chunks of 32 through LDS rows of 36 floats so that a lane fetches its
// four operands of consecutive MFMAs with one ds_read_b128; next chunk / next tile prefetched into registers behind the
// MFMA block), a split-K TN GEMM for the weight gradients and the thin first / last layers.  They serve generic MLP shapes
// (DirectVoxGO twin) and A/B runs (PP_MLP_FUSED=0); the Voxurf shapes run through the layer-fused kernels of
// pp_mlp_fused.hip.  In the accumulator layout a lane holds ONE feature column and rows (reg&3) + 8*(reg>>2) +
// 4*(lane>>5): the four rows of a sample are registers 4q..4q+3 of the same lane, so the 4-row masking needs no
// cross-lane traffic.
#include "pp_common.h"
#include "pp_mlp_fused.h"
#include <stdlib.h>

#include "pp_gemm.h"

// ------------------------------------------------------------------------------------------------ small layers
// warp layer 0 (3 -> 128) in 4-row form.  block = 2 samples x 128 features.
__global__ __launch_bounds__(256) void k_warp_l0_fwd(const float* __restrict__ W0, const float* __restrict__ b0,
                                                     const float* __restrict__ pts, const int32_t* __restrict__ count,
                                                     int capacity, float* __restrict__ X1) {
  int M = min(count[0], capacity);
  int m = blockIdx.x * 2 + (threadIdx.x >> 7), j = threadIdx.x & 127;
  if (m >= M) return;
  float w0 = W0[j * 3], w1 = W0[j * 3 + 1], w2 = W0[j * 3 + 2];
  float y = pts[m * 3] * w0 + pts[m * 3 + 1] * w1 + pts[m * 3 + 2] * w2 + b0[j];
  bool on = y > 0.f;
  size_t base = (size_t)m * 4 * 128 + j;
  X1[base] = on ? y : 0.f;
  X1[base + 128] = on ? w0 : 0.f;
  X1[base + 256] = on ? w1 : 0.f;
  X1[base + 384] = on ? w2 : 0.f;
}

// warp output layer (128 -> 4) on 4 rows: one wavefront per sample, 16 lanes per row.
__global__ __launch_bounds__(256) void k_warp_l4_fwd(const float* __restrict__ W4, const float* __restrict__ b4,
                                                     const float* __restrict__ X4, const int32_t* __restrict__ count,
                                                     int capacity, float out_range, float* __restrict__ out) {
  int M = min(count[0], capacity);
  int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  int c = lane >> 4, sub = lane & 15;
  const float4* xp = reinterpret_cast<const float4*>(X4 + ((size_t)m * 4 + c) * 128 + sub * 8);
  float4 xa = xp[0], xb = xp[1];
  float acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float4* wp = reinterpret_cast<const float4*>(W4 + o * 128 + sub * 8);
    float4 wa = wp[0], wb = wp[1];
    float s = xa.x * wa.x + xa.y * wa.y + xa.z * wa.z + xa.w * wa.w + xb.x * wb.x + xb.y * wb.y + xb.z * wb.z + xb.w * wb.w;
    s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    acc[o] = s;
  }
  if (sub == 0) {
    float4 r;
    r.x = (acc[0] + (c == 0 ? b4[0] : 0.f)) * out_range;
    r.y = (acc[1] + (c == 0 ? b4[1] : 0.f)) * out_range;
    r.z = (acc[2] + (c == 0 ? b4[2] : 0.f)) * out_range;
    r.w = (acc[3] + (c == 0 ? b4[3] : 0.f)) * out_range;
    *reinterpret_cast<float4*>(out + (size_t)m * 16 + c * 4) = r;
  }
}

// backward of the output layer: Ybar4 = mask(X4) * (out_grad*range) W4 ; W4bar, b4bar accumulated over a strip.
#define STRIP 64
__global__ __launch_bounds__(256) void k_warp_l4_bwd(const float* __restrict__ W4, const float* __restrict__ X4,
                                                     const float* __restrict__ out_grad,
                                                     const int32_t* __restrict__ count, int capacity, float out_range,
                                                     float* __restrict__ Ybar, float* __restrict__ W4bar,
                                                     float* __restrict__ b4bar) {
  __shared__ float red[4 * 128];
  int M = min(count[0], capacity);
  int m0 = blockIdx.x * STRIP;
  if (m0 >= M) return;
  int h = threadIdx.x >> 7, j = threadIdx.x & 127;
  float w[4] = {W4[j], W4[128 + j], W4[256 + j], W4[384 + j]};
  float wacc[4] = {0, 0, 0, 0}, bacc = 0.f;
  int mend = min(m0 + STRIP, M);
  for (int m = m0 + h; m < mend; m += 2) {
    const float* og = out_grad + (size_t)m * 16;
    size_t base = (size_t)m * 4 * 128 + j;
    bool on = X4[base] > 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float g0 = og[c * 4] * out_range, g1 = og[c * 4 + 1] * out_range, g2 = og[c * 4 + 2] * out_range,
            g3 = og[c * 4 + 3] * out_range;
      float x = X4[base + c * 128];
      wacc[0] += g0 * x; wacc[1] += g1 * x; wacc[2] += g2 * x; wacc[3] += g3 * x;
      float xb = g0 * w[0] + g1 * w[1] + g2 * w[2] + g3 * w[3];
      Ybar[base + c * 128] = on ? xb : 0.f;
    }
    if (j < 4) bacc += og[j] * out_range;
  }
  if (h == 1) { for (int o = 0; o < 4; ++o) red[o * 128 + j] = wacc[o]; }
  __syncthreads();
  if (h == 0) { for (int o = 0; o < 4; ++o) atomicAdd(&W4bar[o * 128 + j], wacc[o] + red[o * 128 + j]); }
  if (j < 4 && bacc != 0.f) atomicAdd(&b4bar[j], bacc);
}

// backward of warp layer 0, part (a): pts_grad[m][i] += sum_j Ybar1[4m][j] * W0[j][i]   (16 lanes per sample)
__global__ __launch_bounds__(256) void k_warp_l0_bwd_pts(const float* __restrict__ W0, const float* __restrict__ Ybar,
                                                         const int32_t* __restrict__ count, int capacity,
                                                         float* __restrict__ pts_grad) {
  int M = min(count[0], capacity);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, sub = t & 15;
  bool live = m < M;
  float acc[3] = {0.f, 0.f, 0.f};
  if (live) {
    const float4* yp = reinterpret_cast<const float4*>(Ybar + (size_t)m * 4 * 128 + sub * 8);
    float4 ya = yp[0], yb = yp[1];
    float y[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float* w = W0 + (sub * 8 + q) * 3;
      acc[0] += y[q] * w[0]; acc[1] += y[q] * w[1]; acc[2] += y[q] * w[2];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float v = acc[i];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    if (live && sub == 0) pts_grad[m * 3 + i] += v;
  }
}

// part (b): W0bar[j][i] += sum_m (Ybar1[4m][j] p_i + Ybar1[4m+1+i][j]) ; b0bar[j] += sum_m Ybar1[4m][j]
#define STRIP0 128
__global__ __launch_bounds__(256) void k_warp_l0_bwd_w(const float* __restrict__ pts, const float* __restrict__ Ybar,
                                                       const int32_t* __restrict__ count, int capacity,
                                                       float* __restrict__ W0bar, float* __restrict__ b0bar) {
  __shared__ float red[4 * 128];
  int M = min(count[0], capacity);
  int m0 = blockIdx.x * STRIP0;
  if (m0 >= M) return;
  int h = threadIdx.x >> 7, j = threadIdx.x & 127;
  float wacc[3] = {0, 0, 0}, bacc = 0.f;
  int mend = min(m0 + STRIP0, M);
  for (int m = m0 + h; m < mend; m += 2) {
    size_t base = (size_t)m * 4 * 128 + j;
    float y0 = Ybar[base];
    wacc[0] += y0 * pts[m * 3] + Ybar[base + 128];
    wacc[1] += y0 * pts[m * 3 + 1] + Ybar[base + 256];
    wacc[2] += y0 * pts[m * 3 + 2] + Ybar[base + 384];
    bacc += y0;
  }
  if (h == 1) { for (int i = 0; i < 3; ++i) red[i * 128 + j] = wacc[i]; red[3 * 128 + j] = bacc; }
  __syncthreads();
  if (h == 0) {
    for (int i = 0; i < 3; ++i) atomicAdd(&W0bar[j * 3 + i], wacc[i] + red[i * 128 + j]);
    atomicAdd(&b0bar[j], bacc + red[3 * 128 + j]);
  }
}

// rgbnet output layer (128 -> 3) + sigmoid: 16 lanes per sample.
__global__ __launch_bounds__(256) void k_rgb_out_fwd(const float* __restrict__ W3, const float* __restrict__ b3,
                                                     const float* __restrict__ H3, const int32_t* __restrict__ count,
                                                     int capacity, const float* __restrict__ logit_add, int add_ld,
                                                     float* __restrict__ rgb) {
  int M = min(count[0], capacity);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, sub = t & 15;
  bool live = m < M;
  float4 xa = make_float4(0, 0, 0, 0), xb = xa;
  if (live) {
    const float4* xp = reinterpret_cast<const float4*>(H3 + (size_t)m * 128 + sub * 8);
    xa = xp[0]; xb = xp[1];
  }
  float acc[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float4* wp = reinterpret_cast<const float4*>(W3 + o * 128 + sub * 8);
    float4 wa = wp[0], wb = wp[1];
    float s = xa.x * wa.x + xa.y * wa.y + xa.z * wa.z + xa.w * wa.w + xb.x * wb.x + xb.y * wb.y + xb.z * wb.z + xb.w * wb.w;
    s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    acc[o] = s;
  }
  if (live && sub == 0)
    for (int o = 0; o < 3; ++o) rgb[m * 3 + o] = pp_sigmoid(acc[o] + b3[o] + (logit_add ? logit_add[(size_t)m * add_ld + o] : 0.f));
}

__global__ __launch_bounds__(256) void k_rgb_out_bwd(const float* __restrict__ W3, const float* __restrict__ H3,
                                                     const float* __restrict__ rgb, const float* __restrict__ rgb_grad,
                                                     const int32_t* __restrict__ count, int capacity,
                                                     float* __restrict__ Ybar, float* __restrict__ W3bar,
                                                     float* __restrict__ b3bar, float* __restrict__ logit_grad, int lg_ld) {
  __shared__ float red[3 * 128];
  int M = min(count[0], capacity);
  int m0 = blockIdx.x * STRIP;
  if (m0 >= M) return;
  int h = threadIdx.x >> 7, j = threadIdx.x & 127;
  float w[3] = {W3[j], W3[128 + j], W3[256 + j]};
  float wacc[3] = {0, 0, 0}, bacc = 0.f;
  int mend = min(m0 + STRIP, M);
  for (int m = m0 + h; m < mend; m += 2) {
    float gl[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) { float r = rgb[m * 3 + o]; gl[o] = rgb_grad[m * 3 + o] * r * (1.f - r); }
    float x = H3[(size_t)m * 128 + j];
    wacc[0] += gl[0] * x; wacc[1] += gl[1] * x; wacc[2] += gl[2] * x;
    float hb = gl[0] * w[0] + gl[1] * w[1] + gl[2] * w[2];
    Ybar[(size_t)m * 128 + j] = (x > 0.f) ? hb : 0.f;
    if (j < 3) { bacc += gl[j]; if (logit_grad) logit_grad[(size_t)m * lg_ld + j] = gl[j]; }
  }
  if (h == 1) { for (int o = 0; o < 3; ++o) red[o * 128 + j] = wacc[o]; }
  __syncthreads();
  if (h == 0) { for (int o = 0; o < 3; ++o) atomicAdd(&W3bar[o * 128 + j], wacc[o] + red[o * 128 + j]); }
  if (j < 3 && bacc != 0.f) atomicAdd(&b3bar[j], bacc);
}

// ------------------------------------------------------------------------------------------------ host side
#define RG_W0 0
#define RG_B0 (128 * 64)
#define RG_W1 (RG_B0 + 128)
#define RG_B1 (RG_W1 + 128 * 128)
#define RG_W2 (RG_B1 + 128)
#define RG_B2 (RG_W2 + 128 * 128)
#define RG_W3 (RG_B2 + 128)
#define RG_B3 (RG_W3 + 3 * 128)

#define WP_W0 0
#define WP_B0 (128 * 3)
#define WP_W1 (WP_B0 + 128)
#define WP_B1 (WP_W1 + 128 * 128)
#define WP_W2 (WP_B1 + 128)
#define WP_B2 (WP_W2 + 128 * 128)
#define WP_W3 (WP_B2 + 128)
#define WP_B3 (WP_W3 + 128 * 128)
#define WP_W4 (WP_B3 + 128)
#define WP_B4 (WP_W4 + 4 * 128)

// weight-gradient GEMM: a FIXED number of work-groups splits the (device-side) row count evenly; measured optimum on
// MI355X ~ 450 work-groups (more: the 64 KB of contended atomics per work-group dominates; fewer: idle CUs).
static const int TN_WGS = 448;

// ------------------------------------------------------------------------------------------------ side-stream context
// The weight-gradient GEMM (k_gemm_tn) and the data-gradient GEMM of one layer both consume Ybar_k and are otherwise
// independent; each keeps the matrix pipe only ~55 % busy on its own (K = 128 leaves a tile prologue / epilogue per
// 128 MFMAs).  With a context the backward chains fork the weight-gradient GEMM onto an auxiliary HIP stream so the
// two kernels co-reside on the CUs and fill each other's MFMA bubbles.  The context is explicit caller-owned state
// (no globals); fork / join are event edges, so the sequence stays capturable in a hipGraph.
// Fused paths: the weight-gradient kernel of a chain depends only on what the data-gradient kernel left in `scratch` and
// nothing downstream needs it before the optimiser, so with a context it is launched on the auxiliary stream and NOT
// joined here: the caller's next kernels (small, latency-bound ones: colour-feature / geometry backward, ray and pose
// backward) run beside it, and pp_context_join() is called before the scratch buffer or the gradients are touched again.
static hipStream_t deferred_fork(void* ctx, hipStream_t main, int min_mode = 1) {
  PPContext* c = static_cast<PPContext*>(ctx);
  if (!c || c->opt[PP_OPT_SIDE_STREAM] < min_mode || c->pending >= 4 || !pp_context_aux(c)) return main;
  hipEventRecord(c->dfork[c->pending], main);
  hipStreamWaitEvent(c->aux, c->dfork[c->pending], 0);
  return c->aux;
}
static void deferred_forked(void* ctx, hipStream_t used, hipStream_t main) {
  PPContext* c = static_cast<PPContext*>(ctx);
  if (!c || used == main) return;
  hipEventRecord(c->djoin[c->pending], c->aux);
  ++c->pending;
}

extern "C" int pp_context_join(void* ctx, void* stream) {
  PPContext* c = static_cast<PPContext*>(ctx);
  if (!c || !c->have_aux) return PP_OK;
  for (int i = 0; i < c->pending; ++i) hipStreamWaitEvent(pp_stream(stream), c->djoin[i], 0);
  c->pending = 0;
  return PP_OK;
}

struct SideLane {
  PPContext* c;
  hipStream_t main;
  int n = 0;        // forks issued
  int waited = 0;   // joins already waited for
  SideLane(void* ctx, hipStream_t m) : c(static_cast<PPContext*>(ctx)), main(m) {
    if (c && (c->opt[PP_OPT_SIDE_STREAM] == 0 || !pp_context_aux(c))) c = nullptr;
  }
  // stream on which the next side kernel must be launched (after everything enqueued on `main` so far)
  hipStream_t fork() {
    if (!c) return main;
    hipEventRecord(c->fork[n], main);
    hipStreamWaitEvent(c->aux, c->fork[n], 0);
    return c->aux;
  }
  void forked() { if (c) { hipEventRecord(c->join[n], c->aux); ++n; } }
  // main waits for every side kernel issued so far except the `keep` most recent ones
  void join(int keep = 0) {
    if (!c) return;
    for (; waited < n - keep; ++waited) hipStreamWaitEvent(main, c->join[waited], 0);
  }
};

// option "mlp_fused" = 0 selects the layer-by-layer kernels (A/B measurements, generic shapes always use them)
static bool mlp_fused_enabled() { return pp_opt(PP_OPT_MLP_FUSED) == 1; }

static const int GEMM_MAX_WG = 256 * 5;     // 5 resident work-groups per CU at BM=64 (25 KB LDS, 90 regs)
static const int GEMM_MAX_WG_SHARED = 256 * 3;   // when a weight-gradient GEMM runs beside it (register file: 2 x 96 + 2 x 144)
static inline int gemm_grid(int rows, int bm, bool shared = false) {
  int t = pp_div_up(rows, bm), cap = shared ? GEMM_MAX_WG_SHARED : GEMM_MAX_WG;
  return t < cap ? t : cap;
}
#ifndef PP_GEMM_BM
#define PP_GEMM_BM 64
#endif

// Generic ReLU MLP  in_ld -> 128 -> ... -> 128 -> 3 (+ optional sigmoid), n_gemm = number of 128-wide hidden layers.
// Parameter block: W0[128*in_ld] b0[128] | (W[128*128] b[128]) x (n_gemm-1) | Wout[3*128] bout[3].
static inline size_t mlp_off_hidden(int in_ld, int l) { return (size_t)128 * in_ld + 128 + (size_t)(l - 1) * (128 * 128 + 128); }
static inline size_t mlp_off_out(int in_ld, int n_gemm) { return mlp_off_hidden(in_ld, n_gemm); }

extern "C" int pp_mlp_fwd(const float* params, const float* feat, int32_t in_ld, int32_t n_gemm, const int32_t* count,
                          int32_t capacity, const float* logit_add, int32_t logit_add_ld, float* acts, float* out,
                          void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && feat && count && out, "null pointer");
  PP_REQUIRE(capacity > 0 && in_ld % 32 == 0 && in_ld <= 128 && n_gemm >= 1 && n_gemm <= 8, "bad sizes");
  // acts == NULL: forward only (no backward pass will follow: the activations are not written) - the split-precision fused kernel only
  PP_REQUIRE(acts || (in_ld == 64 && n_gemm == 3 && mlp_fused_enabled() && (pp_opt(PP_OPT_MLP_SPLIT) & 4)),
             "acts may be NULL only for the rgbnet shape with the split-precision forward kernel (option mlp_split bit 4)");
  hipStream_t st = pp_stream(stream);
  if (in_ld == 64 && n_gemm == 3 && mlp_fused_enabled()) {       // the Voxurf rgbnet shape: layer-fused kernel
    if (pp_opt(PP_OPT_MLP_SPLIT) & 4) pp_launch_rgb_fused_fwd_s(params, feat, count, capacity, logit_add, logit_add_ld, acts, out, st);
    else pp_launch_rgb_fused_fwd(params, feat, count, capacity, logit_add, logit_add_ld, acts, out, st);
    PP_CHECK_LAUNCH();
    return PP_OK;
  }
  const size_t LS = (size_t)capacity * 128;
  dim3 g(gemm_grid(capacity, PP_GEMM_BM)), b(256);
  hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_RELU, 1, PP_GEMM_BM>), g, b, 0, st, feat, in_ld, params, in_ld, in_ld, 128,
                     params + (size_t)128 * in_ld, nullptr, 0, acts, 128, count, 1, capacity);
  for (int l = 1; l < n_gemm; ++l) {
    const float* W = params + mlp_off_hidden(in_ld, l);
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_RELU, 1, PP_GEMM_BM>), g, b, 0, st, acts + (l - 1) * LS, 128, W, 128, 128, 128,
                       W + 128 * 128, nullptr, 0, acts + l * LS, 128, count, 1, capacity);
  }
  const float* Wo = params + mlp_off_out(in_ld, n_gemm);
  hipLaunchKernelGGL(k_rgb_out_fwd, dim3(pp_div_up(capacity * 16, 256)), b, 0, st, Wo, Wo + 3 * 128,
                     acts + (n_gemm - 1) * LS, count, capacity, logit_add, logit_add_ld, out);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_mlp_bwd(const float* params, const float* feat, int32_t in_ld, int32_t n_gemm, const float* acts,
                          const float* out, const float* out_grad, const int32_t* count, int32_t capacity,
                          float* scratch, float* params_grad, float* feat_grad, float* logit_add_grad,
                          int32_t logit_add_ld, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && feat && acts && out && out_grad && count && scratch && params_grad, "null pointer");
  PP_REQUIRE(capacity > 0 && in_ld % 32 == 0 && in_ld <= 128 && n_gemm >= 1 && n_gemm <= 8, "bad sizes");
  hipStream_t st = pp_stream(stream);
  if (in_ld == 64 && n_gemm == 3 && feat_grad && mlp_fused_enabled()) {
    const bool sb = (pp_opt(PP_OPT_MLP_SPLIT) & 8) != 0;   // split-precision data-gradient kernel: b0..b2 come from the weight-gradient kernel
    if (sb) pp_launch_rgb_fused_bwd_s(params, acts, out, out_grad, count, capacity, scratch, params_grad, feat_grad, logit_add_grad,
                                      logit_add_ld, st);
    else pp_launch_rgb_fused_bwd(params, acts, out, out_grad, count, capacity, scratch, params_grad, feat_grad, logit_add_grad,
                                 logit_add_ld, st);
    const size_t FLS = (size_t)capacity * 128;
    hipStream_t ws = deferred_fork(ctx, st, 1);
    pp_launch_wgrad_chain(scratch, acts + FLS, params_grad + RGF_W2, scratch + FLS, acts, params_grad + RGF_W1,
                          scratch + 2 * FLS, feat, params_grad + RGF_W0, 64, count, 1, capacity, ws,
                          sb ? params_grad + RGF_B2 : nullptr, sb ? params_grad + RGF_B1 : nullptr, sb ? params_grad + RGF_B0 : nullptr,
                          ws != st ? pp_opt(PP_OPT_WGRAD_SIDE_WGS) : 0);
    deferred_forked(ctx, ws, st);
    PP_CHECK_LAUNCH();
    return PP_OK;
  }
  SideLane side(ctx, st);
  const size_t LS = (size_t)capacity * 128;
  float* cur = scratch;
  float* nxt = scratch + LS;
  float* wt = scratch + 2 * LS;      // one transposed weight matrix at a time (128*128 floats)
  dim3 g(gemm_grid(capacity, PP_GEMM_BM, side.c != nullptr)), gt(TN_WGS), b(256);
  const size_t oo = mlp_off_out(in_ld, n_gemm);
  hipLaunchKernelGGL(k_rgb_out_bwd, dim3(pp_div_up(capacity, STRIP)), b, 0, st, params + oo, acts + (n_gemm - 1) * LS, out,
                     out_grad, count, capacity, cur, params_grad + oo, params_grad + oo + 3 * 128, logit_add_grad,
                     logit_add_ld);
  for (int l = n_gemm - 1; l >= 1; --l) {
    const size_t ow = mlp_off_hidden(in_ld, l);
    hipStream_t ss = side.fork();                      // weight gradient of layer l beside its data gradient
    hipLaunchKernelGGL((k_gemm_tn<1>), gt, b, 0, ss, cur, 128, acts + (l - 1) * LS, 128, 128, params_grad + ow, 128,
                       params_grad + ow + 128 * 128, count, 1, capacity);
    side.forked();
    side.join(1);                                      // the previous layer's side GEMM still reads `nxt`
    hipLaunchKernelGGL(k_transpose, dim3(64), b, 0, st, params + ow, wt, 128, 128);
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_MASK, 1, PP_GEMM_BM>), g, b, 0, st, cur, 128, wt, 128, 128, 128, nullptr,
                       acts + (l - 1) * LS, 128, nxt, 128, count, 1, capacity);
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  hipStream_t ss = side.fork();
  hipLaunchKernelGGL((k_gemm_tn<1>), gt, b, 0, ss, cur, 128, feat, in_ld, in_ld, params_grad, in_ld,
                     params_grad + (size_t)128 * in_ld, count, 1, capacity);
  side.forked();
  if (feat_grad) {
    hipLaunchKernelGGL(k_transpose, dim3(pp_div_up(128 * in_ld, 256)), b, 0, st, params, wt, 128, in_ld);
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_PLAIN, 1, PP_GEMM_BM>), g, b, 0, st, cur, 128, wt, 128, 128, in_ld, nullptr,
                       nullptr, 0, feat_grad, in_ld, count, 1, capacity);
  }
  side.join(0);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// rgbnet of the Voxurf configuration = generic MLP with a 64-wide (57 used) input and three 128-wide layers.
extern "C" int pp_rgbnet_fwd(const float* params, const float* feat, const int32_t* count, int32_t capacity,
                             float* acts, float* rgb, void* ctx, void* stream) {
  return pp_mlp_fwd(params, feat, 64, 3, count, capacity, nullptr, 0, acts, rgb, ctx, stream);
}

extern "C" int pp_rgbnet_bwd(const float* params, const float* feat, const float* acts, const float* rgb,
                             const float* rgb_grad, const int32_t* count, int32_t capacity, float* scratch,
                             float* params_grad, float* feat_grad, void* ctx, void* stream) {
  PP_REQUIRE(feat_grad, "null pointer");
  return pp_mlp_bwd(params, feat, 64, 3, acts, rgb, rgb_grad, count, capacity, scratch, params_grad, feat_grad, nullptr, 0,
                    ctx, stream);
}

extern "C" int pp_warp_fwd(const float* params, const float* pts, const int32_t* count, int32_t capacity,
                           float out_range, float* acts, float* out, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && pts && count && out, "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  PP_REQUIRE(acts || (mlp_fused_enabled() && (pp_opt(PP_OPT_MLP_SPLIT) & 1)),
             "acts may be NULL (forward only) only with the split-precision forward kernel (option mlp_split bit 1)");
  hipStream_t st = pp_stream(stream);
  if (mlp_fused_enabled()) {
    if (pp_opt(PP_OPT_MLP_SPLIT) & 1) pp_launch_warp_fused_fwd_s(params, pts, count, capacity, out_range, acts, out, st);
    else pp_launch_warp_fused_fwd(params, pts, count, capacity, out_range, acts, out, st);
    PP_CHECK_LAUNCH();
    return PP_OK;
  }
  const int rcap = capacity * 4;
  const size_t LS = (size_t)rcap * 128;
  dim3 g(gemm_grid(rcap, PP_GEMM_BM)), b(256);
  hipLaunchKernelGGL(k_warp_l0_fwd, dim3(pp_div_up(capacity, 2)), b, 0, st, params + WP_W0, params + WP_B0, pts, count,
                     capacity, acts);
  hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_RELU, 4, PP_GEMM_BM>), g, b, 0, st, acts, 128, params + WP_W1, 128, 128, 128,
                     params + WP_B1, nullptr, 0, acts + LS, 128, count, 4, rcap);
  hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_RELU, 4, PP_GEMM_BM>), g, b, 0, st, acts + LS, 128, params + WP_W2, 128, 128, 128,
                     params + WP_B2, nullptr, 0, acts + 2 * LS, 128, count, 4, rcap);
  hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_RELU, 4, PP_GEMM_BM>), g, b, 0, st, acts + 2 * LS, 128, params + WP_W3, 128, 128, 128,
                     params + WP_B3, nullptr, 0, acts + 3 * LS, 128, count, 4, rcap);
  hipLaunchKernelGGL(k_warp_l4_fwd, dim3(pp_div_up(capacity, 4)), b, 0, st, params + WP_W4, params + WP_B4,
                     acts + 3 * LS, count, capacity, out_range, out);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_warp_bwd(const float* params, const float* pts, const float* acts, const float* out_grad,
                           const int32_t* count, int32_t capacity, float out_range, float* scratch,
                           float* params_grad, float* pts_grad, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && pts && acts && out_grad && count && scratch && params_grad && pts_grad, "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  hipStream_t st = pp_stream(stream);
  const int rcap = capacity * 4;
  const size_t LS = (size_t)rcap * 128;
  if (mlp_fused_enabled()) {
    // one fused data-gradient kernel (+ thin layers), then the three weight-gradient GEMMs on the Ybar it left behind
    if (pp_opt(PP_OPT_MLP_SPLIT) & 2) pp_launch_warp_fused_bwd_s(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad, st);
    else pp_launch_warp_fused_bwd(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad, st);
    hipStream_t ws = deferred_fork(ctx, st, 1);                // option side_stream = 2 keeps the warp chain sequential
    const bool sb = (pp_opt(PP_OPT_MLP_SPLIT) & 2) != 0;   // the split-precision data-gradient kernel leaves b1..b3 to this one
    pp_launch_wgrad_chain(scratch, acts + 2 * LS, params_grad + WP_W3, scratch + LS, acts + LS, params_grad + WP_W2,
                          scratch + 2 * LS, acts, params_grad + WP_W1, 128, count, 4, rcap, ws,
                          sb ? params_grad + WP_B3 : nullptr, sb ? params_grad + WP_B2 : nullptr, sb ? params_grad + WP_B1 : nullptr,
                          ws != st ? pp_opt(PP_OPT_WGRAD_SIDE_WGS) : 0);
    deferred_forked(ctx, ws, st);
    PP_CHECK_LAUNCH();
    return PP_OK;
  }
  SideLane side(ctx, st);
  float* cur = scratch;
  float* nxt = scratch + LS;
  float* wt = scratch + 2 * LS;          // transposed weights W3^T, W2^T, W1^T
  dim3 g(gemm_grid(rcap, PP_GEMM_BM, side.c != nullptr)), gt(TN_WGS), b(256);
  hipLaunchKernelGGL(k_transpose, dim3(64), b, 0, st, params + WP_W3, wt, 128, 128);
  hipLaunchKernelGGL(k_transpose, dim3(64), b, 0, st, params + WP_W2, wt + 16384, 128, 128);
  hipLaunchKernelGGL(k_transpose, dim3(64), b, 0, st, params + WP_W1, wt + 32768, 128, 128);
  hipLaunchKernelGGL(k_warp_l4_bwd, dim3(pp_div_up(capacity, STRIP)), b, 0, st, params + WP_W4, acts + 3 * LS, out_grad,
                     count, capacity, out_range, cur, params_grad + WP_W4, params_grad + WP_B4);
  const int w_off[4] = {0, WP_W1, WP_W2, WP_W3};
  const int b_off[4] = {0, WP_B1, WP_B2, WP_B3};
  for (int l = 3; l >= 1; --l) {
    hipStream_t ss = side.fork();                      // weight gradient of layer l beside its data gradient
    hipLaunchKernelGGL((k_gemm_tn<4>), gt, b, 0, ss, cur, 128, acts + (l - 1) * LS, 128, 128, params_grad + w_off[l], 128,
                       params_grad + b_off[l], count, 4, rcap);
    side.forked();
    side.join(1);                                      // the previous layer's side GEMM still reads `nxt`
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI_MASK, 4, PP_GEMM_BM>), g, b, 0, st, cur, 128, wt + (3 - l) * 16384, 128, 128,
                       128, nullptr, acts + (l - 1) * LS, 128, nxt, 128, count, 4, rcap);
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  // layer 0 (cur = Ybar1)
  hipLaunchKernelGGL(k_warp_l0_bwd_pts, dim3(pp_div_up(capacity * 16, 256)), b, 0, st, params + WP_W0, cur, count, capacity,
                     pts_grad);
  hipLaunchKernelGGL(k_warp_l0_bwd_w, dim3(pp_div_up(capacity, STRIP0)), b, 0, st, pts, cur, count, capacity,
                     params_grad + WP_W0, params_grad + WP_B0);
  side.join(0);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Two-stage forms of the layer-fused backward chains: the data-gradient kernel (which also produces the thin layers' and
// all bias gradients and leaves Ybar of the hidden layers in `scratch`) and the weight-gradient kernel are separate entry
// points, so that a caller can time them, place other work between them, or run the second on another stream.
// pp_warp_bwd / pp_rgbnet_bwd are exactly stage 1 followed by stage 2.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int pp_warp_bwd_data(const float* params, const float* pts, const float* acts, const float* out_grad,
                                const int32_t* count, int32_t capacity, float out_range, float* scratch,
                                float* params_grad, float* pts_grad, int32_t* stage2_host, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && pts && acts && out_grad && count && scratch && params_grad && pts_grad && stage2_host, "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  if (!mlp_fused_enabled()) { pp_set_error("pp_warp_bwd_data: option mlp_fused = 0 has no two-stage form"); return PP_ERR_UNSUPPORTED; }
  *stage2_host = (pp_opt(PP_OPT_MLP_SPLIT) & 2) ? 1 : 0;       // 1: the hidden layers' bias gradients are stage 2's to produce
  if (pp_opt(PP_OPT_MLP_SPLIT) & 2)
    pp_launch_warp_fused_bwd_s(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad,
                               pp_stream(stream));
  else
    pp_launch_warp_fused_bwd(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad,
                             pp_stream(stream));
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_warp_bwd_weights(const float* acts, const float* scratch, const int32_t* count, int32_t capacity,
                                   float* params_grad, int32_t stage2, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(acts && scratch && count && params_grad, "null pointer");
  PP_REQUIRE(capacity > 0 && (stage2 == 0 || stage2 == 1), "capacity<=0 or stage2 is not what pp_warp_bwd_data returned");
  if (!mlp_fused_enabled()) { pp_set_error("pp_warp_bwd_weights: option mlp_fused = 0 has no two-stage form"); return PP_ERR_UNSUPPORTED; }
  const int rcap = capacity * 4;
  const size_t LS = (size_t)rcap * 128;
  const bool sb = stage2 != 0;     // who owns b1..b3 was decided by stage 1 and is handed over explicitly (never re-read from the options)
  pp_launch_wgrad_chain(scratch, acts + 2 * LS, params_grad + WP_W3, scratch + LS, acts + LS, params_grad + WP_W2,
                        scratch + 2 * LS, acts, params_grad + WP_W1, 128, count, 4, rcap, pp_stream(stream),
                        sb ? params_grad + WP_B3 : nullptr, sb ? params_grad + WP_B2 : nullptr, sb ? params_grad + WP_B1 : nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_rgbnet_bwd_data(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                                  const int32_t* count, int32_t capacity, float* scratch, float* params_grad,
                                  float* feat_grad, int32_t* stage2_host, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && acts && rgb && rgb_grad && count && scratch && params_grad && feat_grad && stage2_host, "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  if (!mlp_fused_enabled()) { pp_set_error("pp_rgbnet_bwd_data: option mlp_fused = 0 has no two-stage form"); return PP_ERR_UNSUPPORTED; }
  *stage2_host = (pp_opt(PP_OPT_MLP_SPLIT) & 8) ? 1 : 0;
  if (pp_opt(PP_OPT_MLP_SPLIT) & 8)
    pp_launch_rgb_fused_bwd_s(params, acts, rgb, rgb_grad, count, capacity, scratch, params_grad, feat_grad, nullptr, 0,
                              pp_stream(stream));
  else
    pp_launch_rgb_fused_bwd(params, acts, rgb, rgb_grad, count, capacity, scratch, params_grad, feat_grad, nullptr, 0,
                            pp_stream(stream));
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_rgbnet_bwd_weights(const float* feat, const float* acts, const float* scratch, const int32_t* count,
                                     int32_t capacity, float* params_grad, int32_t stage2, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(feat && acts && scratch && count && params_grad, "null pointer");
  PP_REQUIRE(capacity > 0 && (stage2 == 0 || stage2 == 1), "capacity<=0 or stage2 is not what pp_rgbnet_bwd_data returned");
  if (!mlp_fused_enabled()) { pp_set_error("pp_rgbnet_bwd_weights: option mlp_fused = 0 has no two-stage form"); return PP_ERR_UNSUPPORTED; }
  const size_t FLS = (size_t)capacity * 128;
  const bool sb = stage2 != 0;     // see pp_warp_bwd_weights
  pp_launch_wgrad_chain(scratch, acts + FLS, params_grad + RGF_W2, scratch + FLS, acts, params_grad + RGF_W1,
                        scratch + 2 * FLS, feat, params_grad + RGF_W0, 64, count, 1, capacity, pp_stream(stream),
                        sb ? params_grad + RGF_B2 : nullptr, sb ? params_grad + RGF_B1 : nullptr, sb ? params_grad + RGF_B0 : nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Workspace queries: floats the caller must provide for `acts` (kept from forward to backward) and `scratch` (backward
// only) of the MLP entry points at a given sample capacity.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int pp_rgbnet_workspace(int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats) {
  PP_REQUIRE(acts_floats && scratch_floats && capacity > 0, "bad arguments");
  *acts_floats = (int64_t)3 * capacity * 128;
  *scratch_floats = (int64_t)3 * capacity * 128 + 49152;          // Ybar of the three hidden layers (+ transposed weights, layered path)
  return PP_OK;
}

extern "C" int pp_warp_workspace(int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats) {
  PP_REQUIRE(acts_floats && scratch_floats && capacity > 0, "bad arguments");
  *acts_floats = (int64_t)4 * capacity * 4 * 128;                 // four hidden activations of the 4-row form
  *scratch_floats = (int64_t)3 * capacity * 4 * 128 + 49152;
  return PP_OK;
}

extern "C" int pp_mlp_workspace(int32_t in_ld, int32_t n_gemm, int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats) {
  PP_REQUIRE(acts_floats && scratch_floats && capacity > 0 && in_ld % 32 == 0 && in_ld <= 128 && n_gemm >= 1 && n_gemm <= 8,
             "bad arguments");
  *acts_floats = (int64_t)n_gemm * capacity * 128;
  *scratch_floats = (int64_t)(n_gemm > 3 ? n_gemm : 3) * capacity * 128 + 49152;
  return PP_OK;
}
