// object_losses (lib/losses.py:6-74): forward values and the gradients w.r.t. the render outputs in one pass.
// Ray-level terms (masked MSE :26-29, entropy on alphainv_cum :42-45, BCE mask :66) and sample-level terms
// (eikonal :6-10, deformation priors :11-23) are reduced per block and accumulated with one atomic per block.
#include "pp_common.h"

__device__ __forceinline__ float block_sum256(float v, float* sm) {
  v = pp_wave_sum(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void k_loss_rays(const float* __restrict__ rgbm, const float* __restrict__ alast,
                                                   const float* __restrict__ cw, const float* __restrict__ target,
                                                   const float* __restrict__ mask_px, float* __restrict__ mask_sum,
                                                   int n_rays, float w_main, float w_ent, float w_mask, float ls,
                                                   float* __restrict__ g_rgbm, float* __restrict__ g_alast,
                                                   float* __restrict__ g_cw, float* __restrict__ loss_out,
                                                   const float* __restrict__ batch_norm) {
  __shared__ float sm[4];
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  float l_mse = 0.f, l_ent = 0.f, l_bce = 0.f;
  // number of masked-in pixels of the batch: every block sums the (few KB of) masks itself, in k_sum's order, instead of
  // waiting for a separate single-block kernel
  float part = 0.f;
  for (int i = threadIdx.x; i < n_rays; i += 256) part += mask_px[i];
  const float msum_local = block_sum256(part, sm);
  if (blockIdx.x == 0 && threadIdx.x == 0) mask_sum[0] = msum_local;
  // ray-sharded data parallelism: the masked MSE is normalised by the UNION batch's masked-pixel count / world size
  const float msum = batch_norm ? batch_norm[0] : msum_local;
  const float invN = 1.f / (float)n_rays;
  if (r < n_rays) {
    float y = mask_px[r];
    for (int k = 0; k < 3; ++k) {
      float d = rgbm[r * 3 + k] * y - target[r * 3 + k] * y;
      l_mse += d * d;
      g_rgbm[r * 3 + k] = ls * w_main * 2.f * d * y / (msum * 3.f);
    }
    float a = alast[r];
    float p = fminf(fmaxf(a, 1e-6f), 1.f - 1e-6f);
    float lp = logf(p), lq = logf(1.f - p);
    l_ent = -(p * lp + (1.f - p) * lq);
    g_alast[r] = (a >= 1e-6f && a <= 1.f - 1e-6f) ? ls * w_ent * (-(lp - lq)) * invN : 0.f;
    float c0 = cw[r];
    float c = fminf(fmaxf(c0, 1e-3f), 1.f - 1e-3f);
    l_bce = -(y * logf(c) + (1.f - y) * logf(1.f - c));
    g_cw[r] = (c0 >= 1e-3f && c0 <= 1.f - 1e-3f) ? ls * w_mask * (-(y / c) + (1.f - y) / (1.f - c)) * invN : 0.f;
  }
  l_mse = block_sum256(l_mse, sm);
  l_ent = block_sum256(l_ent, sm);
  l_bce = block_sum256(l_bce, sm);
  if (threadIdx.x == 0 && loss_out) {
    atomicAdd(&loss_out[0], l_mse / (msum * 3.f));
    atomicAdd(&loss_out[1], l_ent * invN);
    atomicAdd(&loss_out[6], l_bce * invN);
    // [7]: the WEIGHTED sum of all terms (both kernels add their share): object_losses' `loss` without the TV term
    atomicAdd(&loss_out[7], w_main * (l_mse / (msum * 3.f)) + w_ent * (l_ent * invN) + w_mask * (l_bce * invN));
  }
}

__global__ __launch_bounds__(256) void k_loss_samples(const float* __restrict__ gradient, const float* __restrict__ gdef,
                                                      const float* __restrict__ warp_out, const float* __restrict__ sdef,
                                                      const int32_t* __restrict__ count, int capacity, float w_eik,
                                                      float w_dyn, float ls, float* __restrict__ g_gradient,
                                                      float* __restrict__ g_gdef, float* __restrict__ g_corr,
                                                      float* __restrict__ g_sdef, float* __restrict__ loss_out,
                                                      const float* __restrict__ batch_norm) {
  __shared__ float sm[4];
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  float invM = M > 0 ? 1.f / (float)M : 0.f;
  if (batch_norm) invM = batch_norm[1] > 0.f ? 1.f / batch_norm[1] : 0.f;   // union batch's sample count / world size
  float l_eik = 0.f, l_gd = 0.f, l_c = 0.f, l_sd = 0.f;
  if (m < M) {
    float g[3] = {gradient[m * 3], gradient[m * 3 + 1], gradient[m * 3 + 2]};
    float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    float e = gn - 1.f;
    l_eik = fabsf(e);
    float sg = (e > 0.f) ? 1.f : (e < 0.f ? -1.f : 0.f);
    for (int k = 0; k < 3; ++k) g_gradient[m * 3 + k] += (gn > 0.f) ? ls * w_eik * sg * g[k] / gn * invM : 0.f;
    for (int i = 0; i < 3; ++i) {
      float a0 = gdef[m * 9 + i * 3], a1 = gdef[m * 9 + i * 3 + 1], a2 = gdef[m * 9 + i * 3 + 2];
      float an = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
      l_gd += an;
      float s = (an > 0.f) ? ls * w_dyn * invM / (3.f * an) : 0.f;
      g_gdef[m * 9 + i * 3] = a0 * s; g_gdef[m * 9 + i * 3 + 1] = a1 * s; g_gdef[m * 9 + i * 3 + 2] = a2 * s;
    }
    float c = warp_out[(size_t)m * 16 + 3];
    l_c = fabsf(c);
    g_corr[m] = ls * w_dyn * invM * ((c > 0.f) ? 1.f : (c < 0.f ? -1.f : 0.f));
    float sd = sdef[m];
    l_sd = fabsf(sd);
    g_sdef[m] = ls * w_dyn * invM * ((sd > 0.f) ? 1.f : (sd < 0.f ? -1.f : 0.f));
  }
  l_eik = block_sum256(l_eik, sm);
  l_gd = block_sum256(l_gd, sm);
  l_c = block_sum256(l_c, sm);
  l_sd = block_sum256(l_sd, sm);
  if (threadIdx.x == 0 && loss_out && m - (int)threadIdx.x < M) {
    atomicAdd(&loss_out[2], l_eik * invM);
    atomicAdd(&loss_out[3], l_gd * invM / 3.f);
    atomicAdd(&loss_out[4], l_c * invM);
    atomicAdd(&loss_out[5], l_sd * invM);
    atomicAdd(&loss_out[7], w_eik * (l_eik * invM) + w_dyn * ((l_gd * invM / 3.f + l_c * invM) + l_sd * invM));
  }
}

extern "C" int pp_loss_rays(const float* rgb_marched, const float* alphainv_last, const float* cum_weights,
                            const float* target, const float* mask_px, float* mask_sum, int32_t n_rays, float w_main,
                            float w_entropy, float w_mask, float loss_scale, float* g_rgb_marched,
                            float* g_alphainv_last, float* g_cum_weights, float* loss_out, const float* batch_norm,
                            void* stream) {
  PP_REQUIRE(rgb_marched && alphainv_last && cum_weights && target && mask_px && mask_sum && g_rgb_marched &&
                 g_alphainv_last && g_cum_weights,
             "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipStream_t st = pp_stream(stream);
  hipLaunchKernelGGL(k_loss_rays, dim3(pp_div_up(n_rays, 256)), dim3(256), 0, st, rgb_marched, alphainv_last,
                     cum_weights, target, mask_px, mask_sum, n_rays, w_main, w_entropy, w_mask, loss_scale,
                     g_rgb_marched, g_alphainv_last, g_cum_weights, loss_out, batch_norm);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_loss_samples(const float* gradient, const float* grad_deform, const float* warp_out,
                               const float* sdf_deform, const int32_t* count, int32_t capacity, float w_eikonal,
                               float w_deform, float loss_scale, float* g_gradient, float* g_grad_deform,
                               float* g_correction, float* g_sdf_deform, float* loss_out, const float* batch_norm,
                               void* stream) {
  PP_REQUIRE(gradient && grad_deform && warp_out && sdf_deform && count && g_gradient && g_grad_deform &&
                 g_correction && g_sdf_deform,
             "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  hipLaunchKernelGGL(k_loss_samples, dim3(pp_div_up(capacity, 256)), dim3(256), 0, pp_stream(stream), gradient,
                     grad_deform, warp_out, sdf_deform, count, capacity, w_eikonal, w_deform, loss_scale, g_gradient,
                     g_grad_deform, g_correction, g_sdf_deform, loss_out, batch_norm);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
