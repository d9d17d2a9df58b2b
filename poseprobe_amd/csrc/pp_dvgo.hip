// DVGO-surface operators of the reference's native extensions that the PoseProbe live loop never calls but that
// belong to the extension API (lib/cuda/render_utils.cpp:170-184, adam_upd.cpp, total_variation.cpp, ub360_utils.cpp).
// Elementwise / per-ray, HBM bound, grid-stride; semantics (incl. the reference's quirks) restated from the .cu text.
#include "pp_common.h"

#define GS_LOOP(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int gs_blocks(long long n) { long long b = (n + 255) / 256; return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096); }

// ---- raw2alpha (render_utils_kernel.cu:431-574): alpha = 1 - (1 + exp(d + shift))^(-interval) -------------------
__global__ void k_raw2alpha(const float* __restrict__ density, float shift, float interval,
                            const float* __restrict__ interval_v, long long n, float* __restrict__ exp_d,
                            float* __restrict__ alpha) {
  GS_LOOP(i, n) {
    float e = expf(density[i] + shift);       // may be inf, as in the reference
    float iv = interval_v ? interval_v[i] : interval;
    exp_d[i] = e;
    alpha[i] = 1.f - powf(1.f + e, -iv);
  }
}
__global__ void k_raw2alpha_bwd(const float* __restrict__ exp_d, const float* __restrict__ grad_back, float interval,
                                const float* __restrict__ interval_v, long long n, float* __restrict__ grad) {
  GS_LOOP(i, n) {
    float e = exp_d[i];
    float iv = interval_v ? interval_v[i] : interval;
    grad[i] = fminf(e, 1e10f) * powf(1.f + e, -iv - 1.f) * iv * grad_back[i];
  }
}

// ---- maskcache_lookup (render_utils_kernel.cu:374-424): nearest-voxel bool lookup, out-of-grid -> false -----------
__global__ void k_maskcache_lookup(const uint8_t* __restrict__ world, const float* __restrict__ xyz, int sx, int sy, int sz,
                                   float s0, float s1, float s2, float t0, float t1, float t2, long long n,
                                   uint8_t* __restrict__ out) {
  GS_LOOP(p, n) {
    int i = (int)roundf(xyz[p * 3] * s0 + t0), j = (int)roundf(xyz[p * 3 + 1] * s1 + t1), k = (int)roundf(xyz[p * 3 + 2] * s2 + t2);
    bool in = (0 <= i) && (i < sx) && (0 <= j) && (j < sy) && (0 <= k) && (k < sz);
    out[p] = in ? world[((size_t)i * sy + j) * sz + k] : 0;
  }
}

// ---- NDC / inverse-sphere background samplers (render_utils_kernel.cu:245-360) ------------------------------------
__global__ void k_sample_ndc(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float mn0, float mn1,
                             float mn2, float mx0, float mx1, float mx2, int S, long long n, float* __restrict__ pts,
                             uint8_t* __restrict__ mask_out) {
  GS_LOOP(idx, n) {
    long long r = idx / S;
    int s = (int)(idx - r * S);
    float dist = ((float)s) / (float)(S - 1);
    float px = rays_o[r * 3] + rays_d[r * 3] * dist, py = rays_o[r * 3 + 1] + rays_d[r * 3 + 1] * dist,
          pz = rays_o[r * 3 + 2] + rays_d[r * 3 + 2] * dist;
    pts[idx * 3] = px; pts[idx * 3 + 1] = py; pts[idx * 3 + 2] = pz;
    mask_out[idx] = (mn0 > px) | (mn1 > py) | (mn2 > pz) | (mx0 < px) | (mx1 < py) | (mx2 < pz);
  }
}
__global__ void k_sample_bg(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            const float* __restrict__ t_max, float bg_preserve, int S, long long n,
                            float* __restrict__ pts) {
  GS_LOOP(idx, n) {
    long long r = idx / S;
    int s = (int)(idx - r * S);
    // double-typed literals of the .cu (`1.`) are kept: the outer radius is evaluated in double then rounded
    float t_outer0 = (float)((double)t_max[r] - 1. + 1. / (1. - (double)(((float)s) / (float)S)));
    float x = rays_o[r * 3] + rays_d[r * 3] * t_outer0, y = rays_o[r * 3 + 1] + rays_d[r * 3 + 1] * t_outer0,
          z = rays_o[r * 3 + 2] + rays_d[r * 3 + 2] * t_outer0;
    float t_outer = sqrtf(x * x + y * y + z * z);
    float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    float R = t_outer / m;
    float o2i = (float)((double)(R * R / (t_outer * t_outer)) * (1. - (double)bg_preserve) + (double)(R / t_outer * bg_preserve));
    pts[idx * 3] = x * o2i; pts[idx * 3 + 1] = y * o2i; pts[idx * 3 + 2] = z * o2i;
  }
}

// ---- adam_upd / masked_adam_upd / adam_upd_with_perlr (adam_upd_kernel.cu:8-133) -----------------------------------
template <int MODE>   // 0 plain, 1 skip where grad == 0, 2 per-element lr multiplier
__global__ void k_adam_upd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                           const float* __restrict__ perlr, long long n, float step_size, float b1, float b2, float eps) {
  GS_LOOP(i, n) {
    float gi = g[i];
    if (MODE == 1 && gi == 0.f) continue;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float lr = (MODE == 2) ? step_size * perlr[i] : step_size;
    p[i] -= lr * mi / (sqrtf(vi) + eps);
  }
}

// The same update for up to 32 SMALL tensors of one optimiser group in one launch (an optimiser step over the two MLPs is 20
// tensors of 3 .. 16 384 floats: one kernel instead of twenty 5 us launches); blockIdx.y = tensor.  Same arithmetic as k_adam_upd<0>.
struct AdamJobs { float* p[32]; const float* g[32]; float* m[32]; float* v[32]; int n[32]; };
__global__ void k_adam_upd_multi(AdamJobs J, float step_size, float b1, float b2, float eps) {
  const int q = blockIdx.y, n = J.n[q];
  float* __restrict__ p = J.p[q];
  const float* __restrict__ g = J.g[q];
  float* __restrict__ m = J.m[q];
  float* __restrict__ v = J.v[q];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) + eps);
  }
}

// ---- total_variation_add_grad{,_new} (total_variation_kernel.cu:13-134) on the channels-last layout ---------------
// Reference quirk kept: the unmasked kernel weights the LAST axis and the FIRST axis both with wz (wx is unused);
// the masked ("_new") kernel uses wx for the last axis, wy, and wz for the first.
template <bool MASKED>
__global__ void k_tv_add_grad(const float* __restrict__ p, float* __restrict__ grad, const float* __restrict__ mask,
                              int X, int Y, int Z, int C, float wx, float wy, float wz, int dense_mode, long long n) {
  GS_LOOP(e, n) {
    if (!dense_mode && grad[e] == 0.f) continue;
    long long vox = e / C;
    int z = (int)(vox % Z);
    long long t = vox / Z;
    int y = (int)(t % Y), x = (int)(t / Y);
    const long long sz = C, sy = (long long)Z * C, sx = (long long)Y * Z * C;
    const float wk = MASKED ? wx : wz;
    float v = p[e], me = MASKED ? mask[e] : 1.f, add = 0.f;
    auto term = [&](long long o, float w) {
      float d = fminf(fmaxf(v - p[e + o], -1.f), 1.f);
      return w * d * (MASKED ? me * mask[e + o] : 1.f);
    };
    if (z > 0) add += term(-sz, wk);
    if (z < Z - 1) add += term(sz, wk);
    if (y > 0) add += term(-sy, wy);
    if (y < Y - 1) add += term(sy, wy);
    if (x > 0) add += term(-sx, wz);
    if (x < X - 1) add += term(sx, wz);
    grad[e] += add;
  }
}

// ---- cumdist_thres (ub360_utils_kernel.cu:12-48): per-ray running distance threshold ----------------------------
__global__ void k_cumdist_thres(const float* __restrict__ dist, float thres, int n_rays, int n_pts, uint8_t* __restrict__ mask) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float cum = 0.f;
  for (int i = 0; i < n_pts; ++i) {
    cum += dist[(size_t)r * n_pts + i];
    bool over = cum > thres;
    if (over) cum = 0.f;
    mask[(size_t)r * n_pts + i] = over;
  }
}

#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(gs_blocks(n)), dim3(256), 0, pp_stream(stream), __VA_ARGS__)

extern "C" int pp_raw2alpha_fwd(const float* density, float shift, float interval, const float* interval_v, int32_t n,
                                float* exp_d, float* alpha, void* stream) {
  PP_REQUIRE(density && exp_d && alpha, "null pointer");
  if (n <= 0) return PP_OK;
  LAUNCH(k_raw2alpha, n, density, shift, interval, interval_v, (long long)n, exp_d, alpha);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_raw2alpha_bwd(const float* exp_d, const float* grad_back, float interval, const float* interval_v,
                                int32_t n, float* grad, void* stream) {
  PP_REQUIRE(exp_d && grad_back && grad, "null pointer");
  if (n <= 0) return PP_OK;
  LAUNCH(k_raw2alpha_bwd, n, exp_d, grad_back, interval, interval_v, (long long)n, grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_maskcache_lookup(const uint8_t* world, const float* xyz, int32_t size_x, int32_t size_y, int32_t size_z,
                                   float scale_x, float scale_y, float scale_z, float shift_x, float shift_y,
                                   float shift_z, int32_t n, uint8_t* out, void* stream) {
  PP_REQUIRE(world && xyz && out, "null pointer");
  if (n <= 0) return PP_OK;
  LAUNCH(k_maskcache_lookup, n, world, xyz, size_x, size_y, size_z, scale_x, scale_y, scale_z, shift_x, shift_y, shift_z,
         (long long)n, out);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_sample_ndc(const pp_scene* sc, const float* rays_o, const float* rays_d, int32_t n_rays,
                             int32_t n_samples, float* pts, uint8_t* mask_outbbox, void* stream) {
  PP_REQUIRE(sc && rays_o && rays_d && pts && mask_outbbox, "null pointer");
  PP_REQUIRE(n_samples >= 2, "n_samples must be >= 2");
  long long n = (long long)n_rays * n_samples;
  if (n <= 0) return PP_OK;
  LAUNCH(k_sample_ndc, n, rays_o, rays_d, sc->xyz_min[0], sc->xyz_min[1], sc->xyz_min[2], sc->xyz_max[0], sc->xyz_max[1],
         sc->xyz_max[2], n_samples, n, pts, mask_outbbox);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_sample_bg(const float* rays_o, const float* rays_d, const float* t_max, float bg_preserve,
                            int32_t n_rays, int32_t n_samples, float* pts, void* stream) {
  PP_REQUIRE(rays_o && rays_d && t_max && pts, "null pointer");
  long long n = (long long)n_rays * n_samples;
  if (n <= 0) return PP_OK;
  LAUNCH(k_sample_bg, n, rays_o, rays_d, t_max, bg_preserve, n_samples, n, pts);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* perlr,
                           int32_t n, int32_t step, float beta1, float beta2, float lr, float eps, int32_t mode,
                           void* stream) {
  PP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
  PP_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || perlr) && step >= 1, "bad mode / step");
  if (n <= 0) return PP_OK;
  // adam_upd_kernel.cu:72: step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in float
  const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
  if (mode == 0) LAUNCH((k_adam_upd<0>), n, param, grad, exp_avg, exp_avg_sq, perlr, (long long)n, step_size, beta1, beta2, eps);
  else if (mode == 1) LAUNCH((k_adam_upd<1>), n, param, grad, exp_avg, exp_avg_sq, perlr, (long long)n, step_size, beta1, beta2, eps);
  else LAUNCH((k_adam_upd<2>), n, param, grad, exp_avg, exp_avg_sq, perlr, (long long)n, step_size, beta1, beta2, eps);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_adam_upd_multi(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host,
                                 float* const* exp_avg_sq_host, const int32_t* sizes_host, int32_t n_tensors, int32_t step,
                                 float beta1, float beta2, float lr, float eps, void* stream) {
  PP_REQUIRE(params_host && grads_host && exp_avg_host && exp_avg_sq_host && sizes_host, "null pointer");
  PP_REQUIRE(n_tensors >= 1 && n_tensors <= 32 && step >= 1, "1 <= n_tensors <= 32, step >= 1");
  AdamJobs J;
  int largest = 0;
  for (int i = 0; i < n_tensors; ++i) {
    PP_REQUIRE(params_host[i] && grads_host[i] && exp_avg_host[i] && exp_avg_sq_host[i] && sizes_host[i] > 0, "null tensor or empty size");
    J.p[i] = params_host[i]; J.g[i] = grads_host[i]; J.m[i] = exp_avg_host[i]; J.v[i] = exp_avg_sq_host[i]; J.n[i] = sizes_host[i];
    largest = sizes_host[i] > largest ? sizes_host[i] : largest;
  }
  const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));   // as pp_adam_upd
  const int bx = pp_div_up(largest, 256) < 64 ? pp_div_up(largest, 256) : 64;
  hipLaunchKernelGGL(k_adam_upd_multi, dim3(bx, n_tensors), dim3(256), 0, pp_stream(stream), J, step_size, beta1, beta2, eps);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_tv_add_grad(const float* param_cl, float* grad_cl, const float* mask_cl, int32_t size_x, int32_t size_y,
                              int32_t size_z, int32_t channels, float wx, float wy, float wz, int32_t dense_mode,
                              void* stream) {
  PP_REQUIRE(param_cl && grad_cl, "null pointer");
  long long n = (long long)size_x * size_y * size_z * channels;
  if (n <= 0) return PP_OK;
  if (mask_cl) LAUNCH((k_tv_add_grad<true>), n, param_cl, grad_cl, mask_cl, size_x, size_y, size_z, channels, wx, wy, wz, dense_mode, n);
  else LAUNCH((k_tv_add_grad<false>), n, param_cl, grad_cl, mask_cl, size_x, size_y, size_z, channels, wx, wy, wz, dense_mode, n);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_cumdist_thres(const float* dist, float thres, int32_t n_rays, int32_t n_pts, uint8_t* mask, void* stream) {
  PP_REQUIRE(dist && mask, "null pointer");
  if (n_rays <= 0 || n_pts <= 0) return PP_OK;
  hipLaunchKernelGGL(k_cumdist_thres, dim3(pp_div_up(n_rays, 64)), dim3(64), 0, pp_stream(stream), dist, thres, n_rays, n_pts, mask);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
