// Transmittance scan (alpha -> weights) and fused compositing, forward and backward.
// One wavefront per ray: lanes hold 64 consecutive samples (coalesced), the running transmittance is advanced in
// the reference's *sequential* order inside the wave (v_readlane broadcast), so the 1e-3 early stop and every
// rounding of lib/cuda/render_utils_kernel.cu:577-605 / :654-707 are reproduced exactly while memory traffic is
// coalesced (the reference walks each ray from a single thread).
#include "pp_common.h"

// value of lane j (j wave-uniform): v_readlane_b32 instead of the ds_bpermute_b32 that __shfl compiles to - inside the sequential
// loops below an LDS crossbar round trip per sample was most of the iteration (k_march_fwd 17.6 us for a 186-sample ray)
__device__ __forceinline__ float lane_value(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ double lane_value(double v, int j) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}

template <bool FUSED, bool DVGO>
__global__ __launch_bounds__(256) void k_march_fwd(const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                   const float* __restrict__ step_w, const float* __restrict__ nrm_in,
                                                   const int32_t* __restrict__ ray_start, int n_rays, float bg,
                                                   float* __restrict__ weights, float* __restrict__ Tout,
                                                   float* __restrict__ alphainv_last, int32_t* __restrict__ i_end,
                                                   float* __restrict__ rgb_marched, float* __restrict__ rgb_pre,
                                                   float* __restrict__ cum_weights, float* __restrict__ depth_acc,
                                                   float* __restrict__ normal_marched) {
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the loop control below is scalar
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  const int b = ray_start[r], e = ray_start[r + 1];
  float Tc = 1.f;
  int stop = e;            // absolute index one past the last sample that received a weight
  bool stopped = false;
  float acc_rgb[3] = {0, 0, 0}, acc_w = 0.f, acc_d = 0.f, acc_n[3] = {0, 0, 0};
  float a_next = (b + lane < e) ? alpha[b + lane] : 0.f;
  for (int c0 = b; c0 < e; c0 += 64) {
    int i = c0 + lane;
    const float a = a_next;
    // everything the chunk needs from memory is requested before the sequential loop (and the next chunk's alpha with it): the
    // loop is ~100 ns per sample of pure latency, the loads land underneath it
    a_next = (i + 64 < e) ? alpha[i + 64] : 0.f;
    float cr[3] = {0.f, 0.f, 0.f}, cs = 0.f, cn[3] = {0.f, 0.f, 0.f};
    if (FUSED && i < e) {
      if (rgb) for (int k = 0; k < 3; ++k) cr[k] = rgb[i * 3 + k];
      if (step_w) cs = step_w[i];
      if (nrm_in) for (int k = 0; k < 3; ++k) cn[k] = nrm_in[i * 3 + k];
    }
    float myT = 1.f, myw = 0.f;
    if (!stopped) {
      int n = e - c0 < 64 ? e - c0 : 64;
      // the transmittance is a sequential product; a sample's factor (1 - alpha, in double for the reference's arithmetic) is
      // formed by its own lane before the loop, and its weight T * alpha after it: the serial part is one multiply per sample
      const double omd = 1.0 - (double)a;
      const float omf = fmaxf(1.f - a, 1e-10f);
      bool reached = false;
      for (int j = 0; j < n; ++j) {
        if (lane == j) { myT = Tc; reached = true; }
        if (DVGO) {           // cumprod_exclusive: p.clamp_min(1e-10).cumprod(-1), fp32, no early stop (dvgo_ori.py:478-485)
          Tc = Tc * lane_value(omf, j);
        } else {
          Tc = (float)((double)Tc * lane_value(omd, j));
          if (Tc < 1e-3f) { stop = c0 + j + 1; stopped = true; break; }   // == ((double)Tc < 1e-3): float(1e-3) is the first float above it
        }
      }
      if (reached) myw = myT * a;
    }
    if (i < e) {
      weights[i] = myw;
      if (Tout) Tout[i] = myT;
      if (FUSED) {
        acc_w += myw;
        if (rgb) for (int k = 0; k < 3; ++k) acc_rgb[k] += myw * cr[k];
        if (step_w) acc_d += myw * cs;
        if (nrm_in) for (int k = 0; k < 3; ++k) acc_n[k] += myw * cn[k];
      }
    }
  }
  if (FUSED) {
    acc_w = pp_wave_sum(acc_w);
    acc_d = pp_wave_sum(acc_d);
    for (int k = 0; k < 3; ++k) { acc_rgb[k] = pp_wave_sum(acc_rgb[k]); acc_n[k] = pp_wave_sum(acc_n[k]); }
  }
  if (lane == 0) {
    alphainv_last[r] = Tc;
    i_end[r] = stop;
    if (FUSED) {
      cum_weights[r] = acc_w;
      if (depth_acc) depth_acc[r] = acc_d;
      for (int k = 0; k < 3; ++k) {
        float pre = acc_rgb[k] + (1.f - acc_w) * bg;
        if (rgb_pre) rgb_pre[r * 3 + k] = pre;
        if (rgb_marched) rgb_marched[r * 3 + k] = fminf(fmaxf(pre, 0.f), 1.f);
        if (normal_marched) normal_marched[r * 3 + k] = acc_n[k];
      }
    }
  }
}

template <bool FUSED>
__global__ __launch_bounds__(256) void k_march_bwd(const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                   const float* __restrict__ step_w, const float* __restrict__ weights,
                                                   const float* __restrict__ Tin, const float* __restrict__ alphainv_last,
                                                   const int32_t* __restrict__ ray_start, const int32_t* __restrict__ i_end,
                                                   int n_rays, float bg, const float* __restrict__ rgb_pre,
                                                   const float* __restrict__ g_rgbm, const float* __restrict__ g_cw,
                                                   const float* __restrict__ g_last, const float* __restrict__ g_depth,
                                                   const float* __restrict__ g_w_in, float* __restrict__ g_alpha,
                                                   float* __restrict__ g_rgb) {
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  const int b = ray_start[r], e = ray_start[r + 1], stop = i_end[r];
  float gm[3] = {0, 0, 0}, gcw = 0.f, gd = 0.f;
  if (FUSED) {
    for (int k = 0; k < 3; ++k) {
      float pre = rgb_pre ? rgb_pre[r * 3 + k] : 0.5f;
      float g = g_rgbm ? g_rgbm[r * 3 + k] : 0.f;
      gm[k] = (pre >= 0.f && pre <= 1.f) ? g : 0.f;
    }
    gcw = (g_cw ? g_cw[r] : 0.f) - bg * (gm[0] + gm[1] + gm[2]);
    gd = g_depth ? g_depth[r] : 0.f;
  }
  float back = (g_last ? g_last[r] : 0.f) * alphainv_last[r];
  int nchunk = (e - b + 63) / 64;
  // a chunk's operands are requested one chunk ahead, so that they land under the sequential loop of the chunk before
  struct Chunk { float a, w, T, gw, v[3], sw; };
  auto fetch = [&](int c) {
    Chunk k;
    const int i = b + c * 64 + lane;
    const bool live = i < e;
    k.a = live ? alpha[i] : 0.f;
    k.w = live ? weights[i] : 0.f;
    k.T = live ? Tin[i] : 1.f;
    k.gw = (live && g_w_in) ? g_w_in[i] : 0.f;
    k.v[0] = k.v[1] = k.v[2] = 0.f; k.sw = 0.f;
    if (FUSED && live) {
      if (rgb) for (int q = 0; q < 3; ++q) k.v[q] = rgb[i * 3 + q];
      if (step_w) k.sw = step_w[i];
    }
    return k;
  };
  Chunk nx = fetch(nchunk > 0 ? nchunk - 1 : 0);
  for (int c = nchunk - 1; c >= 0; --c) {
    int c0 = b + c * 64;
    int i = c0 + lane;
    bool live = i < e;
    const Chunk ck = nx;
    if (c > 0) nx = fetch(c - 1);
    const float a = ck.a, w = ck.w, T = ck.T;
    float gw = ck.gw;
    if (FUSED && live) {
      float wr = 0.f;
      if (rgb) for (int k = 0; k < 3; ++k) { wr += gm[k] * ck.v[k]; if (g_rgb) g_rgb[i * 3 + k] = w * gm[k]; }
      gw += wr + gcw + (step_w ? gd * ck.sw : 0.f);
    }
    int hi = stop - c0;           // samples [c0, stop) of this chunk take part
    if (hi > 64) hi = 64;
    // the running sum `back` is sequential (fp32, last sample first); everything else of a sample's gradient - the double-precision
    // division included - only needs the value `back` had when the sequence reached it: the serial loop is one add per sample and
    // hands lane j its value, the rest is evaluated by all lanes at once (same operations per sample, same results)
    const float gww = gw * w;
    float myback = 0.f;
    for (int j = hi - 1; j >= 0; --j) {
      if (lane == j) myback = back;
      back += lane_value(gww, j);
    }
    float ga = 0.f;
    if (lane < hi) {
      const float gwT = gw * T;
      const double denom = (double)(1.f - a) + 1e-10;
      ga = (float)((double)gwT - (double)myback / denom);
    }
    if (live) g_alpha[i] = ga;
  }
}

extern "C" int pp_alpha2weight_fwd(const float* alpha, const int32_t* ray_start, int32_t n_rays, float* weights,
                                   float* T, float* alphainv_last, int32_t* i_end, void* stream) {
  PP_REQUIRE(alpha && ray_start && weights && T && alphainv_last && i_end, "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipLaunchKernelGGL((k_march_fwd<false, false>), dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), alpha, nullptr,
                     nullptr, nullptr, ray_start, n_rays, 0.f, weights, T, alphainv_last, i_end, nullptr, nullptr,
                     nullptr, nullptr, nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_alpha2weight_bwd(const float* alpha, const float* weights, const float* T, const float* alphainv_last,
                                   const int32_t* ray_start, const int32_t* i_end, int32_t n_rays,
                                   const float* grad_weights, const float* grad_last, float* grad_alpha, void* stream) {
  PP_REQUIRE(alpha && weights && T && alphainv_last && ray_start && i_end && grad_weights && grad_last && grad_alpha,
             "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipLaunchKernelGGL((k_march_bwd<false>), dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), alpha, nullptr,
                     nullptr, weights, T, alphainv_last, ray_start, i_end, n_rays, 0.f, nullptr, nullptr, nullptr,
                     grad_last, nullptr, grad_weights, grad_alpha, nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_march_fwd(const float* alpha, const float* rgb, const float* step_w, const float* nrm_in,
                            const int32_t* ray_start, int32_t n_rays, float bg, float* weights, float* T,
                            float* alphainv_last, int32_t* i_end, float* rgb_marched, float* rgb_pre,
                            float* cum_weights, float* depth_acc, float* normal_marched, void* stream) {
  PP_REQUIRE(alpha && rgb && ray_start && weights && T && alphainv_last && i_end && cum_weights, "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipLaunchKernelGGL((k_march_fwd<true, false>), dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), alpha, rgb,
                     step_w, nrm_in, ray_start, n_rays, bg, weights, T, alphainv_last, i_end, rgb_marched, rgb_pre,
                     cum_weights, depth_acc, normal_marched);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_march_bwd(const float* alpha, const float* rgb, const float* step_w, const float* weights,
                            const float* T, const float* alphainv_last, const int32_t* ray_start,
                            const int32_t* i_end, int32_t n_rays, float bg, const float* rgb_pre,
                            const float* g_rgb_marched, const float* g_cum_weights, const float* g_alphainv_last,
                            const float* g_depth_acc, const float* g_weights, float* grad_alpha, float* grad_rgb,
                            void* stream) {
  PP_REQUIRE(alpha && rgb && weights && T && alphainv_last && ray_start && i_end && grad_alpha, "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipLaunchKernelGGL((k_march_bwd<true>), dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), alpha, rgb,
                     step_w, weights, T, alphainv_last, ray_start, i_end, n_rays, bg, rgb_pre, g_rgb_marched,
                     g_cum_weights, g_alphainv_last, g_depth_acc, g_weights, grad_alpha, grad_rgb);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Surface-point query: first sign change of the SDF along a ray + linear zero crossing
// (Voxurf.query_sdf_point_wocuda / _wodeform, lib/voxurf_coarse.py:766-795, :809-837).  One wavefront per ray.
// Compact mode (ray_start != NULL): sdf[M] holds the in-bbox samples, the dense row is rebuilt with the reference's
// default value 1 for out-of-bbox slots (:752).  Dense mode: sdf is already [N,S].
// ------------------------------------------------------------------------------------------------------------------
#define PP_MAX_S 1024
__global__ __launch_bounds__(256) void k_first_crossing(const float* __restrict__ sdf, const int32_t* __restrict__ ray_start,
                                                        const int32_t* __restrict__ step_k, int n_rays, int S, float dist,
                                                        const float* __restrict__ t_min, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, float* __restrict__ sdf_dense,
                                                        float* __restrict__ pts, uint8_t* __restrict__ mask,
                                                        float* __restrict__ zval) {
  __shared__ float rows[4][PP_MAX_S];
  int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int r = blockIdx.x * 4 + wid;
  if (r >= n_rays) return;
  float* row = rows[wid];
  if (ray_start) {
    for (int k = lane; k < S; k += 64) row[k] = 1.f;
    __builtin_amdgcn_wave_barrier();
    for (int i = ray_start[r] + lane; i < ray_start[r + 1]; i += 64) row[step_k[i]] = sdf[i];
  } else {
    for (int k = lane; k < S; k += 64) row[k] = sdf[(size_t)r * S + k];
  }
  __builtin_amdgcn_wave_barrier();
  if (sdf_dense) for (int k = lane; k < S; k += 64) sdf_dense[(size_t)r * S + k] = row[k];
  int prev = 0;                                        // argmax of an all-zero row is 0 (voxurf_coarse.py:771)
  for (int k0 = 0; k0 < S - 1; k0 += 64) {
    int k = k0 + lane;
    bool hit = (k < S - 1) && (row[k] * row[k + 1] <= 0.f);
    unsigned long long bal = __ballot(hit);
    if (bal) { prev = k0 + __ffsll((long long)bal) - 1; break; }
  }
  if (lane == 0) {
    float s1 = row[prev], s2 = row[prev + 1];
    float z1 = (float)prev * dist + dist * 0.5f, z2 = (float)(prev + 1) * dist + dist * 0.5f;
    float z0 = (s1 * z2 - s2 * z1) / (s1 - s2 + 1e-10f);
    if (z0 < z1) z0 = 0.f;
    if (z0 > z2) z0 = 0.f;
    bool ok = (z0 > 1e-10f) && (s1 * s2 < 0.f);
    float dx = rays_d[r * 3], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
    float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    float interpx = t_min[r] + z0 / nrm;
    pts[r * 3] = rays_o[r * 3] + dx * interpx;
    pts[r * 3 + 1] = rays_o[r * 3 + 1] + dy * interpx;
    pts[r * 3 + 2] = rays_o[r * 3 + 2] + dz * interpx;
    mask[r] = ok ? 1 : 0;
    if (zval) zval[r] = z0;
  }
}

extern "C" int pp_sdf_first_crossing(const float* sdf, const int32_t* ray_start, const int32_t* step_k, int32_t n_rays,
                                     int32_t n_samples, float dist, const float* t_min, const float* rays_o,
                                     const float* rays_d, float* sdf_dense, float* pts, uint8_t* mask, float* zval,
                                     void* stream) {
  PP_REQUIRE(sdf && t_min && rays_o && rays_d && pts && mask, "null pointer");
  PP_REQUIRE((ray_start == nullptr) == (step_k == nullptr), "ray_start and step_k go together");
  PP_REQUIRE(n_rays > 0 && n_samples >= 2 && n_samples <= PP_MAX_S, "need n_rays>0 and 2<=n_samples<=1024");
  hipLaunchKernelGGL(k_first_crossing, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), sdf, ray_start, step_k,
                     n_rays, n_samples, dist, t_min, rays_o, rays_d, sdf_dense, pts, mask, zval);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// DirectVoxGO compositing (lib/dvgo_ori.py:478-489, :361-366): exclusive cumprod of clamp_min(1-alpha, 1e-10) without
// early termination; rgb_marched = sum w*rgb + alphainv_last*bg (clamped), depth_acc = sum w*step_w.
extern "C" int pp_march_dvgo_fwd(const float* alpha, const float* rgb, const float* step_w, const int32_t* ray_start,
                                 int32_t n_rays, float* weights, float* T, float* alphainv_last, int32_t* i_end,
                                 float* rgb_acc, float* cum_weights, float* depth_acc, void* stream) {
  PP_REQUIRE(alpha && ray_start && weights && T && alphainv_last && i_end && cum_weights, "null pointer");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  hipLaunchKernelGGL((k_march_fwd<true, true>), dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), alpha, rgb,
                     step_w, nullptr, ray_start, n_rays, 0.f, weights, T, alphainv_last, i_end, nullptr, rgb_acc,
                     cum_weights, depth_acc, nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
