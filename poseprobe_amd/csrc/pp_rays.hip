// Pose algebra, index-first ray generation, dense / variable-length samplers with ray-major compaction.
// Reference behaviour restated (not translated) from: lib/camera.py:76-99,127-188; lib/recon_scene.py:62-74;
// lib/voxurf_coarse.py:1339-1368,1402-1407,697-719,936-945,661-695; lib/cuda/render_utils_kernel.cu:12-242.
#include "pp_common.h"

// ------------------------------------------------------------------------------------------------
// forward-mode dual numbers with 6 tangents (d/d se3) - one thread per view, a few hundred ops.
// ------------------------------------------------------------------------------------------------
// one tangent per thread: thread (view, k) carries d/d se3[k]
struct D6 {
  float v;
  float d;
};
__device__ __forceinline__ D6 d6c(float c) { D6 r; r.v = c; r.d = 0.f; return r; }
__device__ __forceinline__ D6 operator+(const D6& a, const D6& b) { D6 r; r.v = a.v + b.v; r.d = a.d + b.d; return r; }
__device__ __forceinline__ D6 operator-(const D6& a, const D6& b) { D6 r; r.v = a.v - b.v; r.d = a.d - b.d; return r; }
__device__ __forceinline__ D6 operator-(const D6& a) { D6 r; r.v = -a.v; r.d = -a.d; return r; }
__device__ __forceinline__ D6 operator*(const D6& a, const D6& b) { D6 r; r.v = a.v * b.v; r.d = a.d * b.v + a.v * b.d; return r; }
__device__ __forceinline__ D6 operator*(const D6& a, float s) { D6 r; r.v = a.v * s; r.d = a.d * s; return r; }

// Series of lib/camera.py:165-188 written in t = theta^2 (identical polynomial, smooth derivative at 0).
// kind 0: sin(x)/x, 1: (1-cos x)/x^2, 2: (x-sin x)/x^3 ; 11 terms.
__device__ D6 taylor_t(const D6& t, int kind) {
  D6 ans = d6c(0.f);
  D6 pw = d6c(1.f);
  double denom = 1.0;
  for (int i = 0; i <= 10; ++i) {
    if (kind == 0) { if (i > 0) denom *= (double)((2 * i) * (2 * i + 1)); }
    else if (kind == 1) denom *= (double)((2 * i + 1) * (2 * i + 2));
    else denom *= (double)((2 * i + 2) * (2 * i + 3));
    float c = (float)(((i & 1) ? -1.0 : 1.0) / denom);
    ans = ans + pw * c;
    pw = pw * t;
  }
  return ans;
}

__global__ void k_pose_fwd(const float* __restrict__ se3, const float* __restrict__ w2c_init,
                           const int32_t* __restrict__ refine_mask, int n_views, float* __restrict__ w2c,
                           float* __restrict__ c2w, float* __restrict__ jac) {
  int tidx = blockIdx.x * blockDim.x + threadIdx.x;
  int v = tidx / 6, kd = tidx - (tidx / 6) * 6;       // view, tangent direction
  if (v >= n_views) return;
  const float* P0 = w2c_init + v * 12;
  D6 Rn[3][3], tn[3];
  bool refine = (refine_mask == nullptr) || refine_mask[v] != 0;
  if (!refine) {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Rn[i][j] = d6c(P0[i * 4 + j]); tn[i] = d6c(P0[i * 4 + 3]); }
  } else {
    D6 w[3], u[3];
    for (int k = 0; k < 3; ++k) {
      w[k] = d6c(se3[v * 6 + k]); w[k].d = (kd == k) ? 1.f : 0.f;
      u[k] = d6c(se3[v * 6 + 3 + k]); u[k].d = (kd == 3 + k) ? 1.f : 0.f;
    }
    D6 t = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    D6 A = taylor_t(t, 0), B = taylor_t(t, 1), C = taylor_t(t, 2);
    D6 z = d6c(0.f);
    D6 wx[3][3] = {{z, -w[2], w[1]}, {w[2], z, -w[0]}, {-w[1], w[0], z}};
    D6 wx2[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) wx2[i][j] = wx[i][0] * wx[0][j] + wx[i][1] * wx[1][j] + wx[i][2] * wx[2][j];
    D6 R[3][3], Vm[3][3], tr[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      D6 eye = d6c(i == j ? 1.f : 0.f);
      R[i][j] = eye + A * wx[i][j] + B * wx2[i][j];
      Vm[i][j] = eye + B * wx[i][j] + C * wx2[i][j];
    }
    for (int i = 0; i < 3; ++i) tr[i] = Vm[i][0] * u[0] + Vm[i][1] * u[1] + Vm[i][2] * u[2];
    // compose_pair(pose_a = refine, pose_b = init): R = R_b R_a ; t = R_b t_a + t_b   (camera.py:92-99)
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rn[i][j] = R[0][j] * P0[i * 4 + 0] + R[1][j] * P0[i * 4 + 1] + R[2][j] * P0[i * 4 + 2];
      tn[i] = tr[0] * P0[i * 4 + 0] + tr[1] * P0[i * 4 + 1] + tr[2] * P0[i * 4 + 2] + d6c(P0[i * 4 + 3]);
    }
  }
  // invert: R^T, -R^T t (camera.py:76-82)
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      if (kd == 0) { w2c[v * 12 + i * 4 + j] = Rn[i][j].v; c2w[v * 12 + i * 4 + j] = Rn[j][i].v; }
      jac[(v * 12 + i * 4 + j) * 6 + kd] = Rn[j][i].d;
    }
    D6 ti = -(Rn[0][i] * tn[0] + Rn[1][i] * tn[1] + Rn[2][i] * tn[2]);
    if (kd == 0) { w2c[v * 12 + i * 4 + 3] = tn[i].v; c2w[v * 12 + i * 4 + 3] = ti.v; }
    jac[(v * 12 + i * 4 + 3) * 6 + kd] = ti.d;
  }
}

__global__ void k_pose_bwd(const float* __restrict__ jac, const float* __restrict__ c2w_grad, int n_views,
                           float* __restrict__ se3_grad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_views * 6) return;
  int v = t / 6, k = t % 6;
  float s = 0.f;
  for (int e = 0; e < 12; ++e) s += jac[(v * 12 + e) * 6 + k] * c2w_grad[v * 12 + e];
  se3_grad[t] = s;
}

extern "C" int pp_pose_fwd(const float* se3, const float* w2c_init, const int32_t* refine_mask, int32_t n_views,
                           float* w2c, float* c2w, float* jac, void* stream) {
  PP_REQUIRE(se3 && w2c_init && w2c && c2w && jac && n_views > 0, "null pointer or n_views<=0");
  hipLaunchKernelGGL(k_pose_fwd, dim3(pp_div_up(n_views * 6, 64)), dim3(64), 0, pp_stream(stream), se3, w2c_init,
                     refine_mask, n_views, w2c, c2w, jac);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
extern "C" int pp_pose_bwd(const float* jac, const float* c2w_grad, int32_t n_views, float* se3_grad, void* stream) {
  PP_REQUIRE(jac && c2w_grad && se3_grad && n_views > 0, "null pointer or n_views<=0");
  hipLaunchKernelGGL(k_pose_bwd, dim3(pp_div_up(n_views * 6, 64)), dim3(64), 0, pp_stream(stream), jac, c2w_grad,
                     n_views, se3_grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------
// ray generation for selected pixels
// ------------------------------------------------------------------------------------------------
// torch's CPU norm kernel accumulates with fused multiply-adds: sqrt(fma(z,z,fma(y,y,x*x))) (probed)
__device__ __forceinline__ float pp_norm3(float x, float y, float z) {
  return sqrtf(fmaf(z, z, fmaf(y, y, pp_mul(x, x))));
}

__device__ __forceinline__ void pixel_dir(int idx, int H, int W, const float* __restrict__ intr, int inverse_y,
                                          int& view, float dirs[3]) {
  view = idx / (H * W);
  int rem = idx - view * (H * W);
  int pj = rem / W, pi = rem - pj * W;
  float fi = pp_add((float)pi, 0.5f), fj = pp_add((float)pj, 0.5f);
  const float* K = intr + view * 4;
  dirs[0] = pp_div(pp_sub(fi, K[2]), K[0]);
  float y = pp_div(pp_sub(fj, K[3]), K[1]);
  dirs[1] = inverse_y ? y : -y;
  dirs[2] = inverse_y ? 1.f : -1.f;
}

__global__ void k_raygen_fwd(const int32_t* __restrict__ ray_idx, int n_rays, const float* __restrict__ c2w,
                             const float* __restrict__ intr, int H, int W, int inverse_y, int normalize,
                             const float* __restrict__ images, const float* __restrict__ masks,
                             float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ viewdirs,
                             float* __restrict__ target, float* __restrict__ mask_px) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  int idx = ray_idx[r], view;
  float dirs[3];
  pixel_dir(idx, H, W, intr, inverse_y, view, dirs);
  const float* P = c2w + view * 12;
  float d[3];
  for (int k = 0; k < 3; ++k)  // torch.sum over the last dim of 3 products: (p0+p1)+p2
    d[k] = pp_add(pp_add(pp_mul(dirs[0], P[k * 4 + 0]), pp_mul(dirs[1], P[k * 4 + 1])), pp_mul(dirs[2], P[k * 4 + 2]));
  float nrm = pp_norm3(d[0], d[1], d[2]);
  for (int k = 0; k < 3; ++k) {
    float vd = pp_div(d[k], nrm);
    rays_o[r * 3 + k] = P[k * 4 + 3];
    rays_d[r * 3 + k] = normalize ? vd : d[k];
    viewdirs[r * 3 + k] = vd;
  }
  if (target) for (int k = 0; k < 3; ++k) target[r * 3 + k] = images[(size_t)idx * 3 + k];
  if (mask_px) mask_px[r] = masks[idx];
}

extern "C" int pp_raygen_select_fwd(const pp_scene* sc, const int32_t* ray_idx, int32_t n_rays, const float* c2w,
                                    const float* intr, int32_t n_views, int32_t H, int32_t W, int32_t inverse_y,
                                    int32_t normalize, const float* images, const float* masks, float* rays_o,
                                    float* rays_d, float* viewdirs, float* target, float* mask_px, void* stream) {
  (void)sc; (void)n_views;
  PP_REQUIRE(ray_idx && c2w && intr && rays_o && rays_d && viewdirs && n_rays > 0, "null pointer or n_rays<=0");
  PP_REQUIRE((!target || images) && (!mask_px || masks), "target/mask requested without images/masks");
  hipLaunchKernelGGL(k_raygen_fwd, dim3(pp_div_up(n_rays, 256)), dim3(256), 0, pp_stream(stream), ray_idx, n_rays,
                     c2w, intr, H, W, inverse_y, normalize, images, masks, rays_o, rays_d, viewdirs, target, mask_px);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------
// dense sampler (sample_ray_ori) : one wavefront per ray, lanes = sample slots
// ------------------------------------------------------------------------------------------------
struct RaySlab {
  float o[3], d[3], t_min, t_max, nrm;
  bool miss;
};

__device__ __forceinline__ RaySlab ray_slab(const SceneDev& sc, const float* __restrict__ rays_o,
                                            const float* __restrict__ rays_d, int r) {
  RaySlab s;
  float lo = -INFINITY, hi = INFINITY;
  for (int k = 0; k < 3; ++k) {
    s.o[k] = rays_o[r * 3 + k];
    s.d[k] = rays_d[r * 3 + k];
    float vec = (s.d[k] == 0.f) ? 1e-6f : s.d[k];
    float a = pp_div(pp_sub(sc.mx[k], s.o[k]), vec);
    float b = pp_div(pp_sub(sc.mn[k], s.o[k]), vec);
    lo = fmaxf(lo, fminf(a, b));
    hi = fminf(hi, fmaxf(a, b));
  }
  s.t_min = fminf(fmaxf(lo, sc.near_), sc.far_);
  s.t_max = fminf(fmaxf(hi, sc.near_), sc.far_);
  s.miss = (s.t_max <= s.t_min);
  s.nrm = pp_norm3(s.d[0], s.d[1], s.d[2]);
  return s;
}

// in-bbox test of sample slot k; returns step and point
__device__ __forceinline__ bool dense_sample(const SceneDev& sc, const RaySlab& s, float stepdist, float jit, int k,
                                             float& step, float p[3]) {
  float rng = pp_add((float)k, jit);
  step = pp_mul(stepdist, rng);
  float interpx = pp_add(s.t_min, pp_div(step, s.nrm));
  bool out = s.miss;
  for (int c = 0; c < 3; ++c) {
    p[c] = pp_add(s.o[c], pp_mul(s.d[c], interpx));
    out |= (sc.mn[c] > p[c]) | (p[c] > sc.mx[c]);
  }
  return !out;
}

__global__ __launch_bounds__(256) void k_sample_count(SceneDev sc, const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ jitter, int n_rays,
                                                      float* __restrict__ t_min, float* __restrict__ t_max,
                                                      int32_t* __restrict__ counts) {
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  RaySlab s = ray_slab(sc, rays_o, rays_d, r);
  float stepdist = pp_mul(sc.stepsize, sc.voxel);
  float jit = jitter ? jitter[r] : 0.f;
  int cnt = 0;
  for (int k0 = 0; k0 < sc.S; k0 += 64) {
    int k = k0 + lane;
    float step, p[3];
    bool keep = (k < sc.S) && dense_sample(sc, s, stepdist, jit, k, step, p);
    cnt += __popcll(__ballot(keep));
  }
  if (lane == 0) {
    counts[r] = cnt;
    t_min[r] = s.t_min;
    t_max[r] = s.t_max;
  }
}

// exclusive scan of counts[0..n) into start[0..n], every entry clamped to the capacity: start[n] = count_out[0] =
// min(total, capacity), so every per-ray range [start[r], start[r+1]) a later kernel walks lies inside the allocation (rays past
// the capacity become empty, the ray that straddles it is cut)
__global__ __launch_bounds__(1024) void k_exclusive_scan(const int32_t* __restrict__ counts, int n,
                                                         int32_t* __restrict__ start, int32_t* __restrict__ count_out,
                                                         int capacity) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = (i < n) ? counts[i] : 0;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int carry = carry_s;
    if (i < n) start[i] = min(carry + woff + x - v, capacity);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    const int total = carry_s < capacity ? carry_s : capacity;
    start[n] = total;
    count_out[0] = total;
  }
}

__global__ __launch_bounds__(256) void k_sample_fill(SceneDev sc, const float* __restrict__ rays_o,
                                                     const float* __restrict__ rays_d,
                                                     const float* __restrict__ jitter, int n_rays, int capacity,
                                                     const int32_t* __restrict__ ray_start, float* __restrict__ pts,
                                                     int32_t* __restrict__ ray_id, int32_t* __restrict__ step_k,
                                                     float* __restrict__ step_out, uint8_t* __restrict__ mask_keep) {
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  RaySlab s = ray_slab(sc, rays_o, rays_d, r);
  float stepdist = pp_mul(sc.stepsize, sc.voxel);
  float jit = jitter ? jitter[r] : 0.f;
  int base = ray_start[r];
  for (int k0 = 0; k0 < sc.S; k0 += 64) {
    int k = k0 + lane;
    float step, p[3];
    bool keep = (k < sc.S) && dense_sample(sc, s, stepdist, jit, k, step, p);
    unsigned long long bal = __ballot(keep);
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (mask_keep && k < sc.S) mask_keep[(size_t)r * sc.S + k] = keep ? 1 : 0;
    if (keep && pos < capacity) {
      pts[pos * 3 + 0] = p[0]; pts[pos * 3 + 1] = p[1]; pts[pos * 3 + 2] = p[2];
      ray_id[pos] = r;
      step_k[pos] = k;
      step_out[pos] = step;
    }
    base += __popcll(bal);
  }
}

extern "C" int pp_sample_dense(const pp_scene* sc, const float* rays_o, const float* rays_d, const float* jitter,
                               int32_t n_rays, int32_t capacity, float* t_min, float* t_max, int32_t* ray_start,
                               int32_t* count, float* pts, int32_t* ray_id, int32_t* step_k, float* step,
                               uint8_t* mask_keep, void* stream) {
  PP_REQUIRE(sc && rays_o && rays_d && t_min && t_max && ray_start && count && pts && ray_id && step_k && step,
             "null pointer");
  PP_REQUIRE(n_rays > 0 && capacity > 0, "n_rays/capacity must be positive");
  SceneDev d = pp_scene_dev(sc);
  hipStream_t st = pp_stream(stream);
  // per-ray counts are staged in ray_start[0..N) and scanned in place into a second region: use step_k as scratch
  // is not possible (written later only), so counts live in ray_id's first N entries when capacity >= n_rays.
  PP_REQUIRE(capacity >= n_rays, "capacity must be >= n_rays (scratch reuse)");
  int32_t* counts = ray_id;
  hipLaunchKernelGGL(k_sample_count, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, st, d, rays_o, rays_d, jitter, n_rays,
                     t_min, t_max, counts);
  PP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, counts, n_rays, ray_start, count, capacity);
  PP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sample_fill, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, st, d, rays_o, rays_d, jitter, n_rays,
                     capacity, ray_start, pts, ray_id, step_k, step, mask_keep);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------
// variable-length sampler (sample_pts_on_rays semantics, then Voxurf.sample_ray_cuda's python post-processing)
// ------------------------------------------------------------------------------------------------
struct VarRay {
  float start[3], dir[3], view[3], t_min, t_max;
  int n;
};

__device__ __forceinline__ VarRay var_ray(const SceneDev& sc, const float* __restrict__ rays_o,
                                          const float* __restrict__ rays_d, int r, float stepdist) {
  VarRay v;
  float o[3], d[3];
  float lo = -INFINITY, hi = INFINITY;
  for (int k = 0; k < 3; ++k) {
    o[k] = rays_o[r * 3 + k]; d[k] = rays_d[r * 3 + k];
    float vec = (d[k] == 0.f) ? 1e-6f : d[k];
    float a = pp_div(pp_sub(sc.mx[k], o[k]), vec), b = pp_div(pp_sub(sc.mn[k], o[k]), vec);
    lo = fmaxf(lo, fminf(a, b));
    hi = fminf(hi, fmaxf(a, b));
  }
  const float far = 1e9f;  // voxurf_coarse.py:673
  v.t_min = fmaxf(fminf(lo, far), sc.near_);
  v.t_max = fmaxf(fminf(hi, far), sc.near_);
  // kernel.cu:49-54 plain products, no contraction
  float rn = sqrtf(pp_add(pp_add(pp_mul(d[0], d[0]), pp_mul(d[1], d[1])), pp_mul(d[2], d[2])));
  float ns = ceilf(pp_div(pp_mul(pp_sub(v.t_max, v.t_min), rn), stepdist));
  v.n = (ns < 1.f || !(ns == ns)) ? 1 : (int)fminf(ns, 1.0e6f);
  float tn = pp_norm3(d[0], d[1], d[2]);  // torch-side rays_d.norm() (voxurf_coarse.py:682)
  for (int k = 0; k < 3; ++k) {
    v.start[k] = pp_add(o[k], pp_mul(d[k], v.t_min));
    v.dir[k] = pp_div(d[k], rn);
    v.view[k] = pp_div(d[k], tn);
  }
  return v;
}

__device__ __forceinline__ bool var_keep(const SceneDev& sc, const VarRay& v, float stepdist, int s) {
  float dist = pp_mul(stepdist, (float)s);
  bool out = false;
  for (int k = 0; k < 3; ++k) {
    float p = pp_add(v.start[k], pp_mul(v.dir[k], dist));
    out |= (sc.mn[k] > p) | (sc.mx[k] < p);
  }
  return !out;
}

__global__ __launch_bounds__(256) void k_var_count(SceneDev sc, const float* __restrict__ rays_o,
                                                   const float* __restrict__ rays_d, int n_rays,
                                                   float* __restrict__ t_min, float* __restrict__ t_max,
                                                   int32_t* __restrict__ n_steps, int32_t* __restrict__ counts) {
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  float stepdist = pp_mul(sc.stepsize, sc.voxel);
  VarRay v = var_ray(sc, rays_o, rays_d, r, stepdist);
  int cnt = 0;
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    int s = s0 + lane;
    bool keep = (s < v.n) && var_keep(sc, v, stepdist, s);
    cnt += __popcll(__ballot(keep));
  }
  if (lane == 0) { counts[r] = cnt; n_steps[r] = v.n; t_min[r] = v.t_min; t_max[r] = v.t_max; }
}

__global__ __launch_bounds__(256) void k_var_fill(SceneDev sc, const float* __restrict__ rays_o,
                                                  const float* __restrict__ rays_d, int n_rays, int capacity,
                                                  const int32_t* __restrict__ ray_start, float* __restrict__ pts,
                                                  int32_t* __restrict__ ray_id, int32_t* __restrict__ step_id) {
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  float stepdist = pp_mul(sc.stepsize, sc.voxel);
  VarRay v = var_ray(sc, rays_o, rays_d, r, stepdist);
  int base = ray_start[r];
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    int s = s0 + lane;
    bool keep = (s < v.n) && var_keep(sc, v, stepdist, s);
    unsigned long long bal = __ballot(keep);
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && pos < capacity) {
      // rays_start[ray_id] + rays_view[ray_id] * step_id * stepdist   (voxurf_coarse.py:683)
      for (int k = 0; k < 3; ++k) pts[pos * 3 + k] = pp_add(v.start[k], pp_mul(pp_mul(v.view[k], (float)s), stepdist));
      ray_id[pos] = r;
      step_id[pos] = s;
    }
    base += __popcll(bal);
  }
}

extern "C" int pp_sample_var(const pp_scene* sc, const float* rays_o, const float* rays_d, int32_t n_rays,
                             int32_t capacity, float* t_min, float* t_max, int32_t* n_steps, int32_t* ray_start,
                             int32_t* count, float* pts, int32_t* ray_id, int32_t* step_id, void* stream) {
  PP_REQUIRE(sc && rays_o && rays_d && t_min && t_max && n_steps && ray_start && count && pts && ray_id && step_id,
             "null pointer");
  PP_REQUIRE(n_rays > 0 && capacity >= n_rays, "need n_rays>0 and capacity>=n_rays");
  SceneDev d = pp_scene_dev(sc);
  hipStream_t st = pp_stream(stream);
  int32_t* counts = ray_id;
  hipLaunchKernelGGL(k_var_count, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, st, d, rays_o, rays_d, n_rays, t_min, t_max,
                     n_steps, counts);
  PP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, counts, n_rays, ray_start, count, capacity);
  PP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_var_fill, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, st, d, rays_o, rays_d, n_rays, capacity,
                     ray_start, pts, ray_id, step_id);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------
// backward: samples -> rays -> c2w.  One wavefront per ray.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raygen_bwd(
    SceneDev sc, const int32_t* __restrict__ ray_idx, int n_rays, const float* __restrict__ c2w,
    const float* __restrict__ intr, int n_views, int H, int W, int inverse_y, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ t_min, const int32_t* __restrict__ ray_start,
    const float* __restrict__ pts_grad, const float* __restrict__ step, const float* __restrict__ vgrad_s,
    const float* __restrict__ g_o_in, const float* __restrict__ g_d_in, const float* __restrict__ g_v_in,
    const float* __restrict__ g_depth, float* __restrict__ g_o_out, float* __restrict__ g_d_out,
    float* __restrict__ g_v_out, float* __restrict__ c2w_grad) {
  extern __shared__ float s_c2w[];  // [n_views*12]
  for (int i = threadIdx.x; i < n_views * 12; i += blockDim.x) s_c2w[i] = 0.f;
  __syncthreads();
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r < n_rays) {
    int b = ray_start[r], e = ray_start[r + 1];
    float s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0}, gv[3] = {0, 0, 0};
    for (int i = b + lane; i < e; i += 64) {
      float st = step[i];
      for (int k = 0; k < 3; ++k) {
        float g = pts_grad[i * 3 + k];
        s0[k] += g;
        s1[k] += g * st;
        if (vgrad_s) gv[k] += vgrad_s[i * 3 + k];
      }
    }
    for (int k = 0; k < 3; ++k) { s0[k] = pp_wave_sum(s0[k]); s1[k] = pp_wave_sum(s1[k]); gv[k] = pp_wave_sum(gv[k]); }
    if (lane == 0) {
      float o[3], d[3];
      for (int k = 0; k < 3; ++k) { o[k] = rays_o[r * 3 + k]; d[k] = rays_d[r * 3 + k]; }
      float nrm = pp_norm3(d[0], d[1], d[2]);
      float tm = t_min[r];
      float gdep = g_depth ? g_depth[r] : 0.f;
      float ob[3], db[3];
      float tmin_bar = gdep / nrm, nrm_bar = -gdep * tm / (nrm * nrm);
      float s1d = 0.f;
      for (int k = 0; k < 3; ++k) {
        ob[k] = s0[k];
        db[k] = s0[k] * tm + s1[k] / nrm;
        tmin_bar += s0[k] * d[k];
        s1d += s1[k] * d[k];
      }
      nrm_bar -= s1d / (nrm * nrm);
      for (int k = 0; k < 3; ++k) db[k] += nrm_bar * d[k] / nrm;
      // slab test backward (amax / minimum / clamp with torch's tie handling)
      float ra[3], rb[3], lo[3], vec[3];
      float tm_raw = -INFINITY;
      for (int k = 0; k < 3; ++k) {
        vec[k] = (d[k] == 0.f) ? 1e-6f : d[k];
        ra[k] = (sc.mx[k] - o[k]) / vec[k];
        rb[k] = (sc.mn[k] - o[k]) / vec[k];
        lo[k] = fminf(ra[k], rb[k]);
        tm_raw = fmaxf(tm_raw, lo[k]);
      }
      if (tm_raw >= sc.near_ && tm_raw <= sc.far_ && tmin_bar != 0.f) {
        int nmax = 0;
        for (int k = 0; k < 3; ++k) nmax += (lo[k] == tm_raw);
        for (int k = 0; k < 3; ++k) {
          if (lo[k] != tm_raw) continue;
          float lb = tmin_bar / (float)nmax;
          float wa = ra[k] < rb[k] ? 1.f : (ra[k] == rb[k] ? 0.5f : 0.f);
          float rab = lb * wa, rbb = lb * (1.f - wa);
          ob[k] -= (rab + rbb) / vec[k];
          if (d[k] != 0.f) db[k] -= (rab * ra[k] + rbb * rb[k]) / vec[k];
        }
      }
      if (g_o_in) for (int k = 0; k < 3; ++k) ob[k] += g_o_in[r * 3 + k];
      if (g_d_in) for (int k = 0; k < 3; ++k) db[k] += g_d_in[r * 3 + k];
      if (g_v_in) for (int k = 0; k < 3; ++k) gv[k] += g_v_in[r * 3 + k];
      if (g_o_out) for (int k = 0; k < 3; ++k) g_o_out[r * 3 + k] = ob[k];
      if (g_d_out) for (int k = 0; k < 3; ++k) g_d_out[r * 3 + k] = db[k];
      if (g_v_out) for (int k = 0; k < 3; ++k) g_v_out[r * 3 + k] = gv[k];
      if (c2w_grad) {
        // Voxurf variant: rays_d = viewdirs = normalize(R dirs) -> one tensor (voxurf_coarse.py:1404)
        int view;
        float dirs[3];
        pixel_dir(ray_idx[r], H, W, intr, inverse_y, view, dirs);
        const float* P = c2w + view * 12;
        float Du[3], gt[3];
        for (int k = 0; k < 3; ++k) {
          Du[k] = dirs[0] * P[k * 4 + 0] + dirs[1] * P[k * 4 + 1] + dirs[2] * P[k * 4 + 2];
          gt[k] = db[k] + gv[k];
        }
        float Dn = sqrtf(Du[0] * Du[0] + Du[1] * Du[1] + Du[2] * Du[2]);
        float nh[3] = {Du[0] / Dn, Du[1] / Dn, Du[2] / Dn};
        float dot = nh[0] * gt[0] + nh[1] * gt[1] + nh[2] * gt[2];
        for (int k = 0; k < 3; ++k) {
          float Db = (gt[k] - nh[k] * dot) / Dn;
          for (int j = 0; j < 3; ++j) atomicAdd(&s_c2w[view * 12 + k * 4 + j], Db * dirs[j]);
          atomicAdd(&s_c2w[view * 12 + k * 4 + 3], ob[k]);
        }
      }
    }
  }
  __syncthreads();
  if (c2w_grad)
    for (int i = threadIdx.x; i < n_views * 12; i += blockDim.x)
      if (s_c2w[i] != 0.f) atomicAdd(&c2w_grad[i], s_c2w[i]);
}

extern "C" int pp_raygen_select_bwd(const pp_scene* sc, const int32_t* ray_idx, int32_t n_rays, const float* c2w,
                                    const float* intr, int32_t n_views, int32_t H, int32_t W, int32_t inverse_y,
                                    const float* rays_o, const float* rays_d, const float* t_min,
                                    const int32_t* ray_start, const float* pts_grad, const float* step,
                                    const float* viewdir_grad_s, const float* rays_o_grad, const float* rays_d_grad,
                                    const float* viewdirs_grad, const float* depth_grad, float* rays_o_grad_out,
                                    float* rays_d_grad_out, float* viewdirs_grad_out, float* c2w_grad, void* stream) {
  PP_REQUIRE(sc && rays_o && rays_d && t_min && ray_start && pts_grad && step, "null pointer");
  PP_REQUIRE(!c2w_grad || (ray_idx && c2w && intr && n_views > 0), "c2w_grad requested without camera data");
  PP_REQUIRE(n_rays > 0, "n_rays<=0");
  SceneDev d = pp_scene_dev(sc);
  hipStream_t st = pp_stream(stream);
  if (c2w_grad) {
    if (hipMemsetAsync(c2w_grad, 0, sizeof(float) * 12 * n_views, st) != hipSuccess) {
      pp_set_error("pp_raygen_select_bwd: memset failed");
      return PP_ERR_LAUNCH;
    }
  }
  int nv = n_views > 0 ? n_views : 1;
  hipLaunchKernelGGL(k_raygen_bwd, dim3(pp_div_up(n_rays, 4)), dim3(256), sizeof(float) * 12 * nv, st, d, ray_idx,
                     n_rays, c2w, intr, nv, H, W, inverse_y, rays_o, rays_d, t_min, ray_start, pts_grad, step,
                     viewdir_grad_s, rays_o_grad, rays_d_grad, viewdirs_grad, depth_grad, rays_o_grad_out,
                     rays_d_grad_out, viewdirs_grad_out, c2w_grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
