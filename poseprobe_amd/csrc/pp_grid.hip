// Generic trilinear lookup on a channels-last dense grid [X,Y,Z,C]: DenseGrid.forward (lib/grid.py:47-58, zeros
// padding) and grid_sampler's F.grid_sample path (lib/voxurf_coarse.py:540, dvgo_ori.py:249-261; border padding for
// the SDF).  align_corners=True, the reference's x<->z flip is folded into the indexing (world x -> first grid axis).
// Also the standalone total-variation gradient used by the drop-in autograd path.
#include "pp_common.h"

struct GTri {
  float w0[3], w1[3];
  int i0[3], i1[3];
  bool ok0[3], ok1[3];
};

__device__ __forceinline__ void gtri_setup(const SceneDev& sc, const float p[3], int border, GTri& t) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float u = pp_grid_u(p[a], sc.mn[a], sc.mx[a], sc.sz[a]);
    if (border) u = fminf(fmaxf(u, 0.f), (float)(sc.sz[a] - 1));   // grid_sample clips the coordinate for 'border'
    float f = floorf(u);
    t.w1[a] = pp_sub(u, f);
    t.w0[a] = pp_sub(pp_add(f, 1.f), u);
    float fc = fminf(fmaxf(f, -2.f), (float)sc.sz[a]);
    int i = (int)fc;
    t.ok0[a] = (i >= 0) && (i < sc.sz[a]);
    t.ok1[a] = (i + 1 >= 0) && (i + 1 < sc.sz[a]);
    t.i0[a] = min(max(i, 0), sc.sz[a] - 1);
    t.i1[a] = min(max(i + 1, 0), sc.sz[a] - 1);
  }
}

// 16 lanes per point, lane = channel (C <= 16)
__global__ __launch_bounds__(256) void k_grid_sample_fwd(SceneDev sc, const float* __restrict__ grid, int C,
                                                         const float* __restrict__ pts, int M, int border,
                                                         float* __restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  if (m >= M || ch >= C) return;
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  GTri g;
  gtri_setup(sc, p, border, g);
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    bool ok = ((c & 4) ? g.ok1[0] : g.ok0[0]) && ((c & 2) ? g.ok1[1] : g.ok0[1]) && ((c & 1) ? g.ok1[2] : g.ok0[2]);
    if (!ok) continue;
    int ix = (c & 4) ? g.i1[0] : g.i0[0], iy = (c & 2) ? g.i1[1] : g.i0[1], iz = (c & 1) ? g.i1[2] : g.i0[2];
    float w = ((c & 4) ? g.w1[0] : g.w0[0]) * ((c & 2) ? g.w1[1] : g.w0[1]) * ((c & 1) ? g.w1[2] : g.w0[2]);
    acc += grid[(((size_t)ix * sc.sz[1] + iy) * sc.sz[2] + iz) * C + ch] * w;
  }
  out[(size_t)m * C + ch] = acc;
}

__global__ __launch_bounds__(256) void k_grid_sample_bwd(SceneDev sc, const float* __restrict__ grid, int C,
                                                         const float* __restrict__ pts, int M, int border,
                                                         const float* __restrict__ out_grad, float* __restrict__ grid_grad,
                                                         float* __restrict__ pts_grad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  bool live = (m < M) && (ch < C);
  float pb[3] = {0.f, 0.f, 0.f};
  if (live) {
    float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
    GTri g;
    gtri_setup(sc, p, border, g);
    float go = out_grad[(size_t)m * C + ch];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      bool ok = ((c & 4) ? g.ok1[0] : g.ok0[0]) && ((c & 2) ? g.ok1[1] : g.ok0[1]) && ((c & 1) ? g.ok1[2] : g.ok0[2]);
      if (!ok) continue;
      int ix = (c & 4) ? g.i1[0] : g.i0[0], iy = (c & 2) ? g.i1[1] : g.i0[1], iz = (c & 1) ? g.i1[2] : g.i0[2];
      float wx = (c & 4) ? g.w1[0] : g.w0[0], wy = (c & 2) ? g.w1[1] : g.w0[1], wz = (c & 1) ? g.w1[2] : g.w0[2];
      size_t off = (((size_t)ix * sc.sz[1] + iy) * sc.sz[2] + iz) * C + ch;
      if (grid_grad && go != 0.f) atomicAdd(&grid_grad[off], wx * wy * wz * go);
      if (pts_grad) {
        float v = grid[off] * go;
        float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
        pb[0] += v * sx * wy * wz; pb[1] += v * wx * sy * wz; pb[2] += v * wx * wy * sz;
      }
    }
    if (border) {   // clipped coordinates have zero derivative outside the grid
      for (int a = 0; a < 3; ++a) {
        float u = pp_grid_u(p[a], sc.mn[a], sc.mx[a], sc.sz[a]);
        if (u <= 0.f || u >= (float)(sc.sz[a] - 1)) pb[a] = 0.f;
      }
    }
  }
  if (pts_grad) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = pb[a];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      if (live && ch == 0) pts_grad[m * 3 + a] = v * (float)(sc.sz[a] - 1) / (sc.mx[a] - sc.mn[a]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the dense surface-point query (Voxurf.query_sdf_point_wocuda_wodeform, lib/voxurf_coarse.py:797-837):
//   p_k = o + d (t_min + dist (k + jitter) / |d|),  s_k = border-padded trilinear lookup of the RAW template at p_k,
//   prev = first k with s_k s_{k+1} <= 0,  z0 = (s1 z2 - s2 z1) / (s1 - s2 + 1e-10)  (zeroed outside [z1, z2]),
//   pts = o + d (t_min + z0 / |d|).
// Given d L / d pts (and optionally d L / d s_k for the dense SDF row the query also returns) it produces d L / d o,
// d L / d d and d L / d t_min (t_min's own dependence on the ray, the slab test of :701-705, is differentiated by the
// caller).  One wavefront per ray; lanes walk the samples that carry a gradient (two of them unless g_sdf is given).
// F.grid_sample's 'border' rule: a clipped coordinate (u <= 0 or u >= size-1) has zero derivative.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void border_tri_grad(const SceneDev& sc, const float* __restrict__ grid, const float p[3],
                                                float g[3]) {
  float w0[3], w1[3], mult[3];
  int i0[3], i1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float u = pp_grid_u(p[a], sc.mn[a], sc.mx[a], sc.sz[a]);
    float top = (float)(sc.sz[a] - 1);
    mult[a] = (u <= 0.f || u >= top) ? 0.f : top / (sc.mx[a] - sc.mn[a]);
    u = fminf(fmaxf(u, 0.f), top);
    float f = floorf(u);
    w1[a] = u - f;
    w0[a] = 1.f - w1[a];
    i0[a] = min(max((int)f, 0), sc.sz[a] - 1);
    i1[a] = min(i0[a] + 1, sc.sz[a] - 1);
  }
  g[0] = g[1] = g[2] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int ix = (c & 4) ? i1[0] : i0[0], iy = (c & 2) ? i1[1] : i0[1], iz = (c & 1) ? i1[2] : i0[2];
    float wx = (c & 4) ? w1[0] : w0[0], wy = (c & 2) ? w1[1] : w0[1], wz = (c & 1) ? w1[2] : w0[2];
    float v = grid[((size_t)ix * sc.sz[1] + iy) * sc.sz[2] + iz];
    g[0] += ((c & 4) ? v : -v) * wy * wz;
    g[1] += ((c & 2) ? v : -v) * wx * wz;
    g[2] += ((c & 1) ? v : -v) * wx * wy;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] *= mult[a];
}

__global__ __launch_bounds__(256) void k_crossing_dense_bwd(SceneDev sc, const float* __restrict__ grid,
                                                            const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ t_min, const float* __restrict__ jitter,
                                                            int n_rays, int S, float dist, const float* __restrict__ sdf_dense,
                                                            const float* __restrict__ g_pts, const float* __restrict__ g_sdf,
                                                            float* __restrict__ g_o, float* __restrict__ g_d,
                                                            float* __restrict__ g_tmin) {
  int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int r = blockIdx.x * 4 + wid;
  if (r >= n_rays) return;
  const float* row = sdf_dense + (size_t)r * S;
  int prev = 0;
  for (int k0 = 0; k0 < S - 1; k0 += 64) {
    int k = k0 + lane;
    bool hit = (k < S - 1) && (row[k] * row[k + 1] <= 0.f);
    unsigned long long bal = __ballot(hit);
    if (bal) { prev = k0 + __ffsll((long long)bal) - 1; break; }
  }
  float o[3] = {rays_o[r * 3], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
  float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
  float nrm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
  float tm = t_min[r], jit = jitter ? jitter[r] : 0.f;
  float s1 = row[prev], s2 = row[prev + 1];
  float z1 = (float)prev * dist + dist * 0.5f, z2 = (float)(prev + 1) * dist + dist * 0.5f;
  float den = s1 - s2 + 1e-10f;
  float z0 = (s1 * z2 - s2 * z1) / den;
  bool kept = !(z0 < z1) && !(z0 > z2);               // the two torch.where's replace z0 by a constant 0 otherwise
  float z0c = kept ? z0 : 0.f;
  float G[3] = {0.f, 0.f, 0.f};
  if (g_pts) { G[0] = g_pts[r * 3]; G[1] = g_pts[r * 3 + 1]; G[2] = g_pts[r * 3 + 2]; }
  float gi = G[0] * d[0] + G[1] * d[1] + G[2] * d[2];
  float g_z0 = kept ? gi / nrm : 0.f;
  float g_s1 = g_z0 * (z2 - z0) / den, g_s2 = g_z0 * (z0 - z1) / den;
  float ao[3] = {0.f, 0.f, 0.f}, ad[3] = {0.f, 0.f, 0.f}, atm = 0.f, an = 0.f;
  if (lane == 0) {
    float interp = tm + z0c / nrm;
#pragma unroll
    for (int a = 0; a < 3; ++a) { ao[a] = G[a]; ad[a] = G[a] * interp; }
    atm = gi;
    an = -gi * z0c / (nrm * nrm);
  }
  for (int k = lane; k < S; k += 64) {
    float gs = g_sdf ? g_sdf[(size_t)r * S + k] : 0.f;
    if (k == prev) gs += g_s1;
    if (k == prev + 1) gs += g_s2;
    if (gs == 0.f) continue;
    float st = dist * ((float)k + jit);
    float interp = tm + st / nrm;
    float p[3] = {o[0] + d[0] * interp, o[1] + d[1] * interp, o[2] + d[2] * interp};
    float gp[3];
    border_tri_grad(sc, grid, p, gp);
    float gid = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { gp[a] *= gs; ao[a] += gp[a]; ad[a] += gp[a] * interp; gid += gp[a] * d[a]; }
    atm += gid;
    an -= gid * st / (nrm * nrm);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { ao[a] = pp_wave_sum(ao[a]); ad[a] = pp_wave_sum(ad[a]); }
  atm = pp_wave_sum(atm);
  an = pp_wave_sum(an);
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { g_o[r * 3 + a] = ao[a]; g_d[r * 3 + a] = ad[a] + an * d[a] / nrm; }
    g_tmin[r] = atm;
  }
}

__device__ __forceinline__ float sgn1(float x) { return (x > 0.f) ? 1.f : (x < 0.f ? -1.f : 0.f); }

// grad += scale * g_scalar[0] * d/dp sum|diff|   (total_variation backward, lib/voxurf_coarse.py:1298-1313)
__global__ __launch_bounds__(256) void k_grid_tv_grad(const float* __restrict__ p, int X, int Y, int Z, int C,
                                                      float scale, const float* __restrict__ g_scalar,
                                                      float* __restrict__ grad) {
  const long long n = (long long)X * Y * Z * C;
  const float s = scale * (g_scalar ? g_scalar[0] : 1.f);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    long long vox = e / C;
    int z = (int)(vox % Z);
    long long t = vox / Z;
    int y = (int)(t % Y);
    int x = (int)(t / Y);
    float v = p[e], acc = 0.f;
    const long long sz = C, sy = (long long)Z * C, sx = (long long)Y * Z * C;
    if (z > 0) acc += sgn1(v - p[e - sz]);
    if (z < Z - 1) acc += sgn1(v - p[e + sz]);
    if (y > 0) acc += sgn1(v - p[e - sy]);
    if (y < Y - 1) acc += sgn1(v - p[e + sy]);
    if (x > 0) acc += sgn1(v - p[e - sx]);
    if (x < X - 1) acc += sgn1(v - p[e + sx]);
    grad[e] += s * acc;
  }
}

extern "C" int pp_grid_sample_fwd(const pp_scene* sc, const float* grid_cl, int32_t channels, const float* pts,
                                  int32_t n_pts, int32_t border, float* out, void* stream) {
  PP_REQUIRE(sc && grid_cl && pts && out, "null pointer");
  PP_REQUIRE(channels >= 1 && channels <= 16, "channels must be in [1,16]");
  if (n_pts <= 0) return PP_OK;
  hipLaunchKernelGGL(k_grid_sample_fwd, dim3(pp_div_up(n_pts * 16, 256)), dim3(256), 0, pp_stream(stream),
                     pp_scene_dev(sc), grid_cl, channels, pts, n_pts, border, out);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_grid_sample_bwd(const pp_scene* sc, const float* grid_cl, int32_t channels, const float* pts,
                                  int32_t n_pts, int32_t border, const float* out_grad, float* grid_grad_cl,
                                  float* pts_grad, void* stream) {
  PP_REQUIRE(sc && grid_cl && pts && out_grad, "null pointer");
  PP_REQUIRE(channels >= 1 && channels <= 16, "channels must be in [1,16]");
  if (n_pts <= 0) return PP_OK;
  hipLaunchKernelGGL(k_grid_sample_bwd, dim3(pp_div_up(n_pts * 16, 256)), dim3(256), 0, pp_stream(stream),
                     pp_scene_dev(sc), grid_cl, channels, pts, n_pts, border, out_grad, grid_grad_cl, pts_grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_grid_tv_grad(const float* p, int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels,
                               float scale, const float* g_scalar, float* grad, void* stream) {
  PP_REQUIRE(p && grad, "null pointer");
  if (!pp_launch_tv_march(p, size_x, size_y, size_z, channels, scale, g_scalar, grad, nullptr, pp_stream(stream)))
    hipLaunchKernelGGL(k_grid_tv_grad, dim3(2048), dim3(256), 0, pp_stream(stream), p, size_x, size_y, size_z, channels,
                       scale, g_scalar, grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_sdf_crossing_dense_bwd(const pp_scene* sc, const float* sdf_grid, const float* rays_o, const float* rays_d,
                                         const float* t_min, const float* jitter, int32_t n_rays, int32_t n_samples,
                                         float dist, const float* sdf_dense, const float* g_pts, const float* g_sdf_dense,
                                         float* g_rays_o, float* g_rays_d, float* g_t_min, void* stream) {
  PP_REQUIRE(sc && sdf_grid && rays_o && rays_d && t_min && sdf_dense && g_rays_o && g_rays_d && g_t_min, "null pointer");
  PP_REQUIRE(g_pts || g_sdf_dense, "need at least one upstream gradient");
  PP_REQUIRE(n_rays > 0 && n_samples >= 2, "need n_rays>0 and n_samples>=2");
  hipLaunchKernelGGL(k_crossing_dense_bwd, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), pp_scene_dev(sc),
                     sdf_grid, rays_o, rays_d, t_min, jitter, n_rays, n_samples, dist, sdf_dense, g_pts, g_sdf_dense, g_rays_o,
                     g_rays_d, g_t_min);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
