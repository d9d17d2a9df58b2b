// Optimiser: the reference's Adam (lib/utils.py:82-198, betas (0.9,0.99) from :342) fused, for the dense k0 grid,
// with the total-variation gradient (lib/voxurf_coarse.py:443-456, :1298-1313) and the gradient zero-fill into ONE
// streaming pass: reads p,g,m,v (+ 6 neighbours of p, cache-served), writes p',m,v and g=0 : 384 B/voxel at C=12
// instead of the reference's separate TV forward, TV backward, zero_grad and Adam passes (528 B/voxel).
// Parameters ping-pong between two buffers so that neighbour reads never see updated values.
#include <stdlib.h>

#include "pp_common.h"

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float4 sgn4(float4 a, float4 b) {
  return make_float4(sgnf(a.x - b.x), sgnf(a.y - b.y), sgnf(a.z - b.z), sgnf(a.w - b.w));
}
__device__ __forceinline__ float abs4(float4 a, float4 b) {
  return fabsf(a.x - b.x) + fabsf(a.y - b.y) + fabsf(a.z - b.z) + fabsf(a.w - b.w);
}

__device__ __forceinline__ float adam1(float p, float g, float& m, float& v, float b1, float b2, float eps,
                                       float step_size, float inv_sqrt_bc2) {
  m = m * b1 + (1.f - b1) * g;
  v = v * b2 + (1.f - b2) * g * g;
  float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  return p - step_size * (m / denom);
}

// X-marching formulation.  A thread owns one float4 column position (y,z,q) of the plane and walks a chunk of
// x-planes keeping p[x-1], p[x], p[x+1] in registers: every parameter element is fetched from HBM exactly once (plus a
// two-plane halo per chunk); the +-y / +-z neighbours are re-read from the plane currently being swept, which is L1/L2
// resident because all tiles of a chunk run on ONE XCD (chunk = blockIdx % n_chunks, n_chunks a multiple of 8, and
// blocks are dealt round-robin over the 8 XCDs).  g / m / v and the outputs are touched once: non-temporal accesses
// keep them from evicting the parameter planes out of L2.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt4(const float4* p) {
  v4f r = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void stnt4(float4* p, float4 v) {
  v4f r = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ void acc_sgn(float4& tv, float4 p, float4 n) {
  tv.x += sgnf(p.x - n.x); tv.y += sgnf(p.y - n.y); tv.z += sgnf(p.z - n.z); tv.w += sgnf(p.w - n.w);
}

__global__ __launch_bounds__(256) void k_grid_tv_adam(const float4* __restrict__ p_in, float4* __restrict__ p_out,
                                                      float4* __restrict__ grad, float4* __restrict__ m_,
                                                      float4* __restrict__ v_, int X, int Y, int Z, int q4,
                                                      int x_begin, int x_end, int n_chunks, int chunk_len,
                                                      float tv_scale, float grad_scale, float b1, float b2, float eps,
                                                      float step_size, float inv_sqrt_bc2, float* __restrict__ tv_out,
                                                      const uint8_t* __restrict__ touched,
                                                      uint8_t* __restrict__ touched_clear) {
  __shared__ float sm[4];
  const int plane = Y * Z * q4;                      // float4 per x-plane
  const int chunk = blockIdx.x % n_chunks;
  const int tile = blockIdx.x / n_chunks;
  const int i = tile * 256 + threadIdx.x;            // position inside the plane
  const int xs = x_begin + chunk * chunk_len;
  const int xe = min(xs + chunk_len, x_end);
  float tv_local = 0.f;
  if (i < plane && xs < xe) {
    const int vox = i / q4;
    const int z = vox % Z, y = vox / Z;
    const int sz = q4, sy = Z * q4;
    const bool zl = z > 0, zh = z < Z - 1, yl = y > 0, yh = y < Y - 1;
    size_t e = (size_t)xs * plane + i;
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f), pc = p_in[e], pn;
    if (xs > 0) pm = p_in[e - plane];
    // Sparse gradient: the data-dependent part of the gradient is non-zero only in voxels the scatter marked (~8 % per
    // step); for the others g == 0 is known without reading it and its zero-fill is a no-op.  The mark of plane x+1 is
    // fetched one iteration ahead; the OTHER parity's map (consumed last step) is cleared on the way.
    const int nyz = Y * Z;
    size_t vx = (size_t)xs * nyz + vox;
    uint8_t wcur = touched ? touched[vx] : (uint8_t)1, wnext = wcur;
    const bool clearer = touched_clear && (i % q4) == 0;
    for (int x = xs; x < xe; ++x, e += plane, vx += nyz) {
      const bool xh = x < X - 1;
      pn = xh ? p_in[e + plane] : pc;
      if (touched && x + 1 < xe) wnext = touched[vx + nyz];
      const bool hit = wcur != 0;
      if (clearer) touched_clear[vx] = 0;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hit) g = ldnt4(grad + e);
      float4 m = ldnt4(m_ + e), v = ldnt4(v_ + e);
      float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x > 0) acc_sgn(tv, pc, pm);
      if (xh) { acc_sgn(tv, pc, pn); tv_local += abs4(pc, pn); }
      if (zl) acc_sgn(tv, pc, p_in[e - sz]);
      if (yl) acc_sgn(tv, pc, p_in[e - sy]);
      if (zh) { float4 nb = p_in[e + sz]; acc_sgn(tv, pc, nb); tv_local += abs4(pc, nb); }
      if (yh) { float4 nb = p_in[e + sy]; acc_sgn(tv, pc, nb); tv_local += abs4(pc, nb); }
      g.x = g.x * grad_scale + tv_scale * tv.x; g.y = g.y * grad_scale + tv_scale * tv.y;
      g.z = g.z * grad_scale + tv_scale * tv.z; g.w = g.w * grad_scale + tv_scale * tv.w;
      float4 o;
      o.x = adam1(pc.x, g.x, m.x, v.x, b1, b2, eps, step_size, inv_sqrt_bc2);
      o.y = adam1(pc.y, g.y, m.y, v.y, b1, b2, eps, step_size, inv_sqrt_bc2);
      o.z = adam1(pc.z, g.z, m.z, v.z, b1, b2, eps, step_size, inv_sqrt_bc2);
      o.w = adam1(pc.w, g.w, m.w, v.w, b1, b2, eps, step_size, inv_sqrt_bc2);
      stnt4(p_out + e, o);
      stnt4(m_ + e, m);
      stnt4(v_ + e, v);
      if (hit) stnt4(grad + e, make_float4(0.f, 0.f, 0.f, 0.f));
      pm = pc;
      pc = pn;
      wcur = wnext;
    }
  }
  if (tv_out) {
    tv_local = pp_wave_sum(tv_local);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) sm[wid] = tv_local;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = sm[0] + sm[1] + sm[2] + sm[3];
      if (s != 0.f) atomicAdd(tv_out, s);
    }
  }
}

// The total-variation term ON ITS OWN for the drop-in autograd path (lib/voxurf_coarse.py:443-456, :1298-1313: k0_tv is an
// output of Voxurf.forward and its gradient arrives through loss.backward()), in the same X-marching form as the fused pass:
// every parameter element is fetched once, the +-y / +-z neighbours come from the plane being swept.  GRAD = false: tv_out +=
// sum |forward differences|; GRAD = true: grad += scale * g_scalar[0] * sum of sgn over the 6 neighbours (read-modify-write,
// non-temporal).  Per step at 160^3: gradient 168-177 us (the element-wise kernel it replaces: 385), value 100 us (102: both are bound by
// the latency of the dependent neighbour loads, not by bytes; issuing the next plane's loads one iteration ahead measured SLOWER:
// 261 / 127 us) - bench.py dropin_train_step.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_grid_tv_march(const float4* __restrict__ p_in, int X, int Y, int Z, int q4, int n_chunks,
                                                       int chunk_len, int n_virtual, float scale, const float* __restrict__ g_scalar,
                                                       float4* __restrict__ grad, float* __restrict__ tv_out) {
  __shared__ float sm[4];
  const int plane = Y * Z * q4;
  const float s = GRAD ? scale * (g_scalar ? g_scalar[0] : 1.f) : 0.f;
  if (GRAD && s == 0.f) return;
  float tv_local = 0.f;
  // the value pass ends in ONE same-address atomic per work-group (~10 ns each): it runs a bounded number of persistent
  // work-groups over the (tile, chunk) pairs; the gradient pass has no such tail and launches one work-group per pair
  for (int vb = blockIdx.x; vb < n_virtual; vb += gridDim.x) {
    const int chunk = vb % n_chunks;
    const int tile = vb / n_chunks;
    const int i = tile * 256 + threadIdx.x;
    const int xs = chunk * chunk_len;
    const int xe = min(xs + chunk_len, X);
    if (i >= plane || xs >= xe) continue;
    const int vox = i / q4;
    const int z = vox % Z, y = vox / Z;
    const int sz = q4, sy = Z * q4;
    const bool zl = z > 0, zh = z < Z - 1, yl = y > 0, yh = y < Y - 1;
    size_t e = (size_t)xs * plane + i;
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f), pc = p_in[e], pn;
    if (GRAD && xs > 0) pm = p_in[e - plane];
    for (int x = xs; x < xe; ++x, e += plane) {
      const bool xh = x < X - 1;
      pn = xh ? p_in[e + plane] : pc;
      if (GRAD) {
        float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x > 0) acc_sgn(tv, pc, pm);
        if (xh) acc_sgn(tv, pc, pn);
        if (zl) acc_sgn(tv, pc, p_in[e - sz]);
        if (yl) acc_sgn(tv, pc, p_in[e - sy]);
        if (zh) acc_sgn(tv, pc, p_in[e + sz]);
        if (yh) acc_sgn(tv, pc, p_in[e + sy]);
        float4 g = ldnt4(grad + e);
        g.x += s * tv.x; g.y += s * tv.y; g.z += s * tv.z; g.w += s * tv.w;
        stnt4(grad + e, g);
      } else {
        if (xh) tv_local += abs4(pc, pn);
        if (zh) tv_local += abs4(pc, p_in[e + sz]);
        if (yh) tv_local += abs4(pc, p_in[e + sy]);
      }
      pm = pc;
      pc = pn;
    }
  }
  if (!GRAD) {
    tv_local = pp_wave_sum(tv_local);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) sm[wid] = tv_local;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = sm[0] + sm[1] + sm[2] + sm[3];
      if (t != 0.f) atomicAdd(tv_out, t);
    }
  }
}

// shared launcher (pp_grid_tv_value below, pp_grid_tv_grad in pp_grid.hip); returns false when the shape needs the generic kernels
bool pp_launch_tv_march(const float* p, int X, int Y, int Z, int C, float scale, const float* g_scalar, float* grad, float* tv_out,
                        hipStream_t st) {
  if (C % 4 != 0 || (long long)Y * Z * (C / 4) >= (1ll << 31)) return false;
  const int q4 = C / 4;
  int n_chunks = X >= 128 ? 16 : (X >= 8 ? 8 : X);
  const int chunk_len = (X + n_chunks - 1) / n_chunks;
  n_chunks = (X + chunk_len - 1) / chunk_len;
  const int tiles = (int)(((long long)Y * Z * q4 + 255) / 256);
  const int nv = tiles * n_chunks;
  if (grad)
    hipLaunchKernelGGL((k_grid_tv_march<true>), dim3(nv), dim3(256), 0, st, reinterpret_cast<const float4*>(p), X, Y, Z,
                       q4, n_chunks, chunk_len, nv, scale, g_scalar, reinterpret_cast<float4*>(grad), nullptr);
  else       // a multiple of n_chunks (itself a multiple of 8 when possible) keeps chunk <-> XCD for every pass of the loop
    hipLaunchKernelGGL((k_grid_tv_march<false>), dim3(nv < 1024 ? nv : (1024 / n_chunks) * n_chunks), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(p), X, Y, Z, q4, n_chunks, chunk_len, nv, 0.f, nullptr, nullptr, tv_out);
  return true;
}

__global__ __launch_bounds__(256) void k_grid_tv_value(const float4* __restrict__ p_in, int X, int Y, int Z, int q4,
                                                       float* __restrict__ tv_out) {
  __shared__ float sm[4];
  const long long n = (long long)X * Y * Z * q4;
  float tv_local = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    long long vox = e / q4;
    int z = (int)(vox % Z);
    long long t = vox / Z;
    int y = (int)(t % Y);
    int x = (int)(t / Y);
    float4 p = p_in[e];
    if (z < Z - 1) tv_local += abs4(p, p_in[e + q4]);
    if (y < Y - 1) tv_local += abs4(p, p_in[e + (long long)Z * q4]);
    if (x < X - 1) tv_local += abs4(p, p_in[e + (long long)Y * Z * q4]);
  }
  tv_local = pp_wave_sum(tv_local);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = tv_local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(tv_out, sm[0] + sm[1] + sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void k_adam_flat(float* __restrict__ p, float* __restrict__ grad, float* __restrict__ m_,
                                                   float* __restrict__ v_, int n, const int32_t* __restrict__ seg_end,
                                                   const float* __restrict__ seg_lr, int n_seg, float grad_scale,
                                                   float b1, float b2, float eps, float inv_bc1, float inv_sqrt_bc2,
                                                   int zero_grad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = 0;
  while (s < n_seg - 1 && i >= seg_end[s]) ++s;
  float lr = seg_lr[s];
  float g = grad[i] * grad_scale, m = m_[i], v = v_[i];
  float o = adam1(p[i], g, m, v, b1, b2, eps, lr * inv_bc1, inv_sqrt_bc2);
  if (lr != 0.f) p[i] = o;
  m_[i] = m;
  v_[i] = v;
  if (zero_grad) grad[i] = 0.f;
}

extern "C" int pp_grid_tv_adam_step_sparse(const float* p_in, float* p_out, float* grad, float* exp_avg, float* exp_avg_sq,
                                    int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, int32_t x_begin, int32_t x_end,
                                    float tv_scale, float grad_scale, float lr, float beta1, float beta2, float eps,
                                    int32_t step, float* tv_out, const uint8_t* touched,
                                            uint8_t* touched_clear, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(p_in && p_out && grad && exp_avg && exp_avg_sq, "null pointer");
  const int32_t size[3] = {size_x, size_y, size_z};
  PP_REQUIRE(p_in != p_out, "p_in and p_out must be distinct (ping-pong) buffers");
  PP_REQUIRE(channels > 0 && channels % 4 == 0, "channels must be a positive multiple of 4");
  PP_REQUIRE(0 <= x_begin && x_begin < x_end && x_end <= size[0], "bad x slab");
  PP_REQUIRE(step >= 1, "step must be >= 1");
  const int q4 = channels / 4;
  const long long n = (long long)(x_end - x_begin) * size[1] * size[2] * q4;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int nx = x_end - x_begin;
  const long long plane = (long long)size[1] * size[2] * q4;
  PP_REQUIRE(plane < (1ll << 31) && n < (1ll << 40), "grid too large for 32-bit plane indexing");
  // chunks: a multiple of 8 when possible so that chunk <-> XCD (blocks are dealt round-robin over the 8 XCDs)
  // measured (tools/bench_grid.py, MI355X): 16 chunks win from 128 planes up (160^3: 272 -> 256 us dense), 8 below
  int n_chunks = nx >= 128 ? 16 : (nx >= 8 ? 8 : nx);
  { const int c = pp_opt(PP_OPT_GRID_CHUNKS); if (c > 0 && c <= nx) n_chunks = c; }   // tuning hook (option "grid_chunks")
  const int chunk_len = (nx + n_chunks - 1) / n_chunks;
  n_chunks = (nx + chunk_len - 1) / chunk_len;
  const int tiles = (int)((plane + 255) / 256);
  hipLaunchKernelGGL(k_grid_tv_adam, dim3(tiles * n_chunks), dim3(256), 0, pp_stream(stream),
                     reinterpret_cast<const float4*>(p_in), reinterpret_cast<float4*>(p_out),
                     reinterpret_cast<float4*>(grad), reinterpret_cast<float4*>(exp_avg),
                     reinterpret_cast<float4*>(exp_avg_sq), size[0], size[1], size[2], q4, x_begin, x_end, n_chunks,
                     chunk_len, tv_scale, grad_scale, beta1, beta2, eps, (float)((double)lr / bc1),
                     (float)(1.0 / sqrt(bc2)), tv_out, touched, touched_clear);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_grid_tv_adam_step(const float* p_in, float* p_out, float* grad, float* exp_avg, float* exp_avg_sq,
                                    int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, int32_t x_begin,
                                    int32_t x_end, float tv_scale, float grad_scale, float lr, float beta1, float beta2,
                                    float eps, int32_t step, float* tv_out, void* ctx, void* stream) {
  return pp_grid_tv_adam_step_sparse(p_in, p_out, grad, exp_avg, exp_avg_sq, size_x, size_y, size_z, channels, x_begin, x_end,
                                     tv_scale, grad_scale, lr, beta1, beta2, eps, step, tv_out, nullptr, nullptr, ctx, stream);
}

extern "C" int pp_grid_tv_value(const float* p, int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels,
                                float* out, void* stream) {
  PP_REQUIRE(p && out, "null pointer");
  const int32_t size[3] = {size_x, size_y, size_z};
  PP_REQUIRE(channels > 0 && channels % 4 == 0, "channels must be a positive multiple of 4");
  if (!pp_launch_tv_march(p, size[0], size[1], size[2], channels, 0.f, nullptr, nullptr, out, pp_stream(stream)))
    hipLaunchKernelGGL(k_grid_tv_value, dim3(2048), dim3(256), 0, pp_stream(stream), reinterpret_cast<const float4*>(p),
                       size[0], size[1], size[2], channels / 4, out);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_adam_flat(float* p, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n,
                            const int32_t* seg_end, const float* seg_lr, int32_t n_seg, float grad_scale, float beta1,
                            float beta2, float eps, int32_t step, int32_t zero_grad, void* stream) {
  PP_REQUIRE(p && grad && exp_avg && exp_avg_sq && seg_end && seg_lr, "null pointer");
  PP_REQUIRE(n > 0 && n_seg > 0 && step >= 1, "bad sizes");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adam_flat, dim3(pp_div_up(n, 256)), dim3(256), 0, pp_stream(stream), p, grad, exp_avg,
                     exp_avg_sq, n, seg_end, seg_lr, n_seg, grad_scale, beta1, beta2, eps, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), zero_grad);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
