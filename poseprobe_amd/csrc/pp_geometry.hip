// Geometry stage, one thread per sample: deformed / undeformed SDF lookup with the alpha/beta mapping applied to the
// 8 corner values in registers (the mapped G^3 grid of lib/voxurf_coarse.py:946-949 is never written), analytic
// spatial gradient + mixed second derivatives of the trilinear interpolant (replaces the autograd passes of
// :968-984), NeuS alpha (:483-519).  The sdf template is [X][Y][Z] fp32 (4 B gathers, L2/Infinity-Cache resident:
// 3.5 MB at 96^3, 16 MB at 160^3).
#include "pp_common.h"

struct Tri {
  float w0[3], w1[3];
  int i0[3], i1[3];
};

__device__ __forceinline__ void tri_setup(const SceneDev& sc, const float p[3], Tri& t) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float u = pp_grid_u(p[a], sc.mn[a], sc.mx[a], sc.sz[a]);
    float f = floorf(u);
    t.w1[a] = pp_sub(u, f);                 // weight of the +1 corner, from the UNCLAMPED floor (voxurf_coarse.py:589-596)
    t.w0[a] = pp_sub(pp_add(f, 1.f), u);
    float fc = fminf(fmaxf(f, -2.f), (float)sc.sz[a]);
    int i = (int)fc;
    t.i0[a] = min(max(i, 0), sc.sz[a] - 1);  // indices clamped (voxurf_coarse.py:598-629)
    t.i1[a] = min(max(i + 1, 0), sc.sz[a] - 1);
  }
}

// raw corner values, order c = dx*4 + dy*2 + dz
__device__ __forceinline__ void tri_gather(const SceneDev& sc, const float* __restrict__ grid, const Tri& t, float S[8]) {
  const int Y = sc.sz[1], Z = sc.sz[2];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int ix = (c & 4) ? t.i1[0] : t.i0[0];
    int iy = (c & 2) ? t.i1[1] : t.i0[1];
    int iz = (c & 1) ? t.i1[2] : t.i0[2];
    size_t flat = ((size_t)ix * Y + iy) * Z + iz;
    if (sc.flat_f32) {
      // the reference's fp32 index arithmetic, left to right: ((x * Z) * Y + y * Z) + z, each step rounded to fp32, then
      // truncated (.long()); clamped to the grid (the reference's gather would fault on the one index that rounds past it)
      const float f = __fadd_rn(__fadd_rn(__fmul_rn(__fmul_rn((float)ix, (float)Z), (float)Y), __fmul_rn((float)iy, (float)Z)), (float)iz);
      const size_t total = (size_t)sc.sz[0] * Y * Z;
      flat = (size_t)(long long)f;
      if (flat >= total) flat = total - 1;
    }
    S[c] = grid[flat];
  }
}

struct MapAB {
  float A, B, dA, dB;  // softplus10(alpha_raw), softplus10(beta_raw) and their derivatives
};
__device__ __forceinline__ MapAB map_ab(const float* __restrict__ sdf_ab) {
  MapAB m;
  m.A = pp_softplus10(sdf_ab[0]); m.B = pp_softplus10(sdf_ab[1]);
  m.dA = pp_dsoftplus10(sdf_ab[0]); m.dB = pp_dsoftplus10(sdf_ab[1]);
  return m;
}

__device__ __forceinline__ float tri_value(const Tri& t, const float G[8]) {
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float wx = (c & 4) ? t.w1[0] : t.w0[0], wy = (c & 2) ? t.w1[1] : t.w0[1], wz = (c & 1) ? t.w1[2] : t.w0[2];
    v += G[c] * ((wz * wy) * wx);
  }
  return v;
}

// gradient in voxel units (du) and the three mixed second derivatives
__device__ __forceinline__ void tri_grad(const Tri& t, const float G[8], float g[3], float h[3]) {
  g[0] = g[1] = g[2] = 0.f;
  h[0] = h[1] = h[2] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
    float wx = (c & 4) ? t.w1[0] : t.w0[0], wy = (c & 2) ? t.w1[1] : t.w0[1], wz = (c & 1) ? t.w1[2] : t.w0[2];
    g[0] += G[c] * (sx * wy * wz);
    g[1] += G[c] * (wx * sy * wz);
    g[2] += G[c] * (wx * wy * sz);
    h[0] += G[c] * (sx * sy * wz);  // d2/dxdy
    h[1] += G[c] * (sx * wy * sz);  // d2/dxdz
    h[2] += G[c] * (wx * sy * sz);  // d2/dydz
  }
}

struct GeoFwd {
  float q[3], gq[3], A[3][3], Jc[3], corr, vq, vp, sdf, grad[3];
  float cosv, ic, pc, nc, num, den, a_un;
};

__device__ __forceinline__ void geo_forward(const SceneDev& sc, const float* __restrict__ grid, const MapAB& mp,
                                            const float p[3], const float wo[16], const float v[3], float inv_s,
                                            const float scl[3], Tri& tq, float Sq[8], Tri& tp, float Sp[8],
                                            float Gq[8], float Gp[8], GeoFwd& o) {
#pragma unroll
  for (int k = 0; k < 3; ++k) o.q[k] = p[k] + wo[k];
  o.corr = wo[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o.A[i][j] = (i == j ? 1.f : 0.f) + wo[(1 + i) * 4 + j];
    o.Jc[i] = wo[(1 + i) * 4 + 3];
  }
  tri_setup(sc, o.q, tq);
  tri_gather(sc, grid, tq, Sq);
  tri_setup(sc, p, tp);
  tri_gather(sc, grid, tp, Sp);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    Gq[c] = mp.A * (pp_sigmoid(mp.B * Sq[c]) - 0.5f);
    Gp[c] = mp.A * (pp_sigmoid(mp.B * Sp[c]) - 0.5f);
  }
  o.vq = tri_value(tq, Gq);
  o.vp = tri_value(tp, Gp);
  float gu[3], hu[3];
  tri_grad(tq, Gq, gu, hu);
#pragma unroll
  for (int k = 0; k < 3; ++k) o.gq[k] = gu[k] * scl[k];
  o.sdf = o.vq + o.corr;
#pragma unroll
  for (int i = 0; i < 3; ++i) o.grad[i] = o.A[i][0] * o.gq[0] + o.A[i][1] * o.gq[1] + o.A[i][2] * o.gq[2] + o.Jc[i];
  // NeuS alpha, use_mid, cos_anneal_ratio = 1
  float dist = pp_mul(sc.stepsize, sc.voxel);
  o.cosv = v[0] * o.grad[0] + v[1] * o.grad[1] + v[2] * o.grad[2];
  o.ic = fminf(o.cosv, 0.f);
  float off = (o.ic * dist) * 0.5f;
  float nxt = o.sdf + off, prv = o.sdf - off;
  o.pc = pp_sigmoid(prv * inv_s);
  o.nc = pp_sigmoid(nxt * inv_s);
  o.num = (o.pc - o.nc) + 1e-5f;
  o.den = o.pc + 1e-5f;
  o.a_un = o.num / o.den;
}

__global__ __launch_bounds__(256) void k_geometry_fwd(SceneDev sc, const float* __restrict__ grid,
                                                      const float* __restrict__ sdf_ab, const float* __restrict__ pts,
                                                      const float* __restrict__ warp_out,
                                                      const float* __restrict__ viewdirs,
                                                      const int32_t* __restrict__ ray_id,
                                                      const int32_t* __restrict__ count, int capacity, float inv_s,
                                                      float* __restrict__ alpha, float* __restrict__ gradient,
                                                      float* __restrict__ sdf_final, float* __restrict__ sdf_deform,
                                                      float* __restrict__ grad_deform) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  if (m >= M) return;
  MapAB mp = map_ab(sdf_ab);
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  float wo[16];
  const float4* w4 = reinterpret_cast<const float4*>(warp_out + (size_t)m * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) { float4 t = w4[i]; wo[i * 4] = t.x; wo[i * 4 + 1] = t.y; wo[i * 4 + 2] = t.z; wo[i * 4 + 3] = t.w; }
  int r = ray_id[m];
  float v[3] = {viewdirs[r * 3], viewdirs[r * 3 + 1], viewdirs[r * 3 + 2]};
  float scl[3];
  for (int k = 0; k < 3; ++k) scl[k] = (float)(sc.sz[k] - 1) / (sc.mx[k] - sc.mn[k]);
  Tri tq, tp;
  float Sq[8], Sp[8], Gq[8], Gp[8];
  GeoFwd o;
  geo_forward(sc, grid, mp, p, wo, v, inv_s, scl, tq, Sq, tp, Sp, Gq, Gp, o);
  alpha[m] = fminf(fmaxf(o.a_un, 0.f), 1.f);
  for (int k = 0; k < 3; ++k) gradient[m * 3 + k] = o.grad[k];
  if (sdf_final) sdf_final[m] = o.sdf;
  if (sdf_deform) sdf_deform[m] = o.sdf - o.vp;
  if (grad_deform)
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) grad_deform[m * 9 + i * 3 + j] = o.A[i][j];
}

// PRIORS: the sample-level regularisers of object_losses (lib/losses.py:6-23: eikonal, deformation-Jacobian norm,
// |correction|, |sdf_deform|) are differentiated right here from the recomputed forward instead of by a separate kernel
// that writes g_grad_deform / g_correction / g_sdf_deform to HBM first (pp_loss_samples); same expressions, same
// order of additions, so the two routes are bit-identical.
#ifndef GEO_BWD_THREADS
#define GEO_BWD_THREADS 512
#endif
template <bool PRIORS>
__global__ __launch_bounds__(GEO_BWD_THREADS) void k_geometry_bwd(
    SceneDev sc, const float* __restrict__ grid, const float* __restrict__ sdf_ab, const float* __restrict__ pts,
    const float* __restrict__ warp_out, const float* __restrict__ viewdirs, const int32_t* __restrict__ ray_id,
    const int32_t* __restrict__ count, int capacity, float inv_s, const float* __restrict__ g_alpha,
    const float* __restrict__ g_gradient, const float* __restrict__ g_sdf_final, const float* __restrict__ g_sdf_deform,
    const float* __restrict__ g_grad_deform, const float* __restrict__ g_correction, int accumulate,
    float* __restrict__ warp_out_grad, float* __restrict__ pts_grad, float* __restrict__ vgrad_s,
    float* __restrict__ sdf_ab_grad, float w_eik, float w_dyn, float ls, float* __restrict__ loss_out,
    const float* __restrict__ batch_norm) {
  __shared__ float red[6][GEO_BWD_THREADS / 64];
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  float ga_sum = 0.f, gb_sum = 0.f;
  float l_eik = 0.f, l_gd = 0.f, l_c = 0.f, l_sd = 0.f;
  float invM = M > 0 ? 1.f / (float)M : 0.f;
  if (PRIORS && batch_norm) invM = batch_norm[1] > 0.f ? 1.f / batch_norm[1] : 0.f;   // union batch's sample count / world size
  if (m < M) {
    MapAB mp = map_ab(sdf_ab);
    float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
    float wo[16];
    const float4* w4 = reinterpret_cast<const float4*>(warp_out + (size_t)m * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { float4 t = w4[i]; wo[i * 4] = t.x; wo[i * 4 + 1] = t.y; wo[i * 4 + 2] = t.z; wo[i * 4 + 3] = t.w; }
    int r = ray_id[m];
    float v[3] = {viewdirs[r * 3], viewdirs[r * 3 + 1], viewdirs[r * 3 + 2]};
    float scl[3];
    for (int k = 0; k < 3; ++k) scl[k] = (float)(sc.sz[k] - 1) / (sc.mx[k] - sc.mn[k]);
    Tri tq, tp;
    float Sq[8], Sp[8], Gq[8], Gp[8];
    GeoFwd o;
    geo_forward(sc, grid, mp, p, wo, v, inv_s, scl, tq, Sq, tp, Sp, Gq, Gp, o);
    float dist = pp_mul(sc.stepsize, sc.voxel);
    // ---- alpha backward (clip passes the gradient on the closed interval, as torch.clamp)
    float ga = g_alpha ? g_alpha[m] : 0.f;
    if (!(o.a_un >= 0.f && o.a_un <= 1.f)) ga = 0.f;
    float n_bar = ga / o.den;
    float d_bar = -ga * o.num / (o.den * o.den);
    float pc_bar = n_bar + d_bar, nc_bar = -n_bar;
    float prv_bar = pc_bar * o.pc * (1.f - o.pc) * inv_s;
    float nxt_bar = nc_bar * o.nc * (1.f - o.nc) * inv_s;
    float sdf_bar = prv_bar + nxt_bar + (g_sdf_final ? g_sdf_final[m] : 0.f);
    float ic_bar = (nxt_bar - prv_bar) * (dist * 0.5f);
    float cos_bar = (o.cosv < 0.f) ? ic_bar : 0.f;
    float gg[3], vb[3];
    float eik[3] = {0.f, 0.f, 0.f}, gdef_bar[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, corr_bar = 0.f, sdef_bar = 0.f;
    if (PRIORS) {
      const float gn = sqrtf(o.grad[0] * o.grad[0] + o.grad[1] * o.grad[1] + o.grad[2] * o.grad[2]);
      const float e = gn - 1.f;
      l_eik = fabsf(e);
      const float sg = (e > 0.f) ? 1.f : (e < 0.f ? -1.f : 0.f);
      for (int k = 0; k < 3; ++k) eik[k] = (gn > 0.f) ? ls * w_eik * sg * o.grad[k] / gn * invM : 0.f;
      for (int i = 0; i < 3; ++i) {
        const float a0 = o.A[i][0], a1 = o.A[i][1], a2 = o.A[i][2];
        const float an = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
        l_gd += an;
        const float sc_ = (an > 0.f) ? ls * w_dyn * invM / (3.f * an) : 0.f;
        gdef_bar[i][0] = a0 * sc_; gdef_bar[i][1] = a1 * sc_; gdef_bar[i][2] = a2 * sc_;
      }
      const float c = wo[3];
      l_c = fabsf(c);
      corr_bar = ls * w_dyn * invM * ((c > 0.f) ? 1.f : (c < 0.f ? -1.f : 0.f));
      const float sd = o.sdf - o.vp;
      l_sd = fabsf(sd);
      sdef_bar = ls * w_dyn * invM * ((sd > 0.f) ? 1.f : (sd < 0.f ? -1.f : 0.f));
    }
    for (int k = 0; k < 3; ++k) {
      gg[k] = ((g_gradient ? g_gradient[m * 3 + k] : 0.f) + eik[k]) + cos_bar * v[k];
      vb[k] = cos_bar * o.grad[k];
    }
    // ---- grad = A gq + Jc
    float Abar[3][3], gq_bar[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        Abar[i][j] = gg[i] * o.gq[j] + (PRIORS ? gdef_bar[i][j] : (g_grad_deform ? g_grad_deform[m * 9 + i * 3 + j] : 0.f));
    for (int j = 0; j < 3; ++j) gq_bar[j] = o.A[0][j] * gg[0] + o.A[1][j] * gg[1] + o.A[2][j] * gg[2];
    float sd_up = PRIORS ? sdef_bar : (g_sdf_deform ? g_sdf_deform[m] : 0.f);
    float sdf_tot = sdf_bar + sd_up;
    float c_bar = sdf_tot + (PRIORS ? corr_bar : (g_correction ? g_correction[m] : 0.f));
    float vq_bar = sdf_tot, vp_bar = -sd_up;
    // ---- trilinear backward at q (value + gradient outputs) and p (value only)
    float gu[3], hu[3];
    tri_grad(tq, Gq, gu, hu);
    float gqb_u[3] = {gq_bar[0] * scl[0], gq_bar[1] * scl[1], gq_bar[2] * scl[2]};  // upstream wrt d/du
    float q_bar[3];
    q_bar[0] = (vq_bar * gu[0] + hu[0] * gqb_u[1] + hu[1] * gqb_u[2]) * scl[0];
    q_bar[1] = (vq_bar * gu[1] + hu[0] * gqb_u[0] + hu[2] * gqb_u[2]) * scl[1];
    q_bar[2] = (vq_bar * gu[2] + hu[1] * gqb_u[0] + hu[2] * gqb_u[1]) * scl[2];
    float gpu[3], hpu[3];
    tri_grad(tp, Gp, gpu, hpu);
    float pb[3];
    for (int k = 0; k < 3; ++k) pb[k] = vp_bar * gpu[k] * scl[k] + q_bar[k];
    // ---- alpha / beta gradients through the corner mapping
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
      float wx = (c & 4) ? tq.w1[0] : tq.w0[0], wy = (c & 2) ? tq.w1[1] : tq.w0[1], wz = (c & 1) ? tq.w1[2] : tq.w0[2];
      float coef = vq_bar * (wx * wy * wz) + gqb_u[0] * (sx * wy * wz) + gqb_u[1] * (wx * sy * wz) + gqb_u[2] * (wx * wy * sz);
      float sg = Gq[c] / mp.A + 0.5f;
      ga_sum += coef * mp.dA * (sg - 0.5f);
      gb_sum += coef * mp.A * sg * (1.f - sg) * Sq[c] * mp.dB;
      float wxp = (c & 4) ? tp.w1[0] : tp.w0[0], wyp = (c & 2) ? tp.w1[1] : tp.w0[1], wzp = (c & 1) ? tp.w1[2] : tp.w0[2];
      float coefp = vp_bar * (wxp * wyp * wzp);
      float sgp = Gp[c] / mp.A + 0.5f;
      ga_sum += coefp * mp.dA * (sgp - 0.5f);
      gb_sum += coefp * mp.A * sgp * (1.f - sgp) * Sp[c] * mp.dB;
    }
    // ---- outputs
    float wg[16];
    wg[0] = q_bar[0]; wg[1] = q_bar[1]; wg[2] = q_bar[2]; wg[3] = c_bar;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) wg[(1 + i) * 4 + j] = Abar[i][j]; wg[(1 + i) * 4 + 3] = gg[i]; }
    float4* o4 = reinterpret_cast<float4*>(warp_out_grad + (size_t)m * 16);
    for (int i = 0; i < 4; ++i) o4[i] = make_float4(wg[i * 4], wg[i * 4 + 1], wg[i * 4 + 2], wg[i * 4 + 3]);
    for (int k = 0; k < 3; ++k) {
      if (accumulate) { pts_grad[m * 3 + k] += pb[k]; if (vgrad_s) vgrad_s[m * 3 + k] += vb[k]; }
      else { pts_grad[m * 3 + k] = pb[k]; if (vgrad_s) vgrad_s[m * 3 + k] = vb[k]; }
    }
  }
  // block reduction of the two scalar gradients, one atomic pair per block
  ga_sum = pp_wave_sum(ga_sum);
  gb_sum = pp_wave_sum(gb_sum);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = ga_sum; red[1][wid] = gb_sum; }
  if (PRIORS) {
    l_eik = pp_wave_sum(l_eik); l_gd = pp_wave_sum(l_gd); l_c = pp_wave_sum(l_c); l_sd = pp_wave_sum(l_sd);
    if (lane == 0) { red[2][wid] = l_eik; red[3][wid] = l_gd; red[4][wid] = l_c; red[5][wid] = l_sd; }
  }
  __syncthreads();
  // the sums of a work-group: wavefront partials added in order; ONE atomic per value and work-group - all work-groups finish
  // together and the six addresses serve ~10 ns per atomic, so the kernel runs with GEO_BWD_THREADS = 512 (108 live work-groups
  // at the train step's 55 k samples: 24.4 -> 18.1 us; 14.1 us with no atomics at all; 1024 threads would spill)
  if (threadIdx.x < 6) {
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[threadIdx.x][w];
    const int k = threadIdx.x;
    if (k < 2) {
      if (sdf_ab_grad && s != 0.f) atomicAdd(&sdf_ab_grad[k], s);
    } else if (PRIORS && loss_out && blockIdx.x * blockDim.x < M) {
      atomicAdd(&loss_out[k], k == 3 ? s * invM / 3.f : s * invM);
    }
  }
}

extern "C" int pp_geometry_fwd(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab, const float* pts,
                               const float* warp_out, const float* viewdirs, const int32_t* ray_id,
                               const int32_t* count, int32_t capacity, float inv_s, float* alpha, float* gradient,
                               float* sdf_final, float* sdf_deform, float* grad_deform, void* stream) {
  PP_REQUIRE(sc && sdf_grid && sdf_ab && pts && warp_out && viewdirs && ray_id && count && alpha && gradient,
             "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  hipLaunchKernelGGL(k_geometry_fwd, dim3(pp_div_up(capacity, 256)), dim3(256), 0, pp_stream(stream), pp_scene_dev(sc),
                     sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, alpha, gradient,
                     sdf_final, sdf_deform, grad_deform);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_geometry_bwd(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab, const float* pts,
                               const float* warp_out, const float* viewdirs, const int32_t* ray_id,
                               const int32_t* count, int32_t capacity, float inv_s, const float* g_alpha,
                               const float* g_gradient, const float* g_sdf_final, const float* g_sdf_deform,
                               const float* g_grad_deform, const float* g_correction, int32_t accumulate,
                               float* warp_out_grad, float* pts_grad, float* viewdir_grad_s, float* sdf_ab_grad,
                               void* stream) {
  PP_REQUIRE(sc && sdf_grid && sdf_ab && pts && warp_out && viewdirs && ray_id && count && warp_out_grad && pts_grad,
             "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  hipLaunchKernelGGL(k_geometry_bwd<false>, dim3(pp_div_up(capacity, GEO_BWD_THREADS)), dim3(GEO_BWD_THREADS), 0, pp_stream(stream), pp_scene_dev(sc),
                     sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, g_alpha, g_gradient,
                     g_sdf_final, g_sdf_deform, g_grad_deform, g_correction, accumulate, warp_out_grad, pts_grad,
                     viewdir_grad_s, sdf_ab_grad, 0.f, 0.f, 0.f, (float*)nullptr, (const float*)nullptr);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_geometry_bwd_priors(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab, const float* pts,
                                      const float* warp_out, const float* viewdirs, const int32_t* ray_id,
                                      const int32_t* count, int32_t capacity, float inv_s, const float* g_alpha,
                                      const float* g_gradient, float w_eikonal, float w_deform, float loss_scale,
                                      int32_t accumulate, float* warp_out_grad, float* pts_grad, float* viewdir_grad_s,
                                      float* sdf_ab_grad, float* loss_out, const float* batch_norm, void* stream) {
  PP_REQUIRE(sc && sdf_grid && sdf_ab && pts && warp_out && viewdirs && ray_id && count && warp_out_grad && pts_grad,
             "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  hipLaunchKernelGGL(k_geometry_bwd<true>, dim3(pp_div_up(capacity, GEO_BWD_THREADS)), dim3(GEO_BWD_THREADS), 0, pp_stream(stream), pp_scene_dev(sc),
                     sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, g_alpha, g_gradient,
                     nullptr, nullptr, nullptr, nullptr, accumulate, warp_out_grad, pts_grad, viewdir_grad_s, sdf_ab_grad,
                     w_eikonal, w_deform, loss_scale, loss_out, batch_norm);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
