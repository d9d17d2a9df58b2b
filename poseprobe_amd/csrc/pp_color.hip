// Colour-feature stage: k0 feature lookup (DenseGrid.forward, lib/grid.py:47-58, zeros padding), BARF positional
// encodings (lib/voxurf_coarse.py:721-732, :1009-1025) and the normal feature (:1028-1030), fused into one
// [M,64] operand for the rgbnet GEMM.  k0 lives channels-last [X,Y,Z,C]: each stencil corner is one contiguous
// C*4-byte read (3 x 16 B for C=12) instead of C strided 4-byte reads of the reference's [1,C,X,Y,Z] layout.
#include "pp_common.h"
#include "pp_k0_tri.h"

__device__ __forceinline__ float pp_norm3c(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }

template <int TC, int TLP, int TLV>
__global__ __launch_bounds__(256) void k_color_feat_fwd(SceneDev sc, const float* __restrict__ k0,
                                                        const float* __restrict__ pts, const float* __restrict__ viewdirs,
                                                        const int32_t* __restrict__ ray_id,
                                                        const float* __restrict__ gradient, const float* __restrict__ pe_w,
                                                        const int32_t* __restrict__ count, int capacity,
                                                        float* __restrict__ feat) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  if (m >= M) return;
  float f[PP_FEAT_LD];
#pragma unroll
  for (int i = 0; i < PP_FEAT_LD; ++i) f[i] = 0.f;
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  K0Tri t;
  k0_setup(sc, p, t);
  const int C = TC ? TC : sc.C;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, t, c, off, w)) {
      const float4* src = reinterpret_cast<const float4*>(k0 + off);
#pragma unroll
      for (int q = 0; q < C / 4; ++q) {
        float4 v = src[q];
        f[q * 4 + 0] += v.x * w; f[q * 4 + 1] += v.y * w; f[q * 4 + 2] += v.z * w; f[q * 4 + 3] += v.w * w;
      }
    }
  }
  int o = C;
  const int Lp = TLP ? TLP : sc.Lp, Lv = TLV ? TLV : sc.Lv;
  // xyz embedding: [t(3) | w_k sin(2^k t_a) (a major, k minor) | w_k cos(...)]
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float ta = pp_div(pp_sub(p[a], sc.mn[a]), pp_sub(sc.mx[a], sc.mn[a]));
    f[o + a] = ta;
    float fr = 1.f;
#pragma unroll
    for (int k = 0; k < Lp; ++k) {
      float ang = ta * fr, s, c;
      sincosf(ang, &s, &c);
      f[o + 3 + a * Lp + k] = s * pe_w[k];
      f[o + 3 + 3 * Lp + a * Lp + k] = c * pe_w[k];
      fr *= 2.f;
    }
  }
  o += 3 + 6 * Lp;
  int r = ray_id[m];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float va = viewdirs[r * 3 + a];
    f[o + a] = va;
    float fr = 1.f;
#pragma unroll
    for (int k = 0; k < Lv; ++k) {
      float ang = va * fr, s, c;
      sincosf(ang, &s, &c);
      f[o + 3 + a * Lv + k] = s * pe_w[Lp + k];
      f[o + 3 + 3 * Lv + a * Lv + k] = c * pe_w[Lp + k];
      fr *= 2.f;
    }
  }
  o += 3 + 6 * Lv;
  float g[3] = {gradient[m * 3], gradient[m * 3 + 1], gradient[m * 3 + 2]};
  float gn = pp_norm3c(g[0], g[1], g[2]) + 1e-5f;
  for (int a = 0; a < 3; ++a) f[o + a] = g[a] / gn;
  float4* dst = reinterpret_cast<float4*>(feat + (size_t)m * PP_FEAT_LD);
#pragma unroll
  for (int q = 0; q < PP_FEAT_LD / 4; ++q) dst[q] = make_float4(f[q * 4], f[q * 4 + 1], f[q * 4 + 2], f[q * 4 + 3]);
}

// backward, part 1 (thread per sample): everything except the k0 scatter
template <int TC, int TLP, int TLV>
__global__ __launch_bounds__(256) void k_color_feat_bwd(SceneDev sc, const float* __restrict__ k0,
                                                        const float* __restrict__ pts, const float* __restrict__ viewdirs,
                                                        const int32_t* __restrict__ ray_id,
                                                        const float* __restrict__ gradient, const float* __restrict__ pe_w,
                                                        const int32_t* __restrict__ count, int capacity,
                                                        const float* __restrict__ feat_grad, float* __restrict__ pts_grad,
                                                        float* __restrict__ gradient_grad, float* __restrict__ vgrad_s) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  if (m >= M) return;
  float fg[PP_FEAT_LD];
  const float4* src4 = reinterpret_cast<const float4*>(feat_grad + (size_t)m * PP_FEAT_LD);
#pragma unroll
  for (int q = 0; q < PP_FEAT_LD / 4; ++q) { float4 v = src4[q]; fg[q * 4] = v.x; fg[q * 4 + 1] = v.y; fg[q * 4 + 2] = v.z; fg[q * 4 + 3] = v.w; }
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  K0Tri t;
  k0_setup(sc, p, t);
  const int C = TC ? TC : sc.C;
  float pb[3] = {0, 0, 0};
  // d(k0 feature)/dp : sum_c (K_c . fg) * d w_c / du * scale
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, t, c, off, w)) {
      const float4* src = reinterpret_cast<const float4*>(k0 + off);
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < C / 4; ++q) {
        float4 v = src[q];
        dot += v.x * fg[q * 4] + v.y * fg[q * 4 + 1] + v.z * fg[q * 4 + 2] + v.w * fg[q * 4 + 3];
      }
      float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
      float wx = (c & 4) ? t.w1[0] : t.w0[0], wy = (c & 2) ? t.w1[1] : t.w0[1], wz = (c & 1) ? t.w1[2] : t.w0[2];
      pb[0] += dot * sx * wy * wz;
      pb[1] += dot * wx * sy * wz;
      pb[2] += dot * wx * wy * sz;
    }
  }
  for (int a = 0; a < 3; ++a) pb[a] *= (float)(sc.sz[a] - 1) / (sc.mx[a] - sc.mn[a]);
  int o = C;
  const int Lp = TLP ? TLP : sc.Lp, Lv = TLV ? TLV : sc.Lv;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float ext = pp_sub(sc.mx[a], sc.mn[a]);
    float ta = pp_div(pp_sub(p[a], sc.mn[a]), ext);
    float tb = fg[o + a];
    float fr = 1.f;
#pragma unroll
    for (int k = 0; k < Lp; ++k) {
      float ang = ta * fr, s, c;
      sincosf(ang, &s, &c);
      tb += pe_w[k] * fr * (c * fg[o + 3 + a * Lp + k] - s * fg[o + 3 + 3 * Lp + a * Lp + k]);
      fr *= 2.f;
    }
    pb[a] += tb / ext;
  }
  o += 3 + 6 * Lp;
  int r = ray_id[m];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float va = viewdirs[r * 3 + a];
    float vb = fg[o + a];
    float fr = 1.f;
#pragma unroll
    for (int k = 0; k < Lv; ++k) {
      float ang = va * fr, s, c;
      sincosf(ang, &s, &c);
      vb += pe_w[Lp + k] * fr * (c * fg[o + 3 + a * Lv + k] - s * fg[o + 3 + 3 * Lv + a * Lv + k]);
      fr *= 2.f;
    }
    vgrad_s[m * 3 + a] = vb;
  }
  o += 3 + 6 * Lv;
  float g[3] = {gradient[m * 3], gradient[m * 3 + 1], gradient[m * 3 + 2]};
  float gn = pp_norm3c(g[0], g[1], g[2]);
  float gne = gn + 1e-5f;
  float nb[3] = {fg[o], fg[o + 1], fg[o + 2]};
  float dot = nb[0] * g[0] + nb[1] * g[1] + nb[2] * g[2];
  for (int a = 0; a < 3; ++a) {
    float gb = nb[a] / gne;
    if (gn > 0.f) gb -= g[a] * dot / (gn * gne * gne);
    gradient_grad[m * 3 + a] = gb;
    pts_grad[m * 3 + a] = pb[a];
  }
}

// backward, part 2: k0 gradient scatter.  16 lanes per sample, lane = channel: every atomic wave-instruction
// carries 4 contiguous C*4-byte segments instead of 64 scattered dwords (MI355X float atomics execute at the memory
// side in 64-B requests; one lane per row is ~17x slower - guide 'Global float atomics').
// `touched` (optional): one byte per voxel, set to 1 for every voxel that receives a contribution (lane 0 of a sample marks all
// in-range corners, whatever the gradient values - over-marking is harmless).  Plain byte stores: every writer stores the
// same value, so no atomics are needed (a bitmap with atomicOr costs +20 us of same-word serialisation here).  The
// fused optimiser pass then neither reads nor re-zeroes the gradient of unmarked voxels (pp_grid_tv_adam_step_sparse).
__device__ __forceinline__ void k0_mark(const SceneDev& sc, const K0Tri& tr, uint8_t* __restrict__ touched) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, tr, c, off, w)) touched[off / (size_t)sc.C] = 1;
  }
}

__global__ __launch_bounds__(256) void k_k0_scatter(SceneDev sc, const float* __restrict__ pts,
                                                    const int32_t* __restrict__ count, int capacity,
                                                    const float* __restrict__ feat_grad, float* __restrict__ k0_grad,
                                                    uint8_t* __restrict__ touched) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  int M = min(count[0], capacity);
  if (m >= M || ch >= sc.C) return;
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  K0Tri tr;
  k0_setup(sc, p, tr);
  if (touched && ch == 0) k0_mark(sc, tr, touched);
  float g = feat_grad[(size_t)m * PP_FEAT_LD + ch];
  if (g == 0.f) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, tr, c, off, w)) atomicAdd(&k0_grad[off + ch], w * g);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Multi-GPU exchange of the k0 gradient at SAMPLE granularity.  The dense gradient of a 160^3 x 12 grid is 196 MB, yet a
// rank's rays touch it through ~55 k samples only; what travels is therefore the scatter's INPUT, 64 bytes per sample:
// packed[m] = { d loss / d k0-feature [12], pts xyz [3], pad }, with the rank's sample count stored in packed[0][15]
// (bit pattern of an int32).  After one all-gather every rank replays the scatter for all shards into its own full
// gradient grid.  DESIGN.md 7.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_k0_pack(const float* __restrict__ pts, const float* __restrict__ feat_grad,
                                                 const int32_t* __restrict__ count, int capacity, int C,
                                                 float* __restrict__ packed) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  int M = min(count[0], capacity);
  if (t == 15) packed[15] = __int_as_float(M);          // row 0, pad slot: this shard's sample count
  if (m >= M) return;
  float v = 0.f;
  if (ch < C) v = feat_grad[(size_t)m * PP_FEAT_LD + ch];
  else if (ch >= 12 && ch < 15) v = pts[m * 3 + (ch - 12)];
  if (!(m == 0 && ch == 15)) packed[(size_t)m * PP_PACK_LD + ch] = v;
}

__global__ __launch_bounds__(256) void k_k0_scatter_packed(SceneDev sc, const float* __restrict__ packed, int capacity,
                                                           float* __restrict__ k0_grad, uint8_t* __restrict__ touched) {
  const float* __restrict__ shard = packed + (size_t)blockIdx.y * capacity * PP_PACK_LD;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  int M = min(__float_as_int(shard[15]), capacity);
  if (m >= M || ch >= sc.C) return;
  const float* row = shard + (size_t)m * PP_PACK_LD;
  float p[3] = {row[12], row[13], row[14]};
  K0Tri tr;
  k0_setup(sc, p, tr);
  if (touched && ch == 0) k0_mark(sc, tr, touched);
  float g = row[ch];
  if (g == 0.f) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, tr, c, off, w)) atomicAdd(&k0_grad[off + ch], w * g);
  }
}

extern "C" int pp_k0_pack_samples(const float* pts, const float* feat_grad, const int32_t* count, int32_t capacity,
                                  int32_t k0_dim, float* packed, void* stream) {
  PP_REQUIRE(pts && feat_grad && count && packed, "null pointer");
  PP_REQUIRE(capacity > 0 && k0_dim > 0 && k0_dim <= 12, "bad sizes");
  hipLaunchKernelGGL(k_k0_pack, dim3(pp_div_up(capacity * 16, 256)), dim3(256), 0, pp_stream(stream), pts, feat_grad, count,
                     capacity, k0_dim, packed);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_k0_scatter_samples(const pp_scene* sc, const float* pts, const int32_t* count, int32_t capacity,
                                     const float* feat_grad, float* k0_grad_cl, uint8_t* touched, void* stream) {
  PP_REQUIRE(sc && pts && count && feat_grad && k0_grad_cl, "null pointer");
  PP_REQUIRE(capacity > 0 && sc->k0_dim <= 16, "bad sizes");
  hipLaunchKernelGGL(k_k0_scatter, dim3(pp_div_up(capacity * 16, 256)), dim3(256), 0, pp_stream(stream), pp_scene_dev(sc), pts,
                     count, capacity, feat_grad, k0_grad_cl, touched);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_k0_scatter_packed(const pp_scene* sc, const float* packed, int32_t n_shards, int32_t capacity,
                                    float* k0_grad_cl, uint8_t* touched, void* stream) {
  PP_REQUIRE(sc && packed && k0_grad_cl, "null pointer");
  PP_REQUIRE(capacity > 0 && n_shards > 0 && n_shards <= 65535 && sc->k0_dim <= 12, "bad sizes");
  hipLaunchKernelGGL(k_k0_scatter_packed, dim3(pp_div_up(capacity * 16, 256), n_shards), dim3(256), 0, pp_stream(stream),
                     pp_scene_dev(sc), packed, capacity, k0_grad_cl, touched);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Generic feature builder (DirectVoxGO twin, lib/dvgo_ori.py:336-352): [k0 (C - k0_skip channels) | xyz, sin, cos |
// view, sin, cos | optional normal] with runtime widths, un-weighted encodings when pe_w == NULL, row stride `ld`.
// `sel[M]` (uint8, optional) marks the samples that take part (weights > fast_color_thres); others get zero rows.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_feat_generic_fwd(SceneDev sc, const float* __restrict__ k0,
                                                          const float* __restrict__ pts, const float* __restrict__ viewdirs,
                                                          const int32_t* __restrict__ ray_id,
                                                          const float* __restrict__ gradient, const float* __restrict__ pe_w,
                                                          const uint8_t* __restrict__ sel, int k0_skip, int ld,
                                                          const int32_t* __restrict__ count, int capacity,
                                                          float* __restrict__ feat, float* __restrict__ k0_raw) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  int M = min(count[0], capacity);
  if (m >= M) return;
  float* f = feat + (size_t)m * ld;
  for (int i = 0; i < ld; ++i) f[i] = 0.f;
  if (k0_raw) for (int c = 0; c < sc.C; ++c) k0_raw[(size_t)m * sc.C + c] = 0.f;
  if (sel && !sel[m]) return;
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  K0Tri t;
  k0_setup(sc, p, t);
  const int C = sc.C, Lp = sc.Lp, Lv = sc.Lv;
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, t, c, off, w))
      for (int ch = 0; ch < C; ++ch) {
        float v = k0[off + ch] * w;
        if (ch >= k0_skip) f[ch - k0_skip] += v;
        if (k0_raw) k0_raw[(size_t)m * C + ch] += v;
      }
  }
  int o = C - k0_skip;
  for (int a = 0; a < 3; ++a) {
    float ta = pp_div(pp_sub(p[a], sc.mn[a]), pp_sub(sc.mx[a], sc.mn[a]));
    f[o + a] = ta;
    float fr = 1.f;
    for (int k = 0; k < Lp; ++k) {
      float s, c;
      sincosf(ta * fr, &s, &c);
      float w = pe_w ? pe_w[k] : 1.f;
      f[o + 3 + a * Lp + k] = s * w;
      f[o + 3 + 3 * Lp + a * Lp + k] = c * w;
      fr *= 2.f;
    }
  }
  o += 3 + 6 * Lp;
  int r = ray_id[m];
  for (int a = 0; a < 3; ++a) {
    float va = viewdirs[r * 3 + a];
    f[o + a] = va;
    float fr = 1.f;
    for (int k = 0; k < Lv; ++k) {
      float s, c;
      sincosf(va * fr, &s, &c);
      float w = pe_w ? pe_w[Lp + k] : 1.f;
      f[o + 3 + a * Lv + k] = s * w;
      f[o + 3 + 3 * Lv + a * Lv + k] = c * w;
      fr *= 2.f;
    }
  }
  o += 3 + 6 * Lv;
  if (gradient) {
    float g[3] = {gradient[m * 3], gradient[m * 3 + 1], gradient[m * 3 + 2]};
    float gn = pp_norm3c(g[0], g[1], g[2]) + 1e-5f;
    for (int a = 0; a < 3; ++a) f[o + a] = g[a] / gn;
  }
}

// backward of the generic builder w.r.t. the k0 grid only (the DVGO twin has no pose / point gradients):
// feat_grad[M,ld] (+ optional k0_raw_grad[M,C] for the diffuse channels) -> k0_grad (atomic +=). 16 lanes per sample.
__global__ __launch_bounds__(256) void k_feat_generic_bwd_k0(SceneDev sc, const float* __restrict__ pts,
                                                             const uint8_t* __restrict__ sel, int k0_skip, int ld,
                                                             const int32_t* __restrict__ count, int capacity,
                                                             const float* __restrict__ feat_grad,
                                                             const float* __restrict__ k0_raw_grad,
                                                             float* __restrict__ k0_grad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int m = t >> 4, ch = t & 15;
  int M = min(count[0], capacity);
  if (m >= M || ch >= sc.C) return;
  if (sel && !sel[m]) return;
  float g = (ch >= k0_skip) ? feat_grad[(size_t)m * ld + ch - k0_skip] : 0.f;
  if (k0_raw_grad) g += k0_raw_grad[(size_t)m * sc.C + ch];
  if (g == 0.f) return;
  float p[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
  K0Tri tr;
  k0_setup(sc, p, tr);
  for (int c = 0; c < 8; ++c) {
    size_t off; float w;
    if (k0_corner(sc, tr, c, off, w)) atomicAdd(&k0_grad[off + ch], w * g);
  }
}

extern "C" int pp_feat_generic_fwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                                   const int32_t* ray_id, const float* gradient, const float* pe_w, const uint8_t* sel,
                                   int32_t k0_skip, int32_t ld, const int32_t* count, int32_t capacity, float* feat,
                                   float* k0_raw, void* stream) {
  PP_REQUIRE(sc && k0_cl && pts && viewdirs && ray_id && count && feat, "null pointer");
  PP_REQUIRE(capacity > 0 && sc->k0_dim <= 16, "capacity<=0 or k0_dim>16");
  int width = sc->k0_dim - k0_skip + 3 + 6 * sc->pos_pe + 3 + 6 * sc->view_pe + (gradient ? 3 : 0);
  PP_REQUIRE(ld % 32 == 0 && width <= ld && k0_skip >= 0 && k0_skip <= sc->k0_dim, "feature width does not fit ld (multiple of 32)");
  hipLaunchKernelGGL(k_feat_generic_fwd, dim3(pp_div_up(capacity, 256)), dim3(256), 0, pp_stream(stream), pp_scene_dev(sc),
                     k0_cl, pts, viewdirs, ray_id, gradient, pe_w, sel, k0_skip, ld, count, capacity, feat, k0_raw);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_feat_generic_bwd_k0(const pp_scene* sc, const float* pts, const uint8_t* sel, int32_t k0_skip, int32_t ld,
                                      const int32_t* count, int32_t capacity, const float* feat_grad,
                                      const float* k0_raw_grad, float* k0_grad_cl, void* stream) {
  PP_REQUIRE(sc && pts && count && feat_grad && k0_grad_cl, "null pointer");
  PP_REQUIRE(capacity > 0 && sc->k0_dim <= 16, "capacity<=0 or k0_dim>16");
  hipLaunchKernelGGL(k_feat_generic_bwd_k0, dim3(pp_div_up(capacity * 16, 256)), dim3(256), 0, pp_stream(stream),
                     pp_scene_dev(sc), pts, sel, k0_skip, ld, count, capacity, feat_grad, k0_raw_grad, k0_grad_cl);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_color_feat_fwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                                 const int32_t* ray_id, const float* gradient, const float* pe_w,
                                 const int32_t* count, int32_t capacity, float* feat, void* stream) {
  PP_REQUIRE(sc && k0_cl && pts && viewdirs && ray_id && gradient && pe_w && count && feat, "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  PP_REQUIRE(sc->k0_dim % 4 == 0 && sc->k0_dim <= 16, "k0_dim must be a multiple of 4 and <= 16");
  PP_REQUIRE(sc->k0_dim + 3 + 6 * sc->pos_pe + 3 + 6 * sc->view_pe + 3 <= PP_FEAT_LD, "feature width exceeds 64");
  if (sc->k0_dim == 12 && sc->pos_pe == 5 && sc->view_pe == 1)
    hipLaunchKernelGGL((k_color_feat_fwd<12, 5, 1>), dim3(pp_div_up(capacity, 256)), dim3(256), 0, pp_stream(stream),
                       pp_scene_dev(sc), k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat);
  else
    hipLaunchKernelGGL((k_color_feat_fwd<0, 0, 0>), dim3(pp_div_up(capacity, 256)), dim3(256), 0, pp_stream(stream),
                       pp_scene_dev(sc), k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_color_feat_bwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                                 const int32_t* ray_id, const float* gradient, const float* pe_w,
                                 const int32_t* count, int32_t capacity, const float* feat_grad, float* k0_grad_cl,
                                 float* pts_grad, float* gradient_grad, float* viewdir_grad_s, void* stream) {
  PP_REQUIRE(sc && k0_cl && pts && viewdirs && ray_id && gradient && pe_w && count && feat_grad && pts_grad &&
                 gradient_grad && viewdir_grad_s,
             "null pointer");
  PP_REQUIRE(capacity > 0, "capacity<=0");
  PP_REQUIRE(sc->k0_dim % 4 == 0 && sc->k0_dim <= 16, "k0_dim must be a multiple of 4 and <= 16");
  hipStream_t st = pp_stream(stream);
  if (sc->k0_dim == 12 && sc->pos_pe == 5 && sc->view_pe == 1)
    hipLaunchKernelGGL((k_color_feat_bwd<12, 5, 1>), dim3(pp_div_up(capacity, 256)), dim3(256), 0, st, pp_scene_dev(sc),
                       k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat_grad, pts_grad,
                       gradient_grad, viewdir_grad_s);
  else
    hipLaunchKernelGGL((k_color_feat_bwd<0, 0, 0>), dim3(pp_div_up(capacity, 256)), dim3(256), 0, st, pp_scene_dev(sc),
                       k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat_grad, pts_grad,
                       gradient_grad, viewdir_grad_s);
  PP_CHECK_LAUNCH();
  if (k0_grad_cl) {
    hipLaunchKernelGGL(k_k0_scatter, dim3(pp_div_up(capacity * 16, 256)), dim3(256), 0, st, pp_scene_dev(sc), pts, count,
                       capacity, feat_grad, k0_grad_cl, (uint8_t*)nullptr);
    PP_CHECK_LAUNCH();
  }
  return PP_OK;
}
