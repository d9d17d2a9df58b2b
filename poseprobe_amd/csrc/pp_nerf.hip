// Scene branch (lib/bg_nerf): the 8 x 256 NeRF MLP with BARF positional encoding and exp-cumsum compositing, forward and
// backward.  fp32 in memory, fp32 accumulation, fp32-accurate matrix products: every product runs as three fp16 products on
// v_mfma_f32_32x32x16_f16 (pp_gemm_split.h; error against fp64 equal to the fp32 matrix instructions'); option
// "nerf_split" = 0 selects the fp32 instructions (pp_gemm.h) for all of them, "nerf_split_tn" = 0 for the weight gradients only.
//
//   reference: lib/bg_nerf/source/models/frequency_nerf.py
//     :42-69    FrequencyEmbedder          (sin / cos of 2^l * pi * x, layout [coordinate][sin|cos][band])
//     :152-170  compute_raw_density        (63 -> 256 x 8, skip connection at layer 4, row 0 of the last layer = density)
//     :189-227  forward                    (softplus density, unit view direction encoding, 283 -> 128 -> 3, sigmoid)
//     :239-266  positional_encoding        (coarse-to-fine band weights)
//     :290-343  composite                  (alpha = 1 - exp(-sigma * dist), T = exp(-exclusive cumsum), weights = T * alpha)
//
// A 256-wide layer carries 64 FLOP per byte of fp32 activation traffic - twice the machine ridge on the fp32 matrix
// instructions - so unlike the 128-wide object-branch MLPs this network gains nothing from layer fusion: every layer is one
// launch of a persistent 128 x 128-tile NT GEMM (two column blocks over the same row tiles, which meet in L2), all
// activations stay resident in HBM for the backward pass (9.3 KB per sample, 3.7 GB at 3072 rays x 128 samples), ReLU masks
// as one bit per activation, and the data- and weight-gradient GEMMs of the backward pass use the same tile structure.
// With the three-product scheme the matrix time drops to a third and the GEMMs become load-bound.  The thin
// ends of the network (encoding, density head, 128 -> 3 colour head, compositing, encoding backward) are bandwidth-bound
// streaming kernels.
#define PP_NERF_TU 1
#include "pp_common.h"
#include "pp_gemm.h"
#include "pp_gemm_split.h"
#include "pp_gemm_planes.h"
#include "pp_gemm_tn256.h"

#define NERF_L3D 10
#define NERF_LV 4
#define NERF_PI 3.14159274101257324f      // float32(pi): the reference builds its frequencies as 2^l * float32(pi)


// ------------------------------------------------------------------------------------------------ parameter layout
// W0[256][64] b0 | W1..W3[256][256] b | W4[256][320] b | W5,W6[256][256] b | wd[256] W7[256][256] | bd b7[256] pad |
// R0[128][288] br0[128] | R1[3][128] br1[3] pad
static const int NERF_IN_LD[8] = {64, 256, 256, 256, 320, 256, 256, 256};
static const int NERF_OUT_LD[8] = {256, 256, 256, 320, 256, 256, 256, 288};
struct NerfLayout {
  int64_t w[8], b[8], wd, bd, r0, br0, r1, br1, total;
};
static NerfLayout nerf_layout() {
  NerfLayout L;
  int64_t o = 0;
  for (int l = 0; l < 7; ++l) { L.w[l] = o; o += 256 * NERF_IN_LD[l]; L.b[l] = o; o += 256; }
  // the reference's last feature layer is one [257][256] matrix whose row 0 is the density: wd sits right in front of the
  // feature rows and bd in front of their biases, so that layer is ONE contiguous tensor for checkpoints / optimisers
  L.wd = o; o += 256; L.w[7] = o; o += 256 * 256;
  L.bd = o; o += 1; L.b[7] = o; o += 256; o += 63;
  L.r0 = o; o += 128 * 288; L.br0 = o; o += 128;
  L.r1 = o; o += 3 * 128; L.br1 = o; o += 64;
  L.total = o;
  return L;
}

extern "C" int pp_nerf_layout(int64_t* offsets) {
  PP_REQUIRE(offsets, "null pointer");
  NerfLayout L = nerf_layout();
  for (int l = 0; l < 8; ++l) { offsets[2 * l] = L.w[l]; offsets[2 * l + 1] = L.b[l]; }
  offsets[16] = L.wd; offsets[17] = L.bd; offsets[18] = L.r0; offsets[19] = L.br0; offsets[20] = L.r1; offsets[21] = L.br1;
  offsets[22] = L.total;
  return PP_OK;
}

// activations kept for the backward pass; rows = samples
// mx: largest magnitudes of the GEMM operands for the split-precision path (pp_gemm_split.h); slots below
// bits[l]: ReLU mask of layer l's output, one bit per activation, 32 bytes per sample (layout: pp_gemm.h gemm_epilogue)
// wimg[l]: the layer's weights as split-precision LDS images (pp_gemm_planes.h), rebuilt by every forward pass
struct NerfActs { float* enc; float* a[8]; float* h; float* raw; float* mx; uint16_t* bits[8]; _Float16* wimg[8]; };
enum { MX_ENC = 0, MX_A0 = 1 /* .. MX_A0 + 7 */, MX_DH = 9, MX_P = 10, MX_DY6 = 11 /* dY6 .. dY0 = 11 .. 17 */, MX_DHSUM = 18,
       MX_W0 = 32 /* .. 39 */, MX_R0 = 40, MX_SLOTS = 64 };
static NerfActs nerf_acts(float* base, int64_t M) {
  NerfActs A;
  float* p = base;
  A.enc = p; p += M * 64;
  for (int l = 0; l < 8; ++l) { A.a[l] = p; p += M * NERF_OUT_LD[l]; }
  A.h = p; p += M * 128;
  A.raw = p; p += M;
  A.mx = p; p += MX_SLOTS;
  for (int l = 0; l < 8; ++l) { A.bits[l] = reinterpret_cast<uint16_t*>(p); p += (M + 127) / 128 * 128 * 8; }
  for (int l = 0; l < 8; ++l) { A.wimg[l] = reinterpret_cast<_Float16*>(p); p += 256 * NERF_IN_LD[l]; }   // hi + lo halfs = 4 B per weight
  return A;
}
static const int64_t NERF_WIMG_FLOATS = 256 * (64 + 256 * 6 + 320) + 10 * 8192;   // + the head stage's ten steps of the fused chain's weight stream (pp_nerf_trunk.h)
static int64_t nerf_acts_floats(int64_t M) {
  return M * (64 + 256 * 6 + 320 + 288 + 128 + 1) + MX_SLOTS + (M + 127) / 128 * 128 * 64 + NERF_WIMG_FLOATS;
}
static const int64_t NERF_WT_FLOATS = 5 * 65536 + 320 * 256 + 256 * 288 + 64 * 256 + 288 * 128;
// images of the transposed weights for the data-gradient GEMMs: R0^T (K = 128), W7^T (K = 288), W1^T .. W6^T (K = 256)
static const int64_t NERF_WTIMG_FLOATS = 256 * (128 + 288 + 6 * 256);
static const int64_t NERF_PART_FLOATS = 512 * 392;       // NERF_PART_WGS x NERF_PART_LD: partial rows of the thin heads' backward kernels
static int64_t nerf_scratch_floats(int64_t M, int64_t R) {
  // (the last term: d(layer 6) .. d(layer 0) side by side for the fused data-gradient chain, whose weight-gradient products run after it)
  return M * (320 * 2 + 64 * 2) + R * (128 + 32) + NERF_WT_FLOATS + NERF_WTIMG_FLOATS + NERF_PART_FLOATS + M * 256 * 7;
}

extern "C" int pp_nerf_workspace(int64_t n_samples, int64_t n_rays, int64_t* acts_floats, int64_t* scratch_floats) {
  PP_REQUIRE(acts_floats && scratch_floats && n_samples > 0 && n_rays > 0, "bad arguments");
  *acts_floats = nerf_acts_floats(n_samples);
  *scratch_floats = nerf_scratch_floats(n_samples, n_rays);
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ encoding
// One wavefront per sample: lane j produces element j of the 64-wide (63 used) encoded point and the 32-wide (27 used)
// encoded unit view direction; both are written where the consuming layers read them (layer 0 input, the skip columns of
// layer 4's input, the view columns of the colour head's input), 256-byte coalesced rows.
__device__ __forceinline__ float nerf_enc_elem(int j, int L, const float* __restrict__ x, const float* __restrict__ w) {
  if (j < 3) return x[j];
  const int q = j - 3;
  if (q >= 6 * L) return 0.f;
  const int c = q / (2 * L), r = q - c * 2 * L, s = r / L, l = r - s * L;
  const float ph = x[c] * (NERF_PI * (float)(1 << l));
  return (s == 0 ? sinf(ph) : cosf(ph)) * w[l];
}

__global__ __launch_bounds__(256) void k_nerf_encode(const float* __restrict__ center, const float* __restrict__ ray,
                                                     const float* __restrict__ depth, const float* __restrict__ bands, int M, int S,
                                                     float* __restrict__ enc, float* __restrict__ a3, float* __restrict__ a7) {
  // a wavefront takes four consecutive samples: the view encoding (sines / cosines of the ray's unit direction) is evaluated once
  // per ray it meets, not once per sample
  const int mw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  const int j = threadIdx.x & 63;
  int rprev = -1;
  float d[3] = {0.f, 0.f, 0.f}, o[3] = {0.f, 0.f, 0.f}, ve = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mw + i;
    if (m >= M) return;
    const int r = m / S;
    if (r != rprev) {
      rprev = r;
#pragma unroll
      for (int c = 0; c < 3; ++c) { d[c] = ray[r * 3 + c]; o[c] = center[r * 3 + c]; }
      if (j < 32) {
        const float n = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
        const float u[3] = {d[0] / n, d[1] / n, d[2] / n};
        ve = nerf_enc_elem(j, NERF_LV, u, bands + NERF_L3D);
      }
    }
    const float t = depth[m];
    float x[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = o[c] + d[c] * t;
    const float e = nerf_enc_elem(j, NERF_L3D, x, bands);
    enc[(size_t)m * 64 + j] = e;
    a3[(size_t)m * 320 + 256 + j] = e;
    if (j < 32) a7[(size_t)m * 288 + 256 + j] = ve;
  }
}

// split-precision path: bound of the encoded points' magnitude, max(1, |center + ray * t|_inf) at the smallest and largest depth of every ray
// (the sines / cosines and the view encoding are bounded by 1) -> operand-maximum slots of the three consumers
__global__ __launch_bounds__(256) void k_nerf_enc_bound(const float* __restrict__ center, const float* __restrict__ ray,
                                                        const float* __restrict__ depth, int R, int S, float* __restrict__ mx) {
  // one wavefront per ray, a few rays per wavefront (<= 64 work-groups): the three slots see one atomic per WORK-GROUP - with a
  // wavefront per ray the 3 R same-address atomics were the whole run time (37 us at 1023 rays)
  __shared__ float red[4];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float v = 1.f;
  for (int r = blockIdx.x * 4 + wid; r < R; r += gridDim.x * 4) {
    float t0 = 3.0e38f, t1 = -3.0e38f;            // smallest / largest depth of the ray, whatever the sample order
    for (int i = lane; i < S; i += 64) { const float t = depth[(size_t)r * S + i]; t0 = fminf(t0, t); t1 = fmaxf(t1, t); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t0 = fminf(t0, __shfl_xor(t0, o, 64)); t1 = fmaxf(t1, __shfl_xor(t1, o, 64)); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float o = center[r * 3 + c], d = ray[r * 3 + c];
      v = fmaxf(v, fmaxf(fabsf(o + d * t0), fabsf(o + d * t1)));
    }
  }
  if (lane == 0) red[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned int*>(mx + MX_ENC), __float_as_uint(v));
    atomicMax(reinterpret_cast<unsigned int*>(mx + MX_A0 + 3), __float_as_uint(v));
    atomicMax(reinterpret_cast<unsigned int*>(mx + MX_A0 + 7), __float_as_uint(1.f));
  }
}

// largest |w| of the nine GEMM weight matrices (blockIdx.x selects; the last feature layer includes its density row)
struct NerfWmaxJobs { const float* src[9]; int n[9]; };
__global__ __launch_bounds__(256) void k_nerf_wmax(NerfWmaxJobs J, float* __restrict__ mx) {
  const int q = blockIdx.y;
  const float* __restrict__ p = J.src[q];
  // (every matrix starts 16-byte aligned and has a multiple of four entries: pp_nerf_layout)
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
  float v = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < (J.n[q] >> 2); i += gridDim.x * 256) {
    const float4 x = p4[i];
    v = fmaxf(fmaxf(v, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
  }
  // one atomic per WORK-GROUP: the same-address atomics were the kernel's time (15 us with 128 per slot, 28 us with 256)
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned int*>(mx + MX_W0 + q), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// ------------------------------------------------------------------------------------------------ thin heads
__device__ __forceinline__ float nerf_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float nerf_dsoftplus(float x) { return x > 20.f ? 1.f : pp_sigmoid(x); }

// density head: raw = a6 . wd + bd ; one wavefront per sample (float4 per lane)
__global__ __launch_bounds__(256) void k_nerf_density_fwd(const float* __restrict__ a6, const float* __restrict__ wd,
                                                          const float* __restrict__ bd, int M, float* __restrict__ raw,
                                                          float* __restrict__ density) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int j = threadIdx.x & 63;
  if (m >= M) return;
  const float4 x = *reinterpret_cast<const float4*>(a6 + (size_t)m * 256 + j * 4);
  const float4 w = *reinterpret_cast<const float4*>(wd + j * 4);
  float s = pp_wave_sum(x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w);
  if (j == 0) { s += bd[0]; raw[m] = s; density[m] = nerf_softplus(s); }
}

// colour head: rgb = sigmoid(h . R1^T + br1) ; 16 lanes per sample
#include "pp_nerf_trunk.h"

__global__ __launch_bounds__(256) void k_nerf_rgb_fwd(const float* __restrict__ R1, const float* __restrict__ br1,
                                                      const float* __restrict__ h, int M, float* __restrict__ rgb) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = t >> 4, sub = t & 15;
  const bool live = m < M;
  float4 xa = make_float4(0, 0, 0, 0), xb = xa;
  if (live) {
    const float4* xp = reinterpret_cast<const float4*>(h + (size_t)m * 128 + sub * 8);
    xa = xp[0]; xb = xp[1];
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float4* wp = reinterpret_cast<const float4*>(R1 + o * 128 + sub * 8);
    const float4 wa = wp[0], wb = wp[1];
    float s = xa.x * wa.x + xa.y * wa.y + xa.z * wa.z + xa.w * wa.w + xb.x * wb.x + xb.y * wb.y + xb.z * wb.z + xb.w * wb.w;
    s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if (live && sub == 0) rgb[(size_t)m * 3 + o] = pp_sigmoid(s + br1[o]);
  }
}

// The two thin heads' backward kernels leave their parameter-gradient sums and operand maxima as per-work-group PARTIAL rows
// (part[blockIdx.x][NERF_PART_LD], no atomics), k_nerf_part_finish adds the rows up in a fixed order: deterministic, and the
// kernels can run with as many work-groups as the streaming needs (round 1: 257 / 387 same-address atomics per work-group
// forced 1024-row strips on 128 of the 256 CUs: 77 us for a 134 MB read).
#define NERF_PART_WGS 512
#define NERF_PART_LD 392          // rgb head: 3 x 128 weight sums, 3 bias sums, max | density head: 256 weight sums, bias sum, max

// colour head backward: dH[m][j] = [h > 0] * sum_o gl_o R1[o][j], R1bar, br1bar ; 256-row strips, thread = (row phase of 4, feature),
// eight rows in flight per thread
#define NERF_STRIP 256
__global__ __launch_bounds__(512) void k_nerf_rgb_bwd(const float* __restrict__ R1, const float* __restrict__ h,
                                                      const float* __restrict__ rgb, const float* __restrict__ g_rgb, int M,
                                                      float* __restrict__ dH, float* __restrict__ part) {
  __shared__ float red[3][3 * 128 + 4];
  __shared__ float redm[8];
  const int ph = threadIdx.x >> 7, j = threadIdx.x & 127;
  const float w[3] = {R1[j], R1[128 + j], R1[256 + j]};
  float wacc[3] = {0, 0, 0}, bacc = 0.f, hmax = 0.f;
  for (int m0 = blockIdx.x * NERF_STRIP; m0 < M; m0 += gridDim.x * NERF_STRIP) {
    const int mend = min(m0 + NERF_STRIP, M);
    for (int mb = m0 + ph; mb < mend; mb += 32) {
      float x[8], g0[8], g1[8], g2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = mb + 4 * u;
        const bool ok = m < mend;
        const int mm = ok ? m : m0;
        x[u] = h[(size_t)mm * 128 + j];
        const float r0 = rgb[(size_t)mm * 3], r1 = rgb[(size_t)mm * 3 + 1], r2 = rgb[(size_t)mm * 3 + 2];
        g0[u] = ok ? g_rgb[(size_t)mm * 3] * r0 * (1.f - r0) : 0.f;
        g1[u] = ok ? g_rgb[(size_t)mm * 3 + 1] * r1 * (1.f - r1) : 0.f;
        g2[u] = ok ? g_rgb[(size_t)mm * 3 + 2] * r2 * (1.f - r2) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = mb + 4 * u;
        wacc[0] += g0[u] * x[u]; wacc[1] += g1[u] * x[u]; wacc[2] += g2[u] * x[u];
        const float hb = g0[u] * w[0] + g1[u] * w[1] + g2[u] * w[2];
        if (m < mend) dH[(size_t)m * 128 + j] = (x[u] > 0.f) ? hb : 0.f;
        hmax = fmaxf(hmax, fabsf(hb));
        bacc += (j == 0) ? g0[u] : (j == 1) ? g1[u] : g2[u];          // used by j < 3 only
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = hmax;
  if (ph > 0) {
    for (int o = 0; o < 3; ++o) red[ph - 1][o * 128 + j] = wacc[o];
    if (j < 3) red[ph - 1][384 + j] = bacc;
  }
  __syncthreads();
  float* __restrict__ row = part + (size_t)blockIdx.x * NERF_PART_LD;
  if (ph == 0) {
    for (int o = 0; o < 3; ++o) row[o * 128 + j] = (wacc[o] + red[0][o * 128 + j]) + (red[1][o * 128 + j] + red[2][o * 128 + j]);
    if (j < 3) row[384 + j] = (bacc + red[0][384 + j]) + (red[1][384 + j] + red[2][384 + j]);
  }
  if (threadIdx.x == 0) {
    float v = redm[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = fmaxf(v, redm[i]);
    row[387] = v;
  }
}

// density head backward: column 256 of the last feature layer's output gradient carries d raw; wd / bd gradients.
// 256-row strips on eight wavefronts; a lane owns 4 consecutive columns (float4 loads), a wavefront every 8th row, eight rows in
// flight per lane.
#define NERF_DSTRIP 256
__global__ __launch_bounds__(512) void k_nerf_density_bwd(const float* __restrict__ a6, const float* __restrict__ raw,
                                                          const float* __restrict__ g_density, int M,
                                                          float* __restrict__ dY7, float* __restrict__ part) {
  __shared__ float4 red[8][64];
  __shared__ float redb[8], redm[8];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float bacc = 0.f, gmax = 0.f;
  for (int m0 = blockIdx.x * NERF_DSTRIP; m0 < M; m0 += gridDim.x * NERF_DSTRIP) {
    const int mend = min(m0 + NERF_DSTRIP, M);
    for (int mb = m0 + wid; mb < mend; mb += 64) {
      float4 x[8];
      float g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = mb + 8 * u;
        const bool ok = m < mend;
        const int mm = ok ? m : m0;
        x[u] = *reinterpret_cast<const float4*>(a6 + (size_t)mm * 256 + lane * 4);
        g[u] = ok ? g_density[mm] * nerf_dsoftplus(raw[mm]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = mb + 8 * u;
        acc.x += g[u] * x[u].x; acc.y += g[u] * x[u].y; acc.z += g[u] * x[u].z; acc.w += g[u] * x[u].w;
        if (m < mend && lane < 32) dY7[(size_t)m * 288 + 256 + lane] = (lane == 0) ? g[u] : 0.f;
        bacc += g[u];
        gmax = fmaxf(gmax, fabsf(g[u]));
      }
    }
  }
  red[wid][lane] = acc;
  if (lane == 0) { redb[wid] = bacc; redm[wid] = gmax; }        // g is uniform over the wavefront
  __syncthreads();
  float* __restrict__ row = part + (size_t)blockIdx.x * NERF_PART_LD;
  if (wid == 0) {
    float4 s = red[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float4 t = red[w][lane];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(row + lane * 4) = s;
    if (lane == 0) {
      float sb = redb[0], sm = redm[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) { sb += redb[w]; sm = fmaxf(sm, redm[w]); }
      row[256] = sb;
      row[257] = sm;
    }
  }
}

// dst0[c] += sum_rows part[row][c] (c < n0), dst1[c - n0] += ... (n0 <= c < n0 + n1), slot = max(slot, max_rows part[row][n0 + n1]);
// rows are added in a fixed order (sixteen interleaved runs, then the run sums in order)
__global__ __launch_bounds__(1024) void k_nerf_part_finish(const float* __restrict__ part, int rows, float* __restrict__ dst0, int n0,
                                                           float* __restrict__ dst1, int n1, float* __restrict__ mx_slot) {
  __shared__ float red[16][64];
  const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, q = threadIdx.x >> 6;
  const int ncols = n0 + n1 + 1;
  const bool is_max = c == n0 + n1;
  float acc = 0.f;
  if (c < ncols) {
#pragma unroll 8
    for (int r = q; r < rows; r += 16) {
      const float v = part[(size_t)r * NERF_PART_LD + c];
      acc = is_max ? fmaxf(acc, v) : acc + v;
    }
  }
  red[q][l] = acc;
  __syncthreads();
  if (q != 0 || c >= ncols) return;
  float v = red[0][l];
#pragma unroll
  for (int i = 1; i < 16; ++i) v = is_max ? fmaxf(v, red[i][l]) : v + red[i][l];
  if (is_max) {
    if (mx_slot) atomicMax(reinterpret_cast<unsigned int*>(mx_slot), __float_as_uint(v));
  } else if (c < n0) {
    dst0[c] += v;
  } else {
    dst1[c - n0] += v;
  }
}

// dst[c * ldd + r] = src[r * lds + c] for all nine weight matrices of the backward pass in ONE launch: blockIdx.y selects the matrix
struct NerfTransposeJobs { const float* src[9]; float* dst[9]; int lds[9], rows[9], cols[9], ldd[9]; };
__global__ __launch_bounds__(256) void k_nerf_transpose_all(NerfTransposeJobs J) {
  const int q = blockIdx.y;
  const int rows = J.rows[q], cols = J.cols[q];
  const float* __restrict__ src = J.src[q];
  float* __restrict__ dst = J.dst[q];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int c = i / rows, r = i - c * rows;
    dst[(size_t)c * J.ldd[q] + r] = src[(size_t)r * J.lds[q] + c];
  }
}

// column 256 of the transposed last feature layer = density row wd ; columns 257..287 stay zero
__global__ void k_nerf_wd_column(const float* __restrict__ wd, float* __restrict__ w7t) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 256) return;
  w7t[(size_t)k * 288 + 256] = wd[k];
  for (int c = 257; c < 288; ++c) w7t[(size_t)k * 288 + c] = 0.f;
}

// dHsum[r][j] = sum over the S samples of ray r of dH[m][j] ; 4 wavefront pairs per ray walk every 4th sample; a work-group takes
// every gridDim.x-th ray and records its maximum once
__global__ __launch_bounds__(512) void k_nerf_ray_sum(const float* __restrict__ dH, int R, int S, float* __restrict__ out,
                                                      float* __restrict__ mx_sum) {
  __shared__ float red[4][128];
  __shared__ float redm[2];
  const int j = threadIdx.x & 127, q = threadIdx.x >> 7;
  float vmax = 0.f;
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    float acc = 0.f;
    const float* p = dH + (size_t)r * S * 128 + j;
#pragma unroll 8
    for (int s = q; s < S; s += 4) acc += p[(size_t)s * 128];
    red[q][j] = acc;
    __syncthreads();
    if (q == 0) {
      acc = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
      out[(size_t)r * 128 + j] = acc;
      vmax = fmaxf(vmax, fabsf(acc));
    }
    __syncthreads();
  }
  if (mx_sum && q == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = vmax;
  }
  __syncthreads();
  if (mx_sum && threadIdx.x == 0) pp_record_max_lane(mx_sum, fmaxf(redm[0], redm[1]));
}

// ------------------------------------------------------------------------------------------------ compositing
// One wavefront per ray; lane i owns samples i, i + 64, ... ; exclusive prefix sums of sigma * dist by wave scan.
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_nerf_composite_fwd(const float* __restrict__ rgb_s, const float* __restrict__ density,
                                                            const float* __restrict__ depth, const float* __restrict__ ray,
                                                            int R, int S, int white_bg, float* __restrict__ rgb,
                                                            float* __restrict__ depth_out, float* __restrict__ opacity,
                                                            float* __restrict__ weights, float* __restrict__ all_cum,
                                                            float* __restrict__ rgb_var, float* __restrict__ depth_var) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float len = sqrtf(ray[r * 3] * ray[r * 3] + ray[r * 3 + 1] * ray[r * 3 + 1] + ray[r * 3 + 2] * ray[r * 3 + 2]);
  const size_t base = (size_t)r * S;
  float carry = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_o = 0.f, t_pen = 1.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    float sd = 0.f, t = 0.f;
    if (s < S) {
      t = depth[base + s];
      const float intv = (s + 1 < S) ? depth[base + s + 1] - t : 1e10f;
      sd = density[base + s] * (intv * len);
    }
    const float incl = wave_incl_scan(sd, lane);
    const float prev = __shfl_up(incl, 1, 64);                    // exclusive prefix without "incl - sd": the last interval
    const float T = expf(-(carry + (lane > 0 ? prev : 0.f)));     // is 1e10 long and would cancel the prefix away
    if (s < S) {
      const float w = T * (1.f - expf(-sd));
      weights[base + s] = w;
      acc_r += w * rgb_s[(base + s) * 3]; acc_g += w * rgb_s[(base + s) * 3 + 1]; acc_b += w * rgb_s[(base + s) * 3 + 2];
      acc_d += w * t; acc_o += w;
      if (s == S - 2) t_pen = T;
    }
    carry += __shfl(incl, 63, 64);
  }
  acc_r = pp_wave_sum(acc_r); acc_g = pp_wave_sum(acc_g); acc_b = pp_wave_sum(acc_b);
  acc_d = pp_wave_sum(acc_d); acc_o = pp_wave_sum(acc_o);
  // second pass for the two variance outputs (forward only, they feed logging in the reference)
  float vr = 0.f, vd = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float w = weights[base + s];
    const float dt = depth[base + s] - acc_d;
    vd += w * dt * dt;
    vr += w * ((rgb_s[(base + s) * 3] - acc_r) + (rgb_s[(base + s) * 3 + 1] - acc_g) + (rgb_s[(base + s) * 3 + 2] - acc_b));
  }
  vr = pp_wave_sum(vr); vd = pp_wave_sum(vd);
  const int pen_lane = (S - 2) & 63;
  const float tp = __shfl(t_pen, pen_lane, 64);
  if (lane == 0) {
    const float bgc = white_bg ? (1.f - acc_o) : 0.f;
    rgb[r * 3] = acc_r + bgc; rgb[r * 3 + 1] = acc_g + bgc; rgb[r * 3 + 2] = acc_b + bgc;
    depth_out[r] = acc_d; opacity[r] = acc_o; all_cum[r] = (S >= 2) ? tp : 1.f;
    rgb_var[r] = vr; depth_var[r] = vd;
  }
}

// g_sd[j] = gw[j] * T[j+1] - sum_{i > j} gw[i] * w[i]  with  gw[i] = g_rgb . rgb_s[i] + g_depth * t[i] + g_op + g_w[i]
__global__ __launch_bounds__(256) void k_nerf_composite_bwd(const float* __restrict__ rgb_s, const float* __restrict__ density,
                                                            const float* __restrict__ depth, const float* __restrict__ ray,
                                                            const float* __restrict__ weights, int R, int S, int white_bg,
                                                            const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_op, const float* __restrict__ g_w,
                                                            float* __restrict__ g_rgb_s, float* __restrict__ g_density,
                                                            float* __restrict__ g_ray) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float d0 = ray[r * 3], d1 = ray[r * 3 + 1], d2 = ray[r * 3 + 2];
  const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
  const size_t base = (size_t)r * S;
  const float gr = g_rgb[r * 3], gg = g_rgb[r * 3 + 1], gb = g_rgb[r * 3 + 2];
  const float gd = g_depth[r];
  const float go = g_op[r] - (white_bg ? (gr + gg + gb) : 0.f);
  // forward prefix (for T) chunk by chunk, kept per chunk in registers would need S/64 slots: recompute T from the
  // exclusive prefix in a first sweep, the suffix sum in a second (reverse) sweep.
  float total_after = 0.f;      // sum_{i > chunk} gw[i] * w[i]
  float g_len = 0.f;
  const int nchunk = (S + 63) >> 6;
  // prefix of sd at chunk starts
  float carry_arr[8];
  {
    float carry = 0.f;
    for (int c = 0; c < nchunk && c < 8; ++c) {
      carry_arr[c] = carry;
      const int s = c * 64 + lane;
      float sd = 0.f;
      if (s < S) {
        const float t = depth[base + s];
        const float intv = (s + 1 < S) ? depth[base + s + 1] - t : 1e10f;
        sd = density[base + s] * (intv * len);
      }
      carry += pp_wave_sum(sd);
    }
  }
  for (int c = nchunk - 1; c >= 0; --c) {
    const int s = c * 64 + lane;
    float sd = 0.f, t = 0.f, intv = 0.f, dens = 0.f, w = 0.f, gw = 0.f;
    if (s < S) {
      t = depth[base + s];
      intv = (s + 1 < S) ? depth[base + s + 1] - t : 1e10f;
      dens = density[base + s];
      sd = dens * (intv * len);
      w = weights[base + s];
      const float cr = rgb_s[(base + s) * 3], cg = rgb_s[(base + s) * 3 + 1], cb = rgb_s[(base + s) * 3 + 2];
      gw = gr * cr + gg * cg + gb * cb + gd * t + go + (g_w ? g_w[base + s] : 0.f);
      g_rgb_s[(base + s) * 3] = gr * w; g_rgb_s[(base + s) * 3 + 1] = gg * w; g_rgb_s[(base + s) * 3 + 2] = gb * w;
    }
    const float incl = wave_incl_scan(sd, lane);
    const float Tnext = expf(-(carry_arr[c] + incl));             // T[s + 1]
    const float gww = gw * w;
    const float incl_g = wave_incl_scan(gww, lane);
    // read the chunk total at the LAST VALID lane: there "chunk_sum - incl_g" is exactly zero, which the final sample needs
    // (its interval is 1e10 long - any rounding residue of two differently associated sums would be multiplied by it)
    const float chunk_sum = __shfl(incl_g, min(63, S - 1 - c * 64), 64);
    const float after = total_after + (chunk_sum - incl_g);       // sum over i > s
    if (s < S) {
      const float gsd = gw * Tnext - after;
      g_density[base + s] = gsd * (intv * len);
      g_len += gsd * dens * intv;
    }
    total_after += chunk_sum;
  }
  g_len = pp_wave_sum(g_len);
  if (lane == 0) {
    const float inv = (len > 0.f) ? g_len / len : 0.f;
    g_ray[r * 3] = inv * d0; g_ray[r * 3 + 1] = inv * d1; g_ray[r * 3 + 2] = inv * d2;
  }
}

// ------------------------------------------------------------------------------------------------ encoding backward
// One work-group (4 wavefronts) per ray; a wavefront walks every 4th sample, lane j owns element j of the encoding
// gradient (layer 0's plus the skip layer's).  d/dx of w sin(f x) is f * (w cos(f x)) - the partner element of the stored
// encoding - so no trigonometry is re-evaluated.  The tail adds the view-direction path (through the normalisation) and
// writes d center = sum_s d pts, d ray = sum_s depth * d pts + view term.
__global__ __launch_bounds__(256) void k_nerf_encode_bwd(const float* __restrict__ enc, const float* __restrict__ dEnc0,
                                                         const float* __restrict__ dEncS, const float* __restrict__ dView,
                                                         const float* __restrict__ a7, const float* __restrict__ ray,
                                                         const float* __restrict__ depth, int R, int S,
                                                         float* __restrict__ g_center, float* __restrict__ g_ray) {
  __shared__ float red[4][6];
  const int r = blockIdx.x;
  const int wid = threadIdx.x >> 6, j = threadIdx.x & 63;
  // static role of lane j
  int c = -1, sgn = 0, partner = j;
  float f = 0.f;
  if (j < 3) { c = j; }
  else if (j < 3 + 6 * NERF_L3D) {
    const int q = j - 3;
    c = q / (2 * NERF_L3D);
    const int rr = q - c * 2 * NERF_L3D, s = rr / NERF_L3D, l = rr - s * NERF_L3D;
    f = NERF_PI * (float)(1 << l);
    sgn = (s == 0) ? 1 : -1;
    partner = (s == 0) ? j + NERF_L3D : j - NERF_L3D;
  }
  // a lane sums ITS element's contributions over the samples it meets (d/d center) and the depth-weighted ones (d/d ray); the
  // three per-coordinate sums over the lanes are formed once per wavefront at the end, not once per sample
  float lc = 0.f, lr = 0.f;
#pragma unroll 4
  for (int s = wid; s < S; s += 4) {
    const size_t m = (size_t)r * S + s;
    const float g = dEnc0[m * 64 + j] + dEncS[m * 64 + j];
    const float e = enc[m * 64 + j];
    const float ep = __shfl(e, partner, 64);
    const float contrib = (j < 3) ? g : (float)sgn * f * ep * g;
    lc += contrib;
    lr += depth[m] * contrib;
  }
  float gc[3], gr[3];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    gc[cc] = pp_wave_sum(c == cc ? lc : 0.f);
    gr[cc] = pp_wave_sum(c == cc ? lr : 0.f);
  }
  if (j == 0) { for (int i = 0; i < 3; ++i) { red[wid][i] = gc[i]; red[wid][3 + i] = gr[i]; } }
  __syncthreads();
  if (threadIdx.x == 0) {
    float o[6];
    for (int i = 0; i < 6; ++i) o[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    // view-direction path: venc = (u, w sin(f u), w cos(f u)), u = ray / |ray|
    const float d[3] = {ray[r * 3], ray[r * 3 + 1], ray[r * 3 + 2]};
    const float n = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
    const float* ve = a7 + (size_t)r * S * 288 + 256;          // encoded view direction of the ray's first sample
    const float* gv = dView + (size_t)r * 32;
    float gu[3], u[3], dot = 0.f;
    for (int cc = 0; cc < 3; ++cc) {
      u[cc] = d[cc] / n;
      float acc = gv[cc];
      for (int l = 0; l < NERF_LV; ++l) {
        const float fl = NERF_PI * (float)(1 << l);
        const int js = 3 + cc * 2 * NERF_LV + l, jc = js + NERF_LV;
        acc += fl * (ve[jc] * gv[js] - ve[js] * gv[jc]);
      }
      gu[cc] = acc;
      dot += u[cc] * acc;
    }
    for (int cc = 0; cc < 3; ++cc) {
      g_center[r * 3 + cc] = o[cc];
      g_ray[r * 3 + cc] = o[3 + cc] + (gu[cc] - u[cc] * dot) / n;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static const int NERF_BM = 128;
// Tuning options (pp_set_option; defaults = the measured best on MI355X):
//   nerf_gemm_wgs      persistent work-groups per column block (256 = 512 > 384 > 128)
//   nerf_tn_ch         rows per LDS chunk of the fp32 weight-gradient GEMM (32 | 64)
//   nerf_tn_split_wgs  row splits of the split-precision weight-gradient kernel
//   nerf_tn_wgs        row splits of a weight-gradient block: 128 x 4 blocks = 2 work-groups per CU, one round
//   nerf_bn = 256      128 x 256 tiles for the fp32 NT GEMM (activation tile read once, half the barriers per MFMA).  Measured
//                      SLOWER (3072 x 128 samples: 15.8 vs 13.8 ms per step): 128 accumulators + operand staging do not fit 256
//                      registers without spills inside the K loop.  Kept for A/B runs.
//   nerf_split         NT GEMMs (forward and data gradients) as three fp16 products with fp32 accumulation (pp_gemm_split.h:
//                      error against fp64 equal to the fp32 matrix instructions', a third of their matrix-pipe time); 0 puts
//                      them on the fp32 matrix instructions (A/B runs, bisecting)
//   nerf_split_tn      the same for the weight-gradient products
//   nerf_bitmask       one-bit ReLU masks (pp_gemm.h gemm_epilogue): the forward epilogue packs a lane's 16 rows of a column into
//                      a 16-bit word, the data-gradient epilogue reads that word instead of 16 floats of the forward activation
//                      (neutral for the exact-fp32 path, -11..-17 % with the split-precision path)
#define NERF_GEMM_WGS pp_opt(PP_OPT_NERF_GEMM_WGS)
static const int NERF_GEMM_WGS_WIDE = 512;   // 128 x 256 tiles: 2 resident per CU (55 KB LDS, ~220 registers)
#define NERF_TN_CH pp_opt(PP_OPT_NERF_TN_CH)
#define NERF_TN_SPLIT_WGS pp_opt(PP_OPT_NERF_TN_SPLIT_WGS)
#define NERF_TN_WGS pp_opt(PP_OPT_NERF_TN_WGS)
static int nerf_wide_tiles() { return pp_opt(PP_OPT_NERF_BN) == 256; }
#define NERF_SPLIT (pp_opt(PP_OPT_NERF_SPLIT) == 1)
#define NERF_SPLIT_TN (pp_opt(PP_OPT_NERF_SPLIT_TN) == 1)
#define NERF_TN256 (pp_opt(PP_OPT_NERF_TN256) == 1)
#define NERF_BITMASK (pp_opt(PP_OPT_NERF_BITMASK) == 1)
//   nerf_planes        256-wide layers on the second-generation kernel (pp_gemm_planes.h: weights pre-split into LDS images once
//                      per pass, 128 x 256 tile on eight wavefronts); needs nerf_split and nerf_bitmask; 0 = first generation
#define NERF_PLANES (pp_opt(PP_OPT_NERF_PLANES) == 1 && NERF_SPLIT && NERF_BITMASK)
//   nerf_chain         forward pass: the eight feature layers and the density head as ONE kernel that keeps a 128-sample tile in LDS
//                      across the layers (pp_nerf_trunk.h); needs nerf_planes; 0 = one GEMM per layer
//                      bit 2 (value 3): the data-gradient chain of the backward pass likewise; the ReLU masks then travel in the
//                      fused kernels' own layout, so both passes of a step must see the same value
#define NERF_CHAIN ((pp_opt(PP_OPT_NERF_CHAIN) & 1) && NERF_PLANES)
#define NERF_CHAIN_BWD (pp_opt(PP_OPT_NERF_CHAIN) == 3 && NERF_PLANES)
//   nerf_chain_nw      wavefronts per work-group of the fused chains: 8 = one work-group on a 128-sample tile per CU, 4 = two work-groups on
//                      64-sample tiles per CU (one's epilogue beside the other's matrix instructions; twice the weight traffic from L2)
#define NERF_CHAIN_NW pp_opt(PP_OPT_NERF_CHAIN_NW)
//   nerf_chain_head    1: the colour head's 288 -> 128 layer rides as a ninth stage of the fused forward chain
#define NERF_CHAIN_HEAD (pp_opt(PP_OPT_NERF_CHAIN_HEAD) == 1)

template <int EPI>
static void nerf_gemm(hipStream_t st, const float* A, int lda, const float* W, int ldw, int K, int Nout, const float* bias,
                      const float* mask, int ldm, float* C, int ldc, const int32_t* count, int rows,
                      const float* a_max = nullptr, const float* w_max = nullptr, float* c_max = nullptr,
                      uint16_t* bits = nullptr, const _Float16* wimg = nullptr) {
  const int tiles = pp_div_up(rows, NERF_BM);
  dim3 b(256);
  if (!NERF_BITMASK) bits = nullptr;
  if (wimg && bits && EPI != EPI_PLAIN && Nout == 256 && (K & 31) == 0 && a_max && w_max && NERF_PLANES) {
    const int cus = pp_num_cus();
    constexpr int E = (EPI == EPI_PLAIN) ? EPI_MASK : EPI;
    hipLaunchKernelGGL((k_gemm256p<E>), dim3(tiles < cus ? tiles : cus), dim3(512), 0, st, A, lda, wimg, K, bias, C, ldc, count, rows,
                       a_max, w_max, c_max, bits);
    return;
  }
  if (NERF_SPLIT && a_max && w_max) {
    if (Nout <= 64) {
      dim3 g(tiles < 2 * NERF_GEMM_WGS ? tiles : 2 * NERF_GEMM_WGS, 1);
      hipLaunchKernelGGL((k_gemm128s<EPI, 64>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C, ldc, count, rows, a_max,
                         w_max, c_max);
    } else {
      dim3 g(tiles < NERF_GEMM_WGS ? tiles : NERF_GEMM_WGS, pp_div_up(Nout, 128));
      hipLaunchKernelGGL((k_gemm128s<EPI, 128>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C, ldc, count, rows, a_max,
                         w_max, c_max, bits);
    }
    return;
  }
  if (Nout == 256 && nerf_wide_tiles()) {    // 128 x 256 tile: the activation tile is read once, half the barriers per MFMA
    dim3 g(tiles < NERF_GEMM_WGS_WIDE ? tiles : NERF_GEMM_WGS_WIDE, 1);
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI, 1, NERF_BM, 256>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C,
                       ldc, count, 1, rows);
    return;
  }
  if (Nout <= 64) {                          // encoding / view-direction gradients: 64-column tile instead of a half-empty one
    dim3 g(tiles < 2 * NERF_GEMM_WGS ? tiles : 2 * NERF_GEMM_WGS, 1);
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI, 1, NERF_BM, 64>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C, ldc,
                       count, 1, rows);
    return;
  }
  dim3 g(tiles < NERF_GEMM_WGS ? tiles : NERF_GEMM_WGS, pp_div_up(Nout, 128));
  if (bits && EPI != EPI_PLAIN)
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI, 1, NERF_BM, 128, true>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C,
                       ldc, count, 1, rows, bits);
  else
    hipLaunchKernelGGL((k_gemm128<MODE_NT, EPI, 1, NERF_BM>), g, b, 0, st, A, lda, W, ldw, K, Nout, bias, mask, ldm, C, ldc,
                       count, 1, rows);
}

static void nerf_gemm_tn(hipStream_t st, const float* Y, int ldy, int N, const float* X, int ldx, int Kx, float* Wbar,
                         float* bbar, const int32_t* count, int rows, const float* y_max = nullptr, const float* x_max = nullptr) {
  if (NERF_SPLIT && NERF_SPLIT_TN && NERF_TN256 && y_max && x_max && N == 256 && Kx >= 256) {
    // all 256 x 256 outputs of a row range in one work-group (pp_gemm_tn256.h); the 64 skip columns of layer 4 go the old way
    const int tiles = pp_div_up(rows, TN256_ROWS);
    const int grid = tiles < pp_num_cus() ? tiles : pp_num_cus();
    hipLaunchKernelGGL(k_gemm_tn256, dim3(grid), dim3(512), 0, st, Y, ldy, X, ldx, Wbar, ldx, bbar, count, rows, y_max, x_max);
    if (Kx > 256) nerf_gemm_tn(st, Y, ldy, N, X + 256, ldx, Kx - 256, Wbar + 256, nullptr, count, rows, y_max, x_max);
    return;
  }
  const int blocks = (N / 128) * pp_div_up(Kx, 128);
  dim3 b(256);
  if (NERF_SPLIT && NERF_SPLIT_TN && y_max && x_max) {     // two work-groups per CU (option nerf_tn_split_wgs = 128) since the operand conversion
                                                           // is three instructions per pair: 2.90 vs 2.99 ms per scene step (round 1, with the
                                                           // compiler's conversion: one per CU was best, 3.69 vs 3.85 ms)
    dim3 gs(NERF_TN_SPLIT_WGS * 4 / blocks, blocks);
    if (pp_opt(PP_OPT_NERF_TN_TR) == 1) hipLaunchKernelGGL(k_gemm_tn_tr, gs, b, 0, st, Y, ldy, X, ldx, Kx, Wbar, ldx, bbar, count, rows, y_max, x_max);
    else hipLaunchKernelGGL(k_gemm_tn_split, gs, b, 0, st, Y, ldy, X, ldx, Kx, Wbar, ldx, bbar, count, rows, y_max, x_max);
    return;
  }
  dim3 g(NERF_TN_WGS * 4 / blocks, blocks);               // ~ 4 x NERF_TN_WGS work-groups whatever the block count (2, 3, 4 or 6)
  if (NERF_TN_CH == 64)
    hipLaunchKernelGGL((k_gemm_tn<1, 64>), g, b, 0, st, Y, ldy, X, ldx, Kx, Wbar, ldx, bbar, count, 1, rows);
  else
    hipLaunchKernelGGL((k_gemm_tn<1>), g, b, 0, st, Y, ldy, X, ldx, Kx, Wbar, ldx, bbar, count, 1, rows);
}

extern "C" int pp_nerf_fwd(const float* params, const float* center, const float* ray, const float* depth,
                           const float* bands, const int32_t* count, int32_t n_rays, int32_t n_samples, float* acts,
                           float* rgb_samples, float* density_samples, void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && center && ray && depth && bands && count && acts && rgb_samples && density_samples, "null pointer");
  PP_REQUIRE(n_rays > 0 && n_samples > 0 && (int64_t)n_rays * n_samples < (1LL << 30), "bad sizes");
  hipStream_t st = pp_stream(stream);
  const int M = n_rays * n_samples;
  const NerfLayout L = nerf_layout();
  NerfActs A = nerf_acts(acts, M);
  float* mx = NERF_SPLIT ? A.mx : nullptr;
  if (mx) {
    hipMemsetAsync(mx, 0, MX_SLOTS * sizeof(float), st);
    NerfWmaxJobs J;
    for (int l = 0; l < 7; ++l) { J.src[l] = params + L.w[l]; J.n[l] = 256 * NERF_IN_LD[l]; }
    J.src[7] = params + L.wd; J.n[7] = 257 * 256;
    J.src[8] = params + L.r0; J.n[8] = 128 * 288;
    hipLaunchKernelGGL(k_nerf_wmax, dim3(16, 9), dim3(256), 0, st, J, mx);
    hipLaunchKernelGGL(k_nerf_enc_bound, dim3(n_rays < 256 ? pp_div_up(n_rays, 4) : 64), dim3(256), 0, st, center, ray, depth, n_rays, n_samples, mx);
    if (NERF_CHAIN) {
      TrunkPackJobs P;
      for (int l = 0; l < 8; ++l) { P.src[l] = params + L.w[l]; P.ld[l] = NERF_IN_LD[l]; P.mx_w[l] = MX_W0 + l; }
      P.src[8] = params + L.r0; P.ld[8] = 288; P.mx_w[8] = MX_R0;        // the colour head's hidden layer rides as a ninth stage
      P.nsteps = NERF_CHAIN_HEAD ? TR_STEPS + 10 : TR_STEPS; P.nw = NERF_CHAIN_NW == 4 ? 4 : 8;
      hipLaunchKernelGGL(k_pack_trunk<false>, dim3(P.nsteps * 4), dim3(256), 0, st, P, mx, reinterpret_cast<unsigned char*>(A.wimg[0]));
    } else if (NERF_PLANES) {
      PlanePackJobs P;
      P.n = 8;
      for (int l = 0; l < 8; ++l) {
        P.src[l] = params + L.w[l]; P.dst[l] = A.wimg[l]; P.ld[l] = NERF_IN_LD[l]; P.K[l] = NERF_IN_LD[l]; P.mx_slot[l] = MX_W0 + l;
      }
      hipLaunchKernelGGL(k_pack_planes, dim3(16, 8), dim3(256), 0, st, P, mx);
    }
  }
  hipLaunchKernelGGL(k_nerf_encode, dim3(pp_div_up(M, 16)), dim3(256), 0, st, center, ray, depth, bands, M,
                     n_samples, A.enc, A.a[3], A.a[7]);
  if (mx && NERF_CHAIN) {
    TrunkArgs T;
    memset(&T, 0, sizeof(T));
    T.in = A.enc; T.in_ld = 64;
    for (int l = 0; l < 8; ++l) {
      T.out[l] = A.a[l]; T.ld[l] = NERF_OUT_LD[l]; T.bias[l] = params + L.b[l]; T.bits[l] = reinterpret_cast<uint32_t*>(A.bits[l]);
      T.bitsr[l] = A.bits[l]; T.mx_w[l] = MX_W0 + l; T.mx_out[l] = MX_A0 + l;
    }
    T.wstream = reinterpret_cast<const unsigned char*>(A.wimg[0]);
    T.wd = params + L.wd; T.bd = params + L.bd; T.raw = A.raw; T.density = density_samples;
    T.mx = mx; T.mx_in = MX_ENC;
    T.head = NERF_CHAIN_HEAD; T.out[8] = A.h; T.ld[8] = 128; T.bias[8] = params + L.br0; T.mx_w[8] = MX_R0; T.in2 = A.a[7] + 256; T.in2_ld = 288;
    const int cus = pp_num_cus();
    if (NERF_CHAIN_NW == 4) {                       // 64-row tiles, two work-groups per CU
      const int tiles = pp_div_up(M, 64), grid = tiles < 2 * cus ? tiles : 2 * cus;
      if (NERF_CHAIN_BWD) hipLaunchKernelGGL((k_nerf_trunk<false, true, 4>), dim3(grid), dim3(256), 0, st, T, count, M);
      else hipLaunchKernelGGL((k_nerf_trunk<false, false, 4>), dim3(grid), dim3(256), 0, st, T, count, M);
    } else {
      const int tiles = pp_div_up(M, 128), grid = tiles < cus ? tiles : cus;
      if (NERF_CHAIN_BWD) hipLaunchKernelGGL((k_nerf_trunk<false, true, 8>), dim3(grid), dim3(512), 0, st, T, count, M);
      else hipLaunchKernelGGL((k_nerf_trunk<false, false, 8>), dim3(grid), dim3(512), 0, st, T, count, M);
    }
  } else {
    const float* in = A.enc;
    for (int l = 0; l < 8; ++l) {
      nerf_gemm<EPI_RELU>(st, in, NERF_IN_LD[l], params + L.w[l], NERF_IN_LD[l], NERF_IN_LD[l], 256, params + L.b[l], nullptr, 0,
                          A.a[l], NERF_OUT_LD[l], count, M, mx ? mx + (l == 0 ? MX_ENC : MX_A0 + l - 1) : nullptr,
                          mx ? mx + MX_W0 + l : nullptr, mx ? mx + MX_A0 + l : nullptr, A.bits[l], A.wimg[l]);
      in = A.a[l];
    }
    hipLaunchKernelGGL(k_nerf_density_fwd, dim3(pp_div_up(M, 4)), dim3(256), 0, st, A.a[6], params + L.wd, params + L.bd, M,
                       A.raw, density_samples);
  }
  if (!(mx && NERF_CHAIN && NERF_CHAIN_HEAD))         // (else the fused chain has produced the head's hidden layer as its ninth stage)
    nerf_gemm<EPI_RELU>(st, A.a[7], 288, params + L.r0, 288, 288, 128, params + L.br0, nullptr, 0, A.h, 128, count, M,
                        mx ? mx + MX_A0 + 7 : nullptr, mx ? mx + MX_R0 : nullptr, nullptr);
  hipLaunchKernelGGL(k_nerf_rgb_fwd, dim3(pp_div_up(M * 16, 256)), dim3(256), 0, st, params + L.r1, params + L.br1, A.h, M,
                     rgb_samples);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_nerf_bwd(const float* params, const float* ray, const float* depth, const int32_t* count,
                           int32_t n_rays, int32_t n_samples, const float* acts, const float* rgb_samples, const float* g_rgb_samples,
                           const float* g_density_samples, float* scratch, float* params_grad, float* g_center, float* g_ray,
                           void* ctx, void* stream) {
  PPOptScope scope(ctx);
  PP_REQUIRE(params && ray && depth && count && acts && rgb_samples && g_rgb_samples && g_density_samples && scratch &&
                 params_grad && g_center && g_ray, "null pointer");
  PP_REQUIRE(n_rays > 0 && n_samples > 0 && (int64_t)n_rays * n_samples < (1LL << 30), "bad sizes");
  hipStream_t st = pp_stream(stream);
  const int R = n_rays, S = n_samples, M = R * S;
  const NerfLayout L = nerf_layout();
  NerfActs A = nerf_acts(const_cast<float*>(acts), M);
  float* P = scratch;
  float* Q = P + (size_t)M * 320;
  float* dEnc0 = Q + (size_t)M * 320;
  float* dEncS = dEnc0 + (size_t)M * 64;
  float* dHsum = dEncS + (size_t)M * 64;
  float* dView = dHsum + (size_t)R * 128;
  float* wt = dView + (size_t)R * 32;
  float* WT[8];
  {
    float* p = wt;
    for (int l = 0; l < 8; ++l) { WT[l] = p; p += (l == 7) ? 256 * 288 : 256 * NERF_IN_LD[l]; }
  }
  float* R0T = WT[7] + 256 * 288;                  // [288][128]
  _Float16* r0t_img = reinterpret_cast<_Float16*>(wt + NERF_WT_FLOATS);
  float* part = wt + NERF_WT_FLOATS + NERF_WTIMG_FLOATS;      // [NERF_PART_WGS][NERF_PART_LD]
  _Float16* wt7_img = r0t_img + 2 * 256 * 128;
  _Float16* wt_img[7];
  for (int l = 1; l <= 6; ++l) wt_img[l] = wt7_img + 2 * 256 * 288 + (size_t)(l - 1) * 2 * 256 * 256;
  dim3 b(256);
  // transposed weights for the data-gradient GEMMs (2 MB, L2 resident)
  {
    NerfTransposeJobs J;
    for (int l = 0; l < 8; ++l) {
      J.src[l] = params + L.w[l]; J.dst[l] = WT[l]; J.lds[l] = NERF_IN_LD[l]; J.rows[l] = 256; J.cols[l] = NERF_IN_LD[l];
      J.ldd[l] = (l == 7) ? 288 : 256;
    }
    J.src[8] = params + L.r0; J.dst[8] = R0T; J.lds[8] = 288; J.rows[8] = 128; J.cols[8] = 288; J.ldd[8] = 128;
    hipLaunchKernelGGL(k_nerf_transpose_all, dim3(64, 9), b, 0, st, J);
  }
  hipLaunchKernelGGL(k_nerf_wd_column, dim3(1), b, 0, st, params + L.wd, WT[7]);

  // split-precision path: operand maxima of the gradient tensors are recorded by their producers (slots MX_DH .. MX_DHSUM)
  float* mx = NERF_SPLIT ? A.mx : nullptr;
  if (mx) hipMemsetAsync(mx + MX_DH, 0, (MX_DHSUM - MX_DH + 1) * sizeof(float), st);
  const bool planes = mx && NERF_PLANES;
  const bool chain = mx && NERF_CHAIN_BWD;
  float* DY[7];                                    // d(pre-activation of layer l), l = 0 .. 6, for the fused chain
  for (int l = 0; l < 7; ++l) DY[l] = part + NERF_PART_FLOATS + (size_t)l * M * 256;
  if (chain) {
    // the chain's weights in its order of use, read transposed straight from the parameters: R0 (rows = hidden units), W7 .. W1
    TrunkPackJobs J;
    J.src[0] = params + L.r0; J.ld[0] = 288; J.mx_w[0] = MX_R0;
    for (int s_ = 1; s_ < 8; ++s_) { J.src[s_] = params + L.w[8 - s_]; J.ld[s_] = NERF_IN_LD[8 - s_]; J.mx_w[s_] = MX_W0 + 8 - s_; }
    J.nsteps = TR_STEPS; J.nw = 0;
    hipLaunchKernelGGL(k_pack_trunk<true>, dim3(TR_STEPS * 4), dim3(256), 0, st, J, mx, reinterpret_cast<unsigned char*>(r0t_img));
  } else if (planes) {
    PlanePackJobs P;
    P.n = 8;
    P.src[0] = R0T; P.dst[0] = r0t_img; P.ld[0] = 128; P.K[0] = 128; P.mx_slot[0] = MX_R0;
    P.src[7] = WT[7]; P.dst[7] = wt7_img; P.ld[7] = 288; P.K[7] = 288; P.mx_slot[7] = MX_W0 + 7;
    for (int l = 1; l <= 6; ++l) { P.src[l] = WT[l]; P.dst[l] = wt_img[l]; P.ld[l] = 256; P.K[l] = 256; P.mx_slot[l] = MX_W0 + l; }
    hipLaunchKernelGGL(k_pack_planes, dim3(16, 8), dim3(256), 0, st, P, mx);
  }
  auto slot = [&](int i) -> float* { return mx ? mx + i : nullptr; };

  // colour head
  float* dH = Q;                                   // [M][128]
  {
    const int wgs = min(pp_div_up(M, NERF_STRIP), NERF_PART_WGS);
    hipLaunchKernelGGL(k_nerf_rgb_bwd, dim3(wgs), dim3(512), 0, st, params + L.r1, A.h, rgb_samples, g_rgb_samples, M, dH, part);
    hipLaunchKernelGGL(k_nerf_part_finish, dim3(pp_div_up(388, 64)), dim3(1024), 0, st, part, wgs, params_grad + L.r1, 384, params_grad + L.br1, 3,
                       slot(MX_DH));
  }
  nerf_gemm_tn(st, dH, 128, 128, A.a[7], 288, 288, params_grad + L.r0, params_grad + L.br0, count, M, slot(MX_DH), slot(MX_A0 + 7));
  hipLaunchKernelGGL(k_nerf_ray_sum, dim3(R < 512 ? R : 512), dim3(512), 0, st, dH, R, S, dHsum, slot(MX_DHSUM));
  nerf_gemm<EPI_PLAIN>(st, dHsum, 128, R0T + 256 * 128, 128, 128, 32, nullptr, nullptr, 0, dView, 32, count, R, slot(MX_DHSUM),
                       slot(MX_R0), nullptr);
  if (chain) {
    // d raw (column 256 of P) and the density head's own gradients first: the chain reads that column
    {
      const int wgs = min(pp_div_up(M, NERF_DSTRIP), NERF_PART_WGS);
      hipLaunchKernelGGL(k_nerf_density_bwd, dim3(wgs), dim3(512), 0, st, A.a[6], A.raw, g_density_samples, M, P, part);
      hipLaunchKernelGGL(k_nerf_part_finish, dim3(pp_div_up(258, 64)), dim3(1024), 0, st, part, wgs, params_grad + L.wd, 256, params_grad + L.bd, 1,
                         slot(MX_P));
    }
    // d(layer 7) .. d(layer 0) in one kernel (pp_nerf_trunk.h), each written once for the weight-gradient products below
    TrunkArgs T;
    memset(&T, 0, sizeof(T));
    T.in = dH; T.in_ld = 128;
    T.out[0] = P; T.ld[0] = 288; T.mx_w[0] = MX_R0; T.mx_out[0] = MX_P; T.bitsr[0] = A.bits[7];
    for (int s_ = 1; s_ < 8; ++s_) {
      T.out[s_] = DY[7 - s_]; T.ld[s_] = 256; T.mx_w[s_] = MX_W0 + 8 - s_; T.mx_out[s_] = MX_DY6 + s_ - 1; T.bitsr[s_] = A.bits[7 - s_];
    }
    T.wstream = reinterpret_cast<const unsigned char*>(r0t_img);
    T.wd = params + L.wd; T.draw = P + 256; T.draw_ld = 288;
    T.mx = mx; T.mx_in = MX_DH;
    const int cus = pp_num_cus();
    if (NERF_CHAIN_NW == 4) {
      const int tiles = pp_div_up(M, 64);
      hipLaunchKernelGGL((k_nerf_trunk<true, true, 4>), dim3(tiles < 2 * cus ? tiles : 2 * cus), dim3(256), 0, st, T, count, M);
    } else {
      const int tiles = pp_div_up(M, 128);
      hipLaunchKernelGGL((k_nerf_trunk<true, true, 8>), dim3(tiles < cus ? tiles : cus), dim3(512), 0, st, T, count, M);
    }
    nerf_gemm_tn(st, P, 288, 256, A.a[6], 256, 256, params_grad + L.w[7], params_grad + L.b[7], count, M, slot(MX_P), slot(MX_A0 + 6));
    for (int l = 6; l >= 1; --l)
      nerf_gemm_tn(st, DY[l], 256, 256, A.a[l - 1], NERF_OUT_LD[l - 1], NERF_IN_LD[l], params_grad + L.w[l], params_grad + L.b[l], count, M,
                   slot(MX_DY6 + 6 - l), slot(MX_A0 + l - 1));
    nerf_gemm<EPI_PLAIN>(st, DY[4], 256, WT[4] + 256 * 256, 256, 256, 64, nullptr, nullptr, 0, dEncS, 64, count, M,
                         slot(MX_DY6 + 2), slot(MX_W0 + 4), nullptr);
    nerf_gemm_tn(st, DY[0], 256, 256, A.enc, 64, 64, params_grad + L.w[0], params_grad + L.b[0], count, M, slot(MX_DY6 + 6),
                 slot(MX_ENC));
    nerf_gemm<EPI_PLAIN>(st, DY[0], 256, WT[0], 256, 256, 64, nullptr, nullptr, 0, dEnc0, 64, count, M, slot(MX_DY6 + 6),
                         slot(MX_W0), nullptr);
  } else {
    // last feature layer: columns 0..255 through the colour head, column 256 from the density
    nerf_gemm<EPI_MASK>(st, dH, 128, R0T, 128, 128, 256, nullptr, A.a[7], 288, P, 288, count, M, slot(MX_DH), slot(MX_R0),
                        slot(MX_P), A.bits[7], planes ? r0t_img : nullptr);
    {
      const int wgs = min(pp_div_up(M, NERF_DSTRIP), NERF_PART_WGS);
      hipLaunchKernelGGL(k_nerf_density_bwd, dim3(wgs), dim3(512), 0, st, A.a[6], A.raw, g_density_samples, M, P, part);
      hipLaunchKernelGGL(k_nerf_part_finish, dim3(pp_div_up(258, 64)), dim3(1024), 0, st, part, wgs, params_grad + L.wd, 256, params_grad + L.bd, 1,
                         slot(MX_P));
    }
    nerf_gemm_tn(st, P, 288, 256, A.a[6], 256, 256, params_grad + L.w[7], params_grad + L.b[7], count, M, slot(MX_P), slot(MX_A0 + 6));
    nerf_gemm<EPI_MASK>(st, P, 288, WT[7], 288, 288, 256, nullptr, A.a[6], 256, Q, 256, count, M, slot(MX_P), slot(MX_W0 + 7),
                        slot(MX_DY6), A.bits[6], planes ? wt7_img : nullptr);
    float* cur = Q;
    float* nxt = P;
    for (int l = 6; l >= 1; --l) {                   // cur = d(pre-activation of layer l), [M][256]
      const float* x = A.a[l - 1];
      const int ldx = NERF_OUT_LD[l - 1];            // 320 for layer 4's input (features + skip columns)
      nerf_gemm_tn(st, cur, 256, 256, x, ldx, NERF_IN_LD[l], params_grad + L.w[l], params_grad + L.b[l], count, M,
                   slot(MX_DY6 + 6 - l), slot(MX_A0 + l - 1));
      nerf_gemm<EPI_MASK>(st, cur, 256, WT[l], 256, 256, 256, nullptr, x, ldx, nxt, 256, count, M, slot(MX_DY6 + 6 - l),
                          slot(MX_W0 + l), slot(MX_DY6 + 7 - l), A.bits[l - 1], planes ? wt_img[l] : nullptr);
      if (l == 4)                                    // skip columns: gradient of the encoding, no activation in between
        nerf_gemm<EPI_PLAIN>(st, cur, 256, WT[4] + 256 * 256, 256, 256, 64, nullptr, nullptr, 0, dEncS, 64, count, M,
                             slot(MX_DY6 + 2), slot(MX_W0 + 4), nullptr);
      float* t = cur; cur = nxt; nxt = t;
    }
    nerf_gemm_tn(st, cur, 256, 256, A.enc, 64, 64, params_grad + L.w[0], params_grad + L.b[0], count, M, slot(MX_DY6 + 6),
                 slot(MX_ENC));
    nerf_gemm<EPI_PLAIN>(st, cur, 256, WT[0], 256, 256, 64, nullptr, nullptr, 0, dEnc0, 64, count, M, slot(MX_DY6 + 6),
                         slot(MX_W0), nullptr);
  }
  hipLaunchKernelGGL(k_nerf_encode_bwd, dim3(R), b, 0, st, A.enc, dEnc0, dEncS, dView, A.a[7], ray, depth, R, S, g_center,
                     g_ray);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// ------------------------------------------------------------------------------------------------ schedule + photometric loss
// The pieces of a scene-branch step that are a dozen one-line elementwise launches in torch (4.6 us of stream time each).

// BARF coarse-to-fine weights of the point bands then the view bands (frequency_nerf.py:250-253), same op order:
// alpha = (progress - start) / (end - start) * L ; w_k = (1 - cos(clamp(alpha - k, 0, 1) * pi)) / 2 ; `width` = end - start formed
// by the caller in double and rounded once, as torch rounds the Python scalar
__global__ void k_nerf_band_weights(const float* __restrict__ progress, float start, float width, int L3, int LV,
                                    float* __restrict__ bands) {
  const int i = threadIdx.x;
  if (i >= L3 + LV) return;
  const int L = i < L3 ? L3 : LV, k = i < L3 ? i : i - L3;
  const float alpha = (progress[0] - start) / width * (float)L;
  const float c = fminf(fmaxf(alpha - (float)k, 0.f), 1.f) * NERF_PI;
  bands[i] = (1.f - cosf(c)) / 2.f;
}

extern "C" int pp_nerf_band_weights(const float* progress, float start, float width, int32_t l_3d, int32_t l_view, float* bands,
                                    void* stream) {
  PP_REQUIRE(progress && bands, "null pointer");
  PP_REQUIRE(l_3d >= 0 && l_view >= 0 && l_3d + l_view > 0 && l_3d + l_view <= 64 && width != 0.f, "bad arguments");
  hipLaunchKernelGGL(k_nerf_band_weights, dim3(1), dim3(64), 0, pp_stream(stream), progress, start, width, l_3d, l_view, bands);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

// weight * huber_loss(pred, label, delta, reduction = mean) and its gradient w.r.t. pred (base_losses.py:155-156: delta = 0.5,
// weight = 2).  One work-group, fixed summation order (strided partial sums, then a tree over the 1024 partials).
__global__ __launch_bounds__(1024) void k_nerf_huber(const float* __restrict__ pred, const float* __restrict__ label, int n,
                                                     float delta, float weight, float* __restrict__ loss,
                                                     float* __restrict__ g_pred) {
  __shared__ float red[1024];
  float acc = 0.f;
  const float gs = weight / (float)n;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float d = pred[i] - label[i], ad = fabsf(d);
    acc += ad <= delta ? 0.5f * d * d : delta * (ad - 0.5f * delta);
    g_pred[i] = gs * fminf(fmaxf(d, -delta), delta);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = weight * (red[0] / (float)n);
}

extern "C" int pp_nerf_huber_loss(const float* pred, const float* label, int32_t n, float delta, float weight, float* loss,
                                  float* g_pred, void* stream) {
  PP_REQUIRE(pred && label && loss && g_pred, "null pointer");
  PP_REQUIRE(n > 0 && delta > 0.f, "bad arguments");
  hipLaunchKernelGGL(k_nerf_huber, dim3(1), dim3(1024), 0, pp_stream(stream), pred, label, n, delta, weight, loss, g_pred);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_nerf_composite_fwd(const float* rgb_samples, const float* density_samples, const float* depth,
                                     const float* ray, int32_t n_rays, int32_t n_samples, int32_t white_bg, float* rgb,
                                     float* depth_out, float* opacity, float* weights, float* all_cumulated, float* rgb_var,
                                     float* depth_var, void* stream) {
  PP_REQUIRE(rgb_samples && density_samples && depth && ray && rgb && depth_out && opacity && weights && all_cumulated &&
                 rgb_var && depth_var, "null pointer");
  PP_REQUIRE(n_rays > 0 && n_samples > 0 && n_samples <= 512, "bad sizes");
  hipLaunchKernelGGL(k_nerf_composite_fwd, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), rgb_samples,
                     density_samples, depth, ray, n_rays, n_samples, white_bg, rgb, depth_out, opacity, weights, all_cumulated,
                     rgb_var, depth_var);
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_nerf_composite_bwd(const float* rgb_samples, const float* density_samples, const float* depth,
                                     const float* ray, const float* weights, int32_t n_rays, int32_t n_samples,
                                     int32_t white_bg, const float* g_rgb, const float* g_depth, const float* g_opacity,
                                     const float* g_weights, float* g_rgb_samples, float* g_density_samples, float* g_ray,
                                     void* stream) {
  PP_REQUIRE(rgb_samples && density_samples && depth && ray && weights && g_rgb && g_depth && g_opacity && g_rgb_samples &&
                 g_density_samples && g_ray, "null pointer");
  PP_REQUIRE(n_rays > 0 && n_samples > 0 && n_samples <= 512, "bad sizes");
  hipLaunchKernelGGL(k_nerf_composite_bwd, dim3(pp_div_up(n_rays, 4)), dim3(256), 0, pp_stream(stream), rgb_samples,
                     density_samples, depth, ray, weights, n_rays, n_samples, white_bg, g_rgb, g_depth, g_opacity, g_weights,
                     g_rgb_samples, g_density_samples, g_ray);
  PP_CHECK_LAUNCH();
  return PP_OK;
}
