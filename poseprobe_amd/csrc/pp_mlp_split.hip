// Layer-fused object-branch MLP kernels, split-precision variant (option "mlp_split"): the structure of pp_mlp_fused.hip
// - one persistent work-group per CU walks a 64-row tile through all layers, hidden-layer weights stationary in registers,
// activations in LDS between the layers - with every 128 x 128 product evaluated as THREE fp16 products
// lo.hi + hi.lo + hi.hi on v_mfma_f32_32x32x16_f16 (fp32 accumulation) instead of v_mfma_f32_32x32x2_f32:
// 3 x 32 matrix-pipe cycles per 32 x 32 x 16 block instead of 8 x 64.  Operands and results in memory stay fp32; the error
// against an fp64 product equals the fp32 instructions' (pp_gemm_split.h, tools/split_gemm_probe.hip).
//
// Scales.  x * s = hi + lo needs a power of two s with max|x| * s < 2^16.  Weights: s from max|W| of the layer, found once
// per work-group in the prologue.  Activations live only inside the kernel, one tile at a time, so their scale is PER TILE AND
// LAYER and must be known when the producing epilogue converts - before the tile's maximum exists.  It comes from a bound
// instead: |y[r][n]| <= max|x| * max_n sum_k |W[n][k]| + max|b|, with max|x| the EXACT maximum of the previous layer's tile
// (every lane folds what it writes into an LDS slot by ds_max_u32 on the float bits; the barrier that publishes the tile
// publishes the slot).  The bound is loose by 2^3..2^6, which costs nothing: hi / lo carry 22 significant bits down to
// 2^-18 of the largest representable value and an absolute floor of 2^-40 of it below that.
//
// LDS image of a tile: hi plane [64][136] halfs, lo plane [64][136] halfs (272-byte rows: the 16 rows of a ds_read_b128
// lane group start 4 banks apart and cover all 64 banks once).
//
// Orientation.  The fp32 kernels compute D = X . W^T with the activations as the MFMA A operand, so a lane ends up with 16
// ROWS of one feature: one scalar LDS / HBM store per element, ~17 instructions per element in the epilogue - with the matrix
// time cut to a fifth that is what the kernel then waits for.  Here the roles are swapped: A = weights (feature l31 of the
// wavefront's 32, k = 16 ks + 8 lh + j; register content as before), B = activations (row l31), D[reg][lane] = out[row l31]
// [feature 32 w + (reg & 3) + 8 (reg >> 2) + 4 lh]: a lane holds four runs of four CONSECUTIVE features of one row, so the
// epilogue converts pairs (v_cvt_pk_f16_f32, v_fma_mixlo / mixhi_f16) and stores 8 bytes per LDS write and 16 bytes per HBM
// write (NOT v_pk_*_f32 arithmetic: those cost 16 cycles beside MFMAs, tools/mfma_valu_probe.hip - the file is compiled with
// -fno-slp-vectorize).  The four rows of a warp sample are the four lanes of a quad; the primal row's ReLU state reaches the tangent rows
// by a quad-broadcast DPP operand.  The input layer (3 -> 128 on [p, 1] / unit tangents) is one more MFMA with K padded to 16.
//
// Schedule.  With the matrix time cut to a fifth the epilogues (scale, gate, split, store: ~8 instructions per element) cost
// more than the MFMAs; a lone wavefront per SIMD can only hide them if they are ISSUED between MFMAs.  A tile is therefore
// processed as two 32-row halves, each with its own accumulators, scale and maximum slot, in a software pipeline
//   A_l: MFMAs of (layer l, half 0)  interleaved with the epilogue of (layer l-1, half 1)
//   B_l: MFMAs of (layer l, half 1)  interleaved with the epilogue of (layer l,   half 0)
// with one LDS-only barrier after each stage (the rows a stage's MFMAs read were published by the barrier before it).
#include "pp_common.h"
#include "pp_mlp_fused.h"
#include "pp_gemm_split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDA 132                 // fp32 view of a tile (output layer input), as in pp_mlp_fused.hip
#define TILE_ROWS 64
#define LDH2 136                // halfs per row of a split tile
#define PLANE (TILE_ROWS * LDH2)
#ifndef MS_DBG
#define MS_DBG 0      // experiments only (forward kernels): 1 = no MFMAs, 2 = no HBM activation stores, 3 = no LDS image writes
#endif
#ifdef MS_TIMERS      // phase timers (experiments): wave 0 of every work-group sums s_memtime deltas per phase
__device__ unsigned long long g_ms_t[16];
extern "C" int pp_debug_read_timers(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ms_t), sizeof(g_ms_t)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ms_t), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#define TICK(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tsum[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define TICK(i) do {} while (0)
#endif
#define PP_WAIT_VMEM() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)

namespace {

struct SplitW { pp_half8 h[8], l[8]; };      // one layer's share of a lane: 64 registers, as the fp32 layout

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide maximum of a non-negative value (prologues): DPP row maxima, sixteen partials through LDS, ONE barrier per call - the
// calls alternate between two slot sets (`red`: [2][16] floats, `phase` counts the calls), so the readers of one call are done
// with their set before the call after the next writes it again
__device__ __forceinline__ float block_max(float v, float* red, int tid, int& phase) {
  float* set = red + 16 * (phase & 1);
  ++phase;
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true)));
  if ((tid & 15) == 0) set[tid >> 4] = v;
  __syncthreads();
  const float4 a = *reinterpret_cast<const float4*>(set), b = *reinterpret_cast<const float4*>(set + 4),
               c = *reinterpret_cast<const float4*>(set + 8), d = *reinterpret_cast<const float4*>(set + 12);
  return fmaxf(fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))),
               fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
}
// exponent e of the power of two with mx * 2^e in [2^14, 2^15), clamped to +-60 so that products of two scales stay finite;
// scales are kept as exponents: reciprocals and products are integer arithmetic + one v_ldexp_f32
__device__ __forceinline__ int scale_exp(float mx) {
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 0;
  int e;
  frexpf(mx, &e);
  return min(max(15 - e, -60), 60);
}
__device__ __forceinline__ float pow2(int e) { return ldexpf(1.f, e); }
__device__ __forceinline__ float tile_scale(float mx) { return pow2(scale_exp(mx)); }
__device__ __forceinline__ void split1(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  l = (_Float16)(x - (float)h);
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, float s, pp_half8& h, pp_half8& l) {
  const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 hh, ll;
    split1(v[i], hh, ll);
    h[i] = hh; l[i] = ll;
  }
}
__device__ __forceinline__ float amax4(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ float asum4(const float4& v) { return (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)); }

// forward weights of feature n: k = 16 ks + 8 lh + j.  Returns the exponent of the layer's scale; l1 = max_n sum_k |W[n][k]|.
__device__ __forceinline__ int load_w_rows_split(SplitW& w, float& l1, const float* __restrict__ W, int n, int lh, float* red,
                                                   int tid, int& phase) {
  float4 v[16];
  float mx = 0.f, sum = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    v[2 * ks] = *reinterpret_cast<const float4*>(W + (size_t)n * 128 + 16 * ks + 8 * lh);
    v[2 * ks + 1] = *reinterpret_cast<const float4*>(W + (size_t)n * 128 + 16 * ks + 8 * lh + 4);
    mx = fmaxf(mx, fmaxf(amax4(v[2 * ks]), amax4(v[2 * ks + 1])));
    sum += asum4(v[2 * ks]) + asum4(v[2 * ks + 1]);
  }
  sum += __shfl_xor(sum, 32, 64);
  const int e = scale_exp(block_max(mx, red, tid, phase));
  const float s = pow2(e);
  l1 = block_max(sum, red, tid, phase) * 1.0001f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) split8(v[2 * ks], v[2 * ks + 1], s, w.h[ks], w.l[ks]);
  return e;
}

struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// acc[reg] += sum_k W[feature(reg)][k] X[row l31][k]  for 32 rows of a split tile (`rows` = hi plane of the first of them, lo
// plane PLANE halfs behind); hook(ks), ks = 0..7, is called after the three MFMAs of every operand group - work placed there is
// issued in the shadow of the matrix pipe (KS operand groups of 16, row stride LD halfs, lo plane PL halfs behind the hi plane)
template <int KS = 8, int LD = LDH2, int PL = PLANE, class Hook = NoHook>
__device__ __forceinline__ void mma_half(const _Float16* __restrict__ rows, const SplitW& w, f32x16& acc, int l31, int lh,
                                         Hook hook = Hook()) {
  const _Float16* p = rows + l31 * LD + 8 * lh;
  pp_half8 h = *reinterpret_cast<const pp_half8*>(p), l = *reinterpret_cast<const pp_half8*>(p + PL);
#pragma unroll
  for (int ks = 0; ks < (MS_DBG == 1 ? 0 : KS); ++ks) {
    pp_half8 nh = h, nl = l;
    if (ks + 1 < KS) {
      nh = *reinterpret_cast<const pp_half8*>(p + 16 * (ks + 1));
      nl = *reinterpret_cast<const pp_half8*>(p + 16 * (ks + 1) + PL);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[ks], l, acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.l[ks], h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[ks], h, acc, 0, 0, 0);
    h = nh; l = nl;
    hook(ks);
  }
}

__device__ __forceinline__ void zero16(f32x16& acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// Epilogue of one 32-row half, lane = row, registers = features fb + 8 q + c, as EIGHT batched steps so that it can be issued
// between the MFMAs of another half (step i after operand group i).  y = acc * inv + bias; COLS == 4: `bias` is zero in the
// tangent lanes and the sign of the primal lane's pre-activation gates the whole quad (y == +0 counts as open: rows that are
// exactly zero have zero tangents as well); fp32 copy to HBM for the backward pass (rows with `ok`), next tile to LDS as a split
// image scaled by snext (SPLIT) or as the fp32 view.
// Every step is a batch over all 16 elements with no instruction depending on its predecessor: a lone wavefront per SIMD has
// nothing else to issue while a VALU result matures (DPP needs two wait states after the write of its source, v_cndmask one
// after v_cmp writes VCC - hence sign-bit arithmetic instead of compares), and the scheduler is fenced between steps so that it
// cannot re-serialise them.
template <int COLS, bool SPLIT>
struct HalfEpilogue {
  float4 b[4];
  float y[16];              // y -> gated value v
  float p[16];              // primal pre-activation of the quad -> scaled value x
  unsigned h[8], l[8];
  float xmax;
  // brow: this lane's bias row in LDS + fb (the real biases in primal lanes, a row of zeros in tangent lanes when COLS == 4)
  __device__ __forceinline__ void begin(const float* __restrict__ brow) {
    xmax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const float4*>(brow + 8 * q);
  }
  __device__ __forceinline__ void step(int i, const f32x16& acc, float inv, float snext, bool ok, float* __restrict__ Crow,
                                       _Float16* __restrict__ Arow) {
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y[4 * q] = fmaf(acc[4 * q], inv, b[q].x); y[4 * q + 1] = fmaf(acc[4 * q + 1], inv, b[q].y);
        y[4 * q + 2] = fmaf(acc[4 * q + 2], inv, b[q].z); y[4 * q + 3] = fmaf(acc[4 * q + 3], inv, b[q].w);
      }
    } else if (i == 1) {
      // (inline assembly from here on: the compiler rewrites the sign-bit arithmetic into v_cmp / v_cndmask pairs through VCC)
      if (COLS == 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          asm("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(p[e]) : "v"(y[e]));
#pragma unroll
        for (int e = 0; e < 16; ++e) asm("v_ashrrev_i32 %0, 31, %1" : "=v"(p[e]) : "v"(p[e]));      // all ones where the primal row is negative
      }
    } else if (i == 2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (COLS == 4) asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(y[e]) : "v"(p[e]), "v"(y[e]));          // y & ~mask
        else y[e] = fmaxf(y[e], 0.f);
      }
    } else if (SPLIT && i == 3) {
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = y[e] * snext;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(xmax) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
      }
    } else if (SPLIT && i == 4) {
      // lo = x - hi: fp32 fma on the f16 hi half, rounded once into the packed result
#pragma unroll
      for (int k = 0; k < 8; ++k) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[k]) : "v"(h[k]), "v"(p[2 * k]));
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[k]) : "v"(h[k]), "v"(p[2 * k + 1]));
    } else if (i == 5) {
      // the LDS image two steps before the end of the stage: its writes have retired when the barrier's lgkmcnt(0) comes
      if (SPLIT && MS_DBG != 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<uint2*>(Arow + 8 * q) = make_uint2(h[2 * q], h[2 * q + 1]);
          *reinterpret_cast<uint2*>(Arow + 8 * q + PLANE) = make_uint2(l[2 * q], l[2 * q + 1]);
        }
      }
      if (!SPLIT) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Arow) + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
    } else if (i == 6) {
      if (ok && MS_DBG != 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(Crow + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void all(const f32x16& acc, float inv, float snext, bool ok, float* __restrict__ Crow, _Float16* __restrict__ Arow) {
#pragma unroll
    for (int i = 0; i < 8; ++i) step(i, acc, inv, snext, ok, Crow, Arow);
  }
  // max |v| of what this lane wrote (xmax holds it in scaled units; ninv = 1 / snext)
  __device__ __forceinline__ float vmax(float ninv) const { return xmax * ninv; }
};

// maximum over the 16 lanes of a DPP row (all 16 lanes receive it): four VALU instructions, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_max16(float v) {       // v >= 0
  v = fmaxf(v, dpp_f<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_f<0x140>(v));     // row_mirror
  return v;
}
// fold a lane's non-negative value into an LDS slot (float bits as unsigned).  Written as one ds_max_u32 from four lanes
// in inline assembly: the compiler's atomic optimiser would turn a plain atomicMax into a 64-iteration scalar readlane loop.
__device__ __forceinline__ void slot_max(unsigned* slot, float v, int lane) {
  v = row_max16(v);
  if ((lane & 15) == 0) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)slot;
    asm volatile("ds_max_u32 %0, %1" ::"v"(a), "v"(__float_as_uint(v)) : "memory");
  }
}
__device__ __forceinline__ float slot_get(const unsigned* slot) { return __uint_as_float(*slot); }

}  // namespace

// ------------------------------------------------------------------------------------------------ warp net, forward
// Same contract as k_warp_fused_fwd: pts[M][3] -> out[M][4][4], hidden activations X0..X3 ([4M][128] fp32 each) for backward.
__global__ __launch_bounds__(256) void k_warp_fused_fwd_s(const float* __restrict__ params, const float* __restrict__ pts,
                                                          const int32_t* __restrict__ count, int capacity, float out_range,
                                                          float* __restrict__ acts, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) _Float16 At[2][2 * PLANE];      // tile 1 doubles as the fp32 [64][LDA] view
  __shared__ __attribute__((aligned(16))) float W4s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float Red[4 * 64 * 4];
  __shared__ float Ps[2][52];                                              // 48 coordinates + max |.| of each 16 at [48..50]
  __shared__ unsigned Mx[2][8];                                            // maxima of (X0, X1, X2) x (half 0, half 1), two parities
  __shared__ __attribute__((aligned(16))) float red4[32];
  __shared__ __attribute__((aligned(16))) float Bs[4][128];                // hidden-layer biases, row 3 = zeros
  static_assert(2 * PLANE * 2 >= TILE_ROWS * LDA * 4, "fp32 view must fit into a split tile");
  const int M = min(count[0], capacity);
  const int R = 4 * M;
  const int ntiles = (M + 15) >> 4;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;                    // feature whose weights this lane holds (A operand)
  const int fb = wid * 32 + 4 * lh;                  // first of the features this lane holds in the accumulators
  const bool primal = (l31 & 3) == 0;
  const size_t LS = (size_t)capacity * 4 * 128;

  SplitW w1, w2, w3;
  int phase = 0;                        // call counter of block_max
  float l1_1, l1_2, l1_3;
  const int ew1 = load_w_rows_split(w1, l1_1, params + WPF_W1, col, lh, red4, tid, phase);
  const int ew2 = load_w_rows_split(w2, l1_2, params + WPF_W2, col, lh, red4, tid, phase);
  const int ew3 = load_w_rows_split(w3, l1_3, params + WPF_W3, col, lh, red4, tid, phase);
  // biases of the hidden layers in LDS: row l = the layer's biases, row 3 = zeros (tangent lanes, and the input layer whose
  // bias rides in the product); a lane reads its four runs of four from `brow(l)`
  if (tid < 128) { Bs[0][tid] = params[WPF_B1 + tid]; Bs[1][tid] = params[WPF_B2 + tid]; Bs[2][tid] = params[WPF_B3 + tid]; Bs[3][tid] = 0.f; }
  auto brow = [&](int l) -> const float* { return &Bs[primal ? l : 3][fb]; };
  const float b1mx = block_max(fabsf(params[WPF_B1 + col]), red4, tid, phase), b2mx = block_max(fabsf(params[WPF_B2 + col]), red4, tid, phase);
  // input layer as a K = 16 product: A = [w0x w0y w0z b0 0 ...] (lanes lh = 0), B = [px py pz 1 0 ...] / unit tangents
  const float w0x = params[WPF_W0 + col * 3], w0y = params[WPF_W0 + col * 3 + 1], w0z = params[WPF_W0 + col * 3 + 2];
  const float b0 = params[WPF_B0 + col];
  const float w0l1 = block_max((fabsf(w0x) + fabsf(w0y)) + fabsf(w0z), red4, tid, phase) * 1.0001f;
  const float w0mx = block_max(fmaxf(fmaxf(fabsf(w0x), fabsf(w0y)), fabsf(w0z)), red4, tid, phase);
  const float b0mx = block_max(fabsf(b0), red4, tid, phase);
  const int ew0 = scale_exp(fmaxf(w0mx, b0mx));
  pp_half8 a0h, a0l;
  {
    const float z = 0.f;
    const float4 wa = lh == 0 ? make_float4(w0x, w0y, w0z, b0) : make_float4(z, z, z, z);
    split8(wa, make_float4(z, z, z, z), pow2(ew0), a0h, a0l);
  }
  for (int i = tid; i < 512; i += 256) W4s[(i >> 7) * LDA + (i & 127)] = params[WPF_W4 + i];
  const float b4 = (((tid >> 2) & 3) == 0) ? params[WPF_B4 + (tid & 3)] : 0.f;   // bias on the primal row only
  // positions of a tile are fetched one tile ahead and parked in LDS together with their largest magnitudes (wave 0)
  auto park = [&](int slot, float p) {
    if (wid == 0) {
      const float m = row_max16(fabsf(p));
      if (lane < 48) Ps[slot][lane] = p;
      if ((lane & 15) == 0 && lane < 48) Ps[slot][48 + (lane >> 4)] = m;
    }
  };
  float pnext = 0.f;
  if (tid < 48 && (int)blockIdx.x * 48 + tid < M * 3) pnext = pts[blockIdx.x * 48 + tid];
  park(0, pnext);
  if (tid < 16) Mx[tid >> 3][tid & 7] = 0u;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
#ifdef MS_TIMERS
  unsigned long long tsum[16] = {0}, tprev = __builtin_readcyclecounter();
#endif
  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int s0 = tile * 16, r0 = tile * TILE_ROWS;
    TICK(15);
    {
      const int nt = tile + gridDim.x;
      pnext = 0.f;
      if (tid < 48 && nt < ntiles && nt * 48 + tid < M * 3) pnext = pts[nt * 48 + tid];
    }
    // acts == nullptr: forward only (inference) - the activations are not written
    const bool ok0 = acts != nullptr && r0 + l31 < R, ok1 = acts != nullptr && r0 + 32 + l31 < R;
    float* __restrict__ crow = acts + (acts != nullptr ? (size_t)(r0 + l31) * 128 + fb : 0);   // this lane's row of half 0 in X0
    _Float16* const arow0 = &At[0][l31 * LDH2 + fb];                            // ... in LDS tile 0 / 1
    _Float16* const arow1 = &At[1][l31 * LDH2 + fb];
    float* const frow1 = reinterpret_cast<float*>(At[1]) + l31 * LDA + fb;      // fp32 view of tile 1
    f32x16 acc0, acc1;
    // ---- layer 0 (3 -> 128), 4-row form: row 4s = W0 [p, 1], rows 4s+1.. = W0 e_i (gated by the primal row's state)
    const float pmx = fmaxf(fmaxf(Ps[par][48], Ps[par][49]), Ps[par][50]);
    const int ein = scale_exp(fmaxf(pmx, 1.f));
    const int e0 = scale_exp(fmaxf(fmaf(w0l1, pmx, b0mx), w0mx));
    const float sin = pow2(ein), sc0 = pow2(e0);
    auto layer0 = [&](int t, f32x16& acc) {
      const int sl = 8 * t + (l31 >> 2), c = l31 & 3;                      // sample of this lane's row, row kind
      const bool valid = s0 + sl < M && lh == 0;
      const float px = Ps[par][sl * 3], py = Ps[par][sl * 3 + 1], pz = Ps[par][sl * 3 + 2];
      float4 x = make_float4(c == 0 ? px : (c == 1 ? 1.f : 0.f), c == 0 ? py : (c == 2 ? 1.f : 0.f),
                             c == 0 ? pz : (c == 3 ? 1.f : 0.f), c == 0 ? 1.f : 0.f);
      if (!valid) x = make_float4(0.f, 0.f, 0.f, 0.f);
      pp_half8 xh, xl;
      split8(x, make_float4(0.f, 0.f, 0.f, 0.f), sin, xh, xl);
      zero16(acc);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, xh, acc, 0, 0, 0);
    };
    layer0(0, acc0);
    layer0(1, acc1);
    const float inv0 = pow2(-(ein + ew0)), ninv0 = pow2(-e0);
    HalfEpilogue<4, true> ea;          // epilogue state of the half that is being written out
    ea.begin(brow(3));
    ea.all(acc0, inv0, sc0, ok0, crow, arow0);
    slot_max(&Mx[par][0], ea.vmax(ninv0), lane);
    TICK(0);
    __syncthreads();
    TICK(1);
    if (tid < 8) Mx[par ^ 1][tid] = 0u;                   // the next tile's slots (last read a tile ago)
    // ---- stage A1: layer 1 on half 0  ||  epilogue of (layer 0, half 1)
    zero16(acc0);
    ea.begin(brow(3));
    mma_half(&At[0][0], w1, acc0, l31, lh, [&](int ks) { ea.step(ks, acc1, inv0, sc0, ok1, crow + 32 * 128, arow0 + 32 * LDH2); });
    slot_max(&Mx[par][1], ea.vmax(ninv0), lane);
    park(par ^ 1, pnext);                                  // next tile's positions (read after >= 3 barriers)
    TICK(2);
    __syncthreads();
    TICK(3);
    // ---- stage B1: layer 1 on half 1  ||  epilogue of (layer 1, half 0)
    const int e10 = scale_exp(fmaf(slot_get(&Mx[par][0]), l1_1, b1mx));
    const float sc10 = pow2(e10), inv1 = pow2(-(e0 + ew1));
    zero16(acc1);
    ea.begin(brow(0));
    mma_half(&At[0][32 * LDH2], w1, acc1, l31, lh, [&](int ks) { ea.step(ks, acc0, inv1, sc10, ok0, crow + LS, arow1); });
    slot_max(&Mx[par][2], ea.vmax(pow2(-e10)), lane);
    TICK(4);
    __syncthreads();
    TICK(5);
    // ---- stage A2: layer 2 on half 0  ||  epilogue of (layer 1, half 1)
    const int e11 = scale_exp(fmaf(slot_get(&Mx[par][1]), l1_1, b1mx));
    const float sc11 = pow2(e11);
    zero16(acc0);
    ea.begin(brow(0));
    mma_half(&At[1][0], w2, acc0, l31, lh, [&](int ks) { ea.step(ks, acc1, inv1, sc11, ok1, crow + LS + 32 * 128, arow1 + 32 * LDH2); });
    slot_max(&Mx[par][3], ea.vmax(pow2(-e11)), lane);
    TICK(6);
    __syncthreads();
    TICK(7);
    // ---- stage B2: layer 2 on half 1  ||  epilogue of (layer 2, half 0)
    const int e20 = scale_exp(fmaf(slot_get(&Mx[par][2]), l1_2, b2mx));
    const float sc20 = pow2(e20), inv20 = pow2(-(e10 + ew2)), inv21 = pow2(-(e11 + ew2));
    zero16(acc1);
    ea.begin(brow(1));
    mma_half(&At[1][32 * LDH2], w2, acc1, l31, lh, [&](int ks) { ea.step(ks, acc0, inv20, sc20, ok0, crow + 2 * LS, arow0); });
    TICK(8);
    __syncthreads();
    TICK(9);
    // ---- stage A3: layer 3 on half 0  ||  epilogue of (layer 2, half 1)
    const int e21 = scale_exp(fmaf(slot_get(&Mx[par][3]), l1_2, b2mx));
    const float sc21 = pow2(e21);
    zero16(acc0);
    ea.begin(brow(1));
    mma_half(&At[0][0], w3, acc0, l31, lh, [&](int ks) { ea.step(ks, acc1, inv21, sc21, ok1, crow + 2 * LS + 32 * 128, arow0 + 32 * LDH2); });
    TICK(10);
    __syncthreads();
    TICK(11);
    // ---- stage B3: layer 3 on half 1  ||  epilogue of (layer 3, half 0) into the fp32 view, then that of half 1
    HalfEpilogue<4, false> ef;
    const float inv30 = pow2(-(e20 + ew3)), inv31 = pow2(-(e21 + ew3));
    zero16(acc1);
    ef.begin(brow(2));
    mma_half(&At[0][32 * LDH2], w3, acc1, l31, lh, [&](int ks) { ef.step(ks, acc0, inv30, 1.f, ok0, crow + 3 * LS, reinterpret_cast<_Float16*>(frow1)); });
    ef.begin(brow(2));
    ef.all(acc1, inv31, 1.f, ok1, crow + 3 * LS + 32 * 128, reinterpret_cast<_Float16*>(frow1 + 32 * LDA));
    TICK(12);
    __syncthreads();
    TICK(13);
    // ---- output layer (128 -> 4) on v_mfma_f32_4x4x1, fp32 view of tile 1 (as in k_warp_fused_fwd)
    {
      const float* As1 = reinterpret_cast<const float*>(At[1]);
      const float* xr = &As1[lane * LDA + 32 * wid];
      const float* wr = &W4s[(lane & 3) * LDA + 32 * wid];
      float4 xv[8], wv[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        xv[g] = *reinterpret_cast<const float4*>(xr + 4 * g);
        wv[g] = *reinterpret_cast<const float4*>(wr + 4 * g);
      }
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].x, wv[g].x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].y, wv[g].y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].z, wv[g].z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].w, wv[g].w, d1, 0, 0, 0);
      }
      *reinterpret_cast<float4*>(&Red[(wid * 64 + lane) * 4]) = make_float4(d0[0] + d1[0], d0[1] + d1[1], d0[2] + d1[2], d0[3] + d1[3]);
    }
    __syncthreads();
    {
      const int row = tid >> 2, o = tid & 3;
      const int idx = ((row >> 2) * 4 + o) * 4 + (row & 3);
      const float sum = (Red[idx] + Red[256 + idx]) + (Red[512 + idx] + Red[768 + idx]);
      if (r0 + row < R) out[(size_t)r0 * 4 + tid] = (sum + b4) * out_range;
    }
    TICK(14);
    // the next tile's first barrier orders these reads of tile 1 / Red before they are overwritten
  }
#ifdef MS_TIMERS
  if (tid == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ms_t[i], tsum[i]);
#endif
}

int pp_launch_warp_fused_fwd_s(const float* params, const float* pts, const int32_t* count, int capacity, float out_range,
                               float* acts, float* out, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, 16);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_warp_fused_fwd_s, dim3(grid), dim3(256), 0, st, params, pts, count, capacity, out_range, acts, out);
  return 0;
}

// ------------------------------------------------------------------------------------------------ warp net, backward
// Same contract as k_warp_fused_bwd (pp_mlp_fused.hip): data-gradient chain Ybar3 -> Ybar2 -> Ybar1 -> Ybar0 with the transposed
// weights stationary, the output layer's backward (Ybar3, W4bar, b4bar), the input layer's backward (pts_grad, W0bar, b0bar) and
// all bias gradients; Ybar3 / Ybar2 / Ybar1 go to `ybar` ([3][4 cap][128] fp32) for the weight-gradient kernel.
// Differences to the forward kernel above: the epilogue gates by the STORED activation of the layer's input (primal row of the
// quad, staged by LDS-direct loads one stage ahead into GT) instead of a quad broadcast, there is no bias, and the lanes
// accumulate the bias gradients (lane = row, so the per-feature sums are reduced over the primal lanes once, at the end).
// Everything in a tile is linear in out_grad, so the scale of the first image (from the tile's largest |out_grad|) would do for
// all of them; the per-half maxima are tracked anyway, which keeps the bounds one product deep.
namespace {

// transposed weights of input feature j (A operand of the data-gradient product): k = n = 16 ks + 8 lh + jj, value W[n][j].
// Returns the exponent of the layer's scale; l1 = max_j sum_n |W[n][j]|.
__device__ __forceinline__ int load_w_cols_split(SplitW& w, float& l1, const float* __restrict__ W, int j, int lh, float* red, int tid, int& phase) {
  float4 v[16];
  float mx = 0.f, sum = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const float* p = W + (size_t)(16 * ks + 8 * lh) * 128 + j;
    v[2 * ks] = make_float4(p[0], p[128], p[256], p[384]);
    v[2 * ks + 1] = make_float4(p[512], p[640], p[768], p[896]);
    mx = fmaxf(mx, fmaxf(amax4(v[2 * ks]), amax4(v[2 * ks + 1])));
    sum += asum4(v[2 * ks]) + asum4(v[2 * ks + 1]);
  }
  sum += __shfl_xor(sum, 32, 64);
  const int e = scale_exp(block_max(mx, red, tid, phase));
  const float s = pow2(e);
  l1 = block_max(sum, red, tid, phase) * 1.0001f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) split8(v[2 * ks], v[2 * ks + 1], s, w.h[ks], w.l[ks]);
  return e;
}

// Epilogue of one 32-row half of a data-gradient product, eight batched steps (see HalfEpilogue): v = gate ? acc * inv : 0 with
// gate = stored activation of the quad's primal row > 0 (grow: that row in GT / XS + fb), fp32 copy to HBM (`ok` rows;
// Crow == nullptr: none), next tile to LDS as a split image (SPLIT) or as the fp32 view.  The bias gradients of the hidden layers
// (column sums of Ybar over the primal rows) are NOT formed here - with lane = row they would cost 48 accumulator registers and
// a cross-lane reduction; the weight-gradient kernel, which has every Ybar tile in LDS anyway, adds them (pp_launch_wgrad_chain).
template <bool SPLIT>
struct HalfEpilogueB {
  float4 g[4];
  float y[16], p[16];
  unsigned h[8], l[8];
  unsigned long long m[16];
  float xmax;
  __device__ __forceinline__ void begin(const float* __restrict__ grow) {
    xmax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = *reinterpret_cast<const float4*>(grow + 8 * q);
  }
  __device__ __forceinline__ void step(int i, const f32x16& acc, float inv, float snext, bool ok, float* __restrict__ Crow,
                                       _Float16* __restrict__ Arow) {
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) y[e] = acc[e] * inv;
    } else if (i == 1) {
      // gates into sixteen SGPR pairs, consumed a step later: no VCC round trip between a compare and its select
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m[4 * q]) : "v"(g[q].x));
        asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m[4 * q + 1]) : "v"(g[q].y));
        asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m[4 * q + 2]) : "v"(g[q].z));
        asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m[4 * q + 3]) : "v"(g[q].w));
      }
    } else if (i == 2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y[e]) : "v"(y[e]), "s"(m[e]));
    } else if (SPLIT && i == 3) {
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = y[e] * snext;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(xmax) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
      }
    } else if (SPLIT && i == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[k]) : "v"(h[k]), "v"(p[2 * k]));
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[k]) : "v"(h[k]), "v"(p[2 * k + 1]));
    } else if (i == 5) {
      if (SPLIT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<uint2*>(Arow + 8 * q) = make_uint2(h[2 * q], h[2 * q + 1]);
          *reinterpret_cast<uint2*>(Arow + 8 * q + PLANE) = make_uint2(l[2 * q], l[2 * q + 1]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Arow) + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
    } else if (i == 6) {
      // HBM copy last: the loads a stage issues after this step (step 7 is left to the caller) are then the youngest memory
      // operations, and a vmcnt(0) placed before step 6 of the NEXT stage waits for nothing younger than them
      if (Crow != nullptr && ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(Crow + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void all(const f32x16& acc, float inv, float snext, bool ok, float* __restrict__ Crow, _Float16* __restrict__ Arow) {
#pragma unroll
    for (int i = 0; i < 8; ++i) step(i, acc, inv, snext, ok, Crow, Arow);
  }
  __device__ __forceinline__ float vmax(float ninv) const { return xmax * ninv; }
};

#define PP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

}  // namespace

__global__ __launch_bounds__(256) void k_warp_fused_bwd_s(const float* __restrict__ params, const float* __restrict__ pts,
                                                          const float* __restrict__ acts, const float* __restrict__ out_grad,
                                                          const int32_t* __restrict__ count, int capacity, float out_range,
                                                          float* __restrict__ ybar, float* __restrict__ params_grad,
                                                          float* __restrict__ pts_grad) {
  __shared__ __attribute__((aligned(16))) _Float16 At[2][2 * PLANE];      // tile 1 doubles as the fp32 [64][LDA] view (Ybar0)
  __shared__ __attribute__((aligned(16))) float XS[TILE_ROWS * 128];      // X3 rows of the NEXT tile (unpadded, lane-contiguous)
  __shared__ __attribute__((aligned(16))) float GT[2][16 * 128];          // primal rows of the gating activation of a layer
  __shared__ __attribute__((aligned(16))) float W0s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float G[2][256];                // raw out_grad of the current / next tile
  __shared__ __attribute__((aligned(16))) float Ps[2][64];                // sample positions of the current / next tile
  __shared__ __attribute__((aligned(16))) float Red[4 * 64];
  __shared__ unsigned Mx[2][8];      // per parity: max |Ybar3| (halves 0, 1), |Ybar2| (0, 1), |Ybar1| (0, 1), max |out_grad| of the tile
  __shared__ __attribute__((aligned(16))) float red4[32];
  const int M = min(count[0], capacity);
  const int R = 4 * M;
  const int ntiles = (M + 15) >> 4;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const int fb = wid * 32 + 4 * lh;
  const int pr = l31 >> 2;                            // primal row (sample) of this lane's quad inside a 32-row half
  const size_t LS = (size_t)capacity * 4 * 128;
  const float* __restrict__ X0 = acts;
  const float* __restrict__ X1 = acts + LS;
  const float* __restrict__ X2 = acts + 2 * LS;
  const float* __restrict__ X3 = acts + 3 * LS;

  SplitW w3, w2, w1;
  int phase = 0;                        // call counter of block_max
  float l1_3, l1_2, l1_1;
  const int ew3 = load_w_cols_split(w3, l1_3, params + WPF_W3, col, lh, red4, tid, phase);
  const int ew2 = load_w_cols_split(w2, l1_2, params + WPF_W2, col, lh, red4, tid, phase);
  const int ew1 = load_w_cols_split(w1, l1_1, params + WPF_W1, col, lh, red4, tid, phase);
  // output layer backward as a K = 16 product: A = [w4_0 w4_1 w4_2 w4_3 0 ...] * out_range of feature `col` (lanes lh = 0),
  // B = the row's four out_grad entries
  const float w4a = params[WPF_W4 + col] * out_range, w4b = params[WPF_W4 + 128 + col] * out_range,
              w4c = params[WPF_W4 + 256 + col] * out_range, w4d = params[WPF_W4 + 384 + col] * out_range;
  const float w4l1 = block_max((fabsf(w4a) + fabsf(w4b)) + (fabsf(w4c) + fabsf(w4d)), red4, tid, phase) * 1.0001f;
  const int ew4 = scale_exp(block_max(fmaxf(fmaxf(fabsf(w4a), fabsf(w4b)), fmaxf(fabsf(w4c), fabsf(w4d))), red4, tid, phase));
  pp_half8 a4h, a4l;
  {
    const float z = 0.f;
    const float4 wa = lh == 0 ? make_float4(w4a, w4b, w4c, w4d) : make_float4(z, z, z, z);
    split8(wa, make_float4(z, z, z, z), pow2(ew4), a4h, a4l);
  }
  const int j0 = tid & 127, h0 = tid >> 7;
  for (int i = tid; i < 512; i += 256) {
    const int r = i >> 7, j = i & 127;
    W0s[r * LDA + j] = (r < 3) ? params[WPF_W0 + j * 3 + r] : 0.f;
  }
  float wacc4[4] = {0.f, 0.f, 0.f, 0.f}, bacc4 = 0.f, wacc0[3] = {0.f, 0.f, 0.f}, bacc0 = 0.f;

  // ---- LDS-direct staging of tile t into parity slot b: X3 rows -> XS, out_grad -> G[b], positions -> Ps[b] (10 pieces);
  // rows / samples past the end are clamped, their out_grad slot is zeroed after the wait
  auto stage_piece = [&](int t, int b, int i) {
    const int rn0 = t * TILE_ROWS, sn0 = t * 16;
    if (i < 8) {
      const int rl = 16 * wid + 2 * i;
      const int row = min(rn0 + rl + lh, R - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X3 + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(&XS[rl * 128]), 16, 0, 0);
    } else if (i == 8) {
      const int e = min(sn0 * 16 + tid, M * 16 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(out_grad + e), PP_LDS_PTR(&G[b][wid * 64]), 4, 0, 0);
    } else if (wid == 0) {
      const int e = min(sn0 * 3 + lane, M * 3 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(pts + e), PP_LDS_PTR(&Ps[b][0]), 4, 0, 0);
    }
  };
  // primal rows (every fourth) of activation X of tile rows r0.. -> GT[gb]: 16 rows x 512 B, two pieces per wavefront
  auto stage_gate_piece = [&](const float* __restrict__ X, int r0, int gb, int i) {
    const int prow = 4 * wid + 2 * i;
    const int row = min(r0 + 4 * (prow + lh), R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(&GT[gb][prow * 128]), 16, 0, 0);
  };
  // samples past M contribute nothing; largest |out_grad| of the tile into its slot
  auto prepare_grad = [&](int t, int b) {
    float gv = G[b][tid];
    if (t * 16 + (tid >> 4) >= M) { gv = 0.f; G[b][tid] = 0.f; }
    slot_max(&Mx[b][6], fabsf(gv), lane);
  };
  // W4bar / b4bar of the staged tile (thread = feature j0, rows of half h0)
  auto w4_accumulate = [&](int b) {
    const float* __restrict__ xs = &XS[(h0 * 32) * 128 + j0];
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float4 gq = *reinterpret_cast<const float4*>(&G[b][(h0 * 32 + r) * 4]);
      const float x = xs[r * 128];
      wacc4[0] = fmaf(gq.x, x, wacc4[0]); wacc4[1] = fmaf(gq.y, x, wacc4[1]);
      wacc4[2] = fmaf(gq.z, x, wacc4[2]); wacc4[3] = fmaf(gq.w, x, wacc4[3]);
      if (j0 < 4 && (r & 3) == 0) bacc4 += G[b][(h0 * 32 + r) * 4 + j0];
    }
  };
  // Ybar3 (pre-gate) of half t of the staged tile: three MFMAs
  auto out_layer = [&](int b, int t, float sg, f32x16& acc) {
    float4 x = *reinterpret_cast<const float4*>(&G[b][(t * 32 + l31) * 4]);
    if (lh != 0) x = make_float4(0.f, 0.f, 0.f, 0.f);
    pp_half8 xh, xl;
    split8(x, make_float4(0.f, 0.f, 0.f, 0.f), sg, xh, xl);
    zero16(acc);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4l, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, xh, acc, 0, 0, 0);
  };
  // Ybar3 of the staged tile (parity b) into LDS tile 0 and `ybar`: both halves; returns the exponents of the two images
  int e3n0 = 0, e3n1 = 0;
  auto out_layer_bwd = [&](int t, int b) {
    const float gmx = slot_get(&Mx[b][6]);
    const int eg = scale_exp(gmx);
    const int e3 = scale_exp(gmx * w4l1);
    const float sg = pow2(eg), s3 = pow2(e3), inv = pow2(-(eg + ew4)), n3 = pow2(-e3);
    const int r0 = t * TILE_ROWS;
    f32x16 a0, a1;
    out_layer(b, 0, sg, a0);
    out_layer(b, 1, sg, a1);
    HalfEpilogueB<true> eb;
    float* crow = ybar + (size_t)(r0 + l31) * 128 + fb;
    eb.begin(&XS[(4 * pr) * 128 + fb]);
    eb.all(a0, inv, s3, r0 + l31 < R, crow, &At[0][l31 * LDH2 + fb]);
    slot_max(&Mx[b][0], eb.vmax(n3), lane);
    eb.begin(&XS[(32 + 4 * pr) * 128 + fb]);
    eb.all(a1, inv, s3, r0 + 32 + l31 < R, crow + 32 * 128, &At[0][(32 + l31) * LDH2 + fb]);
    slot_max(&Mx[b][1], eb.vmax(n3), lane);
    e3n0 = e3; e3n1 = e3;
  };

  if (tid < 16) Mx[tid >> 3][tid & 7] = 0u;
#pragma unroll
  for (int i = 0; i < 10; ++i) stage_piece(blockIdx.x, 0, i);
  __builtin_amdgcn_s_waitcnt(0);        // all prologue loads landed
  __syncthreads();
  prepare_grad(blockIdx.x, 0);
  __syncthreads();
  w4_accumulate(0);
  out_layer_bwd(blockIdx.x, 0);
  __syncthreads();

  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS, s0 = tile * 16;
    const int tnext = tile + gridDim.x;
    const bool more = tnext < ntiles;
    const bool ok0 = r0 + l31 < R, ok1 = r0 + 32 + l31 < R;
    const int e30 = e3n0, e31 = e3n1;                     // exponents of this tile's Ybar3 image (set by out_layer_bwd)
    float* const crow = ybar + (size_t)(r0 + l31) * 128 + fb;
    _Float16* const arow0 = &At[0][l31 * LDH2 + fb];
    _Float16* const arow1 = &At[1][l31 * LDH2 + fb];
    float* const frow1 = reinterpret_cast<float*>(At[1]) + l31 * LDA + fb;
    const float* const grow0 = &GT[0][pr * 128 + fb];     // gate rows of half 0 (half 1: + 8 rows)
    const float* const grow1 = &GT[1][pr * 128 + fb];
    f32x16 acc0, acc1;
    HalfEpilogueB<true> eb;
    if (tid < 7) Mx[par ^ 1][tid] = 0u;                   // the next tile's slots
    // ---- stage A3: layer 3 on half 0; loads: gates X2 -> GT[0], the next tile's X3 / out_grad / positions
    zero16(acc0);
    mma_half(&At[0][0], w3, acc0, l31, lh, [&](int ks) {
      if (ks < 2) stage_gate_piece(X2, r0, 0, ks);
      if (more) { stage_piece(tnext, par ^ 1, ks); if (ks < 2) stage_piece(tnext, par ^ 1, 8 + ks); }
    });
    PP_WAIT_VMEM();
    __syncthreads();
    // ---- stage B3: layer 3 on half 1  ||  epilogue of (Ybar2, half 0); loads: gates X1 -> GT[1]
    const int e20 = scale_exp(slot_get(&Mx[par][0]) * l1_3), e21 = scale_exp(slot_get(&Mx[par][1]) * l1_3);
    if (more) prepare_grad(tnext, par ^ 1);
    zero16(acc1);
    eb.begin(grow0);
    mma_half(&At[0][32 * LDH2], w3, acc1, l31, lh, [&](int ks) {
      eb.step(ks, acc0, pow2(-(e30 + ew3)), pow2(e20), ok0, crow + LS, arow1);
      if (ks == 7) { stage_gate_piece(X1, r0, 1, 0); stage_gate_piece(X1, r0, 1, 1); }      // after the stage's stores
    });
    slot_max(&Mx[par][2], eb.vmax(pow2(-e20)), lane);
    __syncthreads();
    // ---- stage A2: layer 2 on half 0  ||  epilogue of (Ybar2, half 1)
    zero16(acc0);
    eb.begin(grow0 + 8 * 128);
    mma_half(&At[1][0], w2, acc0, l31, lh, [&](int ks) {
      if (ks == 5) PP_WAIT_VMEM();               // the X1 gates (and nothing younger: this stage's stores come in step 6)
      eb.step(ks, acc1, pow2(-(e31 + ew3)), pow2(e21), ok1, crow + LS + 32 * 128, arow1 + 32 * LDH2);
    });
    slot_max(&Mx[par][3], eb.vmax(pow2(-e21)), lane);
    __syncthreads();
    // ---- stage B2: layer 2 on half 1  ||  epilogue of (Ybar1, half 0); loads: gates X0 -> GT[0]
    const int e10 = scale_exp(slot_get(&Mx[par][2]) * l1_2), e11 = scale_exp(slot_get(&Mx[par][3]) * l1_2);
    zero16(acc1);
    eb.begin(grow1);
    mma_half(&At[1][32 * LDH2], w2, acc1, l31, lh, [&](int ks) {
      eb.step(ks, acc0, pow2(-(e20 + ew2)), pow2(e10), ok0, crow + 2 * LS, arow0);
      if (ks == 7) { stage_gate_piece(X0, r0, 0, 0); stage_gate_piece(X0, r0, 0, 1); }
    });
    __syncthreads();
    // ---- stage A1: layer 1 on half 0  ||  epilogue of (Ybar1, half 1)
    zero16(acc0);
    eb.begin(grow1 + 8 * 128);
    mma_half(&At[0][0], w1, acc0, l31, lh, [&](int ks) {
      if (ks == 5) PP_WAIT_VMEM();               // the X0 gates
      eb.step(ks, acc1, pow2(-(e21 + ew2)), pow2(e11), ok1, crow + 2 * LS + 32 * 128, arow0 + 32 * LDH2);
    });
    __syncthreads();
    // ---- stage B1: layer 1 on half 1  ||  epilogue of (Ybar0, half 0) into the fp32 view; then that of half 1
    HalfEpilogueB<false> ef;
    zero16(acc1);
    ef.begin(grow0);
    mma_half(&At[0][32 * LDH2], w1, acc1, l31, lh, [&](int ks) {
      ef.step(ks, acc0, pow2(-(e10 + ew1)), 1.f, false, nullptr, reinterpret_cast<_Float16*>(frow1));
    });
    ef.begin(grow0 + 8 * 128);
    ef.all(acc1, pow2(-(e11 + ew1)), 1.f, false, nullptr, reinterpret_cast<_Float16*>(frow1 + 32 * LDA));
    __syncthreads();
    // ---- layer 0: W0bar[j][i] += Ybar0[4s][j] p_i + Ybar0[4s+1+i][j], b0bar[j] += Ybar0[4s][j]  (thread = feature j0)
    const float* As1 = reinterpret_cast<const float*>(At[1]);
    {
      const float* __restrict__ yb = &As1[(4 * h0 * 8) * LDA + j0];
      const float* ps = &Ps[par][h0 * 24];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float y0 = yb[(4 * q) * LDA], y1 = yb[(4 * q + 1) * LDA], y2 = yb[(4 * q + 2) * LDA], y3 = yb[(4 * q + 3) * LDA];
        wacc0[0] += y0 * ps[q * 3] + y1;
        wacc0[1] += y0 * ps[q * 3 + 1] + y2;
        wacc0[2] += y0 * ps[q * 3 + 2] + y3;
        bacc0 += y0;
      }
    }
    // pts_grad[s][i] = sum_j Ybar0[4s][j] W0[j][i] on v_mfma_f32_4x4x1 (lane = row, K slice per wavefront)
    {
      const float* xr = &As1[lane * LDA + 32 * wid];
      const float* wr = &W0s[(lane & 3) * LDA + 32 * wid];
      float4 xv[8], wv[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        xv[g] = *reinterpret_cast<const float4*>(xr + 4 * g);
        wv[g] = *reinterpret_cast<const float4*>(wr + 4 * g);
      }
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].x, wv[g].x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].y, wv[g].y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].z, wv[g].z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].w, wv[g].w, d1, 0, 0, 0);
      }
      Red[wid * 64 + lane] = d0[0] + d1[0];       // register 0 = primal row of sample lane>>2, output lane&3
    }
    // ---- the next tile's output-layer backward: W4bar sums, Ybar3 image into tile 0 (last read by stage B1's MFMAs)
    if (more) {
      w4_accumulate(par ^ 1);
      out_layer_bwd(tnext, par ^ 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int s = tid >> 2, i = tid & 3;
      if (i < 3 && s0 + s < M) {
        const float v = (Red[tid] + Red[64 + tid]) + (Red[128 + tid] + Red[192 + tid]);
        atomicAdd(&pts_grad[(s0 + s) * 3 + i], v);
      }
    }
  }

  // ---- flush the thin-layer weight gradients (one atomic per entry and work-group)
  __syncthreads();
  float* red = reinterpret_cast<float*>(At[0]);
  if (h0 == 1) {
#pragma unroll
    for (int o = 0; o < 4; ++o) red[o * 128 + j0] = wacc4[o];
#pragma unroll
    for (int i = 0; i < 3; ++i) red[(4 + i) * 128 + j0] = wacc0[i];
    red[7 * 128 + j0] = bacc0;
    if (j0 < 4) red[8 * 128 + j0] = bacc4;
  }
  __syncthreads();
  if (h0 == 0) {
#pragma unroll
    for (int o = 0; o < 4; ++o) atomicAdd(&params_grad[WPF_W4 + o * 128 + j0], (wacc4[o] + red[o * 128 + j0]) * out_range);
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(&params_grad[WPF_W0 + j0 * 3 + i], wacc0[i] + red[(4 + i) * 128 + j0]);
    atomicAdd(&params_grad[WPF_B0 + j0], bacc0 + red[7 * 128 + j0]);
    if (j0 < 4) atomicAdd(&params_grad[WPF_B4 + j0], (bacc4 + red[8 * 128 + j0]) * out_range);
  }
}

int pp_launch_warp_fused_bwd_s(const float* params, const float* pts, const float* acts, const float* out_grad,
                               const int32_t* count, int capacity, float out_range, float* ybar, float* params_grad,
                               float* pts_grad, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, 16);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_warp_fused_bwd_s, dim3(grid), dim3(256), 0, st, params, pts, acts, out_grad, count, capacity, out_range,
                     ybar, params_grad, pts_grad);
  return 0;
}

// ================================================================================================ rgbnet (64 -> 128 x3 -> 3)
// Same contract as k_rgb_fused_fwd (pp_mlp_fused.hip): feat[M][64] -> rgb[M][3] = sigmoid(MLP(feat) (+ logit_add)); hidden
// activations H0..H2 ([cap][128] fp32 each) kept for backward.  One row per sample, so the gate is a plain ReLU (COLS == 1).
// The input tile arrives by LDS-direct loads as fp32 and is split by the threads that fetched it: four threads per row, the row's
// own power of two as scale (rows are the N index of the product, so the scale may differ per row; the epilogue's factor is then
// per lane).
namespace {

#define FLD 72                          // halfs per row of the split feature tile (144 B: conflict-free ds_read_b128 fragments)
#define FPL (TILE_ROWS * FLD)

// forward weights of feature n of the 64-wide input layer: k = 16 ks + 8 lh + j, ks < 4 (the upper half of `w` stays unused)
__device__ __forceinline__ int load_w0_rows_split(SplitW& w, float& l1, const float* __restrict__ W, int n, int lh, float* red, int tid, int& phase) {
  float4 v[8];
  float mx = 0.f, sum = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    v[2 * ks] = *reinterpret_cast<const float4*>(W + (size_t)n * 64 + 16 * ks + 8 * lh);
    v[2 * ks + 1] = *reinterpret_cast<const float4*>(W + (size_t)n * 64 + 16 * ks + 8 * lh + 4);
    mx = fmaxf(mx, fmaxf(amax4(v[2 * ks]), amax4(v[2 * ks + 1])));
    sum += asum4(v[2 * ks]) + asum4(v[2 * ks + 1]);
  }
  sum += __shfl_xor(sum, 32, 64);
  const int e = scale_exp(block_max(mx, red, tid, phase));
  const float s = pow2(e);
  l1 = block_max(sum, red, tid, phase) * 1.0001f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) split8(v[2 * ks], v[2 * ks + 1], s, w.h[ks], w.l[ks]);
#pragma unroll
  for (int ks = 4; ks < 8; ++ks) { w.h[ks] = w.h[0]; w.l[ks] = w.l[0]; }
  return e;
}

// LDS-direct load of a [64][64] fp32 tile into an unpadded buffer whose 16-byte slots are XOR-swizzled (pp_mlp_fused.hip)
__device__ __forceinline__ void stage_feat_tile_s(const float* __restrict__ feat, int r0, int R, float* Fs, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = 16 * wid + 4 * i + (lane >> 4);            // four rows per instruction
    const int c4 = (lane & 15) ^ (rl & 15);
    const int row = min(r0 + rl, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(feat + (size_t)row * 64 + c4 * 4), PP_LDS_PTR(Fs + (16 * wid + 4 * i) * 64), 16, 0, 0);
  }
}

}  // namespace

__global__ __launch_bounds__(256) void k_rgb_fused_fwd_s(const float* __restrict__ params, const float* __restrict__ feat,
                                                         const int32_t* __restrict__ count, int capacity,
                                                         const float* __restrict__ logit_add, int add_ld,
                                                         float* __restrict__ acts, float* __restrict__ rgb) {
  __shared__ __attribute__((aligned(16))) _Float16 At[2][2 * PLANE];      // tile 0 doubles as the fp32 [64][LDA] view (H2)
  __shared__ __attribute__((aligned(16))) float Fs[2][TILE_ROWS * 64];    // fp32 feature tiles (current / next), swizzled slots
  __shared__ __attribute__((aligned(16))) _Float16 Fh[2 * FPL];           // split image of the current feature tile
  __shared__ __attribute__((aligned(16))) float W3s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float Red[4 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float Bs[3][128];
  __shared__ int Er[TILE_ROWS];                                            // scale exponent of every row of the feature tile
  __shared__ unsigned Mx[2][8];      // per parity: max |feat| (halves 0, 1), |H0| (0, 1), |H1| (0, 1)
  __shared__ __attribute__((aligned(16))) float red4[32];
  const int R = min(count[0], capacity);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const int fb = wid * 32 + 4 * lh;
  const size_t LS = (size_t)capacity * 128;

  SplitW w0, w1, w2;
  int phase = 0;                        // call counter of block_max
  float l1_0, l1_1, l1_2;
  const int ew0 = load_w0_rows_split(w0, l1_0, params + RGF_W0, col, lh, red4, tid, phase);
  const int ew1 = load_w_rows_split(w1, l1_1, params + RGF_W1, col, lh, red4, tid, phase);
  const int ew2 = load_w_rows_split(w2, l1_2, params + RGF_W2, col, lh, red4, tid, phase);
  if (tid < 128) { Bs[0][tid] = params[RGF_B0 + tid]; Bs[1][tid] = params[RGF_B1 + tid]; Bs[2][tid] = params[RGF_B2 + tid]; }
  const float b0mx = block_max(fabsf(params[RGF_B0 + col]), red4, tid, phase), b1mx = block_max(fabsf(params[RGF_B1 + col]), red4, tid, phase);
  for (int i = tid; i < 512; i += 256) {
    const int r = i >> 7, j = i & 127;
    W3s[r * LDA + j] = (r < 3) ? params[RGF_W3 + r * 128 + j] : 0.f;
  }
  const float b3 = ((tid & 3) < 3) ? params[RGF_B3 + (tid & 3)] : 0.f;
  if (tid < 16) Mx[tid >> 3][tid & 7] = 0u;
  stage_feat_tile_s(feat, blockIdx.x * TILE_ROWS, R, Fs[0], wid, lane);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS;
    const int tnext = tile + gridDim.x;
    const bool ok0 = acts != nullptr && r0 + l31 < R, ok1 = acts != nullptr && r0 + 32 + l31 < R;     // acts == nullptr: forward only
    float* __restrict__ crow = acts + (acts != nullptr ? (size_t)(r0 + l31) * 128 + fb : 0);
    _Float16* const arow0 = &At[0][l31 * LDH2 + fb];
    _Float16* const arow1 = &At[1][l31 * LDH2 + fb];
    float* const frow0 = reinterpret_cast<float*>(At[0]) + l31 * LDA + fb;      // fp32 view of tile 0
    // ---- split the feature tile: thread = (row tid >> 2, 16 columns); the rows a wavefront converts are the rows it loaded
    PP_WAIT_VMEM();
    {
      const int row = tid >> 2, qd = tid & 3;
      float4 f[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const float4*>(&Fs[par][row * 64 + (((4 * qd + i) ^ (row & 15)) << 2)]);
      float mx = fmaxf(fmaxf(amax4(f[0]), amax4(f[1])), fmaxf(amax4(f[2]), amax4(f[3])));
      if (r0 + row >= R) { mx = 0.f; f[0] = f[1] = f[2] = f[3] = make_float4(0.f, 0.f, 0.f, 0.f); }
      mx = fmaxf(mx, dpp_f<0xB1>(mx));
      mx = fmaxf(mx, dpp_f<0x4E>(mx));                                  // the row's maximum in all four of its lanes
      const int er = scale_exp(mx);
      const float sr = pow2(er);
      pp_half8 h0, l0, h1, l1;
      split8(f[0], f[1], sr, h0, l0);
      split8(f[2], f[3], sr, h1, l1);
      _Float16* d = &Fh[row * FLD + 16 * qd];
      *reinterpret_cast<pp_half8*>(d) = h0; *reinterpret_cast<pp_half8*>(d + 8) = h1;
      *reinterpret_cast<pp_half8*>(d + FPL) = l0; *reinterpret_cast<pp_half8*>(d + FPL + 8) = l1;
      if (qd == 0) Er[row] = er;
      slot_max(&Mx[par][wid >> 1], mx, lane);                           // rows 16 wid .. 16 wid + 15 belong to half wid >> 1
    }
    if (tnext < ntiles) stage_feat_tile_s(feat, tnext * TILE_ROWS, R, Fs[par ^ 1], wid, lane);   // lands during this tile
    __syncthreads();
    if (tid < 6) Mx[par ^ 1][tid] = 0u;
    f32x16 acc0, acc1;
    HalfEpilogue<1, true> ea;
    // ---- layer 0 (64 -> 128): half 0, then half 1 beside the epilogue of half 0
    const int e00 = scale_exp(fmaf(slot_get(&Mx[par][0]), l1_0, b0mx)), e01 = scale_exp(fmaf(slot_get(&Mx[par][1]), l1_0, b0mx));
    const float inv00 = pow2(-(Er[l31] + ew0)), inv01 = pow2(-(Er[32 + l31] + ew0));          // per lane: the row's own scale
    zero16(acc0);
    mma_half<4, FLD, FPL>(&Fh[0], w0, acc0, l31, lh);
    zero16(acc1);
    ea.begin(&Bs[0][fb]);
    mma_half<4, FLD, FPL>(&Fh[32 * FLD], w0, acc1, l31, lh, [&](int ks) {
      ea.step(2 * ks, acc0, inv00, pow2(e00), ok0, crow, arow0);
      ea.step(2 * ks + 1, acc0, inv00, pow2(e00), ok0, crow, arow0);
    });
    slot_max(&Mx[par][2], ea.vmax(pow2(-e00)), lane);
    __syncthreads();
    // ---- stage A1: layer 1 on half 0  ||  epilogue of (layer 0, half 1)
    zero16(acc0);
    ea.begin(&Bs[0][fb]);
    mma_half(&At[0][0], w1, acc0, l31, lh, [&](int ks) { ea.step(ks, acc1, inv01, pow2(e01), ok1, crow + 32 * 128, arow0 + 32 * LDH2); });
    slot_max(&Mx[par][3], ea.vmax(pow2(-e01)), lane);
    __syncthreads();
    // ---- stage B1: layer 1 on half 1  ||  epilogue of (layer 1, half 0)
    const int e10 = scale_exp(fmaf(slot_get(&Mx[par][2]), l1_1, b1mx));
    zero16(acc1);
    ea.begin(&Bs[1][fb]);
    mma_half(&At[0][32 * LDH2], w1, acc1, l31, lh, [&](int ks) { ea.step(ks, acc0, pow2(-(e00 + ew1)), pow2(e10), ok0, crow + LS, arow1); });
    __syncthreads();
    // ---- stage A2: layer 2 on half 0  ||  epilogue of (layer 1, half 1)
    const int e11 = scale_exp(fmaf(slot_get(&Mx[par][3]), l1_1, b1mx));
    zero16(acc0);
    ea.begin(&Bs[1][fb]);
    mma_half(&At[1][0], w2, acc0, l31, lh, [&](int ks) { ea.step(ks, acc1, pow2(-(e01 + ew1)), pow2(e11), ok1, crow + LS + 32 * 128, arow1 + 32 * LDH2); });
    __syncthreads();
    // ---- stage B2: layer 2 on half 1  ||  epilogue of (layer 2, half 0) into the fp32 view; then that of half 1
    HalfEpilogue<1, false> ef;
    zero16(acc1);
    ef.begin(&Bs[2][fb]);
    mma_half(&At[1][32 * LDH2], w2, acc1, l31, lh, [&](int ks) { ef.step(ks, acc0, pow2(-(e10 + ew2)), 1.f, ok0, crow + 2 * LS, reinterpret_cast<_Float16*>(frow0)); });
    ef.begin(&Bs[2][fb]);
    ef.all(acc1, pow2(-(e11 + ew2)), 1.f, ok1, crow + 2 * LS + 32 * 128, reinterpret_cast<_Float16*>(frow0 + 32 * LDA));
    __syncthreads();
    // ---- output layer (128 -> 3) on v_mfma_f32_4x4x1: lane = row, lane&3 = output, K slice per wavefront
    {
      const float* As0 = reinterpret_cast<const float*>(At[0]);
      const float* xr = &As0[lane * LDA + 32 * wid];
      const float* wr = &W3s[(lane & 3) * LDA + 32 * wid];
      float4 xv[8], wv[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        xv[g] = *reinterpret_cast<const float4*>(xr + 4 * g);
        wv[g] = *reinterpret_cast<const float4*>(wr + 4 * g);
      }
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].x, wv[g].x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].y, wv[g].y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].z, wv[g].z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[g].w, wv[g].w, d1, 0, 0, 0);
      }
      *reinterpret_cast<float4*>(&Red[(wid * 64 + lane) * 4]) = make_float4(d0[0] + d1[0], d0[1] + d1[1], d0[2] + d1[2], d0[3] + d1[3]);
    }
    __syncthreads();
    {
      const int row = tid >> 2, o = tid & 3;
      const int idx = ((row >> 2) * 4 + o) * 4 + (row & 3);
      const float sum = (Red[idx] + Red[256 + idx]) + (Red[512 + idx] + Red[768 + idx]);
      if (o < 3 && r0 + row < R) {
        const size_t m = (size_t)(r0 + row);
        rgb[m * 3 + o] = pp_sigmoid(sum + b3 + (logit_add ? logit_add[m * add_ld + o] : 0.f));
      }
    }
  }
}

int pp_launch_rgb_fused_fwd_s(const float* params, const float* feat, const int32_t* count, int capacity,
                              const float* logit_add, int add_ld, float* acts, float* rgb, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, TILE_ROWS);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_rgb_fused_fwd_s, dim3(grid), dim3(256), 0, st, params, feat, count, capacity, logit_add, add_ld, acts, rgb);
  return 0;
}

// ------------------------------------------------------------------------------------------------ rgbnet, backward
// Same contract as k_rgb_fused_bwd: output layer, the two 128 x 128 data-gradient products and the 128 -> 64 product onto the
// features; Ybar2 / Ybar1 / Ybar0 go to `ybar` ([3][cap][128] fp32) for k_wgrad_chain<64>, which also forms b2bar / b1bar / b0bar
// when this kernel ran (pp_launch_wgrad_chain, stride-1 column sums); W3bar and b3bar are accumulated here (thread = feature).
// Gates are per element (the layer input's own activation); the tile of gating activations arrives by LDS-direct loads into ONE
// buffer (H1, then H0 - a second 32 KB buffer does not fit beside the two images and the next tile's H2), so the two halves of
// Ybar0 are written out without MFMAs beside them.
__global__ __launch_bounds__(256) void k_rgb_fused_bwd_s(const float* __restrict__ params, const float* __restrict__ acts,
                                                         const float* __restrict__ rgb, const float* __restrict__ rgb_grad,
                                                         const int32_t* __restrict__ count, int capacity,
                                                         float* __restrict__ ybar, float* __restrict__ params_grad,
                                                         float* __restrict__ feat_grad, float* __restrict__ logit_grad, int lg_ld) {
  __shared__ __attribute__((aligned(16))) _Float16 At[2][2 * PLANE];
  __shared__ __attribute__((aligned(16))) float XS[TILE_ROWS * 128];      // H2 of the NEXT tile
  __shared__ __attribute__((aligned(16))) float GT[TILE_ROWS * 128];      // gates of the current layer (H1, then H0)
  __shared__ __attribute__((aligned(16))) float RG[2][2][192];            // [parity][rgb | rgb_grad] of a tile
  __shared__ __attribute__((aligned(16))) float GL[TILE_ROWS * 4];        // d loss / d logits of the staged tile
  __shared__ unsigned Mx[2][8];      // per parity: max |Ybar2| (halves 0, 1), |Ybar1| (0, 1), max |d logits| of the tile
  __shared__ __attribute__((aligned(16))) float red4[32];
  const int R = min(count[0], capacity);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const int fb = wid * 32 + 4 * lh;
  const size_t LS = (size_t)capacity * 128;
  const float* __restrict__ H0 = acts;
  const float* __restrict__ H1 = acts + LS;
  const float* __restrict__ H2 = acts + 2 * LS;

  SplitW w2, w1, w0;
  int phase = 0;                        // call counter of block_max
  float l1_2, l1_1;
  const int ew2 = load_w_cols_split(w2, l1_2, params + RGF_W2, col, lh, red4, tid, phase);
  const int ew1 = load_w_cols_split(w1, l1_1, params + RGF_W1, col, lh, red4, tid, phase);
  // last product: feat_grad[64 rows][64] = Ybar0 . W0, wavefront = (row half wid >> 1, feature block wid & 1); A operand =
  // transposed W0 (k = hidden feature, value W0[k][f]) of input feature f = 32 (wid & 1) + l31
  const int rb = wid >> 1, fcol = (wid & 1) * 32 + l31;
  int ew0;
  {
    float4 v[16];
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float* p = params + RGF_W0 + (size_t)(16 * ks + 8 * lh) * 64 + fcol;
      v[2 * ks] = make_float4(p[0], p[64], p[128], p[192]);
      v[2 * ks + 1] = make_float4(p[256], p[320], p[384], p[448]);
      mx = fmaxf(mx, fmaxf(amax4(v[2 * ks]), amax4(v[2 * ks + 1])));
    }
    ew0 = scale_exp(block_max(mx, red4, tid, phase));
    const float s = pow2(ew0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) split8(v[2 * ks], v[2 * ks + 1], s, w0.h[ks], w0.l[ks]);
  }
  // output layer backward as a K = 16 product: A = [w3_0 w3_1 w3_2 0 ...] of feature `col` (lanes lh = 0), B = d loss / d logits
  const float w3a = params[RGF_W3 + col], w3b = params[RGF_W3 + 128 + col], w3c = params[RGF_W3 + 256 + col];
  const float w3l1 = block_max((fabsf(w3a) + fabsf(w3b)) + fabsf(w3c), red4, tid, phase) * 1.0001f;
  const int ew3 = scale_exp(block_max(fmaxf(fmaxf(fabsf(w3a), fabsf(w3b)), fabsf(w3c)), red4, tid, phase));
  pp_half8 a3h, a3l;
  {
    const float z = 0.f;
    const float4 wa = lh == 0 ? make_float4(w3a, w3b, w3c, z) : make_float4(z, z, z, z);
    split8(wa, make_float4(z, z, z, z), pow2(ew3), a3h, a3l);
  }
  const int j0 = tid & 127, h0 = tid >> 7;
  float wacc3[3] = {0.f, 0.f, 0.f}, bacc3 = 0.f;

  auto stage_tile = [&](const float* __restrict__ X, int r0, float* dst) {       // [64][128] fp32, rows clamped to R - 1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = 16 * wid + 2 * i;
      const int row = min(r0 + rl + lh, R - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(dst + rl * 128), 16, 0, 0);
    }
  };
  auto stage = [&](int t, int b) {              // H2 rows, rgb and rgb_grad of tile t
    const int r0 = t * TILE_ROWS;
    stage_tile(H2, r0, XS);
    if (wid < 3) {
      const int e = min(r0 * 3 + tid, R * 3 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(rgb + e), PP_LDS_PTR(&RG[b][0][wid * 64]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(rgb_grad + e), PP_LDS_PTR(&RG[b][1][wid * 64]), 4, 0, 0);
    }
  };
  // d loss / d logit = rgb_grad * rgb * (1 - rgb); zero for rows past the end; its largest magnitude into the tile's slot
  auto logit_grads = [&](int t, int b) {
    const int r0 = t * TILE_ROWS;
    float gl = 0.f;
    if (tid < 192) {
      const int m = tid / 3, o = tid - 3 * m;
      const float r = RG[b][0][tid];
      gl = (r0 + m < R) ? RG[b][1][tid] * r * (1.f - r) : 0.f;
      GL[m * 4 + o] = gl;
      if (logit_grad && r0 + m < R) logit_grad[(size_t)(r0 + m) * lg_ld + o] = gl;
    } else {
      GL[(tid - 192) * 4 + 3] = 0.f;
    }
    slot_max(&Mx[b][4], fabsf(gl), lane);
  };
  // W3bar / b3bar of the staged tile (thread = feature j0, rows of half h0)
  auto w3_accumulate = [&]() {
    const float* __restrict__ xs = &XS[(h0 * 32) * 128 + j0];
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float4 g = *reinterpret_cast<const float4*>(&GL[(h0 * 32 + r) * 4]);
      const float x = xs[r * 128];
      wacc3[0] = fmaf(g.x, x, wacc3[0]); wacc3[1] = fmaf(g.y, x, wacc3[1]); wacc3[2] = fmaf(g.z, x, wacc3[2]);
      if (j0 < 3) bacc3 += GL[(h0 * 32 + r) * 4 + j0];
    }
  };
  // Ybar2 of the staged tile (parity b) into LDS tile `dst` and `ybar`: both halves; sets the exponent of the image
  int e2n = 0;
  auto out_layer_bwd = [&](int t, int b, _Float16* dst) {
    const float gmx = slot_get(&Mx[b][4]);
    const int eg = scale_exp(gmx);
    const int e2 = scale_exp(gmx * w3l1);
    const float sg = pow2(eg), s2 = pow2(e2), inv = pow2(-(eg + ew3)), n2 = pow2(-e2);
    const int r0 = t * TILE_ROWS;
    float* crow = ybar + (size_t)(r0 + l31) * 128 + fb;
    HalfEpilogueB<true> eb;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float4 x = *reinterpret_cast<const float4*>(&GL[(hh * 32 + l31) * 4]);
      if (lh != 0) x = make_float4(0.f, 0.f, 0.f, 0.f);
      pp_half8 xh, xl;
      split8(x, make_float4(0.f, 0.f, 0.f, 0.f), sg, xh, xl);
      f32x16 a;
      zero16(a);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3h, xl, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3l, xh, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3h, xh, a, 0, 0, 0);
      eb.begin(&XS[(hh * 32 + l31) * 128 + fb]);
      eb.all(a, inv, s2, r0 + hh * 32 + l31 < R, crow + hh * 32 * 128, dst + (hh * 32 + l31) * LDH2 + fb);
      slot_max(&Mx[b][hh], eb.vmax(n2), lane);
    }
    e2n = e2;
  };

  if (tid < 16) Mx[tid >> 3][tid & 7] = 0u;
  stage(blockIdx.x, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  logit_grads(blockIdx.x, 0);
  __syncthreads();
  w3_accumulate();
  out_layer_bwd(blockIdx.x, 0, At[0]);
  __syncthreads();

  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS;
    const int tnext = tile + gridDim.x;
    const bool more = tnext < ntiles;
    const bool ok0 = r0 + l31 < R, ok1 = r0 + 32 + l31 < R;
    _Float16* const Aa = At[par];                // holds Ybar2 of this tile, later Ybar0
    _Float16* const Ab = At[par ^ 1];            // Ybar1, later the next tile's Ybar2
    const int e2 = e2n;
    float* const crow = ybar + (size_t)(r0 + l31) * 128 + fb;
    f32x16 acc0, acc1;
    HalfEpilogueB<true> eb;
    if (tid < 5) Mx[par ^ 1][tid] = 0u;
    if (more) stage(tnext, par ^ 1);             // XS / RG[par^1]: consumed at the bottom of this iteration
    stage_tile(H1, r0, GT);
    // ---- layer 2 on half 0
    zero16(acc0);
    mma_half(Aa, w2, acc0, l31, lh);
    PP_WAIT_VMEM();
    __syncthreads();                             // gates come from all four wavefronts' loads
    // ---- layer 2 on half 1  ||  epilogue of (Ybar1, half 0)
    const int e10 = scale_exp(slot_get(&Mx[par][0]) * l1_2), e11 = scale_exp(slot_get(&Mx[par][1]) * l1_2);
    zero16(acc1);
    eb.begin(&GT[l31 * 128 + fb]);
    mma_half(Aa + 32 * LDH2, w2, acc1, l31, lh, [&](int ks) {
      eb.step(ks, acc0, pow2(-(e2 + ew2)), pow2(e10), ok0, crow + LS, Ab + l31 * LDH2 + fb);
    });
    slot_max(&Mx[par][2], eb.vmax(pow2(-e10)), lane);
    __syncthreads();
    // ---- layer 1 on half 0  ||  epilogue of (Ybar1, half 1)
    zero16(acc0);
    eb.begin(&GT[(32 + l31) * 128 + fb]);
    mma_half(Ab, w1, acc0, l31, lh, [&](int ks) {
      eb.step(ks, acc1, pow2(-(e2 + ew2)), pow2(e11), ok1, crow + LS + 32 * 128, Ab + (32 + l31) * LDH2 + fb);
    });
    slot_max(&Mx[par][3], eb.vmax(pow2(-e11)), lane);
    __syncthreads();                             // GT (H1) is free
    stage_tile(H0, r0, GT);
    // ---- layer 1 on half 1 (the H0 gates land meanwhile)
    zero16(acc1);
    mma_half(Ab + 32 * LDH2, w1, acc1, l31, lh);
    PP_WAIT_VMEM();
    __syncthreads();
    // ---- epilogues of Ybar0 (image into tile Aa: the operand of the feature-gradient product)
    const int e00 = scale_exp(slot_get(&Mx[par][2]) * l1_1), e01 = scale_exp(slot_get(&Mx[par][3]) * l1_1);
    eb.begin(&GT[l31 * 128 + fb]);
    eb.all(acc0, pow2(-(e10 + ew1)), pow2(e00), ok0, crow + 2 * LS, Aa + l31 * LDH2 + fb);
    eb.begin(&GT[(32 + l31) * 128 + fb]);
    eb.all(acc1, pow2(-(e11 + ew1)), pow2(e01), ok1, crow + 2 * LS + 32 * 128, Aa + (32 + l31) * LDH2 + fb);
    if (more) logit_grads(tnext, par ^ 1);
    __syncthreads();
    // ---- layer 0: feat_grad = Ybar0 . W0   (32 rows x 32 features per wavefront)
    {
      f32x16 fa;
      zero16(fa);
      mma_half(Aa + rb * 32 * LDH2, w0, fa, l31, lh);
      const float inv = pow2(-((rb ? e01 : e00) + ew0));
      const int row = r0 + rb * 32 + l31;
      if (row < R) {
        float* __restrict__ ft = feat_grad + (size_t)row * 64 + (wid & 1) * 32 + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(ft + 8 * q) = make_float4(fa[4 * q] * inv, fa[4 * q + 1] * inv, fa[4 * q + 2] * inv, fa[4 * q + 3] * inv);
      }
    }
    // ---- the next tile's output-layer backward into tile Ab (last read by the layer-1 MFMAs)
    if (more) {
      w3_accumulate();
      out_layer_bwd(tnext, par ^ 1, Ab);
    }
    __syncthreads();
  }

  // ---- flush W3bar, b3bar (thread = feature / output)
  float* red = reinterpret_cast<float*>(At[0]);
  __syncthreads();
  if (h0 == 1) {
#pragma unroll
    for (int o = 0; o < 3; ++o) red[o * 128 + j0] = wacc3[o];
    if (j0 < 3) red[4 * 128 + j0] = bacc3;
  }
  __syncthreads();
  if (h0 == 0) {
#pragma unroll
    for (int o = 0; o < 3; ++o) atomicAdd(&params_grad[RGF_W3 + o * 128 + j0], wacc3[o] + red[o * 128 + j0]);
    if (j0 < 3) atomicAdd(&params_grad[RGF_B3 + j0], bacc3 + red[4 * 128 + j0]);
  }
}

int pp_launch_rgb_fused_bwd_s(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                              const int32_t* count, int capacity, float* ybar, float* params_grad, float* feat_grad,
                              float* logit_grad, int lg_ld, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, TILE_ROWS);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_rgb_fused_bwd_s, dim3(grid), dim3(256), 0, st, params, acts, rgb, rgb_grad, count, capacity, ybar,
                     params_grad, feat_grad, logit_grad, lg_ld);
  return 0;
}

// ================================================================================================ weight gradients
// Wbar_l[n][k] += sum_r Ybar_l[r][n] * X_l[r][k] for three layers in ONE launch, as k_wgrad_chain (pp_mlp_fused.hip: a step = one
// 64-row tile of one layer, both operand tiles by LDS-direct loads into a double buffer, the accumulators resident over the
// work-group's whole row range) - with the product on the fp16 MFMAs, three per fp32 product.
// The reduction runs over ROWS, so an operand fragment is 8 consecutive rows of one column: a lane gathers them from the row-major
// fp32 tile (conflict-free: consecutive lanes read consecutive columns; two rows per ds_read2st64_b32), scales, splits, and feeds
// them to the MFMAs - 20 vector instructions per fragment, issued between the MFMAs of the previous 16-row group.
// Scale: one power of two per operand for a wavefront's whole row range (its accumulators see every tile), kept as a RUNNING
// exponent: the magnitudes of every 16-row group's fragments are checked against what the running scale can hold before they are
// converted (v_max3 + one ballot); a group that exceeds it lowers the exponent (with a factor 4 of headroom) and the accumulators
// are rescaled by the exact power of two - a handful of times per launch.  Rows far below the running maximum are converted with
// its absolute floor (2^-40 of it), which is what a sum needs.
namespace {

// piece i of 8 of the LDS-direct loads of one tile pair (Y: two 512-byte rows; X: two rows, or four 256-byte rows on the first
// four pieces when KX == 64)
template <int KX>
__device__ __forceinline__ void wgs_issue_piece(const WgradOperands& L, int r0, int R, float* Ybuf, float* Xbuf, int wid, int lane, int i) {
  const int l31 = lane & 31, lh = lane >> 5;
  {
    const int rl = 16 * wid + 2 * i;
    const int row = min(r0 + rl + lh, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.Y + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(Ybuf + rl * 128), 16, 0, 0);
  }
  if (KX == 128) {
    const int rl = 16 * wid + 2 * i;
    const int row = min(r0 + rl + lh, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.X + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(Xbuf + rl * 128), 16, 0, 0);
  } else if (i < 4) {
    const int l15 = lane & 15, lq = lane >> 4;
    const int rl = 16 * wid + 4 * i;
    const int row = min(r0 + rl + lq, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.X + (size_t)row * 64 + l15 * 4), PP_LDS_PTR(Xbuf + rl * 64), 16, 0, 0);
  }
}
template <int KX>
__device__ __forceinline__ void wgs_issue(const WgradOperands& L, int r0, int R, float* Ybuf, float* Xbuf, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) wgs_issue_piece<KX>(L, r0, R, Ybuf, Xbuf, wid, lane, i);
}

// eight consecutive rows of one column, scaled and split (v: the raw values, kept for the bias sums)
__device__ __forceinline__ void frag_split(const float (&v)[8], float s, pp_half8& h, pp_half8& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x0 = v[2 * k] * s, x1 = v[2 * k + 1] * s;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh[k]) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ll[k]) : "v"(hh[k]), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ll[k]) : "v"(hh[k]), "v"(x1));
  }
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 hv = {hh[0], hh[1], hh[2], hh[3]}, lv = {ll[0], ll[1], ll[2], ll[3]};
  h = __builtin_bit_cast(pp_half8, hv);
  l = __builtin_bit_cast(pp_half8, lv);
}

// state of one wavefront: running scale exponents of its Y / X fragments (a wavefront's accumulators only ever see its own
// fragments, so the scales are per wavefront and need no work-group-wide maximum), bias partial sums
struct WgsLayer {
  int eY, eX;
  float bsum[2];
};

// largest magnitude of the fragments against what the running scale can hold; true if any lane exceeds it
template <int NF>
__device__ __forceinline__ bool frag_exceeds(const float (&v)[NF][8], float lim, float& m) {
  m = 0.f;
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v[f][2 * k]), "v"(v[f][2 * k + 1]));
  return __builtin_amdgcn_ballot_w64(m > lim) != 0ull;
}

// 64 rows x (64 x 32 NB) outputs of one wavefront: acc[t][u] += Y[:, 64 wr + 32 t + .]^T X[:, 32 NB wc + 32 u + .]; issue(i), i = 0..7,
// is called between the MFMA groups of the first two 16-row groups (the next tile's loads ride there).
// Group -1 only converts.  Cold path: a fragment that exceeds the running scale lowers the exponent, with a factor 4 of headroom,
// and rescales the accumulators by the exact power of two.  (A rolled loop over the groups would hold one copy of it instead of
// ten, but measures 2.2 x slower: the copies of the fragments between iterations and the uniform branches defeat the scheduling.)
template <int KX, int CSTEP, class Issue>
__device__ __forceinline__ void wgs_compute(const float* __restrict__ Ybuf, const float* __restrict__ Xbuf, f32x16 (&acc)[2][KX / 64],
                                            WgsLayer& st, bool bias, int wr, int wc, int l31, int lh, Issue issue) {
  constexpr int NB = KX / 64;
  float sY = pow2(st.eY), sX = pow2(st.eX), limY = pow2(15 - st.eY), limX = pow2(15 - st.eX);
  const float* yp = Ybuf + (8 * lh) * 128 + 64 * wr + l31;
  const float* xp = Xbuf + (8 * lh) * KX + 32 * NB * wc + l31;
  float ry[2][8], rx[NB][8];
  pp_half8 ah[2] = {}, al[2] = {}, bh[NB] = {}, bl[NB] = {};
  auto rescale = [&](int d) {
    const float f = pow2(d);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][u][i] *= f;
  };
#pragma unroll
  for (int ks = -1; ks < 4; ++ks) {
    pp_half8 ch[2] = {ah[0], ah[1]}, cl[2] = {al[0], al[1]}, dh[NB], dl[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) { dh[u] = bh[u]; dl[u] = bl[u]; }
    __builtin_amdgcn_sched_barrier(0);
    if (ks < 3) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) ry[t][j] = yp[(16 * (ks + 1) + j) * 128 + 32 * t];
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[u][j] = xp[(16 * (ks + 1) + j) * KX + 32 * u];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks >= 0) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[0], dh[u], acc[0][u], 0, 0, 0);
        acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[0], dl[u], acc[0][u], 0, 0, 0);
        acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[0], dh[u], acc[0][u], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks >= 0 && ks < 2) { issue(4 * ks); issue(4 * ks + 1); issue(4 * ks + 2); issue(4 * ks + 3); }
    __builtin_amdgcn_sched_barrier(0);
    if (ks >= 0) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[1], dh[u], acc[1][u], 0, 0, 0);
        acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[1], dl[u], acc[1][u], 0, 0, 0);
        acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[1], dh[u], acc[1][u], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next group's fragments, converted while the MFMAs above run.  The scale checks come AFTER both MFMA groups of this
    // 16-row group: a rescale between them would add the second group's products, still at the old scale, to rescaled sums.
    if (ks < 3) {
      float m;
      if (frag_exceeds<2>(ry, limY, m)) {                        // cold
        const int e = scale_exp(wave_max(m)) - 2;
        rescale(e - st.eY);
        st.eY = e; sY = pow2(e); limY = pow2(15 - e);
      }
      if (frag_exceeds<NB>(rx, limX, m)) {                       // cold
        const int e = scale_exp(wave_max(m)) - 2;
        rescale(e - st.eX);
        st.eX = e; sX = pow2(e); limX = pow2(15 - e);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (bias) {               // column sums of Y over the rows that count (every row / the primal rows of the 4-row form)
#pragma unroll
          for (int j = 0; j < 8; j += CSTEP) st.bsum[t] += ry[t][j];
        }
        frag_split(ry[t], sY, ah[t], al[t]);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) frag_split(rx[u], sX, bh[u], bl[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

// One work-group works on ONE layer (blockIdx.y) - the three layers share nothing, and 64 resident accumulators instead of 192
// leave the registers the conversion needs; the persistent work-groups of a layer walk its tiles from the end of the row range
// (what the data-gradient kernel wrote last is what the Infinity Cache still holds).
template <int KX, int CSTEP>
__device__ __forceinline__ void wgs_layer(const WgradOperands& L, int R, int ntiles, int wg, int nwg, float* __restrict__ Yb0,
                                          float* __restrict__ Xb0, float* __restrict__ Yb1, float* __restrict__ Xb1) {
  constexpr int NB = KX / 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][NB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NB; ++u) zero16(acc[t][u]);
  WgsLayer st = {60, 60, {0.f, 0.f}};
  const bool bias = L.bbar != nullptr && wc == 0;
  // after the loads of a tile have landed: zero the Y rows past R (last tile only)
  auto prepare = [&](int r0, float* Ybuf) {
    if (r0 + TILE_ROWS > R) {
      for (int i = tid; i < TILE_ROWS * 128; i += 256)
        if (r0 + (i >> 7) >= R) Ybuf[i] = 0.f;
      __syncthreads();
    }
  };
  const int first = ntiles - 1 - wg;
  wgs_issue<KX>(L, first * TILE_ROWS, R, Yb0, Xb0, wid, lane);
  // two tiles per iteration keep the buffer assignment static
#ifdef MS_TIMERS
  unsigned long long tsum[16] = {0}, tprev = __builtin_readcyclecounter();
#endif
  for (int t0 = first; t0 >= 0; t0 -= 2 * nwg) {
    const int t1 = t0 - nwg, t2 = t1 - nwg;
    TICK(4);
    PP_WAIT_VMEM(); __syncthreads();
    TICK(0);
    prepare(t0 * TILE_ROWS, Yb0);
    TICK(1);
    TICK(2);
    wgs_compute<KX, CSTEP>(Yb0, Xb0, acc, st, bias, wr, wc, l31, lh, [&](int i) {
      if (t1 >= 0) wgs_issue_piece<KX>(L, t1 * TILE_ROWS, R, Yb1, Xb1, wid, lane, i);
    });
    TICK(3);
    if (t1 < 0) break;
    PP_WAIT_VMEM(); __syncthreads();
    TICK(0);
    prepare(t1 * TILE_ROWS, Yb1);
    TICK(1);
    TICK(2);
    wgs_compute<KX, CSTEP>(Yb1, Xb1, acc, st, bias, wr, wc, l31, lh, [&](int i) {
      if (t2 >= 0) wgs_issue_piece<KX>(L, t2 * TILE_ROWS, R, Yb0, Xb0, wid, lane, i);
    });
    TICK(3);
  }
#ifdef MS_TIMERS
  if (tid == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ms_t[i], tsum[i]);
#endif
  // flush: one atomic per entry, scaled back
  const float f = pow2(-(st.eY + st.eX));
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int k = 32 * NB * wc + u * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        atomicAdd(&L.Wbar[(size_t)n * KX + k], acc[t][u][reg] * f);
      }
    }
  if (bias) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v = st.bsum[t] + __shfl_xor(st.bsum[t], 32, 64);
      if (lh == 0) atomicAdd(&L.bbar[64 * wr + 32 * t + l31], v);
    }
  }
}

template <int KXC, int CSTEP>
__global__ __launch_bounds__(256) void k_wgrad_chain_s(WgradOperands LA, WgradOperands LB, WgradOperands LC,
                                                       const int32_t* __restrict__ count, int rmul, int rcap, int nwg_ab, int nwg_c) {
  __shared__ __attribute__((aligned(16))) float Yb0[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Xb0[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Yb1[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Xb1[TILE_ROWS * 128];
  const int R = min(count[0] * rmul, rcap);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  // work-groups 0 .. nwg_ab - 1: layer A, the next nwg_ab: layer B, the last nwg_c: layer C (fewer when its operand is narrower)
  const int bx = blockIdx.x;
  const int layer = bx < nwg_ab ? 0 : (bx < 2 * nwg_ab ? 1 : 2);
  const int wg = bx - layer * nwg_ab, nwg = layer < 2 ? nwg_ab : nwg_c;
  if (wg >= ntiles) return;
  // one copy of the layer code per operand width (three calls would triple the instruction footprint)
  const WgradOperands L = layer == 0 ? LA : (layer == 1 ? LB : LC);
  if (KXC == 128 || layer < 2) wgs_layer<128, CSTEP>(L, R, ntiles, wg, nwg, Yb0, Xb0, Yb1, Xb1);
  else wgs_layer<KXC, CSTEP>(L, R, ntiles, wg, nwg, Yb0, Xb0, Yb1, Xb1);
}

int pp_launch_wgrad_chain_s(const float* YA, const float* XA, float* WA, const float* YB, const float* XB, float* WB,
                            const float* YC, const float* XC, float* WC, int kxc, const int32_t* count, int rmul, int rcap,
                            hipStream_t st, float* bA, float* bB, float* bC, int wgs_) {
  WgradOperands LA{YA, XA, WA, bA}, LB{YB, XB, WB, bB}, LC{YC, XC, WC, bC};
  const int ntiles = pp_div_up(rcap, TILE_ROWS);
  // persistent work-groups, at most one per CU over the three layers, shared out in proportion to the layers' work (wgs_ > 0: the
  // caller's number - a launch on an auxiliary stream that leaves CUs to the kernels running beside it)
  const int wgs = wgs_ > 0 ? (wgs_ < 16 ? 16 : wgs_) : PP_FUSED_WGS;
  // (a 64-wide layer costs 3/4 of a 128-wide one: three instead of four operand fragments to convert per 16-row group - the
  // conversion, not the MFMA count, is what the time follows)
  int nab = kxc == 128 ? wgs / 3 : (wgs * 4) / 11, nc = kxc == 128 ? wgs / 3 : wgs - 2 * ((wgs * 4) / 11);
  if (nab > ntiles) nab = ntiles;
  if (nc > ntiles) nc = ntiles;
  if (kxc == 128)
    hipLaunchKernelGGL((k_wgrad_chain_s<128, 4>), dim3(2 * nab + nc), dim3(256), 0, st, LA, LB, LC, count, rmul, rcap, nab, nc);
  else
    hipLaunchKernelGGL((k_wgrad_chain_s<64, 1>), dim3(2 * nab + nc), dim3(256), 0, st, LA, LB, LC, count, rmul, rcap, nab, nc);
  return 0;
}
