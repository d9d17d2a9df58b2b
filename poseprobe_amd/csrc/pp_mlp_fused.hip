// Layer-fused evaluation of the two shallow MLPs: one persistent kernel walks a 64-row tile through ALL layers.
//
// Why: a 128x128 fp32 layer over R rows moves 2 x R x 512 B through HBM for 2 x R x 128 x 128 FLOP = 32 FLOP/B, which
// is exactly the ridge of the machine (157 TFLOP/s fp32 MFMA vs ~5.5 TB/s attainable) - the layer-by-layer GEMMs can
// never be better than "half memory-, half matrix-bound".  Fused, the activations of a tile stay in LDS between the
// layers (HBM only sees the copies the backward pass needs, write-only) and the weights are *stationary in
// registers*: a wavefront owns 32 output features of every layer, i.e. 32 x 128 weights = 64 VGPRs per layer in the
// MFMA B-operand layout, loaded once per work-group.  The inner loop is then ds_read_b128 (A operand, shared by the
// four wavefronts) + v_mfma_f32_32x32x2_f32 only.
//
// Operand layouts (v_mfma_f32_32x32x2_f32, see pp_mlp.hip): lane = (l31, lh).  A: row l31, B: feature l31, both
// supply k = 8g + 4lh + j to the j-th MFMA of group g ("K-slot permutation": one 16-byte read feeds 4 MFMAs).
// D: feature l31, rows (reg&3) + 8(reg>>2) + 4lh -> the 4 rows of a warp sample sit in 4 registers of one lane.
#include "pp_common.h"
#include "pp_mlp_fused.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDA 132                 // LDS row stride of an activation tile (floats): 16-B aligned, 4-bank skew per row
#define TILE_ROWS 64

namespace {

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
}

// forward weights of one layer for feature n:  w[g] = W[n][8g + 4lh .. +3]
template <int K>
__device__ __forceinline__ void load_w_rows(float4 (&w)[K / 8], const float* __restrict__ W, int ldw, int n, int lh) {
#pragma unroll
  for (int g = 0; g < K / 8; ++g) w[g] = *reinterpret_cast<const float4*>(W + (size_t)n * ldw + 8 * g + 4 * lh);
}

// backward-data weights (B(k = n, j) = W[n][j]) for input feature j:  w[g].{x..w} = W[8g + 4lh + {0..3}][j]
template <int K>
__device__ __forceinline__ void load_w_cols(float4 (&w)[K / 8], const float* __restrict__ W, int ldw, int j, int lh) {
#pragma unroll
  for (int g = 0; g < K / 8; ++g) {
    const float* p = W + (size_t)(8 * g + 4 * lh) * ldw + j;
    w[g] = make_float4(p[0], p[ldw], p[2 * ldw], p[3 * ldw]);
  }
}

// acc[t] += A[t*32 + l31][:] . w   for a 64-row tile in LDS
template <int K>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float4 (&w)[K / 8], f32x16 (&acc)[2],
                                         int l31, int lh) {
  const float* a0p = As + l31 * LDA + 4 * lh;
  const float* a1p = a0p + 32 * LDA;
  // operands of group g+1 are fetched before the MFMAs of group g (LDS latency hides behind 8 x 16 matrix passes)
  float4 a0 = *reinterpret_cast<const float4*>(a0p);
  float4 a1 = *reinterpret_cast<const float4*>(a1p);
#pragma unroll
  for (int g = 0; g < K / 8; ++g) {
    float4 n0 = a0, n1 = a1;
    if (g + 1 < K / 8) {
      n0 = *reinterpret_cast<const float4*>(a0p + 8 * (g + 1));
      n1 = *reinterpret_cast<const float4*>(a1p + 8 * (g + 1));
    }
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w[g].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w[g].x, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w[g].y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w[g].y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w[g].z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w[g].z, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w[g].w, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w[g].w, acc[1], 0, 0, 0);
    a0 = n0; a1 = n1;
  }
}

// ReLU epilogue of a hidden layer: bias (+ 4-row masking), copy for the backward pass to HBM, next A tile to LDS.
template <int COLS, bool FULL>
__device__ __forceinline__ void relu_epilogue_impl(const f32x16 (&acc)[2], float bcol, int r0, int R, int col, int lh,
                                                   float* __restrict__ C, float* __restrict__ Anext) {
  // wave-uniform tile base + 32-bit lane offset (+ compile-time row offsets): no 64-bit address arithmetic per store
  float* __restrict__ Ct = C + (size_t)r0 * 128;
  const unsigned lane_off = (unsigned)(4 * lh) * 128u + (unsigned)col;
  float* __restrict__ At = Anext + (4 * lh) * LDA + col;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
      if (COLS == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[t][4 * q + c] + bcol, 0.f);
      } else {
        const float y0 = acc[t][4 * q] + bcol;             // primal row of this sample (same lane)
        const bool on = y0 > 0.f;
        v[0] = on ? y0 : 0.f;
#pragma unroll
        for (int c = 1; c < 4; ++c) v[c] = on ? acc[t][4 * q + c] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int row = t * 32 + 8 * q + c;                // + 4*lh (lane part)
        At[row * LDA] = v[c];
        if (FULL || r0 + row + 4 * lh < R) Ct[lane_off + (unsigned)(row * 128)] = v[c];
      }
    }
  }
}

template <int COLS>
__device__ __forceinline__ void relu_epilogue(const f32x16 (&acc)[2], float bcol, int r0, int R, int col, int lh,
                                              float* __restrict__ C, float* __restrict__ Anext) {
  if (r0 + TILE_ROWS <= R) relu_epilogue_impl<COLS, true>(acc, bcol, r0, R, col, lh, C, Anext);
  else relu_epilogue_impl<COLS, false>(acc, bcol, r0, R, col, lh, C, Anext);
}

// Backward-data epilogue: gradient w.r.t. the input activations of a layer = (Ybar W) masked by that activation's own
// ReLU state (mk > 0, one value per sample = 4 rows); next A tile to LDS, optional copy to HBM for the weight-gradient GEMM.
template <bool FULL, bool TOGLOBAL>
__device__ __forceinline__ void mask_epilogue_impl(const f32x16 (&acc)[2], const float (&mk)[2][4], int r0, int R, int col,
                                                   int lh, float* __restrict__ C, float* __restrict__ Anext) {
  float* __restrict__ Ct = C + (size_t)r0 * 128;
  const unsigned lane_off = (unsigned)(4 * lh) * 128u + (unsigned)col;
  float* __restrict__ At = Anext + (4 * lh) * LDA + col;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool on = mk[t][q] > 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int row = t * 32 + 8 * q + c;
        const float v = on ? acc[t][4 * q + c] : 0.f;
        At[row * LDA] = v;
        if (TOGLOBAL && (FULL || r0 + row + 4 * lh < R)) Ct[lane_off + (unsigned)(row * 128)] = v;
      }
    }
  }
}

template <bool TOGLOBAL>
__device__ __forceinline__ void mask_epilogue(const f32x16 (&acc)[2], const float (&mk)[2][4], int r0, int R, int col, int lh,
                                              float* __restrict__ C, float* __restrict__ Anext) {
  if (r0 + TILE_ROWS <= R) mask_epilogue_impl<true, TOGLOBAL>(acc, mk, r0, R, col, lh, C, Anext);
  else mask_epilogue_impl<false, TOGLOBAL>(acc, mk, r0, R, col, lh, C, Anext);
}

// primal-row activations (one per sample) that gate the 8 samples a lane owns in the accumulator layout
template <bool FULL>
__device__ __forceinline__ void load_masks_impl(float (&mk)[2][4], const float* __restrict__ X, int r0, int R, int col, int lh) {
  const float* __restrict__ Xt = X + (size_t)r0 * 128;
  const unsigned lane_off = (unsigned)(4 * lh) * 128u + (unsigned)col;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = t * 32 + 8 * q;
      mk[t][q] = (FULL || r0 + row + 4 * lh < R) ? Xt[lane_off + (unsigned)(row * 128)] : 0.f;
    }
}
__device__ __forceinline__ void load_masks(float (&mk)[2][4], const float* __restrict__ X, int r0, int R, int col, int lh) {
  if (r0 + TILE_ROWS <= R) load_masks_impl<true>(mk, X, r0, R, col, lh);
  else load_masks_impl<false>(mk, X, r0, R, col, lh);
}

template <bool B> struct BoolC { static constexpr bool value = B; };
// run `f` with a compile-time copy of a wave-uniform condition (fast path without per-element guards)
#define PP_WITH_FULL(cond, f) do { if (cond) f(BoolC<true>{}); else f(BoolC<false>{}); } while (0)

}  // namespace

// ------------------------------------------------------------------------------------------------ warp net, forward
// pts[M][3] -> (value, Jacobian) out[M][4][4]; writes the four hidden activations X0..X3 ([4M][128] each) for backward.
__global__ __launch_bounds__(256) void k_warp_fused_fwd(const float* __restrict__ params, const float* __restrict__ pts,
                                                        const int32_t* __restrict__ count, int capacity,
                                                        float out_range, float* __restrict__ acts,
                                                        float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float As[2][TILE_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float W4s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float Red[4 * 64 * 4];
  const int M = min(count[0], capacity);
  const int R = 4 * M;
  const int ntiles = (M + 15) >> 4;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;                    // output feature of this lane in every hidden layer
  const size_t LS = (size_t)capacity * 4 * 128;

  float4 w1[16], w2[16], w3[16];
  load_w_rows<128>(w1, params + WPF_W1, 128, col, lh);
  load_w_rows<128>(w2, params + WPF_W2, 128, col, lh);
  load_w_rows<128>(w3, params + WPF_W3, 128, col, lh);
  const float b1 = params[WPF_B1 + col], b2 = params[WPF_B2 + col], b3 = params[WPF_B3 + col];
  // layer 0 / layer 4 are evaluated on the vector ALUs: thread = (feature j, half h) resp. (row = lane, output o = wave)
  const int j0 = tid & 127, h0 = tid >> 7;
  const float w0x = params[WPF_W0 + j0 * 3], w0y = params[WPF_W0 + j0 * 3 + 1], w0z = params[WPF_W0 + j0 * 3 + 2];
  const float b0 = params[WPF_B0 + j0];
  for (int i = tid; i < 512; i += 256) W4s[(i >> 7) * LDA + (i & 127)] = params[WPF_W4 + i];
  const float b4 = (((tid >> 2) & 3) == 0) ? params[WPF_B4 + (tid & 3)] : 0.f;   // bias on the primal row only
  // The 16 sample positions of a tile are fetched one tile ahead and parked in LDS.  On gfx9 a wait for a vector load
  // is a vmcnt(0), i.e. it also drains every activation store in flight - so the load is issued at the top of a tile
  // and consumed right after the first MFMA block, when the stores before it have long retired.
  __shared__ float Ps[48];
  float pnext = 0.f;
  if (tid < 48 && blockIdx.x * 48 + tid < M * 3) pnext = pts[blockIdx.x * 48 + tid];
  if (tid < 48) Ps[tid] = pnext;
  // every prologue load (weights!) has landed before the loop: otherwise the first in-loop use of a loop-invariant
  // register carries a conservative vmcnt(0) that drains the activation stores on every iteration
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s0 = tile * 16, r0 = tile * TILE_ROWS;
    {
      const int nt = tile + gridDim.x;
      pnext = 0.f;
      if (tid < 48 && nt < ntiles && nt * 48 + tid < M * 3) pnext = pts[nt * 48 + tid];
    }
    // ---- layer 0 (3 -> 128), 4-row form: row 0 = relu(W0 p + b0), rows 1..3 = mask * W0[:, i]
    auto layer0 = [&](auto fc) {
      constexpr bool FULL = decltype(fc)::value;
      float* __restrict__ gt = acts + (size_t)s0 * 512;    // wave-uniform tile base
      const unsigned goff = (unsigned)(h0 * 8) * 512u + (unsigned)j0;
      float* __restrict__ at = &As[0][(4 * h0 * 8) * LDA + j0];
      const float* ps = &Ps[h0 * 24];
      float pp[24];
#pragma unroll
      for (int i = 0; i < 24; ++i) pp[i] = ps[i];           // wave-uniform (broadcast) reads, all in flight at once
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float y = pp[q * 3] * w0x + pp[q * 3 + 1] * w0y + pp[q * 3 + 2] * w0z + b0;
        const bool valid = FULL || (s0 + h0 * 8 + q < M);
        const bool on = (y > 0.f) && valid;
        const float x0 = on ? y : 0.f, x1 = on ? w0x : 0.f, x2 = on ? w0y : 0.f, x3 = on ? w0z : 0.f;
        at[(4 * q) * LDA] = x0; at[(4 * q + 1) * LDA] = x1; at[(4 * q + 2) * LDA] = x2; at[(4 * q + 3) * LDA] = x3;
        if (valid) {
          gt[goff + q * 512] = x0; gt[goff + q * 512 + 128] = x1; gt[goff + q * 512 + 256] = x2; gt[goff + q * 512 + 384] = x3;
        }
      }
    };
    PP_WITH_FULL(s0 + 16 <= M, layer0);
    __syncthreads();
    f32x16 acc[2];
    // ---- hidden layers 1..3 on the matrix cores (ping-pong LDS tiles)
    zero_acc(acc);
    mma_tile<128>(As[0], w1, acc, l31, lh);
    if (tid < 48) Ps[tid] = pnext;                        // next tile's positions (read after >= 3 barriers)
    relu_epilogue<4>(acc, b1, r0, R, col, lh, acts + LS, As[1]);
    __syncthreads();
    zero_acc(acc);
    mma_tile<128>(As[1], w2, acc, l31, lh);
    relu_epilogue<4>(acc, b2, r0, R, col, lh, acts + 2 * LS, As[0]);
    __syncthreads();
    zero_acc(acc);
    mma_tile<128>(As[0], w3, acc, l31, lh);
    relu_epilogue<4>(acc, b3, r0, R, col, lh, acts + 3 * LS, As[1]);
    __syncthreads();
    // ---- output layer (128 -> 4) on v_mfma_f32_4x4x1 (16 blocks of 4 rows x 4 outputs per instruction): lane = row
    // for A, lane&3 = output for B; every wavefront contracts a 32-wide K slice, the four partial results meet in LDS.
    {
      const float* xr = &As[1][lane * LDA + 32 * wid];
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
      // lane (block b = lane>>2, output o = lane&3) holds rows 4b..4b+3 of its block in d[0..3]
      *reinterpret_cast<float4*>(&Red[(wid * 64 + lane) * 4]) = make_float4(d0[0] + d1[0], d0[1] + d1[1], d0[2] + d1[2], d0[3] + d1[3]);
    }
    __syncthreads();
    {
      // thread = (row = tid>>2, o = tid&3): partial of wave w sits at Red[w][lane' = 4*(row>>2) + o][row&3]
      const int row = tid >> 2, o = tid & 3;
      const int idx = ((row >> 2) * 4 + o) * 4 + (row & 3);
      const float sum = (Red[idx] + Red[256 + idx]) + (Red[512 + idx] + Red[768 + idx]);
      if (r0 + row < R) out[(size_t)r0 * 4 + tid] = (sum + b4) * out_range;
    }
    // the next tile's first barrier orders these reads of As[1] / Ps before they are overwritten
  }
}

int pp_launch_warp_fused_fwd(const float* params, const float* pts, const int32_t* count, int capacity, float out_range,
                             float* acts, float* out, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, 16);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_warp_fused_fwd, dim3(grid), dim3(256), 0, st, params, pts, count, capacity, out_range, acts, out);
  return 0;
}

// ------------------------------------------------------------------------------------------------ warp net, backward
// Data-gradient chain of the warp net in one persistent kernel (weights W3, W2, W1 stationary in the B-operand layout of
// the transposed product), plus everything that is "thin": the output layer's backward (Ybar3, W4bar, b4bar), the input
// layer's backward (pts_grad, W0bar, b0bar).  Ybar3 / Ybar2 / Ybar1 are written to `ybar` ([3][4*cap][128]) for the three
// weight-gradient GEMMs (k_gemm_tn) that follow; Ybar0 never leaves the chip.
//
// Every global read inside the tile loop is an LDS-direct load (global_load_lds): with ~450 live registers per lane a
// register-staged prefetch would be parked in accumulator registers by the compiler (= waited for on the spot), and on
// gfx9 any wait for a vector load is a vmcnt(0) that also drains the activation stores in flight.  LDS-direct loads are
// issued one MFMA block (or one tile) ahead of their single explicit wait, by which time those stores have retired.
#define PP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PP_WAIT_VMEM() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)   /* vmcnt(0) only (gfx9 encoding) */

__global__ __launch_bounds__(256) void k_warp_fused_bwd(const float* __restrict__ params, const float* __restrict__ pts,
                                                        const float* __restrict__ acts,
                                                        const float* __restrict__ out_grad,
                                                        const int32_t* __restrict__ count, int capacity, float out_range,
                                                        float* __restrict__ ybar, float* __restrict__ params_grad,
                                                        float* __restrict__ pts_grad) {
  __shared__ __attribute__((aligned(16))) float As[3][TILE_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float XS[TILE_ROWS * 128];     // X3 rows of the NEXT tile (unpadded, lane-contiguous)
  __shared__ __attribute__((aligned(16))) float MK[8 * 256];             // ReLU gates of the current layer, one slot per lane
  __shared__ __attribute__((aligned(16))) float W0s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float G[2][256];               // raw out_grad of the current / next tile
  __shared__ __attribute__((aligned(16))) float Ps[2][64];               // sample positions of the current / next tile
  __shared__ __attribute__((aligned(16))) float Red[4 * 64];
  const int M = min(count[0], capacity);
  const int R = 4 * M;
  const int ntiles = (M + 15) >> 4;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const size_t LS = (size_t)capacity * 4 * 128;
  const float* __restrict__ X0 = acts;
  const float* __restrict__ X1 = acts + LS;
  const float* __restrict__ X2 = acts + 2 * LS;
  const float* __restrict__ X3 = acts + 3 * LS;

  float4 w3[16], w2[16], w1[16];
  load_w_cols<128>(w3, params + WPF_W3, 128, col, lh);
  load_w_cols<128>(w2, params + WPF_W2, 128, col, lh);
  load_w_cols<128>(w1, params + WPF_W1, 128, col, lh);
  const int j0 = tid & 127, h0 = tid >> 7;
  // out_grad is used raw; the output range factor is folded into W4 here and into the W4 / b4 gradients at the end
  const float w4a = params[WPF_W4 + j0] * out_range, w4b = params[WPF_W4 + 128 + j0] * out_range,
              w4c = params[WPF_W4 + 256 + j0] * out_range, w4d = params[WPF_W4 + 384 + j0] * out_range;
  for (int i = tid; i < 512; i += 256) {
    const int r = i >> 7, j = i & 127;
    W0s[r * LDA + j] = (r < 3) ? params[WPF_W0 + j * 3 + r] : 0.f;
  }
  float wacc4[4] = {0.f, 0.f, 0.f, 0.f}, bacc4 = 0.f, wacc0[3] = {0.f, 0.f, 0.f}, bacc0 = 0.f;

  // ---- LDS-direct staging of tile t into parity slot b: X3 rows -> XS, out_grad -> G[b], positions -> Ps[b].
  // Rows / samples past the end are clamped to the last valid one; their out_grad slot is zeroed after the wait, which
  // zeroes every contribution of those samples.
  auto stage = [&](int t, int b) {
    const int rn0 = t * TILE_ROWS, sn0 = t * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = 16 * wid + 2 * i;                                   // two rows (2 x 512 B) per instruction
      const int row = min(rn0 + rl + lh, R - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X3 + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(&XS[rl * 128]), 16, 0, 0);
    }
    {
      const int e = min(sn0 * 16 + tid, M * 16 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(out_grad + e), PP_LDS_PTR(&G[b][wid * 64]), 4, 0, 0);
    }
    if (wid == 0) {
      const int e = min(sn0 * 3 + lane, M * 3 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(pts + e), PP_LDS_PTR(&Ps[b][0]), 4, 0, 0);
    }
  };
  // gates of the 8 samples a lane owns (primal rows of X): one dword per (t, q) into this lane's private MK slots
  auto stage_masks = [&](const float* __restrict__ X, int r0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = min(r0 + t * 32 + 8 * q + 4 * lh, R - 1);
        __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X + (size_t)row * 128 + col), PP_LDS_PTR(&MK[(t * 4 + q) * 256 + wid * 64]), 4, 0, 0);
      }
  };
  auto read_masks = [&](float (&mk)[2][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) mk[t][q] = MK[(t * 4 + q) * 256 + tid];
  };
  // output layer backward of the staged tile -> As[0] (+ HBM copy), weight / bias partial sums in registers
  auto out_layer_bwd = [&](int t, int b) {
    const int sn0 = t * 16;
    float* __restrict__ yt = ybar + (size_t)sn0 * 512;
    const unsigned off = (unsigned)(h0 * 8) * 512u + (unsigned)j0;
    float* __restrict__ at = &As[0][(4 * h0 * 8) * LDA + j0];
    const float* __restrict__ xs = &XS[(4 * h0 * 8) * 128 + j0];
    auto body = [&](auto fc) {
      constexpr bool FULL = decltype(fc)::value;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool on = xs[(4 * q) * 128] > 0.f;
        const bool ok = FULL || (sn0 + h0 * 8 + q < M);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 g = *reinterpret_cast<const float4*>(&G[b][(h0 * 8 + q) * 16 + c * 4]);
          const float x = xs[(4 * q + c) * 128];
          wacc4[0] += g.x * x; wacc4[1] += g.y * x; wacc4[2] += g.z * x; wacc4[3] += g.w * x;
          const float yb = g.x * w4a + g.y * w4b + g.z * w4c + g.w * w4d;
          const float v = on ? yb : 0.f;
          at[(4 * q + c) * LDA] = v;
          if (ok) yt[off + q * 512 + c * 128] = v;
        }
        if (j0 < 4) bacc4 += G[b][(h0 * 8 + q) * 16 + j0];
      }
    };
    PP_WITH_FULL(sn0 + 16 <= M, body);
  };
  auto zero_invalid_grad = [&](int t, int b) {                           // samples past M contribute nothing
    if (t * 16 + (tid >> 4) >= M) G[b][tid] = 0.f;
  };

  stage(blockIdx.x, 0);
  __builtin_amdgcn_s_waitcnt(0);        // all prologue loads landed (see k_warp_fused_fwd)
  __syncthreads();
  zero_invalid_grad(blockIdx.x, 0);
  __syncthreads();
  out_layer_bwd(blockIdx.x, 0);
  __syncthreads();

  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS, s0 = tile * 16;
    const int tnext = tile + gridDim.x;
    if (tnext < ntiles) stage(tnext, par ^ 1);       // consumed at the bottom of this iteration, 3 MFMA blocks later
    stage_masks(X2, r0);
    float mk[2][4];
    f32x16 acc[2];
    // ---- layer 3: Ybar2 = gate(X2) . (Ybar3 W3)
    zero_acc(acc);
    mma_tile<128>(As[0], w3, acc, l31, lh);
    PP_WAIT_VMEM();
    read_masks(mk);
    mask_epilogue<true>(acc, mk, r0, R, col, lh, ybar + LS, As[1]);
    stage_masks(X1, r0);                             // (each lane overwrites only its own, already consumed, slots)
    __syncthreads();
    // ---- layer 2
    zero_acc(acc);
    mma_tile<128>(As[1], w2, acc, l31, lh);
    PP_WAIT_VMEM();
    read_masks(mk);
    mask_epilogue<true>(acc, mk, r0, R, col, lh, ybar + 2 * LS, As[2]);
    stage_masks(X0, r0);
    __syncthreads();
    // ---- layer 1: Ybar0 stays in LDS
    zero_acc(acc);
    mma_tile<128>(As[2], w1, acc, l31, lh);
    PP_WAIT_VMEM();                                  // also covers stage(tnext): XS / G / Ps of the next tile have landed
    read_masks(mk);
    mask_epilogue<false>(acc, mk, r0, R, col, lh, ybar, As[1]);
    __syncthreads();
    // ---- layer 0: W0bar[j][i] += Ybar0[4s][j] p_i + Ybar0[4s+1+i][j], b0bar[j] += Ybar0[4s][j]  (thread = feature j)
    {
      const float* __restrict__ yb = &As[1][(4 * h0 * 8) * LDA + j0];
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
      const float* xr = &As[1][lane * LDA + 32 * wid];
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
    if (tnext < ntiles) zero_invalid_grad(tnext, par ^ 1);
    __syncthreads();
    if (tid < 64) {
      const int s = tid >> 2, i = tid & 3;
      if (i < 3 && s0 + s < M) {
        const float v = (Red[tid] + Red[64 + tid]) + (Red[128 + tid] + Red[192 + tid]);
        pts_grad[(s0 + s) * 3 + i] += v;
      }
    }
    // ---- the next tile's output-layer backward into As[0] (last read by this tile's first MFMA block)
    if (tnext < ntiles) out_layer_bwd(tnext, par ^ 1);
    __syncthreads();
  }

  // ---- flush the thin-layer weight gradients (one atomic per entry and work-group)
  float* red = &As[0][0];
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

int pp_launch_warp_fused_bwd(const float* params, const float* pts, const float* acts, const float* out_grad,
                             const int32_t* count, int capacity, float out_range, float* ybar, float* params_grad,
                             float* pts_grad, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, 16);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_warp_fused_bwd, dim3(grid), dim3(256), 0, st, params, pts, acts, out_grad, count, capacity, out_range,
                     ybar, params_grad, pts_grad);
  return 0;
}
