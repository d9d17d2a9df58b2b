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
#include "pp_wgrad_asm.inc"
#include "pp_gemm_split.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PP_WAIT_VMEM() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)   /* vmcnt(0) only (gfx9 encoding) */

#define LDA 132                 // LDS row stride of an activation tile (floats): 16-B aligned, 4-bank skew per row
#define TILE_ROWS 64

int pp_fused_wgs() {
  const int opt = pp_opt(PP_OPT_MLP_WGS);          // of the calling entry point's context
  if (opt > 0) return opt < 16 ? 16 : opt;        // the weight-gradient chains share the work-groups out over three layers
  return pp_num_cus();
}

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

// acc[t] += A[t*32 + l31][:] . w   for a 64-row tile in LDS.  `between(g)` is called after the MFMAs of operand group g
// have been issued: work placed there (LDS-direct load issue, address arithmetic) executes in the shadow of the matrix
// pipe instead of in front of it - with one wavefront per SIMD nothing else would hide it.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int K, class Hook = NoHook>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float4 (&w)[K / 8], f32x16 (&acc)[2],
                                         int l31, int lh, Hook between = Hook()) {
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
    between(g);
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
                                                   int lh, float* __restrict__ C, float* __restrict__ Anext, float& bsum) {
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
        if (c == 0) bsum += v;                              // bias gradient: primal rows only (rows past R are zero)
        At[row * LDA] = v;
        if (TOGLOBAL && (FULL || r0 + row + 4 * lh < R)) Ct[lane_off + (unsigned)(row * 128)] = v;
      }
    }
  }
}

template <bool TOGLOBAL>
__device__ __forceinline__ void mask_epilogue(const f32x16 (&acc)[2], const float (&mk)[2][4], int r0, int R, int col, int lh,
                                              float* __restrict__ C, float* __restrict__ Anext, float& bsum) {
  if (r0 + TILE_ROWS <= R) mask_epilogue_impl<true, TOGLOBAL>(acc, mk, r0, R, col, lh, C, Anext, bsum);
  else mask_epilogue_impl<false, TOGLOBAL>(acc, mk, r0, R, col, lh, C, Anext, bsum);
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
  if (tid < 48 && (int)blockIdx.x * 48 + tid < M * 3) pnext = pts[blockIdx.x * 48 + tid];
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

__global__ __launch_bounds__(256) void k_warp_fused_bwd(const float* __restrict__ params, const float* __restrict__ pts,
                                                        const float* __restrict__ acts,
                                                        const float* __restrict__ out_grad,
                                                        const int32_t* __restrict__ count, int capacity, float out_range,
                                                        float* __restrict__ ybar, float* __restrict__ params_grad,
                                                        float* __restrict__ pts_grad) {
  __shared__ __attribute__((aligned(16))) float As[3][TILE_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float XS[TILE_ROWS * 128];     // X3 rows of the NEXT tile (unpadded, lane-contiguous)
  __shared__ __attribute__((aligned(16))) float MK[2][8 * 256];          // ReLU gates of the current / next layer, one slot per lane
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
  float bacc3 = 0.f, bacc2 = 0.f, bacc1 = 0.f, bdummy = 0.f;   // bias gradients of the hidden layers (bacc3: thread = feature)

  // ---- LDS-direct staging of tile t into parity slot b: X3 rows -> XS, out_grad -> G[b], positions -> Ps[b].
  // Rows / samples past the end are clamped to the last valid one; their out_grad slot is zeroed after the wait, which
  // zeroes every contribution of those samples.
  // piece i of 10: i < 8 two X3 rows, 8 = out_grad, 9 = positions (issued one per MFMA operand group, see mma_tile)
  auto stage_piece = [&](int t, int b, int i) {
    const int rn0 = t * TILE_ROWS, sn0 = t * 16;
    if (i < 8) {
      const int rl = 16 * wid + 2 * i;                                   // two rows (2 x 512 B) per instruction
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
  auto stage = [&](int t, int b) {
#pragma unroll
    for (int i = 0; i < 10; ++i) stage_piece(t, b, i);
  };
  // gates of the 8 samples a lane owns (primal rows of X): one dword per (t, q) into this lane's private MK slots
  auto stage_mask_piece = [&](const float* __restrict__ X, int r0, int mb, int i) {      // i = t*4 + q
    const int row = min(r0 + (i >> 2) * 32 + 8 * (i & 3) + 4 * lh, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X + (size_t)row * 128 + col), PP_LDS_PTR(&MK[mb][i * 256 + wid * 64]), 4, 0, 0);
  };
  auto stage_masks = [&](const float* __restrict__ X, int r0, int mb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_mask_piece(X, r0, mb, i);
  };
  auto read_masks = [&](float (&mk)[2][4], int mb) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) mk[t][q] = MK[mb][(t * 4 + q) * 256 + tid];
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
          wacc4[0] = fmaf(g.x, x, wacc4[0]); wacc4[1] = fmaf(g.y, x, wacc4[1]);
          wacc4[2] = fmaf(g.z, x, wacc4[2]); wacc4[3] = fmaf(g.w, x, wacc4[3]);
          const float yb = fmaf(g.w, w4d, fmaf(g.z, w4c, fmaf(g.y, w4b, g.x * w4a)));
          const float v = on ? yb : 0.f;
          if (c == 0) bacc3 += v;
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
  stage_masks(X2, blockIdx.x * TILE_ROWS, 0);
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
    float mk[2][4];
    f32x16 acc[2];
    // All global reads of the loop are LDS-direct loads issued from INSIDE the MFMA loops (one or two per operand group),
    // one block ahead of their use.  Gate buffers alternate: this tile's X2 gates sit in MK[par]; X1 -> MK[par^1] during
    // block 3; X0 -> MK[par] during block 2 (after its X2 content was consumed); the next tile's X2 -> MK[par^1]
    // during block 1.
    // ---- layer 3: Ybar2 = gate(X2) . (Ybar3 W3)
    zero_acc(acc);
    mma_tile<128>(As[0], w3, acc, l31, lh, [&](int g) {
      if (g < 8) stage_mask_piece(X1, r0, par ^ 1, g);
      if (g < 10 && tnext < ntiles) stage_piece(tnext, par ^ 1, g);
    });
    PP_WAIT_VMEM();
    read_masks(mk, par);
    mask_epilogue<true>(acc, mk, r0, R, col, lh, ybar + LS, As[1], bacc2);
    __syncthreads();
    // ---- layer 2
    zero_acc(acc);
    mma_tile<128>(As[1], w2, acc, l31, lh, [&](int g) {
      if (g < 8) stage_mask_piece(X0, r0, par, g);
    });
    PP_WAIT_VMEM();                                  // also: XS / G / Ps of the next tile have landed
    read_masks(mk, par ^ 1);
    mask_epilogue<true>(acc, mk, r0, R, col, lh, ybar + 2 * LS, As[2], bacc1);
    __syncthreads();
    // ---- layer 1: Ybar0 stays in LDS (the next tile's X2 gates ride in this block)
    zero_acc(acc);
    mma_tile<128>(As[2], w1, acc, l31, lh, [&](int g) {
      if (g < 8 && tnext < ntiles) stage_mask_piece(X2, tnext * TILE_ROWS, par ^ 1, g);
    });
    PP_WAIT_VMEM();
    read_masks(mk, par);
    mask_epilogue<false>(acc, mk, r0, R, col, lh, ybar, As[1], bdummy);
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
        atomicAdd(&pts_grad[(s0 + s) * 3 + i], v);   // no load to wait for (a read-modify-write would drain the stores)
      }
    }
    // ---- the next tile's output-layer backward into As[0] (last read by this tile's first MFMA block).  Measured: issuing
    // it from inside the last MFMA loop instead is slower (its LDS round trips stall the MFMA issue of the lone wave).
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
    red[9 * 128 + j0] = bacc3;
  }
  __syncthreads();
  if (h0 == 0) {
#pragma unroll
    for (int o = 0; o < 4; ++o) atomicAdd(&params_grad[WPF_W4 + o * 128 + j0], (wacc4[o] + red[o * 128 + j0]) * out_range);
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(&params_grad[WPF_W0 + j0 * 3 + i], wacc0[i] + red[(4 + i) * 128 + j0]);
    atomicAdd(&params_grad[WPF_B0 + j0], bacc0 + red[7 * 128 + j0]);
    if (j0 < 4) atomicAdd(&params_grad[WPF_B4 + j0], (bacc4 + red[8 * 128 + j0]) * out_range);
    atomicAdd(&params_grad[WPF_B3 + j0], bacc3 + red[9 * 128 + j0]);
  }
  // b2 / b1: lane = (feature col, row half lh)
  bacc2 += __shfl_xor(bacc2, 32, 64);
  bacc1 += __shfl_xor(bacc1, 32, 64);
  if (lh == 0) {
    atomicAdd(&params_grad[WPF_B2 + col], bacc2);
    atomicAdd(&params_grad[WPF_B1 + col], bacc1);
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

// ------------------------------------------------------------------------------------------------ weight gradients
// Wbar_l[n][k] += sum_r Ybar_l[r][n] * X_l[r][k]  for up to three layers in ONE persistent kernel.  A step = one 64-row
// tile of one layer: both operand tiles arrive by LDS-direct loads (no register staging) into a double buffer while the
// previous step is on the matrix cores; the 128 x KX accumulators of all layers stay in (accumulator) registers over
// the whole row range of the work-group and are flushed with one atomic per entry at the end.  Both MFMA operands
// are read row-wise (k = row), so the unpadded lane-contiguous layout of the LDS-direct loads is conflict-free.
struct WgradLayer {
  const float* Y;      // [R][128]   gradient w.r.t. the layer's pre-activation (already gated)
  const float* X;      // [R][KX]    input activations of the layer
  float* Wbar;         // [128][KX]
  float* bbar;         // [128] or nullptr: += column sums of Y over every FOURTH row (KXC == 128: the primal rows of the warp net's 4-row
                       // form) or over every row (KXC == 64: rgbnet);
                       // used when the data-gradient kernel leaves the hidden-layer bias gradients to this one (pp_mlp_split.hip)
};

namespace {

template <int KX>
__device__ __forceinline__ void wgrad_issue(const WgradLayer& L, int r0, int R, float* Ybuf, float* Xbuf, int wid, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int rl = 16 * wid + 2 * i;
    const int row = min(r0 + rl + lh, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.Y + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(Ybuf + rl * 128), 16, 0, 0);
  }
  if (KX == 128) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = 16 * wid + 2 * i;
      const int row = min(r0 + rl + lh, R - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.X + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(Xbuf + rl * 128), 16, 0, 0);
    }
  } else {               // KX == 64: four 256-byte rows per instruction
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 16 * wid + 4 * i;
      const int row = min(r0 + rl + lq, R - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(L.X + (size_t)row * 64 + l15 * 4), PP_LDS_PTR(Xbuf + rl * 64), 16, 0, 0);
    }
  }
}

// 64 rows x (128 x KX) on the matrix cores; hand-scheduled (tools/gen_wgrad_asm.py -> pp_wgrad_asm.inc)
template <int KX>
__device__ __forceinline__ void wgrad_compute(const float* __restrict__ Ybuf, const float* __restrict__ Xbuf,
                                              f32x16 (&acc)[2][KX / 64], int wr, int wc, int l31, int lh) {
  constexpr int NB = KX / 64;
  // 32-bit LDS byte addresses of this lane's operand streams (row parity lh, feature l31)
  const unsigned y = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Ybuf + lh * 128 + 64 * wr + l31);
  const unsigned x = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Xbuf + lh * KX + 32 * NB * wc + l31);
  float a00, a10, b00, b10, a01, a11, b01, b11;
  if constexpr (NB == 2) {
    asm volatile(PP_WGRAD_BLOCK_NB2
                 : [c00] "+a"(acc[0][0]), [c10] "+a"(acc[1][0]), [c01] "+a"(acc[0][1]), [c11] "+a"(acc[1][1]),
                   [a00] "=&v"(a00), [a10] "=&v"(a10), [b00] "=&v"(b00), [b10] "=&v"(b10),
                   [a01] "=&v"(a01), [a11] "=&v"(a11), [b01] "=&v"(b01), [b11] "=&v"(b11)
                 : [y] "v"(y), [x] "v"(x)
                 : "memory");
  } else {
    asm volatile(PP_WGRAD_BLOCK_NB1
                 : [c00] "+a"(acc[0][0]), [c10] "+a"(acc[1][0]),
                   [a00] "=&v"(a00), [a10] "=&v"(a10), [b00] "=&v"(b00),
                   [a01] "=&v"(a01), [a11] "=&v"(a11), [b01] "=&v"(b01)
                 : [y] "v"(y), [x] "v"(x)
                 : "memory");
  }
}

template <int KX>
__device__ __forceinline__ void wgrad_flush(float* __restrict__ Wbar, const f32x16 (&acc)[2][KX / 64], int wr, int wc, int l31, int lh) {
  constexpr int NB = KX / 64;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int k = 32 * NB * wc + u * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        atomicAdd(&Wbar[(size_t)n * KX + k], acc[t][u][reg]);
      }
    }
}

template <int KX>
__device__ __forceinline__ void zero_acc2(f32x16 (&acc)[2][KX / 64]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < KX / 64; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
}

}  // namespace

// One pipeline step: start the LDS-direct loads of the NEXT step into (Ydst, Xdst), then run the matrix cores on the
// tiles already sitting in (Ysrc, Xsrc).  The __restrict__ qualifiers matter: after inlining they become alias scopes
// on the LDS-DMA writes and the ds_reads, which lets the compiler's wait-count pass see that the reads do not depend on
// the loads just issued (otherwise it inserts a vmcnt(0) in front of the first ds_read and the overlap is gone).
template <int KXN, int KXS>
__device__ __forceinline__ void wgrad_step(bool issue, const WgradLayer& Ln, int r0n, int R, float* __restrict__ Ydst,
                                           float* __restrict__ Xdst, const float* __restrict__ Ysrc,
                                           const float* __restrict__ Xsrc, f32x16 (&acc)[2][KXS / 64], int wid, int lane) {
  if (issue) wgrad_issue<KXN>(Ln, r0n, R, Ydst, Xdst, wid, lane);
  wgrad_compute<KXS>(Ysrc, Xsrc, acc, wid >> 1, wid & 1, lane & 31, lane >> 5);
}

// layers A, B: KX = 128; layer C: KX = KXC (128 for the warp net, 64 for rgbnet's input layer)
template <int KXC>
__global__ __launch_bounds__(256) void k_wgrad_chain(WgradLayer LA, WgradLayer LB, WgradLayer LC,
                                                     const int32_t* __restrict__ count, int rmul, int rcap) {
  __shared__ __attribute__((aligned(16))) float Yb0[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Xb0[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Yb1[TILE_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float Xb1[TILE_ROWS * 128];
  const int R = min(count[0] * rmul, rcap);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  if ((int)blockIdx.x * 2 >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;
  f32x16 accA[2][2], accB[2][2], accC[2][KXC / 64];
  zero_acc2<128>(accA); zero_acc2<128>(accB); zero_acc2<KXC>(accC);

  // rows past R are fetched clamped; their Y rows are zeroed in LDS before use (last tile only)
  auto fix_tail = [&](int r0, float* Ybuf) {
    if (r0 + TILE_ROWS > R) {
      for (int i = tid; i < TILE_ROWS * 128; i += 256)
        if (r0 + (i >> 7) >= R) Ybuf[i] = 0.f;
      __syncthreads();
    }
  };
  // Six steps per iteration keep the buffer assignment static: A0 B1 C0 | A1 B0 C1.  The unit of work is a PAIR of
  // adjacent tiles, so only the very last pair can contain an empty tile (rows clamped, Y zeroed by fix_tail);
  // straight-line control flow keeps the 192 accumulator registers pinned across the hand-scheduled blocks.
  // optional bias gradients: thread = (feature tid & 127, row half tid >> 7); rows past R are zero in LDS (fix_tail).  The eight
  // LDS reads are issued BEFORE the hand-scheduled MFMA block and summed after it (colsum_take: the empty asm is their first
  // use), so their latency hides behind the block instead of in front of it.
  float bA = 0.f, bB = 0.f, bC = 0.f;
  constexpr int NCS = KXC == 64 ? 32 : 8, CSTEP = KXC == 64 ? 1 : 4;     // rgbnet: every row; warp net: the primal rows
  float cs[NCS];
  auto colsum_issue = [&](const float* Ybuf) {
    const float* p = Ybuf + (tid >> 7) * 32 * 128 + (tid & 127);
#pragma unroll
    for (int r = 0; r < NCS; ++r) cs[r] = p[r * CSTEP * 128];
  };
  auto colsum_take = [&]() {
#pragma unroll
    for (int r = 0; r < NCS; ++r) asm volatile("" : "+v"(cs[r]));
    float t[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      t[r] = cs[r];
#pragma unroll
      for (int k = 8; k < NCS; k += 8) t[r] += cs[r + k];
    }
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  };
  const int npairs = (ntiles + 1) >> 1;
  // Tiles are walked from the END of the row range: the backward-data kernel that just ran wrote Ybar front to back, so its
  // most recent output is what the 256 MB Infinity Cache still holds.
  wgrad_issue<128>(LA, (npairs - 1 - (int)blockIdx.x) * 2 * TILE_ROWS, R, Yb0, Xb0, wid, lane);
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int r0 = (npairs - 1 - pair) * 2 * TILE_ROWS;
    const int r1 = r0 + TILE_ROWS;
    const int t2 = pair + gridDim.x;
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r0, Yb0);
    if (LA.bbar) colsum_issue(Yb0);
    wgrad_step<128, 128>(true, LB, r0, R, Yb1, Xb1, Yb0, Xb0, accA, wid, lane);
    if (LA.bbar) bA += colsum_take();
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r0, Yb1);
    if (LB.bbar) colsum_issue(Yb1);
    wgrad_step<KXC, 128>(true, LC, r0, R, Yb0, Xb0, Yb1, Xb1, accB, wid, lane);
    if (LB.bbar) bB += colsum_take();
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r0, Yb0);
    if (LC.bbar) colsum_issue(Yb0);
    wgrad_step<128, KXC>(true, LA, r1, R, Yb1, Xb1, Yb0, Xb0, accC, wid, lane);
    if (LC.bbar) bC += colsum_take();
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r1, Yb1);
    if (LA.bbar) colsum_issue(Yb1);
    wgrad_step<128, 128>(true, LB, r1, R, Yb0, Xb0, Yb1, Xb1, accA, wid, lane);
    if (LA.bbar) bA += colsum_take();
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r1, Yb0);
    if (LB.bbar) colsum_issue(Yb0);
    wgrad_step<KXC, 128>(true, LC, r1, R, Yb1, Xb1, Yb0, Xb0, accB, wid, lane);
    if (LB.bbar) bB += colsum_take();
    PP_WAIT_VMEM(); __syncthreads();
    fix_tail(r1, Yb1);
    if (LC.bbar) colsum_issue(Yb1);
    wgrad_step<128, KXC>(t2 < npairs, LA, (npairs - 1 - t2) * 2 * TILE_ROWS, R, Yb0, Xb0, Yb1, Xb1, accC, wid, lane);
    if (LC.bbar) bC += colsum_take();
  }
  wgrad_flush<128>(LA.Wbar, accA, wr, wc, l31, lh);
  wgrad_flush<128>(LB.Wbar, accB, wr, wc, l31, lh);
  wgrad_flush<KXC>(LC.Wbar, accC, wr, wc, l31, lh);
  if (LA.bbar) atomicAdd(&LA.bbar[tid & 127], bA);
  if (LB.bbar) atomicAdd(&LB.bbar[tid & 127], bB);
  if (LC.bbar) atomicAdd(&LC.bbar[tid & 127], bC);
}

// option "wgrad_split" = 1 replaces the fp32-instruction chain kernel above by three launches of the self-scaling split-precision
// kernel.  Parity suite green, but SLOWER on MI355X (1.405 vs 1.356 ms per step): the chain kernel reads each operand once,
// keeps 192 accumulators resident and is hand-scheduled; three load-bound launches with an extra barrier per chunk are not
// a match for it.  Off by default; kept as the starting point for a fused split-precision chain.
static bool wgrad_split_enabled() { return pp_opt(PP_OPT_WGRAD_SPLIT) == 1; }

int pp_launch_wgrad_chain(const float* YA, const float* XA, float* WA, const float* YB, const float* XB, float* WB,
                          const float* YC, const float* XC, float* WC, int kxc, const int32_t* count, int rmul, int rcap,
                          hipStream_t st, float* bA, float* bB, float* bC, int wgs) {
  if (pp_opt(PP_OPT_MLP_SPLIT) & 16)            // split-precision chain kernel (pp_mlp_split.hip): bias sums included when asked for
    return pp_launch_wgrad_chain_s(YA, XA, WA, YB, XB, WB, YC, XC, WC, kxc, count, rmul, rcap, st, bA, bB, bC, wgs);
  if (wgrad_split_enabled() && !bA) {
    // three launches of the self-scaling split-precision kernel (pp_gemm_split.h): three fp16 products per fp32 product,
    // error against fp64 equal to the fp32 matrix instructions'; load-bound instead of matrix-pipe bound
    const int splits = rcap >= 131072 ? 256 : 128;
    hipLaunchKernelGGL(k_gemm_tn_split_auto, dim3(splits), dim3(256), 0, st, YA, XA, 128, 128, WA, 128, count, rmul, rcap);
    hipLaunchKernelGGL(k_gemm_tn_split_auto, dim3(splits), dim3(256), 0, st, YB, XB, 128, 128, WB, 128, count, rmul, rcap);
    hipLaunchKernelGGL(k_gemm_tn_split_auto, dim3(splits), dim3(256), 0, st, YC, XC, kxc, kxc, WC, kxc, count, rmul, rcap);
    return 0;
  }
  WgradLayer LA{YA, XA, WA, bA}, LB{YB, XB, WB, bB}, LC{YC, XC, WC, bC};
  const int npairs = pp_div_up(rcap, 2 * TILE_ROWS);
  const int cap_wgs = wgs > 0 ? (wgs < 16 ? 16 : wgs) : PP_FUSED_WGS;
  const int grid = npairs < cap_wgs ? npairs : cap_wgs;
  if (kxc == 128)
    hipLaunchKernelGGL((k_wgrad_chain<128>), dim3(grid), dim3(256), 0, st, LA, LB, LC, count, rmul, rcap);
  else
    hipLaunchKernelGGL((k_wgrad_chain<64>), dim3(grid), dim3(256), 0, st, LA, LB, LC, count, rmul, rcap);
  return 0;
}
// ================================================================================================ rgbnet (64 -> 128 x3 -> 3)
namespace {

// LDS-direct load of a [64][64] tile of 256-byte rows into an UNPADDED buffer whose 16-byte slots are XOR-swizzled:
// element (row, c4) lives in slot row*16 + (c4 ^ (row & 15)).  The permutation is applied on the global-address side
// (which quad a lane fetches), the LDS side of an LDS-direct load is always lane-contiguous.  With it the MFMA
// A-operand reads (8 consecutive rows per LDS phase, same c4) hit 8 different bank groups, like the padded tiles.
__device__ __forceinline__ void stage_feat_tile(const float* __restrict__ feat, int r0, int R, float* Fs, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = 16 * wid + 4 * i + (lane >> 4);            // four rows per instruction
    const int c4 = (lane & 15) ^ (rl & 15);
    const int row = min(r0 + rl, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(feat + (size_t)row * 64 + c4 * 4), PP_LDS_PTR(Fs + (16 * wid + 4 * i) * 64), 16, 0, 0);
  }
}

// acc[t] += F[t*32 + l31][0:64] . w   (swizzled feature tile, K = 64)
__device__ __forceinline__ void mma_feat_tile(const float* __restrict__ Fs, const float4 (&w)[8], f32x16 (&acc)[2], int l31, int lh) {
  const int sw = l31 & 15;
  const float* f0 = Fs + l31 * 64;
  const float* f1 = f0 + 32 * 64;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int c4 = ((2 * g + lh) ^ sw) * 4;
    const float4 a0 = *reinterpret_cast<const float4*>(f0 + c4);
    const float4 a1 = *reinterpret_cast<const float4*>(f1 + c4);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w[g].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w[g].x, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w[g].y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w[g].y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w[g].z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w[g].z, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w[g].w, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w[g].w, acc[1], 0, 0, 0);
  }
}

// two rows (2 x 512 B) per instruction, 8 instructions per wavefront: a [64][128] tile, unpadded, rows clamped to R-1
__device__ __forceinline__ void stage_tile128(const float* __restrict__ X, int r0, int R, float* dst, int wid, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int rl = 16 * wid + 2 * i;
    const int row = min(r0 + rl + lh, R - 1);
    __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(X + (size_t)row * 128 + l31 * 4), PP_LDS_PTR(dst + rl * 128), 16, 0, 0);
  }
}

// backward-data epilogue with one ReLU gate per element (gates = the layer input's activations, tile in LDS)
template <bool FULL, bool TOGLOBAL>
__device__ __forceinline__ void gate_epilogue_impl(const f32x16 (&acc)[2], const float* __restrict__ Gt, int r0, int R, int col,
                                                   int lh, float* __restrict__ C, float* __restrict__ Anext, float& bsum) {
  float* __restrict__ Ct = C + (size_t)r0 * 128;
  const unsigned lane_off = (unsigned)(4 * lh) * 128u + (unsigned)col;
  float* __restrict__ At = Anext + (4 * lh) * LDA + col;
  const float* __restrict__ gt = Gt + (4 * lh) * 128 + col;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = t * 32 + (reg & 3) + 8 * (reg >> 2);
      const float v = (gt[row * 128] > 0.f) ? acc[t][reg] : 0.f;
      bsum += v;
      At[row * LDA] = v;
      if (TOGLOBAL && (FULL || r0 + row + 4 * lh < R)) Ct[lane_off + (unsigned)(row * 128)] = v;
    }
  }
}
template <bool TOGLOBAL>
__device__ __forceinline__ void gate_epilogue(const f32x16 (&acc)[2], const float* __restrict__ Gt, int r0, int R, int col, int lh,
                                              float* __restrict__ C, float* __restrict__ Anext, float& bsum) {
  if (r0 + TILE_ROWS <= R) gate_epilogue_impl<true, TOGLOBAL>(acc, Gt, r0, R, col, lh, C, Anext, bsum);
  else gate_epilogue_impl<false, TOGLOBAL>(acc, Gt, r0, R, col, lh, C, Anext, bsum);
}

}  // namespace

// feat[M][64] -> rgb[M][3] = sigmoid(MLP(feat) (+ logit_add)); hidden activations H0..H2 ([cap][128] each) kept for backward
__global__ __launch_bounds__(256) void k_rgb_fused_fwd(const float* __restrict__ params, const float* __restrict__ feat,
                                                       const int32_t* __restrict__ count, int capacity,
                                                       const float* __restrict__ logit_add, int add_ld,
                                                       float* __restrict__ acts, float* __restrict__ rgb) {
  __shared__ __attribute__((aligned(16))) float As[2][TILE_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float Fs[2][TILE_ROWS * 64];
  __shared__ __attribute__((aligned(16))) float W3s[4 * LDA];
  __shared__ __attribute__((aligned(16))) float Red[4 * 64 * 4];
  const int R = min(count[0], capacity);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const size_t LS = (size_t)capacity * 128;

  float4 w0[8], w1[16], w2[16];
  load_w_rows<64>(w0, params + RGF_W0, 64, col, lh);
  load_w_rows<128>(w1, params + RGF_W1, 128, col, lh);
  load_w_rows<128>(w2, params + RGF_W2, 128, col, lh);
  const float b0 = params[RGF_B0 + col], b1 = params[RGF_B1 + col], b2 = params[RGF_B2 + col];
  for (int i = tid; i < 512; i += 256) {
    const int r = i >> 7, j = i & 127;
    W3s[r * LDA + j] = (r < 3) ? params[RGF_W3 + r * 128 + j] : 0.f;
  }
  const float b3 = ((tid & 3) < 3) ? params[RGF_B3 + (tid & 3)] : 0.f;

  stage_feat_tile(feat, blockIdx.x * TILE_ROWS, R, Fs[0], wid, lane);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS;
    const int tnext = tile + gridDim.x;
    if (tnext < ntiles) stage_feat_tile(feat, tnext * TILE_ROWS, R, Fs[par ^ 1], wid, lane);   // lands during this tile
    f32x16 acc[2];
    zero_acc(acc);
    mma_feat_tile(Fs[par], w0, acc, l31, lh);
    relu_epilogue<1>(acc, b0, r0, R, col, lh, acts, As[0]);
    __syncthreads();
    zero_acc(acc);
    mma_tile<128>(As[0], w1, acc, l31, lh);
    relu_epilogue<1>(acc, b1, r0, R, col, lh, acts + LS, As[1]);
    __syncthreads();
    zero_acc(acc);
    mma_tile<128>(As[1], w2, acc, l31, lh);
    PP_WAIT_VMEM();                      // next feature tile has landed (stores in flight are two MFMA blocks old)
    relu_epilogue<1>(acc, b2, r0, R, col, lh, acts + 2 * LS, As[0]);
    __syncthreads();
    // output layer (128 -> 3) on v_mfma_f32_4x4x1: lane = row, lane&3 = output, K slice per wavefront
    {
      const float* xr = &As[0][lane * LDA + 32 * wid];
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

int pp_launch_rgb_fused_fwd(const float* params, const float* feat, const int32_t* count, int capacity,
                            const float* logit_add, int add_ld, float* acts, float* rgb, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, TILE_ROWS);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_rgb_fused_fwd, dim3(grid), dim3(256), 0, st, params, feat, count, capacity, logit_add, add_ld, acts, rgb);
  return 0;
}

// Backward: output layer, the two 128x128 data-gradient products and the 128 -> 64 product onto the features in one
// persistent kernel; Ybar2 / Ybar1 / Ybar0 go to `ybar` ([3][cap][128]) for k_wgrad_chain<64>; all bias gradients and
// W3bar are accumulated in registers.
__global__ __launch_bounds__(256) void k_rgb_fused_bwd(const float* __restrict__ params, const float* __restrict__ acts,
                                                       const float* __restrict__ rgb, const float* __restrict__ rgb_grad,
                                                       const int32_t* __restrict__ count, int capacity,
                                                       float* __restrict__ ybar, float* __restrict__ params_grad,
                                                       float* __restrict__ feat_grad, float* __restrict__ logit_grad, int lg_ld) {
  __shared__ __attribute__((aligned(16))) float As[2][TILE_ROWS * LDA];
  __shared__ __attribute__((aligned(16))) float XS[TILE_ROWS * 128];      // H2 of the NEXT tile
  __shared__ __attribute__((aligned(16))) float GT[TILE_ROWS * 128];      // gates of the current layer (H1, then H0)
  __shared__ __attribute__((aligned(16))) float RG[2][2][192];            // [parity][rgb | rgb_grad] of a tile
  __shared__ __attribute__((aligned(16))) float GL[TILE_ROWS * 4];        // d loss / d logits of the staged tile
  const int R = min(count[0], capacity);
  const int ntiles = (R + TILE_ROWS - 1) / TILE_ROWS;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int col = wid * 32 + l31;
  const size_t LS = (size_t)capacity * 128;
  const float* __restrict__ H0 = acts;
  const float* __restrict__ H1 = acts + LS;
  const float* __restrict__ H2 = acts + 2 * LS;

  float4 w2[16], w1[16], w0[16];
  load_w_cols<128>(w2, params + RGF_W2, 128, col, lh);
  load_w_cols<128>(w1, params + RGF_W1, 128, col, lh);
  // last product: feat_grad[64 rows][64] = Ybar0 . W0, wavefront = (row block wid>>1, column block wid&1)
  const int wr = wid >> 1, fcol = (wid & 1) * 32 + l31;
  load_w_cols<128>(w0, params + RGF_W0, 64, fcol, lh);
  const int j0 = tid & 127, h0 = tid >> 7;
  const float w3a = params[RGF_W3 + j0], w3b = params[RGF_W3 + 128 + j0], w3c = params[RGF_W3 + 256 + j0];
  float wacc3[3] = {0.f, 0.f, 0.f}, bacc3 = 0.f, bacc2 = 0.f, bacc1 = 0.f, bacc0 = 0.f;

  auto stage = [&](int t, int b) {              // H2 rows, rgb and rgb_grad of tile t
    const int r0 = t * TILE_ROWS;
    stage_tile128(H2, r0, R, XS, wid, lane);
    if (wid < 3) {
      const int e = min(r0 * 3 + tid, R * 3 - 1);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(rgb + e), PP_LDS_PTR(&RG[b][0][wid * 64]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(rgb_grad + e), PP_LDS_PTR(&RG[b][1][wid * 64]), 4, 0, 0);
    }
  };
  // d loss / d logit = rgb_grad * rgb * (1 - rgb); zero for rows past the end (which zeroes everything downstream)
  auto logit_grads = [&](int t, int b) {
    const int r0 = t * TILE_ROWS;
    if (tid < 192) {
      const int m = tid / 3, o = tid - 3 * m;
      const float r = RG[b][0][tid];
      const float gl = (r0 + m < R) ? RG[b][1][tid] * r * (1.f - r) : 0.f;
      GL[m * 4 + o] = gl;
      if (logit_grad && r0 + m < R) logit_grad[(size_t)(r0 + m) * lg_ld + o] = gl;
    } else {
      GL[(tid - 192) * 4 + 3] = 0.f;
    }
  };
  // output layer backward of the staged tile -> dst (+ HBM copy): thread = (feature j0, row half h0)
  auto out_layer_bwd = [&](int t, float* __restrict__ dst) {
    const int r0 = t * TILE_ROWS;
    float* __restrict__ yt = ybar + (size_t)r0 * 128;
    const unsigned off = (unsigned)(h0 * 32) * 128u + (unsigned)j0;
    float* __restrict__ at = dst + (h0 * 32) * LDA + j0;
    const float* __restrict__ xs = &XS[(h0 * 32) * 128 + j0];
    auto body = [&](auto fc) {
      constexpr bool FULL = decltype(fc)::value;
#pragma unroll 8
      for (int q = 0; q < 32; ++q) {
        const float4 g = *reinterpret_cast<const float4*>(&GL[(h0 * 32 + q) * 4]);
        const float x = xs[q * 128];
        wacc3[0] += g.x * x; wacc3[1] += g.y * x; wacc3[2] += g.z * x;
        const float hb = g.x * w3a + g.y * w3b + g.z * w3c;
        const float v = (x > 0.f) ? hb : 0.f;
        bacc2 += v;
        at[q * LDA] = v;
        if (FULL || r0 + h0 * 32 + q < R) yt[off + q * 128] = v;
        if (j0 < 3) bacc3 += GL[(h0 * 32 + q) * 4 + j0];
      }
    };
    PP_WITH_FULL(r0 + TILE_ROWS <= R, body);
  };

  stage(blockIdx.x, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  logit_grads(blockIdx.x, 0);
  __syncthreads();
  out_layer_bwd(blockIdx.x, As[0]);
  __syncthreads();

  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const int r0 = tile * TILE_ROWS;
    const int tnext = tile + gridDim.x;
    float* __restrict__ Aa = As[par];            // holds Ybar2 of this tile
    float* __restrict__ Ab = As[par ^ 1];
    if (tnext < ntiles) stage(tnext, par ^ 1);   // XS / RG[par^1]: consumed at the bottom of this iteration
    stage_tile128(H1, r0, R, GT, wid, lane);
    f32x16 acc[2];
    // ---- layer 2: Ybar1 = gate(H1) . (Ybar2 W2)
    zero_acc(acc);
    mma_tile<128>(Aa, w2, acc, l31, lh);
    PP_WAIT_VMEM();
    __syncthreads();                             // gates come from all four wavefronts' loads
    gate_epilogue<true>(acc, GT, r0, R, col, lh, ybar + LS, Ab, bacc1);
    __syncthreads();
    stage_tile128(H0, r0, R, GT, wid, lane);
    // ---- layer 1: Ybar0 = gate(H0) . (Ybar1 W1)
    zero_acc(acc);
    mma_tile<128>(Ab, w1, acc, l31, lh);
    PP_WAIT_VMEM();
    __syncthreads();
    gate_epilogue<true>(acc, GT, r0, R, col, lh, ybar + 2 * LS, Aa, bacc0);
    if (tnext < ntiles) logit_grads(tnext, par ^ 1);
    __syncthreads();
    // ---- layer 0: feat_grad = Ybar0 . W0   (32 rows x 32 features per wavefront)
    {
      f32x16 fa;
#pragma unroll
      for (int i = 0; i < 16; ++i) fa[i] = 0.f;
      const float* ap = Aa + (wr * 32 + l31) * LDA + 4 * lh;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 8 * g);
        fa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0[g].x, fa, 0, 0, 0);
        fa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w0[g].y, fa, 0, 0, 0);
        fa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w0[g].z, fa, 0, 0, 0);
        fa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w0[g].w, fa, 0, 0, 0);
      }
      float* __restrict__ ft = feat_grad + (size_t)r0 * 64;
      const unsigned foff = (unsigned)(wr * 32 + 4 * lh) * 64u + (unsigned)fcol;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2);
        if (r0 + wr * 32 + 4 * lh + row < R) ft[foff + (unsigned)(row * 64)] = fa[reg];
      }
    }
    // ---- the next tile's output-layer backward into the buffer this tile started from (free since the first MFMA block;
    // Aa is still being read by the feature-gradient product, so it goes to Ab, which the next iteration calls Aa)
    if (tnext < ntiles) out_layer_bwd(tnext, Ab);
    __syncthreads();
  }

  // ---- flush: W3bar, b3bar (thread = feature / output), b2bar (thread = feature); b1bar, b0bar (lane = feature, row half)
  float* red = &As[0][0];
  __syncthreads();
  if (h0 == 1) {
#pragma unroll
    for (int o = 0; o < 3; ++o) red[o * 128 + j0] = wacc3[o];
    red[3 * 128 + j0] = bacc2;
    if (j0 < 3) red[4 * 128 + j0] = bacc3;
  }
  __syncthreads();
  if (h0 == 0) {
#pragma unroll
    for (int o = 0; o < 3; ++o) atomicAdd(&params_grad[RGF_W3 + o * 128 + j0], wacc3[o] + red[o * 128 + j0]);
    atomicAdd(&params_grad[RGF_B2 + j0], bacc2 + red[3 * 128 + j0]);
    if (j0 < 3) atomicAdd(&params_grad[RGF_B3 + j0], bacc3 + red[4 * 128 + j0]);
  }
  bacc1 += __shfl_xor(bacc1, 32, 64);
  bacc0 += __shfl_xor(bacc0, 32, 64);
  if (lh == 0) {
    atomicAdd(&params_grad[RGF_B1 + col], bacc1);
    atomicAdd(&params_grad[RGF_B0 + col], bacc0);
  }
}

int pp_launch_rgb_fused_bwd(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                            const int32_t* count, int capacity, float* ybar, float* params_grad, float* feat_grad,
                            float* logit_grad, int lg_ld, hipStream_t st) {
  const int ntiles = pp_div_up(capacity, TILE_ROWS);
  const int grid = ntiles < PP_FUSED_WGS ? ntiles : PP_FUSED_WGS;
  hipLaunchKernelGGL(k_rgb_fused_bwd, dim3(grid), dim3(256), 0, st, params, acts, rgb, rgb_grad, count, capacity, ybar,
                     params_grad, feat_grad, logit_grad, lg_ld);
  return 0;
}

