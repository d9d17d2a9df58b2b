// Split-precision weight-gradient GEMM for the 256-wide layers of the scene branch, second generation:
//   Wbar[n][k] += sum_r Y[r][n] X[r][k],  n, k in [0, 256),  bbar[n] += sum_r Y[r][n]   (fp32 in memory, three fp16 products
// per fp32 product, scales from the operands' recorded maxima as in pp_gemm_split.h).
// k_gemm_tn_split works on 128 x 128 output blocks: every operand tile is fetched and converted by two blocks, and the conversion
// (fp32 tile -> registers -> transposed split image in LDS) sits between the loads and the MFMAs of ONE work-group per CU.  Here
// (the scheme of k_wgrad_chain_s, pp_mlp_split.hip) one work-group of EIGHT wavefronts owns all 256 x 256 outputs of its row
// range: both operand tiles arrive row-major by LDS-direct loads into a double buffer (32 rows each), a lane gathers its operand
// fragments - 8 consecutive rows of one column - straight from the fp32 tile (consecutive lanes read consecutive columns), scales,
// splits (20 vector instructions per fragment) and feeds 24 MFMAs per 16-row group: 5 vector instructions per MFMA, which is
// what a v_mfma_f32_32x32x16_f16 hides (tools/mfma_valu_probe2.hip); the second wavefront of a SIMD covers the LDS latency.
// MEASURED: 127 us per layer at 131 k rows against 123 us for k_gemm_tn_split (scene step 3.22 vs 3.21 ms) - the conversion
// (v_fma_mix* costs 11.6 cycles beside MFMAs) bounds both; OFF by default (option nerf_tn256), kept as the starting point for a
// version whose producers write split operands.
#pragma once
#include "pp_gemm_split.h"

#define TN256_ROWS 32                                 // rows per tile (two 16-row operand groups)
#define PP_TN_GLOBAL(p) ((const __attribute__((address_space(1))) void*)(p))
#define PP_TN_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// grid = persistent work-groups (one per CU: 128 KB of LDS) of 512 threads; Y [R][ldy] (256 columns used), X [R][ldx] (256
// columns used), Wbar [256][ldw]
static __global__ __launch_bounds__(512, 1) void k_gemm_tn256(const float* __restrict__ Y, int ldy, const float* __restrict__ X, int ldx,
                                                              float* __restrict__ Wbar, int ldw, float* __restrict__ bbar,
                                                              const int32_t* __restrict__ count, int rcap,
                                                              const float* __restrict__ y_max, const float* __restrict__ x_max) {
  __shared__ __attribute__((aligned(16))) float Yb[2][TN256_ROWS * 256];
  __shared__ __attribute__((aligned(16))) float Xb[2][TN256_ROWS * 256];
  const int R = min(count[0], rcap);
  const int ntiles = (R + TN256_ROWS - 1) / TN256_ROWS;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;      // outputs n in 64 wr .. +63, k in 128 wc .. +127
  const float sY = pp_split_scale(y_max[0]), sX = pp_split_scale(x_max[0]);
  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const bool bias = bbar != nullptr && wc == 0;

  // one instruction = one 1 KB row; a wavefront loads rows 4 wid .. 4 wid + 3 of both tiles
  auto issue = [&](int tile, int b) {
    const int r0 = tile * TN256_ROWS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 4 * wid + i;
      const int row = min(r0 + rl, R - 1);
      __builtin_amdgcn_global_load_lds(PP_TN_GLOBAL(Y + (size_t)row * ldy + lane * 4), PP_TN_LDS(&Yb[b][rl * 256]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(PP_TN_GLOBAL(X + (size_t)row * ldx + lane * 4), PP_TN_LDS(&Xb[b][rl * 256]), 16, 0, 0);
    }
  };
  issue(blockIdx.x, 0);
  int b = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, b ^= 1) {
    const int r0 = tile * TN256_ROWS;
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this tile's loads (the only memory operations in flight)
    __syncthreads();                                    // ... of all wavefronts; and everybody is done with the other buffer
    if (r0 + TN256_ROWS > R) {                          // last tile: rows past R were fetched clamped, their Y is zeroed
      for (int i = tid; i < TN256_ROWS * 256; i += 512)
        if (r0 + (i >> 8) >= R) Yb[b][i] = 0.f;
      __syncthreads();
    }
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, b ^ 1);
    const float* __restrict__ yp = &Yb[b][(8 * lh) * 256 + 64 * wr + l31];
    const float* __restrict__ xp = &Xb[b][(8 * lh) * 256 + 128 * wc + l31];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pp_half8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = yp[(16 * ks + j) * 256 + 32 * t];
        if (bias) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[t] += v[j];
        }
        pp_split8(v, sY, ah[t], al[t]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xp[(16 * ks + j) * 256 + 32 * u];
        pp_split8(v, sX, bh[u], bl[u]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {        // small terms first
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
    }
  }
  // flush: one atomic per entry and work-group, scaled back
  const float f = 1.0f / (sY * sX);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 128 * wc + 32 * u + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = 64 * wr + 32 * t + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        atomicAdd(&Wbar[(size_t)n * ldw + k], acc[t][u][reg] * f);
      }
    }
  if (bias) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      if (lh == 0) atomicAdd(&bbar[64 * wr + 32 * t + l31], v);
    }
  }
}
