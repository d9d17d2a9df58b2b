// Layer-by-layer fp32 MFMA GEMM templates shared by the object-branch MLPs (pp_mlp.hip) and the scene-branch NeRF
// (pp_nerf.hip).  gridDim.y selects a 128-column block of the output (k_gemm128) or a 128 x 128 block of the weight
// gradient (k_gemm_tn), so layers wider than 128 run as several column blocks over the same row tiles.
#pragma once
#include "pp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDT 36   // row stride (floats): 16-B aligned rows, conflict-free ds_read_b128 / ds_write_b128
enum { MODE_NT = 0 };
enum { EPI_RELU = 0, EPI_MASK = 1, EPI_PLAIN = 2 };

// bits16 (optional, COLS == 1, outputs 256 columns wide): the ReLU mask as ONE BIT per activation.  A lane's 16 accumulator
// registers of one 32 x 32 tile are 16 rows of ONE column, so it packs them into a 16-bit word without any cross-lane work:
//   bits16[(row_group32 * 256 + column) * 2 + half],  bit j <-> row_group32 * 32 + (j & 3) + 8 * (j >> 2) + 4 * half
// (the MFMA accumulator layout, the same in every kernel that uses 32 x 32 tiles).  EPI_RELU writes the words, EPI_MASK reads
// one 2-byte word per tile and lane instead of 16 floats of the forward activation.  `cb` = first column of the work-group.
template <int EPI, int COLS, bool FULL, int TM, int TNW = 2, bool BITS = false>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TNW], int r0, int R, int Nout, int wr, int wc, int l31,
                                              int lh, const float* __restrict__ bias,
                                              const float* __restrict__ Xmask, int ldm, float* __restrict__ C, int ldc,
                                              uint16_t* __restrict__ bits16 = nullptr, int cb = 0) {
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int u = 0; u < TNW; ++u) {
      const int col = wc * (32 * TNW) + u * 32 + l31;
      if (!FULL && col >= Nout) continue;
      const float bcol = (EPI == EPI_RELU && bias) ? bias[col] : 0.f;
      const int rbase = r0 + wr * (32 * TM) + t * 32 + 4 * lh;
      const size_t bidx = (((size_t)(rbase - 4 * lh) >> 5) * 256 + cb + col) * 2 + lh;
      unsigned mbits = 0u;
      if (BITS && COLS == 1 && EPI == EPI_MASK) mbits = bits16[bidx];
      float mk[4] = {1.f, 1.f, 1.f, 1.f};
      if (EPI == EPI_MASK && COLS == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // one mask value per sample (4 rows): row of the primal activation
          const int mrow = rbase + 8 * q;
          mk[q] = (FULL || mrow < R) ? Xmask[(size_t)mrow * ldm + col] : 0.f;
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = rbase + (reg & 3) + 8 * (reg >> 2);
        if (!FULL && row >= R) continue;
        float val = acc[t][u][reg];
        if (EPI == EPI_RELU) {
          if (COLS == 1) {
            val = fmaxf(val + bcol, 0.f);
            if (BITS) mbits |= (val > 0.f ? 1u : 0u) << reg;
          } else {
            const float y0 = acc[t][u][reg & ~3] + bcol;   // primal row of this sample (same lane)
            const float y = val + (((reg & 3) == 0) ? bcol : 0.f);
            val = (y0 > 0.f) ? y : 0.f;
          }
        } else if (EPI == EPI_MASK) {
          if (COLS == 4) val = (mk[reg >> 2] > 0.f) ? val : 0.f;
          else if (BITS) val = ((mbits >> reg) & 1u) ? val : 0.f;
          else val = (Xmask[(size_t)row * ldm + col] > 0.f) ? val : 0.f;
        }
        C[(size_t)row * ldc + col] = val;
      }
      if (BITS && COLS == 1 && EPI == EPI_RELU) bits16[bidx] = (uint16_t)mbits;
    }
  }
}

// C[r][n] = epi( sum_k A[r][k] * B(n,k) ),  NT: B(n,k) = W[n*ldw + k]   NN: B(n,k) = W[k*ldw + n]
// Register budget: the 128-column tile is left to the compiler (it takes 190-280 registers, 1-2 wavefronts per SIMD; capping
// it at 168 for three work-groups per CU measured 10 % slower); the 256-column tile is capped at 256.
template <int MODE, int EPI, int COLS, int BM, int BN = 128, bool BITS = false>
__global__ __launch_bounds__(256, ((BN == 256 || (EPI == EPI_MASK && BM == 128)) ? 2 : 1)) void k_gemm128(const float* __restrict__ A, int lda, const float* __restrict__ W_,
                                                 int ldw, int K, int Nout_, const float* __restrict__ bias_,
                                                 const float* __restrict__ Xmask_, int ldm, float* __restrict__ C_,
                                                 int ldc, const int32_t* __restrict__ count, int rmul, int rcap,
                                                 uint16_t* __restrict__ bits16 = nullptr) {
  constexpr int TM = BM / 64;            // 32-row MFMA tiles per wavefront (waves are arranged 2 x 2)
  constexpr int NA = BM / 32;            // float4 of the A tile per thread and K-chunk
  constexpr int TNW = BN / 64;           // 32-column MFMA tiles per wavefront
  constexpr int NB = BN / 32;            // float4 of the B tile per thread and K-chunk
  __shared__ float As[BM * LDT];
  __shared__ float Bs[BN * LDT];
  const int cb = blockIdx.y * BN;        // column block of the output
  const float* __restrict__ W = W_ + (size_t)cb * ldw;
  const float* __restrict__ bias = bias_ ? bias_ + cb : nullptr;
  const float* __restrict__ Xmask = Xmask_ ? Xmask_ + cb : nullptr;
  float* __restrict__ C = C_ + cb;
  const int Nout = min(BN, Nout_ - cb);
  const int R = min(count[0] * rmul, rcap);
  const int ntiles = (R + BM - 1) / BM;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // Persistent work-group: tiles blockIdx.x, +gridDim.x, ...  Software pipeline: the global loads of the NEXT K-chunk -
  // or of the next tile's first chunk - are issued before the MFMA block of the current chunk (register staging), so
  // HBM/L2 latency hides behind the matrix pipe and the chip-wide load bursts of lock-stepped work-groups disappear.
  float4 ra[NA], rw[NB];
  auto load_chunk = [&](int r0, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int e = tid + i * 256;
      int row = e >> 3, c4 = e & 7;
      int gr = r0 + row;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < R) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gr * lda + k0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int e = tid + i * 256;
      int row = e >> 3, c4 = e & 7;
      rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < Nout) rw[i] = *reinterpret_cast<const float4*>(W + (size_t)row * ldw + k0 + c4 * 4);
    }
  };
  load_chunk(tile * BM, 0);
  for (; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * BM;
    f32x16 acc[TM][TNW];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TNW; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int e = tid + i * 256;
        *reinterpret_cast<float4*>(As + (e >> 3) * LDT + (e & 7) * 4) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int e = tid + i * 256;
        *reinterpret_cast<float4*>(Bs + (e >> 3) * LDT + (e & 7) * 4) = rw[i];
      }
      __syncthreads();
      if (k0 + 32 < K) load_chunk(r0, k0 + 32);
      else if (tile + (int)gridDim.x < ntiles) load_chunk((tile + gridDim.x) * BM, 0);
      // K-slot permutation: half-wave h supplies k = kb + 4h + j to the j-th of four consecutive MFMAs, so every lane
      // fetches its four operands with ONE 16-byte LDS read (A and B use the same map, the sum over k is unchanged).
#pragma unroll
      for (int kb = 0; kb < 32; kb += 8) {
        float4 a[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t)
          a[t] = *reinterpret_cast<const float4*>(As + (wr * (32 * TM) + t * 32 + l31) * LDT + kb + 4 * lh);
        float4 b[TNW];
#pragma unroll
        for (int u = 0; u < TNW; ++u)
          b[u] = *reinterpret_cast<const float4*>(Bs + (wc * (32 * TNW) + u * 32 + l31) * LDT + kb + 4 * lh);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
#pragma unroll
          for (int u = 0; u < TNW; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < TNW; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < TNW; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < TNW; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // epilogue.  Full tiles (all but the last one) take a branch-free instantiation.
    if ((r0 + BM <= R) && (Nout == BN))
      gemm_epilogue<EPI, COLS, true, TM, TNW, BITS>(acc, r0, R, Nout, wr, wc, l31, lh, bias, Xmask, ldm, C, ldc, bits16, cb);
    else
      gemm_epilogue<EPI, COLS, false, TM, TNW, BITS>(acc, r0, R, Nout, wr, wc, l31, lh, bias, Xmask, ldm, C, ldc, bits16, cb);
  }
}

// dst[c][r] = src[r][c]  (weights are tiny: 128x128 / 128x64); lets the backward-data GEMM run in the same NT form
static __global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int c = i / rows, r = i - c * rows;           // consecutive threads write consecutive dst elements
  dst[i] = src[r * cols + c];
}

// Wbar[n][k] += sum_r Y[r][n] * X[r][k]   (n < 128, k < Kx) ;  bbar[n] += sum_{r % COLS == 0} Y[r][n]
template <int COLS, int CH = 32>
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ Y_, int ldy, const float* __restrict__ X_, int ldx,
                                                 int Kx_, float* __restrict__ Wbar_, int ldwb, float* __restrict__ bbar_,
                                                 const int32_t* __restrict__ count, int rmul, int rcap) {
  constexpr int NL = CH / 8;             // float4 per thread, matrix and row chunk (CH rows of 128 floats)
  __shared__ float Ys[CH * 128];
  __shared__ float Xs[CH * 128];
  const int nkb = (Kx_ + 127) >> 7;      // gridDim.y = (N / 128) * nkb blocks of 128 x 128 weight gradients
  const int nb = blockIdx.y / nkb, kb = blockIdx.y - nb * nkb;
  const float* __restrict__ Y = Y_ + nb * 128;
  const float* __restrict__ X = X_ + kb * 128;
  const int Kx = min(128, Kx_ - kb * 128);
  float* __restrict__ Wbar = Wbar_ + (size_t)nb * 128 * ldwb + kb * 128;
  float* __restrict__ bbar = (bbar_ && kb == 0) ? bbar_ + nb * 128 : nullptr;
  const int R = min(count[0] * rmul, rcap);
  const int rows_per_wg = ((R + (int)gridDim.x - 1) / (int)gridDim.x + CH - 1) / CH * CH;   // multiple of CH (and of COLS)
  const int rb = blockIdx.x * rows_per_wg;
  if (rb >= R) return;
  const int re = min(rb + rows_per_wg, R);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const bool active = (wc * 64 < Kx);
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float bsum = 0.f;
  const int kx4 = Kx >> 2;
  float4 ry[NL], rx[NL];
  auto load_rows = [&](int r0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int e = tid + i * 256;
      int rr = e >> 5, c4 = e & 31;
      int gr = r0 + rr;
      ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rx[i] = ry[i];
      if (gr < re) {
        ry[i] = *reinterpret_cast<const float4*>(Y + (size_t)gr * ldy + c4 * 4);
        if (c4 < kx4) rx[i] = *reinterpret_cast<const float4*>(X + (size_t)gr * ldx + c4 * 4);
      }
    }
  };
  // bias gradient: every thread sums half of the chunk's rows of one column (both halves of the work-group share the work)
  const int bcol = tid & 127, bhalf = tid >> 7;
  load_rows(rb);
  for (int r0 = rb; r0 < re; r0 += CH) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int e = tid + i * 256;
      int rr = e >> 5, c4 = e & 31;
      *reinterpret_cast<float4*>(Ys + rr * 128 + c4 * 4) = ry[i];
      *reinterpret_cast<float4*>(Xs + rr * 128 + c4 * 4) = rx[i];
    }
    __syncthreads();
    if (r0 + CH < re) load_rows(r0 + CH);
    if (active) {
#pragma unroll 4
      for (int kk = 0; kk < CH; kk += 2) {
        const int ridx = kk + lh;
        float a0 = Ys[ridx * 128 + wr * 64 + l31];
        float a1 = Ys[ridx * 128 + wr * 64 + 32 + l31];
        float b0 = Xs[ridx * 128 + wc * 64 + l31];
        float b1 = Xs[ridx * 128 + wc * 64 + 32 + l31];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (bbar) {
#pragma unroll
      for (int rr = 0; rr < CH / 2; rr += COLS) bsum += Ys[(bhalf * (CH / 2) + rr) * 128 + bcol];   // chunk starts are multiples of COLS
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = wc * 64 + u * 32 + l31;
        if (k >= Kx) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          atomicAdd(&Wbar[(size_t)n * ldwb + k], acc[t][u][reg]);
        }
      }
  }
  if (bbar) atomicAdd(&bbar[bcol], bsum);
}
