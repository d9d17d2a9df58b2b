// Split-precision NT GEMM for the scene branch (opt-in, PP_NERF_SPLIT=1): C = epi(A . W^T) with fp32 operands in memory,
// computed as three fp16 products  lo.hi + hi.lo + hi.hi  on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The operands are
// split while they are staged into LDS:  x * s = hi + lo,  hi = fp16(x * s),  lo = fp16(x * s - hi),  s = the power of two
// that puts the tensor's largest magnitude into [2^14, 2^15) (22 significant bits per element, no overflow, subnormal floor
// 2e-12 of the tensor maximum).  tools/split_gemm_probe.hip measures the error against fp64 on this network's shapes:
// 1.90e-7 relative rms versus 2.04e-7 for the exact-fp32 MFMA path - the dropped lo.lo term is below fp32 rounding - at a
// third of the matrix-pipe time (3 x 32 cycles per 16 k instead of 8 x 64).
//
// Every kernel that produces a GEMM operand records max |x| of what it wrote in a device slot (float bits as unsigned,
// atomicMax - valid for non-negative floats); the consumer derives its scale from the slot.  Same tile structure as
// k_gemm128 (persistent work-group, 128 x BN tile, K-chunks of 32, next chunk prefetched into registers behind the MFMAs).
#pragma once
#include "pp_gemm.h"

typedef _Float16 pp_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 pp_half4 __attribute__((ext_vector_type(4)));

#ifndef PP_SPLIT_WAVES
#define PP_SPLIT_WAVES 2
#endif
#define LDH 40        // halfs per LDS row: 80 B (16-byte aligned fragments, rows skewed by 20 banks)

__device__ __forceinline__ float pp_split_scale(float mx) {
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
  int e;
  frexpf(mx, &e);                       // mx = m * 2^e, m in [0.5, 1)  ->  mx * s in [2^14, 2^15)
  return ldexpf(1.f, 15 - e);
}

__device__ __forceinline__ void pp_split4(float4 x, float s, pp_half4& hi, pp_half4& lo) {
  const float v[4] = {x.x * s, x.y * s, x.z * s, x.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

__device__ __forceinline__ void pp_record_max(float* slot, float v) {      // v >= 0 ; one atomic per wavefront
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0 && slot) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}

template <int EPI, int BN>
__global__ __launch_bounds__(256, PP_SPLIT_WAVES) void k_gemm128s(const float* __restrict__ A, int lda, const float* __restrict__ W_, int ldw,
                                                  int K, int Nout_, const float* __restrict__ bias_,
                                                  const float* __restrict__ Xmask_, int ldm, float* __restrict__ C_, int ldc,
                                                  const int32_t* __restrict__ count, int rcap,
                                                  const float* __restrict__ a_max, const float* __restrict__ w_max,
                                                  float* __restrict__ c_max, uint16_t* __restrict__ bits16 = nullptr) {
  constexpr int BM = 128, TM = 2, NA = 4;
  constexpr int TNW = BN / 64, NB = BN / 32;
  __shared__ _Float16 Ah[BM * LDH], Al[BM * LDH], Bh[BN * LDH], Bl[BN * LDH];
  const int cb = blockIdx.y * BN;
  const float* __restrict__ W = W_ + (size_t)cb * ldw;
  const float* __restrict__ bias = bias_ ? bias_ + cb : nullptr;
  const float* __restrict__ Xmask = Xmask_ ? Xmask_ + cb : nullptr;
  float* __restrict__ C = C_ + cb;
  const int Nout = min(BN, Nout_ - cb);
  const int R = min(count[0], rcap);
  const int ntiles = (R + BM - 1) / BM;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const float sA = pp_split_scale(a_max[0]), sW = pp_split_scale(w_max[0]);
  const float inv = 1.0f / (sA * sW);

  float4 ra[NA], rw[NB];
  auto load_chunk = [&](int r0, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7, gr = r0 + row;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < R) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gr * lda + k0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < Nout) rw[i] = *reinterpret_cast<const float4*>(W + (size_t)row * ldw + k0 + c4 * 4);
    }
  };
  float vmax = 0.f;
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
        const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
        pp_half4 h, l;
        pp_split4(ra[i], sA, h, l);
        *reinterpret_cast<pp_half4*>(Ah + row * LDH + c4 * 4) = h;
        *reinterpret_cast<pp_half4*>(Al + row * LDH + c4 * 4) = l;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
        pp_half4 h, l;
        pp_split4(rw[i], sW, h, l);
        *reinterpret_cast<pp_half4*>(Bh + row * LDH + c4 * 4) = h;
        *reinterpret_cast<pp_half4*>(Bl + row * LDH + c4 * 4) = l;
      }
      __syncthreads();
      if (k0 + 32 < K) load_chunk(r0, k0 + 32);
      else if (tile + (int)gridDim.x < ntiles) load_chunk((tile + gridDim.x) * BM, 0);
#pragma unroll
      for (int ks = 0; ks < 32; ks += 16) {
        pp_half8 ah[TM], al[TM], bh[TNW], bl[TNW];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int o = (wr * 64 + t * 32 + l31) * LDH + ks + 8 * lh;
          ah[t] = *reinterpret_cast<const pp_half8*>(Ah + o);
          al[t] = *reinterpret_cast<const pp_half8*>(Al + o);
        }
#pragma unroll
        for (int u = 0; u < TNW; ++u) {
          const int o = (wc * (32 * TNW) + u * 32 + l31) * LDH + ks + 8 * lh;
          bh[u] = *reinterpret_cast<const pp_half8*>(Bh + o);
          bl[u] = *reinterpret_cast<const pp_half8*>(Bl + o);
        }
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TNW; ++u) {        // small terms first
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
          }
      }
      __syncthreads();
    }
    // epilogue: un-scale, bias + ReLU / ReLU mask / plain, record the largest magnitude written; the ReLU mask travels as
    // one bit per activation when `bits16` is given (layout: pp_gemm.h gemm_epilogue)
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TNW; ++u) {
        const int col = wc * (32 * TNW) + u * 32 + l31;
        if (col >= Nout) continue;
        const float bcol = (EPI == EPI_RELU && bias) ? bias[col] : 0.f;
        const int rbase = r0 + wr * 64 + t * 32 + 4 * lh;
        const size_t bidx = (((size_t)(rbase - 4 * lh) >> 5) * 256 + cb + col) * 2 + lh;
        unsigned mbits = (EPI == EPI_MASK && bits16) ? bits16[bidx] : 0u;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = rbase + (reg & 3) + 8 * (reg >> 2);
          if (row >= R) continue;
          float val = acc[t][u][reg] * inv;
          if (EPI == EPI_RELU) {
            val = fmaxf(val + bcol, 0.f);
            mbits |= (val > 0.f ? 1u : 0u) << reg;
          } else if (EPI == EPI_MASK) {
            const bool on = bits16 ? ((mbits >> reg) & 1u) != 0u : Xmask[(size_t)row * ldm + col] > 0.f;
            val = on ? val : 0.f;
          }
          C[(size_t)row * ldc + col] = val;
          vmax = fmaxf(vmax, fabsf(val));
        }
        if (EPI == EPI_RELU && bits16) bits16[bidx] = (uint16_t)mbits;
      }
  }
  pp_record_max(c_max, vmax);
}
