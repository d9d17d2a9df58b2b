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

// four values -> scaled hi | lo halves (see pp_split8 below for the instruction choice)
__device__ __forceinline__ void pp_split4(float4 x, float s, pp_half4& hi, pp_half4& lo) {
  const float x0 = x.x * s, x1 = x.y * s, x2 = x.z * s, x3 = x.w * s;
  unsigned h01, h23, l01, l23;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(x0), "v"(x1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(x2), "v"(x3));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(x2));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(x3));
  typedef unsigned pp_u2 __attribute__((ext_vector_type(2)));
  const pp_u2 hv = {h01, h23}, lv = {l01, l23};
  hi = __builtin_bit_cast(pp_half4, hv);
  lo = __builtin_bit_cast(pp_half4, lv);
}

// eight values -> scaled hi | lo halves, three instructions per PAIR after the scaling: hi pair by v_cvt_pk_f16_f32, lo = x - hi by
// v_fma_mixlo / mixhi_f16 (fp32 fma on the f16 hi half, rounded once into the packed result).  Inline assembly: the compiler
// expands the same arithmetic into five instructions per element (and, with the SLP vectoriser on, into v_pk_*_f32, which cost 16
// cycles each beside MFMAs).
__device__ __forceinline__ void pp_split8(const float (&v)[8], float s, pp_half8& h, pp_half8& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x0 = v[2 * k] * s, x1 = v[2 * k + 1] * s;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh[k]) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ll[k]) : "v"(hh[k]), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ll[k]) : "v"(hh[k]), "v"(x1));
  }
  typedef unsigned pp_u4 __attribute__((ext_vector_type(4)));
  const pp_u4 hv = {hh[0], hh[1], hh[2], hh[3]}, lv = {ll[0], ll[1], ll[2], ll[3]};
  h = __builtin_bit_cast(pp_half8, hv);
  l = __builtin_bit_cast(pp_half8, lv);
}

// v >= 0 ; at most one atomic per wavefront, and none when the slot already holds at least v: same-address atomics cost
// ~80 ns each, in sequence (k_nerf_wmax: 15 us with 128 per slot, 28 us with 256, 5 us with 16), the slots only ever grow, and
// an L2-served read that returns an older (smaller) value merely issues an atomic that was not needed
__device__ __forceinline__ void pp_record_max_lane(float* slot, float v) {
  const unsigned bits = __float_as_uint(v);
  if (__hip_atomic_load(reinterpret_cast<unsigned int*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits)
    atomicMax(reinterpret_cast<unsigned int*>(slot), bits);
}
__device__ __forceinline__ void pp_record_max(float* slot, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0 && slot) pp_record_max_lane(slot, v);
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

// ------------------------------------------------------------------------------------------------------------------
// Split-precision weight gradient:  Wbar[n][k] += sum_r Y[r][n] X[r][k]  (one 128 x 128 block per blockIdx.y, row splits
// along blockIdx.x, as k_gemm_tn), three fp16 products per fp32 product.  The reduction runs over ROWS, so an MFMA operand
// fragment is 8 consecutive rows of one column: a thread therefore loads 8 consecutive rows x 4 columns of the chunk (512-byte
// coalesced row segments), splits them and writes each column's 8 rows as ONE 16-byte value into a column-major LDS image
// T[column][64 rows + 8] - the transpose costs nothing but the store pattern.  Column stride 144 B; the eight 16-byte row
// blocks of a column are XOR-swizzled with (column / 4) & 7, which brings the stores (lanes 4 columns apart) to the 4 passes
// a 16-byte store of 32 lanes needs anyway and leaves the fragment reads (ds_read_b128, consecutive columns) at most 2-way.
// ------------------------------------------------------------------------------------------------------------------
#define LDC 72        // halfs per LDS column: 64 rows + 8 (144 B)

static __global__ __launch_bounds__(256) void k_gemm_tn_split(const float* __restrict__ Y_, int ldy, const float* __restrict__ X_, int ldx,
                                                       int Kx_, float* __restrict__ Wbar_, int ldwb, float* __restrict__ bbar_,
                                                       const int32_t* __restrict__ count, int rcap,
                                                       const float* __restrict__ y_max, const float* __restrict__ x_max) {
  constexpr int CH = 64;
  __shared__ _Float16 Yh[128 * LDC], Yl[128 * LDC], Xh[128 * LDC], Xl[128 * LDC];
  const int nkb = (Kx_ + 127) >> 7;
  const int nb = blockIdx.y / nkb, kb = blockIdx.y - nb * nkb;
  const float* __restrict__ Y = Y_ + nb * 128;
  const float* __restrict__ X = X_ + kb * 128;
  const int Kx = min(128, Kx_ - kb * 128);
  float* __restrict__ Wbar = Wbar_ + (size_t)nb * 128 * ldwb + kb * 128;
  float* __restrict__ bbar = (bbar_ && kb == 0) ? bbar_ + nb * 128 : nullptr;
  const int R = min(count[0], rcap);
  const int rows_per_wg = ((R + (int)gridDim.x - 1) / (int)gridDim.x + CH - 1) / CH * CH;
  const int rb = blockIdx.x * rows_per_wg;
  if (rb >= R) return;
  const int re = min(rb + rows_per_wg, R);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const float sY = pp_split_scale(y_max[0]), sX = pp_split_scale(x_max[0]);
  const int c4 = tid & 31, rblk = tid >> 5;                 // this thread: columns 4 c4 .. 4 c4 + 3, rows 8 rblk .. 8 rblk + 7
  const int kx4 = Kx >> 2;
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  float4 ry[8], rx[8];
  auto load_rows = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gr = r0 + rblk * 8 + i;
      ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rx[i] = ry[i];
      if (gr < re) {
        ry[i] = *reinterpret_cast<const float4*>(Y + (size_t)gr * ldy + c4 * 4);
        if (c4 < kx4) rx[i] = *reinterpret_cast<const float4*>(X + (size_t)gr * ldx + c4 * 4);
      }
    }
  };
  auto store_col = [&](const float4 (&r)[8], int j, float s, _Float16* Th, _Float16* Tl) {
    pp_half8 h, l;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = j == 0 ? r[i].x : j == 1 ? r[i].y : j == 2 ? r[i].z : r[i].w;
    pp_split8(v, s, h, l);
    const int o = (c4 * 4 + j) * LDC + ((rblk ^ (c4 & 7)) * 8);
    *reinterpret_cast<pp_half8*>(Th + o) = h;
    *reinterpret_cast<pp_half8*>(Tl + o) = l;
  };
  load_rows(rb);
  for (int r0 = rb; r0 < re; r0 += CH) {
    store_col(ry, 0, sY, Yh, Yl); store_col(ry, 1, sY, Yh, Yl); store_col(ry, 2, sY, Yh, Yl); store_col(ry, 3, sY, Yh, Yl);
    store_col(rx, 0, sX, Xh, Xl); store_col(rx, 1, sX, Xh, Xl); store_col(rx, 2, sX, Xh, Xl); store_col(rx, 3, sX, Xh, Xl);
    if (bbar) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { bsum[0] += ry[i].x; bsum[1] += ry[i].y; bsum[2] += ry[i].z; bsum[3] += ry[i].w; }
    }
    __syncthreads();
    if (r0 + CH < re) load_rows(r0 + CH);
    if (wc * 64 < Kx) {
#pragma unroll
      for (int ks = 0; ks < CH; ks += 16) {
        pp_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = wr * 64 + t * 32 + l31;
          const int o = col * LDC + ((((ks >> 3) + lh) ^ ((col >> 2) & 7)) * 8);
          ah[t] = *reinterpret_cast<const pp_half8*>(Yh + o);
          al[t] = *reinterpret_cast<const pp_half8*>(Yl + o);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = wc * 64 + u * 32 + l31;
          const int o = col * LDC + ((((ks >> 3) + lh) ^ ((col >> 2) & 7)) * 8);
          bh[u] = *reinterpret_cast<const pp_half8*>(Xh + o);
          bl[u] = *reinterpret_cast<const pp_half8*>(Xl + o);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  const float inv = 1.0f / (sY * sX);
  if (wc * 64 < Kx) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = wc * 64 + u * 32 + l31;
        if (k >= Kx) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          atomicAdd(&Wbar[(size_t)n * ldwb + k], acc[t][u][reg] * inv);
        }
      }
  }
  if (bbar) {                              // 8 row blocks x 128 columns of partial sums -> one atomic per column
    float* red = reinterpret_cast<float*>(Yh);            // the operand images are dead (last chunk ended with a barrier)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rblk * 128 + c4 * 4 + j] = bsum[j];
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += red[q * 128 + tid];
      atomicAdd(&bbar[tid], sum);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same product with ROW-MAJOR LDS images and hardware-transposed fragment reads (gfx950 ds_read_b64_tr_b16).
// k_gemm_tn_split transposes on the way INTO LDS: 16-byte stores of eight rows of one column, 4-way bank conflicts by
// construction - on the LDS store path (13 cycles per conflict-free ds_write_b128 already, MI355X_MICROARCH.md LDS table) that
// is ~32 cycles per wave-instruction, 64 of them per chunk and work-group: more LDS time than the chunk's matrix instructions
// (PMC: matrix pipe busy 25 %).  Here a thread stores what it loaded - four consecutive columns of a row, hi and lo, as 8-byte
// conflict-free stores into [64 rows][128 halfs] images with 256-byte rows whose 16-byte chunks are XOR-swizzled
// (cdna_hip_programming.md T10, image (b)) - and a fragment (eight consecutive rows of one column per lane) is two transposed
// reads of 4 rows x 16 columns per 16-lane group.
// ------------------------------------------------------------------------------------------------------------------
#ifndef TN_DBG
#define TN_DBG 0      // experiments only (k_gemm_tn_tr), bit mask: 1 no row fetches in the loop, 2 fragments read once per chunk
#endif
__device__ __forceinline__ int tn_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

#if defined(TN_TIMERS) && defined(PP_NERF_TU)      // phase timers (experiments, scene translation unit only): wave 0 of every work-group sums s_memtime deltas per phase
#define TN_TIMERS_ON 1
__device__ unsigned long long g_tn_t[8];
extern "C" int pp_debug_read_tn_timers(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tn_t), sizeof(g_tn_t)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tn_t), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
static __global__ __launch_bounds__(256, 2) void k_gemm_tn_tr(const float* __restrict__ Y_, int ldy, const float* __restrict__ X_, int ldx,
                                                          int Kx_, float* __restrict__ Wbar_, int ldwb, float* __restrict__ bbar_,
                                                          const int32_t* __restrict__ count, int rcap,
                                                          const float* __restrict__ y_max, const float* __restrict__ x_max) {
  constexpr int CH = 64;
  __shared__ __attribute__((aligned(1024))) unsigned char img[4 * CH * 256];       // Yh | Yl | Xh | Xl
  unsigned char* const Yh = img;
  unsigned char* const Yl = img + CH * 256;
  unsigned char* const Xh = img + 2 * CH * 256;
  unsigned char* const Xl = img + 3 * CH * 256;
  const int nkb = (Kx_ + 127) >> 7;
  const int nb = blockIdx.y / nkb, kb = blockIdx.y - nb * nkb;
  const float* __restrict__ Y = Y_ + nb * 128;
  const float* __restrict__ X = X_ + kb * 128;
  const int Kx = min(128, Kx_ - kb * 128);
  float* __restrict__ Wbar = Wbar_ + (size_t)nb * 128 * ldwb + kb * 128;
  float* __restrict__ bbar = (bbar_ && kb == 0) ? bbar_ + nb * 128 : nullptr;
  const int R = min(count[0], rcap);
  const int rows_per_wg = ((R + (int)gridDim.x - 1) / (int)gridDim.x + CH - 1) / CH * CH;
  const int rb = blockIdx.x * rows_per_wg;
  if (rb >= R) return;
  const int re = min(rb + rows_per_wg, R);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const float sY = pp_split_scale(y_max[0]), sX = pp_split_scale(x_max[0]);
  const int c4 = tid & 31, rblk = tid >> 5;                 // this thread: columns 4 c4 .. 4 c4 + 3, rows 8 rblk .. 8 rblk + 7
  const int kx4 = Kx >> 2;
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // rows i0 .. i1 - 1 of this thread's eight of the chunk at r0: clamped addresses, no branches (rows past the range and columns
  // past Kx are zeroed when they are converted) - the fetches of the next chunk are issued in FOUR pieces between the matrix
  // instructions of this one.  All sixteen in front of them cost 2 k ticks per chunk (phase timers with the fetches removed:
  // matrix phase 4.5 k -> 2.5 k): 8 wavefronts x 16 KB pass the CU's 64 B / clock address-and-data path in ~2 k cycles, and a
  // wavefront's matrix instructions cannot issue before the fetches in front of them have
  float4 ry[8], rx[8];
  auto load_rows = [&](int r0, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < i0 || i >= i1) continue;
      const int gr = min(r0 + rblk * 8 + i, R - 1);
      ry[i] = *reinterpret_cast<const float4*>(Y + (size_t)gr * ldy + c4 * 4);
      if (TN_DBG & 4) rx[i] = ry[i];                        // experiment: half the bytes
      else rx[i] = *reinterpret_cast<const float4*>(X + (size_t)gr * ldx + (c4 < kx4 ? c4 : 0) * 4);
    }
  };
  // transposed fragment: lane (group g = lane / 16, q = (lane & 15) / 4, p = lane & 3) addresses row q, columns 4 p .. 4 p + 3 of its
  // group's 4 x 16 block and receives column (lane & 15) of the block's four rows; two blocks = the eight rows of the lane's half
  typedef __fp16 tn_h4 __attribute__((vector_size(8)));
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  auto frag = [&](const unsigned char* plane, int cb, int ks) -> pp_half8 {        // cb: first of the tile's 32 columns, ks: first of its 16 rows
    const int ch = ((cb + 16 * (fg & 1)) >> 3) + (fp >> 1);
    const int row = ks + 8 * (fg >> 1) + fq;
    const tn_h4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tn_h4*)(plane + tn_off(row, ch) + 8 * (fp & 1)));
    const tn_h4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tn_h4*)(plane + tn_off(row + 4, ch) + 8 * (fp & 1)));
    typedef __fp16 tn_h8 __attribute__((vector_size(16)));
    const tn_h8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(pp_half8, v);
  };
#ifdef TN_TIMERS_ON
  unsigned long long tsum[8] = {0}, tprev = __builtin_readcyclecounter();
#define TN_TICK(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tsum[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define TN_TICK(i) do {} while (0)
#endif
  load_rows(rb, 0, 8);
  for (int r0 = rb; r0 < re; r0 += CH) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TN_TICK(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = tn_off(rblk * 8 + i, c4 >> 1) + 8 * (c4 & 1);
      const float ky = (r0 + rblk * 8 + i < re) ? 1.f : 0.f, kx = (c4 < kx4) ? ky : 0.f;
      ry[i].x *= ky; ry[i].y *= ky; ry[i].z *= ky; ry[i].w *= ky;
      rx[i].x *= kx; rx[i].y *= kx; rx[i].z *= kx; rx[i].w *= kx;
      pp_half4 h, l;
      pp_split4(ry[i], sY, h, l);
      *reinterpret_cast<pp_half4*>(Yh + o) = h;
      *reinterpret_cast<pp_half4*>(Yl + o) = l;
      pp_split4(rx[i], sX, h, l);
      *reinterpret_cast<pp_half4*>(Xh + o) = h;
      *reinterpret_cast<pp_half4*>(Xl + o) = l;
    }
    if (bbar) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { bsum[0] += ry[i].x; bsum[1] += ry[i].y; bsum[2] += ry[i].z; bsum[3] += ry[i].w; }
    }
    TN_TICK(1);
    __syncthreads();
    TN_TICK(2);
    if (wc * 64 < Kx) {                                       // (uniform per wavefront: the transposed reads need all 64 lanes)
      // fragments of the next 16 rows are on their way while the matrix instructions of these 16 issue (left in one loop
      // body, the compiler reads a step's ten fragments only after the previous step's last matrix instruction: four exposed
      // LDS round trips per chunk - phase timers: 4.4 k ticks per chunk of 48 matrix instructions)
      pp_half8 fa[2][4], fb[2][4];                          // [parity][tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
      auto fetch = [&](int par, int ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) { fa[par][2 * t] = frag(Yh, wr * 64 + t * 32, ks); fa[par][2 * t + 1] = frag(Yl, wr * 64 + t * 32, ks); }
#pragma unroll
        for (int u = 0; u < 2; ++u) { fb[par][2 * u] = frag(Xh, wc * 64 + u * 32, ks); fb[par][2 * u + 1] = frag(Xl, wc * 64 + u * 32, ks); }
      };
      fetch(0, 0);
#pragma unroll
      for (int s4 = 0; s4 < CH / 16; ++s4) {
        const int par = s4 & 1;
        if (s4 + 1 < CH / 16 && !(TN_DBG & 2)) fetch(par ^ 1, (s4 + 1) * 16);
        if (TN_DBG & 2) { for (int i = 0; i < 4; ++i) { fa[par ^ 1][i] = fa[par][i]; fb[par ^ 1][i] = fb[par][i]; } }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][2 * t + 1], fb[par][2 * u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][2 * t], fb[par][2 * u + 1], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][2 * t], fb[par][2 * u], acc[t][u], 0, 0, 0);
          }
          if (t == 0 && !(TN_DBG & 1)) {                    // a quarter of the next chunk's rows behind the first six matrix instructions
            __builtin_amdgcn_sched_barrier(0);
            load_rows(r0 + CH, 2 * s4, 2 * s4 + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else if (!(TN_DBG & 1)) {
      load_rows(r0 + CH, 0, 8);                               // a wavefront without columns in this block still stages its rows
    }
    TN_TICK(3);
    __syncthreads();
    TN_TICK(4);
  }
  const float inv = 1.0f / (sY * sX);
  if (wc * 64 < Kx) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = wc * 64 + u * 32 + l31;
        if (k >= Kx) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          atomicAdd(&Wbar[(size_t)n * ldwb + k], acc[t][u][reg] * inv);
        }
      }
  }
  if (bbar) {                              // 8 row blocks x 128 columns of partial sums -> one atomic per column
    float* red = reinterpret_cast<float*>(img);           // the operand images are dead (last chunk ended with a barrier)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rblk * 128 + c4 * 4 + j] = bsum[j];
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += red[q * 128 + tid];
      atomicAdd(&bbar[tid], sum);
    }
  }
#ifdef TN_TIMERS_ON
  TN_TICK(5);
  if (tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_tn_t[i], tsum[i]);
#endif
#undef TN_TICK
}

// ------------------------------------------------------------------------------------------------------------------
// Self-scaling variant of k_gemm_tn_split for callers that have no operand maxima (the object branch's layer-fused kernels
// do not record any):  Wbar[n][k] += sum_r Y[r][n] X[r][k],  n < 128, k < Kx <= 128, Y and X with 128 / ldx floats per row.
// Every 64-row chunk reduces max|Y|, max|X| of what it is about to stage (registers -> wave shuffle -> 8 floats of LDS) and
// converts with the RUNNING MINIMUM of the chunk scales; when a chunk lowers the scale, the accumulators are multiplied by
// the (power-of-two, hence exact) ratio first - the online-rescaling trick of streaming softmax.  No chunk is ever converted
// with a scale above its own safe one, so nothing overflows, and the result carries the precision of a single scale derived
// from the work-group's whole row range.  One extra barrier per chunk.  gridDim.x = row splits.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pp_split_scale_or_huge(float mx) {
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.7e38f;       // an all-zero chunk puts no constraint on the scale
  int e;
  frexpf(mx, &e);
  return ldexpf(1.f, 15 - e);
}

static __global__ __launch_bounds__(256) void k_gemm_tn_split_auto(const float* __restrict__ Y, const float* __restrict__ X, int ldx, int Kx,
                                                            float* __restrict__ Wbar, int ldwb,
                                                            const int32_t* __restrict__ count, int rmul, int rcap) {
  constexpr int CH = 64;
  __shared__ _Float16 Yh[128 * LDC], Yl[128 * LDC], Xh[128 * LDC], Xl[128 * LDC];
  __shared__ float mxs[4][2];
  const int R = min(count[0] * rmul, rcap);
  const int rows_per_wg = ((R + (int)gridDim.x - 1) / (int)gridDim.x + CH - 1) / CH * CH;
  const int rb = blockIdx.x * rows_per_wg;
  if (rb >= R) return;
  const int re = min(rb + rows_per_wg, R);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid & 31, rblk = tid >> 5;
  const int kx4 = Kx >> 2;
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float4 ry[8], rx[8];
  auto load_rows = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gr = r0 + rblk * 8 + i;
      ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rx[i] = ry[i];
      if (gr < re) {
        ry[i] = *reinterpret_cast<const float4*>(Y + (size_t)gr * 128 + c4 * 4);
        if (c4 < kx4) rx[i] = *reinterpret_cast<const float4*>(X + (size_t)gr * ldx + c4 * 4);
      }
    }
  };
  auto store_col = [&](const float4 (&r)[8], int j, float s, _Float16* Th, _Float16* Tl) {
    pp_half8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = (j == 0 ? r[i].x : j == 1 ? r[i].y : j == 2 ? r[i].z : r[i].w) * s;
      const _Float16 hh = (_Float16)x;
      h[i] = hh;
      l[i] = (_Float16)(x - (float)hh);
    }
    const int o = (c4 * 4 + j) * LDC + ((rblk ^ (c4 & 7)) * 8);
    *reinterpret_cast<pp_half8*>(Th + o) = h;
    *reinterpret_cast<pp_half8*>(Tl + o) = l;
  };
  float SY = 1.7e38f, SX = 1.7e38f;             // running (minimum) scales; "huge" = not constrained yet
  bool any = false;
  load_rows(rb);
  for (int r0 = rb; r0 < re; r0 += CH) {
    float my = 0.f, mxv = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      my = fmaxf(my, fmaxf(fmaxf(fabsf(ry[i].x), fabsf(ry[i].y)), fmaxf(fabsf(ry[i].z), fabsf(ry[i].w))));
      mxv = fmaxf(mxv, fmaxf(fmaxf(fabsf(rx[i].x), fabsf(rx[i].y)), fmaxf(fabsf(rx[i].z), fabsf(rx[i].w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { my = fmaxf(my, __shfl_xor(my, o, 64)); mxv = fmaxf(mxv, __shfl_xor(mxv, o, 64)); }
    if (lane == 0) { mxs[wid][0] = my; mxs[wid][1] = mxv; }
    __syncthreads();                            // chunk maxima visible; previous chunk's fragment reads are done
    my = fmaxf(fmaxf(mxs[0][0], mxs[1][0]), fmaxf(mxs[2][0], mxs[3][0]));
    mxv = fmaxf(fmaxf(mxs[0][1], mxs[1][1]), fmaxf(mxs[2][1], mxs[3][1]));
    const float nSY = fminf(SY, pp_split_scale_or_huge(my)), nSX = fminf(SX, pp_split_scale_or_huge(mxv));
    if (any && (nSY != SY || nSX != SX)) {      // uniform over the work-group: rescale what has been accumulated so far
      const float f = (nSY / SY) * (nSX / SX);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][u][i] *= f;
    }
    SY = nSY; SX = nSX;
    const float sy = (SY < 1.0e38f) ? SY : 1.f, sx = (SX < 1.0e38f) ? SX : 1.f;   // an operand that was all zero so far: any scale
    any = any || (SY < 1.0e38f && SX < 1.0e38f);    // from here on the accumulators may be non-zero (both scales are set)
    store_col(ry, 0, sy, Yh, Yl); store_col(ry, 1, sy, Yh, Yl); store_col(ry, 2, sy, Yh, Yl); store_col(ry, 3, sy, Yh, Yl);
    store_col(rx, 0, sx, Xh, Xl); store_col(rx, 1, sx, Xh, Xl); store_col(rx, 2, sx, Xh, Xl); store_col(rx, 3, sx, Xh, Xl);
    __syncthreads();
    if (r0 + CH < re) load_rows(r0 + CH);
    if (wc * 64 < Kx) {
#pragma unroll
      for (int ks = 0; ks < CH; ks += 16) {
        pp_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = wr * 64 + t * 32 + l31;
          const int o = col * LDC + ((((ks >> 3) + lh) ^ ((col >> 2) & 7)) * 8);
          ah[t] = *reinterpret_cast<const pp_half8*>(Yh + o);
          al[t] = *reinterpret_cast<const pp_half8*>(Yl + o);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = wc * 64 + u * 32 + l31;
          const int o = col * LDC + ((((ks >> 3) + lh) ^ ((col >> 2) & 7)) * 8);
          bh[u] = *reinterpret_cast<const pp_half8*>(Xh + o);
          bl[u] = *reinterpret_cast<const pp_half8*>(Xl + o);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
          }
      }
    }
  }
  if (!any || !(wc * 64 < Kx)) return;          // nothing but zeros seen, or a wavefront without real columns
  const float invy = 1.0f / SY, invx = 1.0f / SX;          // applied one after the other: their product may leave the float range
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = wc * 64 + u * 32 + l31;
      if (k >= Kx) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        atomicAdd(&Wbar[(size_t)n * ldwb + k], acc[t][u][reg] * invy * invx);
      }
    }
}
