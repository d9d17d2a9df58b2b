// Roadmap probe (not product): C[M][N] = A[M][K] . W[N][K]^T with fp32 operands in memory, computed
//   (a) exactly in fp32 on v_mfma_f32_32x32x2_f32 (the product path's arithmetic), and
//   (b) as three fp16 products  hi.hi + hi.lo + lo.hi  on v_mfma_f32_32x32x16_f16, operands split on the fly while they are
//       staged into LDS:  x * s = hi + lo,  hi = fp16(x * s),  lo = fp16(x * s - hi),  s a per-tensor power of two.
//   (c) as (b) with the weights pre-split into fp16 planes, 128 x 128 and 128 x 256 tiles.
// Prints time, algorithmic TFLOP/s and the error against an fp64 host product on the first 256 rows.  Measured on MI355X at
// M = 131072, N = K = 256: (a) 284 us, 2.04e-7 rel rms; (b) 84 us, 1.90e-7; (c) 132 / 128 us - SLOWER than converting the fp32
// weights on the fly: 64-byte rows per plane halve the bytes per load request, the kernel is bound by load requests and
// LDS traffic, not by the conversion arithmetic (so: keep 128-byte rows per request, e.g. hi / lo interleaved per 16 k).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/split_gemm_probe tools/split_gemm_probe.hip && tools/_build/split_gemm_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define KC 32
#define LDH 40        // halfs per LDS row (80 B: 16-byte aligned fragments, rows skewed by 20 banks)
#define LDF 36        // floats per LDS row of the fp32 variant

// ---------------------------------------------------------------------------------------------- (a) exact fp32
__global__ __launch_bounds__(256) void k_fp32(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                              int M, int N, int K) {
  __shared__ float As[BM * LDF];
  __shared__ float Bs[BN * LDF];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * BM, c0 = blockIdx.y * BN;
  f32x16 acc[2][2];
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float4 ra[4], rw[4];
  auto load = [&](int k0) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(r0 + row) * K + k0 + c4 * 4);
      rw[i] = *reinterpret_cast<const float4*>(W + (size_t)(c0 + row) * K + k0 + c4 * 4);
    }
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      *reinterpret_cast<float4*>(As + (e >> 3) * LDF + (e & 7) * 4) = ra[i];
      *reinterpret_cast<float4*>(Bs + (e >> 3) * LDF + (e & 7) * 4) = rw[i];
    }
    __syncthreads();
    if (k0 + KC < K) load(k0 + KC);
#pragma unroll
    for (int kb = 0; kb < KC; kb += 8) {
      float4 a[2], b[2];
      for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const float4*>(As + (wr * 64 + t * 32 + l31) * LDF + kb + 4 * lh);
      for (int u = 0; u < 2; ++u) b[u] = *reinterpret_cast<const float4*>(Bs + (wc * 64 + u * 32 + l31) * LDF + kb + 4 * lh);
      for (int t = 0; t < 2; ++t)
        for (int u = 0; u < 2; ++u) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  for (int t = 0; t < 2; ++t)
    for (int u = 0; u < 2; ++u)
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh, col = c0 + wc * 64 + u * 32 + l31;
        C[(size_t)row * N + col] = acc[t][u][reg];
      }
}

// ---------------------------------------------------------------------------------------------- (b) fp16 x 3
__device__ __forceinline__ void split4(float4 x, float s, half4& hi, half4& lo) {
  const float v[4] = {x.x * s, x.y * s, x.z * s, x.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

__global__ __launch_bounds__(256) void k_split(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                               int M, int N, int K, float sA, float sW) {
  __shared__ _Float16 Ah[BM * LDH], Al[BM * LDH], Bh[BN * LDH], Bl[BN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * BM, c0 = blockIdx.y * BN;
  f32x16 acc[2][2];
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float4 ra[4], rw[4];
  auto load = [&](int k0) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(r0 + row) * K + k0 + c4 * 4);
      rw[i] = *reinterpret_cast<const float4*>(W + (size_t)(c0 + row) * K + k0 + c4 * 4);
    }
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      half4 h, l;
      split4(ra[i], sA, h, l);
      *reinterpret_cast<half4*>(Ah + row * LDH + c4 * 4) = h;
      *reinterpret_cast<half4*>(Al + row * LDH + c4 * 4) = l;
      split4(rw[i], sW, h, l);
      *reinterpret_cast<half4*>(Bh + row * LDH + c4 * 4) = h;
      *reinterpret_cast<half4*>(Bl + row * LDH + c4 * 4) = l;
    }
    __syncthreads();
    if (k0 + KC < K) load(k0 + KC);
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
      half8 ah[2], al[2], bh[2], bl[2];
      for (int t = 0; t < 2; ++t) {
        const int o = (wr * 64 + t * 32 + l31) * LDH + ks + 8 * lh;
        ah[t] = *reinterpret_cast<const half8*>(Ah + o);
        al[t] = *reinterpret_cast<const half8*>(Al + o);
      }
      for (int u = 0; u < 2; ++u) {
        const int o = (wc * 64 + u * 32 + l31) * LDH + ks + 8 * lh;
        bh[u] = *reinterpret_cast<const half8*>(Bh + o);
        bl[u] = *reinterpret_cast<const half8*>(Bl + o);
      }
      for (int t = 0; t < 2; ++t)
        for (int u = 0; u < 2; ++u) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);   // small terms first
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  const float inv = 1.0f / (sA * sW);
  for (int t = 0; t < 2; ++t)
    for (int u = 0; u < 2; ++u)
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh, col = c0 + wc * 64 + u * 32 + l31;
        C[(size_t)row * N + col] = acc[t][u][reg] * inv;
      }
}

// (c) as (b) with the weights pre-split once into fp16 hi / lo planes (what a per-step pack kernel would produce)
__global__ void k_presplit(const float* __restrict__ W, _Float16* __restrict__ Wh, _Float16* __restrict__ Wl, int n, float s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = W[i] * s;
  const _Float16 h = (_Float16)v;
  Wh[i] = h;
  Wl[i] = (_Float16)(v - (float)h);
}

template <int TBN>
__global__ __launch_bounds__(256) void k_split_w(const float* __restrict__ A, const _Float16* __restrict__ Wh,
                                                 const _Float16* __restrict__ Wl, float* __restrict__ C, int M, int N, int K,
                                                 float sA, float sW) {
  constexpr int TNW = TBN / 64, NBW = TBN / 64;          // uint4 (8 halfs) per thread, plane and chunk: TBN*32*2/16/256
  __shared__ _Float16 Ah[BM * LDH], Al[BM * LDH], Bh[TBN * LDH], Bl[TBN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * BM, c0 = blockIdx.y * TBN;
  f32x16 acc[2][TNW];
  for (int t = 0; t < 2; ++t) for (int u = 0; u < TNW; ++u) for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float4 ra[4];
  uint4 rh[NBW], rl[NBW];
  auto load = [&](int k0) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(r0 + row) * K + k0 + c4 * 4);
    }
    for (int i = 0; i < NBW; ++i) {
      const int e = tid + i * 256, row = e >> 2, c8 = e & 3;            // 4 x 16 B per row of 32 halfs
      rh[i] = *reinterpret_cast<const uint4*>(Wh + (size_t)(c0 + row) * K + k0 + c8 * 8);
      rl[i] = *reinterpret_cast<const uint4*>(Wl + (size_t)(c0 + row) * K + k0 + c8 * 8);
    }
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, c4 = e & 7;
      half4 h, l;
      split4(ra[i], sA, h, l);
      *reinterpret_cast<half4*>(Ah + row * LDH + c4 * 4) = h;
      *reinterpret_cast<half4*>(Al + row * LDH + c4 * 4) = l;
    }
    for (int i = 0; i < NBW; ++i) {
      const int e = tid + i * 256, row = e >> 2, c8 = e & 3;
      *reinterpret_cast<uint4*>(Bh + row * LDH + c8 * 8) = rh[i];
      *reinterpret_cast<uint4*>(Bl + row * LDH + c8 * 8) = rl[i];
    }
    __syncthreads();
    if (k0 + KC < K) load(k0 + KC);
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
      half8 ah[2], al[2], bh[TNW], bl[TNW];
      for (int t = 0; t < 2; ++t) {
        const int o = (wr * 64 + t * 32 + l31) * LDH + ks + 8 * lh;
        ah[t] = *reinterpret_cast<const half8*>(Ah + o);
        al[t] = *reinterpret_cast<const half8*>(Al + o);
      }
      for (int u = 0; u < TNW; ++u) {
        const int o = (wc * (32 * TNW) + u * 32 + l31) * LDH + ks + 8 * lh;
        bh[u] = *reinterpret_cast<const half8*>(Bh + o);
        bl[u] = *reinterpret_cast<const half8*>(Bl + o);
      }
      for (int t = 0; t < 2; ++t)
        for (int u = 0; u < TNW; ++u) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  const float inv = 1.0f / (sA * sW);
  for (int t = 0; t < 2; ++t)
    for (int u = 0; u < TNW; ++u)
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh, col = c0 + wc * (32 * TNW) + u * 32 + l31;
        C[(size_t)row * N + col] = acc[t][u][reg] * inv;
      }
}

static float pow2_scale(const std::vector<float>& v, float target) {
  float mx = 0.f;
  for (float x : v) mx = fmaxf(mx, fabsf(x));
  return exp2f(floorf(log2f(target / mx)));
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 131072, N = 256, K = 256;
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX; };
  auto gauss = [&] { return sqrtf(-2.f * logf(rnd() + 1e-12f)) * cosf(6.2831853f * rnd()); };
  for (auto& x : hA) { float g = gauss(); x = g > 0.f ? g : 0.f; }                 // post-ReLU activations
  for (auto& x : hW) x = 0.108f * (2.f * rnd() - 1.f) * 1.41421f;                  // Xavier-uniform, ReLU gain, 256 -> 256
  float *dA, *dW, *dC;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dW, hW.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  const float sA = pow2_scale(hA, 16384.f), sW = pow2_scale(hW, 16384.f);
  const int RR = 256;
  std::vector<double> ref((size_t)RR * N);
  for (int r = 0; r < RR; ++r)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)r * K + k] * (double)hW[(size_t)n * K + k];
      ref[(size_t)r * N + n] = s;
    }
  std::vector<float> hC((size_t)RR * N);
  dim3 g(M / BM, N / BN), b(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  _Float16 *dWh, *dWl;
  hipMalloc(&dWh, hW.size() * 2); hipMalloc(&dWl, hW.size() * 2);
  hipLaunchKernelGGL(k_presplit, dim3((N * K + 255) / 256), dim3(256), 0, 0, dW, dWh, dWl, N * K, sW);
  auto launch = [&](int variant) {
    if (variant == 0) hipLaunchKernelGGL(k_fp32, g, b, 0, 0, dA, dW, dC, M, N, K);
    else if (variant == 1) hipLaunchKernelGGL(k_split, g, b, 0, 0, dA, dW, dC, M, N, K, sA, sW);
    else if (variant == 2) hipLaunchKernelGGL((k_split_w<128>), g, b, 0, 0, dA, dWh, dWl, dC, M, N, K, sA, sW);
    else hipLaunchKernelGGL((k_split_w<256>), dim3(M / BM, N / 256), b, 0, 0, dA, dWh, dWl, dC, M, N, K, sA, sW);
  };
  const char* names[4] = {"fp32 MFMA 32x32x2", "fp16 x 3 MFMA 32x32x16", "fp16 x 3, W pre-split", "fp16 x 3, W pre-split, 128x256"};
  for (int variant = 0; variant < 4; ++variant) {
    for (int w = 0; w < 3; ++w) launch(variant);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) launch(variant);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double se = 0, sr = 0, mx = 0, mref = 0;
    for (size_t i = 0; i < hC.size(); ++i) {
      const double d = hC[i] - ref[i];
      se += d * d; sr += ref[i] * ref[i]; mx = fmax(mx, fabs(d)); mref = fmax(mref, fabs(ref[i]));
    }
    printf("%-32s %8.1f us  %6.1f TFLOP/s (algorithmic)  rel rms err %.3e  max abs err %.3e (max |ref| %.2f)  [sA=%g sW=%g]\n",
           names[variant], ms * 1e3, 2.0 * M * N * K / ms / 1e9, sqrt(se / sr), mx, mref,
           sA, sW);
  }
  return 0;
}
