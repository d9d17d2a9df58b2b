// Probe 2: N independent plain VALU instructions per MFMA (v_mfma_f32_32x32x16_f16), ONE or TWO accumulator chains, one wavefront
// per SIMD; and the same with 2 wavefronts per SIMD (512 threads).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_valu_probe2 tools/mfma_valu_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int CHAINS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, long long* cyc) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (CHAINS == 1 || (u & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[i % 16]));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int N, int CHAINS, int THREADS>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<N, CHAINS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, CHAINS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("chains=%d waves/SIMD=%d N=%2d: %.1f ns per MFMA-slot of one wave, %.1f ticks\n", CHAINS, THREADS / 256, N, ms * 1e6 / (iters * 8), (double)c / (iters * 8));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0, 1, 256>(out, cyc); run<4, 1, 256>(out, cyc); run<5, 1, 256>(out, cyc); run<6, 1, 256>(out, cyc); run<8, 1, 256>(out, cyc); run<12, 1, 256>(out, cyc); run<16, 1, 256>(out, cyc);
  run<0, 2, 256>(out, cyc); run<4, 2, 256>(out, cyc); run<6, 2, 256>(out, cyc); run<8, 2, 256>(out, cyc); run<12, 2, 256>(out, cyc); run<16, 2, 256>(out, cyc);
  run<0, 1, 512>(out, cyc); run<4, 1, 512>(out, cyc); run<8, 1, 512>(out, cyc); run<12, 1, 512>(out, cyc); run<16, 1, 512>(out, cyc);
  return 0;
}
