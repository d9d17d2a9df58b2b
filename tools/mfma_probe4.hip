// Calibration probe (not product): dependent MFMA chains - one accumulator vs two interleaved, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float a = 1.f + threadIdx.x, b = 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (NACC == 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc0, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<256, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); probe<NACC><<<256, 256>>>(out, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-32s %.1f us  %.1f TFLOP/s\n", name, ms * 1e3, 256.0 * 4 * 1000 * 32 * 4096.0 / ms / 1e9);
}
int main() { float* out; hipMalloc(&out, 256 * 256 * 4); run<2>("two accumulators interleaved", out); run<1>("one accumulator chain", out); return 0; }
