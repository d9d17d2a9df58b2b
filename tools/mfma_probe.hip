// Calibration probe (not product): achievable v_mfma_f32_32x32x2_f32 rate for the GEMM inner-loop shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREADS>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ float lds[192 * 36];
  for (int i = threadIdx.x; i < 192 * 36; i += 256) lds[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5, wid = threadIdx.x >> 6;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b0 = a, b1 = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 32; kb += 8) {
      if (LDSREADS) {
        a = *reinterpret_cast<const float4*>(lds + ((wid >> 1) * 32 + l31) * 36 + kb + 4 * lh);
        b0 = *reinterpret_cast<const float4*>(lds + (64 + (wid & 1) * 64 + l31) * 36 + kb + 4 * lh);
        b1 = *reinterpret_cast<const float4*>(lds + (64 + (wid & 1) * 64 + 32 + l31) * 36 + kb + 4 * lh);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
    }
    if (LDSREADS == 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, int wgs, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<V><<<wgs, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V><<<wgs, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)wgs * 4 * iters * 32 * 4096.0;
  printf("%-28s wgs=%5d iters=%d: %.1f us  %.1f TFLOP/s\n", name, wgs, iters, ms * 1e3, flop / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int wgs : {256, 512, 1024, 1280}) {
    run<0>("mfma only", wgs, 400, out);
    run<1>("mfma + ds_read_b128", wgs, 400, out);
    run<2>("mfma + ds_read + barrier", wgs, 400, out);
  }
  return 0;
}
