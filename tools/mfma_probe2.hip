// Calibration probe 2 (not product): GEMM-tile structure = per chunk {6 ds_write_b128, barrier, 32 MFMA with 12 ds_read_b128, barrier}
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDT 36

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* out, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2][192 * LDT];
  int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, wid = tid >> 6;
  int wr = wid >> 1, wc = wid & 1;
  float4 r[6];
  for (int i = 0; i < 6; ++i) r[i] = *reinterpret_cast<const float4*>(src + ((tid + i * 256) % 1024) * 4);
  float s = 0.f;
  int buf = 0;
  for (int t = 0; t < tiles; ++t) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int c = 0; c < 4; ++c) {
      float* L = lds[VARIANT == 2 ? buf : 0];
      for (int i = 0; i < 6; ++i) { int e = tid + i * 256; *reinterpret_cast<float4*>(L + (e >> 3) * LDT + (e & 7) * 4) = r[i]; }
      __syncthreads();
      if (VARIANT == 1) for (int i = 0; i < 6; ++i) r[i] = *reinterpret_cast<const float4*>(src + ((tid + i * 256 + t * 7 + c) % 4096) * 4);
#pragma unroll
      for (int kb = 0; kb < 32; kb += 8) {
        float4 a = *reinterpret_cast<const float4*>(L + (wr * 32 + l31) * LDT + kb + 4 * lh);
        float4 b0 = *reinterpret_cast<const float4*>(L + (64 + wc * 64 + l31) * LDT + kb + 4 * lh);
        float4 b1 = *reinterpret_cast<const float4*>(L + (64 + wc * 64 + 32 + l31) * LDT + kb + 4 * lh);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
      }
      if (VARIANT == 2) buf ^= 1; else __syncthreads();
    }
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  }
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char* name, int wgs, int tiles, const float* src, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<V><<<wgs, 256>>>(src, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V><<<wgs, 256>>>(src, out, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)wgs * 4 * tiles * 128 * 4096.0;
  printf("%-44s wgs=%5d: %.1f us  %.1f TFLOP/s\n", name, wgs, ms * 1e3, flop / ms / 1e9);
}

int main() {
  float *out, *src;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&src, 4096 * 16 * 4);
  hipMemset(src, 0, 4096 * 16 * 4);
  for (int wgs : {512, 1280}) {
    run<0>("lds write + 2 barriers/chunk", wgs, 40, src, out);
    run<1>("+ global loads (L2 hits) per chunk", wgs, 40, src, out);
    run<2>("double-buffered LDS, 1 barrier/chunk", wgs, 40, src, out);
  }
  return 0;
}
