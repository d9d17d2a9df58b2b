// Calibration probe (not product): the weight-gradient inner block (pp_wgrad_asm.inc) alone, with / without LDS-direct
// loads in flight, one work-group per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../poseprobe_amd/csrc/pp_wgrad_asm.inc"
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int DMA, int NACC>
__global__ __launch_bounds__(256) void probe(const float* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float Y0[64 * 128], X0[64 * 128], Y1[64 * 128], X1[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 256) { Y0[i] = 0.001f * (i & 7); X0[i] = 0.002f * (i & 3); Y1[i] = 0.f; X1[i] = 0.f; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5, wr = wid >> 1, wc = wid & 1;
  f32x16 acc[NACC][4];
  for (int n = 0; n < NACC; ++n) for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[n][k][i] = 0.f;
  const unsigned y = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Y0 + lh * 128 + 64 * wr + l31);
  const unsigned x = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(X0 + lh * 128 + 64 * wc + l31);
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
      const float* g = src + ((size_t)(blockIdx.x * 64 + (it & 63)) * 64) * 128;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = 16 * wid + 2 * i;
        __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(g + (rl + lh) * 128 + l31 * 4), PP_LDS_PTR(Y1 + rl * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(PP_GLOBAL_PTR(g + 8192 + (rl + lh) * 128 + l31 * 4), PP_LDS_PTR(X1 + rl * 128), 16, 0, 0);
      }
    }
    float a00, a10, b00, b10, a01, a11, b01, b11;
    f32x16(&c)[4] = acc[it % NACC];
    asm volatile(PP_WGRAD_BLOCK_NB2
                 : [c00] "+a"(c[0]), [c10] "+a"(c[1]), [c01] "+a"(c[2]), [c11] "+a"(c[3]), [a00] "=&v"(a00),
                   [a10] "=&v"(a10), [b00] "=&v"(b00), [b10] "=&v"(b10), [a01] "=&v"(a01), [a11] "=&v"(a11),
                   [b01] "=&v"(b01), [b11] "=&v"(b11)
                 : [y] "v"(y), [x] "v"(x)
                 : "memory");
    if (DMA) { __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[n][k][i];
  out[blockIdx.x * 256 + threadIdx.x] = s + Y1[threadIdx.x];
}

template <int DMA, int NACC>
void run(const char* name, const float* src, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 42;
  probe<DMA, NACC><<<256, 256>>>(src, out, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<DMA, NACC><<<256, 256>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = 256.0 * 4 * iters * 128 * 4096.0;
  printf("%-40s %.1f us  %.1f TFLOP/s  (%.0f clk@2.1GHz per block)\n", name, ms * 1e3, flop / ms / 1e9, ms * 1e-3 / iters * 2.1e9);
}

int main() {
  float *src, *out;
  hipMalloc(&src, (size_t)256 * 64 * 64 * 128 * 4 * 2);
  hipMemset(src, 0, (size_t)256 * 64 * 64 * 128 * 4 * 2);
  hipMalloc(&out, 256 * 256 * 4);
  run<0, 1>("block, 1 acc set, no DMA", src, out);
  run<0, 3>("block, 3 acc sets, no DMA", src, out);
  run<1, 1>("block, 1 acc set, DMA 64KB/step", src, out);
  run<1, 3>("block, 3 acc sets, DMA 64KB/step", src, out);
  return 0;
}
