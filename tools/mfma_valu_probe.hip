// Probe: how many independent VALU instructions fit between dependent v_mfma_f32_32x32x16_f16 of ONE wavefront per SIMD before the
// loop slows down (co-issue of vector ALU work in the shadow of the matrix pipe).  Variants: plain fp32 VALU, v_pk_fma_f32,
// v_accvgpr_read, DPP moves, ds_write_b64.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_valu_probe tools/mfma_valu_probe.hip && tools/_build/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  __shared__ float lds[256 * 4];
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc, other;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; other[i] = threadIdx.x + i; }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[threadIdx.x * 2];
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      if (KIND == 0) {          // independent fp32 VALU
        if (N > 0) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v0));
        if (N > 1) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v1));
        if (N > 2) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v2));
        if (N > 3) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v3));
        if (N > 4) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v4));
        if (N > 5) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v5));
        if (N > 6) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v6));
        if (N > 7) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v7));
      } else if (KIND == 1) {   // accvgpr reads of another accumulator
        if (N > 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0) : "a"(other[0]));
        if (N > 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1) : "a"(other[1]));
        if (N > 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v2) : "a"(other[2]));
        if (N > 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v3) : "a"(other[3]));
        if (N > 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v4) : "a"(other[4]));
        if (N > 5) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v5) : "a"(other[5]));
        if (N > 6) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v6) : "a"(other[6]));
        if (N > 7) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v7) : "a"(other[7]));
      } else if (KIND == 2) {   // DPP moves
        if (N > 0) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v0) : "v"(v4));
        if (N > 1) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v1) : "v"(v5));
        if (N > 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v2) : "v"(v6));
        if (N > 3) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v3) : "v"(v7));
      } else if (KIND == 3) {   // LDS writes
        if (N > 0) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(*(double*)&v0) : "memory");
        if (N > 1) asm volatile("ds_write_b64 %0, %1 offset:2048" :: "v"(la), "v"(*(double*)&v0) : "memory");
      } else if (KIND == 4) {   // packed fp32
        double d0 = 0, d1 = 0;
        if (N > 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v0));
        if (N > 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v2));
        if (N > 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v4));
        if (N > 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v6));
      } else if (KIND == 6) {   // mixed-precision fma into a packed half
        if (N > 0) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v0) : "v"(v4), "v"(v5));
        if (N > 1) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v0) : "v"(v4), "v"(v6));
        if (N > 2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v1) : "v"(v4), "v"(v5));
        if (N > 3) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v1) : "v"(v4), "v"(v6));
        if (N > 4) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v2) : "v"(v4), "v"(v5));
        if (N > 5) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v2) : "v"(v4), "v"(v6));
        if (N > 6) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v3) : "v"(v4), "v"(v5));
        if (N > 7) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v3) : "v"(v4), "v"(v6));
      } else if (KIND == 7) {   // ds_read2_b32 (two rows of a column)
        if (N > 0) asm volatile("ds_read2_b32 %0, %1 offset1:128" : "=v"(*(double*)&v0) : "v"(la) : "memory");
        if (N > 1) asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:129" : "=v"(*(double*)&v2) : "v"(la) : "memory");
        if (N > 2) asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:130" : "=v"(*(double*)&v4) : "v"(la) : "memory");
        if (N > 3) asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:131" : "=v"(*(double*)&v6) : "v"(la) : "memory");
        if (N > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (KIND == 5) {   // f16 conversions
        if (N > 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v0) : "v"(v4), "v"(v5));
        if (N > 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v1) : "v"(v4), "v"(v5));
        if (N > 2) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v2) : "v"(v6));
        if (N > 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v3) : "v"(v6));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) s += acc[i] + other[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int N, int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s N=%d: %.1f ns per MFMA, %.1f ticks per MFMA\n", name, N, ms * 1e6 / (iters * 8), (double)c / (iters * 8));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 0>("valu", out, cyc); run<2, 0>("valu", out, cyc); run<4, 0>("valu", out, cyc); run<6, 0>("valu", out, cyc); run<8, 0>("valu", out, cyc);
  run<4, 1>("accread", out, cyc); run<8, 1>("accread", out, cyc);
  run<2, 2>("dpp", out, cyc); run<4, 2>("dpp", out, cyc);
  run<1, 3>("ds_write", out, cyc); run<2, 3>("ds_write", out, cyc);
  run<2, 4>("pk_fma", out, cyc); run<4, 4>("pk_fma", out, cyc);
  run<2, 5>("cvt", out, cyc); run<4, 5>("cvt", out, cyc);
  run<4, 6>("fma_mix", out, cyc); run<8, 6>("fma_mix", out, cyc);
  run<2, 7>("ds_read2", out, cyc); run<4, 7>("ds_read2", out, cyc);
  return 0;
}
