// Probe (not product) for csrc/pp_gemm_planes.h: runs k_pack_planes + k_gemm256p<EPI_RELU> on one 256 -> 256 layer at M rows,
// checks the result against an fp64 host product on sampled rows and prints time / bandwidth; with -DPL_TIMERS the kernel's
// phase timers (s_memtime, summed per work-group) are printed for a few work-groups.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DPL_TIMERS] -I poseprobe_amd/csrc -o tools/_build/gemm256_probe tools/gemm256_probe.hip poseprobe_amd/csrc/pp_error.hip
//   tools/_build/gemm256_probe [M]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "pp_gemm_planes.h"

int pp_fused_wgs() { return 256; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 130944, K = 256, N = 256;
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& v : hA) v = fmaxf(0.f, (float)rand() / RAND_MAX * 2.f - 0.7f);
  for (auto& v : hW) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
  for (auto& v : hb) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  float *A, *W, *b, *C, *mx;
  _Float16* img;
  uint16_t* bits;
  int32_t* count;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMalloc(&mx, 64 * 4); hipMalloc(&img, (size_t)N * K * 4); hipMalloc(&bits, ((size_t)M + 127) / 128 * 128 * 32); hipMalloc(&count, 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(count, &M, 4, hipMemcpyHostToDevice);
  float amax = 0.f, wmax = 0.f;
  for (auto v : hA) amax = fmaxf(amax, fabsf(v));
  for (auto v : hW) wmax = fmaxf(wmax, fabsf(v));
  float hmx[64] = {0};
  hmx[0] = amax; hmx[1] = wmax;
  hipMemcpy(mx, hmx, sizeof(hmx), hipMemcpyHostToDevice);
  PlanePackJobs J;
  J.n = 1; J.src[0] = W; J.dst[0] = img; J.ld[0] = K; J.K[0] = K; J.mx_slot[0] = 1;
  hipLaunchKernelGGL(k_pack_planes, dim3(16, 1), dim3(256), 0, 0, J, mx);
  const int tiles = (M + 127) / 128, grid = tiles < 256 ? tiles : 256;
  auto run = [&]() {
    hipLaunchKernelGGL((k_gemm256p<EPI_RELU>), dim3(grid), dim3(512), 0, 0, A, K, img, K, b, C, N, count, M, mx, mx + 1, mx + 2, bits);
  };
  run();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) run();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("M=%d: %.1f us per launch, %.2f TB/s algorithmic (A read + C write), %.1f TFLOP/s algorithmic\n", M, us,
         2.0 * M * K * 4 / us / 1e6, 2.0 * M * N * K / us / 1e6);
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double se = 0, sr = 0;
  for (int r = 0; r < M; r += 997)
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)r * K + k] * hW[(size_t)n * K + k];
      s = s > 0 ? s : 0;
      const double d = hC[(size_t)r * N + n] - s;
      se += d * d; sr += s * s;
    }
  printf("relative rms error vs fp64: %.3e\n", sqrt(se / sr));
  return 0;
}
