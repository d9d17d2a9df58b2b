// Shared device/host helpers for libposeprobe_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/poseprobe_hip.h"

#define PP_WAVE 64

void pp_set_error(const char* fmt, ...);

// Options (include/poseprobe_hip.h: pp_context_set_option / pp_context_get_option).  They are fields of a caller-owned
// pp_context that travels with every call whose behaviour they select; a NULL context means the compiled-in defaults.
// The library holds NO process-wide mutable state and never reads the environment.
enum PPOption {
  PP_OPT_MLP_FUSED = 0,        // 1: layer-fused object-branch MLP kernels, 0: layer-by-layer GEMMs (A/B runs)
  PP_OPT_WGRAD_SPLIT,          // 1: split-precision weight-gradient kernels in the object branch (measured slower)
  PP_OPT_GRID_CHUNKS,          // x-chunks of the fused TV + Adam pass (0 = heuristic)
  PP_OPT_NERF_SPLIT,           // scene branch: NT products as three fp16 products (1) or on the fp32 instructions (0)
  PP_OPT_NERF_SPLIT_TN,        // scene branch: weight-gradient products likewise
  PP_OPT_NERF_BITMASK,         // scene branch: one-bit ReLU masks
  PP_OPT_NERF_GEMM_WGS,        // scene branch: persistent work-groups per column block
  PP_OPT_NERF_TN_CH,           // scene branch: rows per LDS chunk of the fp32 weight-gradient GEMM (32 | 64)
  PP_OPT_NERF_TN_SPLIT_WGS,    // scene branch: row splits of the split-precision weight-gradient kernel
  PP_OPT_NERF_TN_WGS,          // scene branch: row splits of the fp32 weight-gradient kernel
  PP_OPT_NERF_BN,              // scene branch: 256 selects the 128 x 256 tile of the fp32 NT GEMM
  PP_OPT_NERF_PLANES,          // scene branch: 1 = activations travel as pre-split fp16 hi / lo planes (pp_gemm_planes.h)
  PP_OPT_MLP_SPLIT,            // bit mask: layer-fused object-branch MLP kernels with three fp16 products per fp32 product (pp_mlp_split.hip); 1 warp fwd, 2 warp bwd, 4 rgb fwd, 8 rgb bwd, 16 weight-gradient chains
  PP_OPT_NERF_TN256,           // scene branch: 1 = 256 x 256 weight gradients by the one-work-group-per-row-range kernel (pp_gemm_tn256.h; measured equal: 127 vs 123 us)
  PP_OPT_MLP_WGS,              // work-groups of the persistent object-branch MLP kernels (0 = one per CU); fewer leave CUs to a concurrent HBM-bound kernel
  PP_OPT_WGRAD_SIDE_WGS,       // work-groups of a weight-gradient chain kernel launched on a pp_context's auxiliary stream (0 = as on the main stream):
                               // fewer leave whole CUs to the small kernels that run beside it
  PP_OPT_SIDE_STREAM,          // 1: the weight-gradient kernels of both object-branch MLP chains are forked onto the context's auxiliary stream
                               // (joined by pp_context_join), 2: rgbnet's only, 0 (default): strictly sequential on the caller's stream
  PP_OPT_NERF_CHAIN,           // scene branch: bit 1 = the eight feature layers + density head of the forward pass as one kernel with the tile resident in LDS
                               // (pp_nerf_trunk.h), 3 = the data-gradient chain of the backward pass too (0 = one GEMM per layer)
  PP_OPT_NERF_CHAIN_NW,        // scene branch: wavefronts per work-group of the fused chains (8: one 128-sample tile per CU, 4: two 64-sample tiles per CU)
  PP_OPT_NERF_CHAIN_HEAD,      // scene branch: 1 = the colour head's hidden layer as a ninth stage of the fused forward chain
  PP_OPT_NERF_TN_TR,           // scene branch: 1 = weight-gradient kernel with row-major LDS images and transposed fragment reads (k_gemm_tn_tr)
  PP_OPT_COUNT
};

// Caller-owned context: the option values + (created on first use) one auxiliary HIP stream with its fork / join events.
struct PPContext {
  int opt[PP_OPT_COUNT];
  bool have_aux;
  hipStream_t aux;
  hipEvent_t fork[16], join[16];
  int pending;                      // deferred side launches of the fused paths not yet joined (pp_context_join)
  hipEvent_t dfork[4], djoin[4];
};
bool pp_context_aux(PPContext* c);   // creates the auxiliary stream + events on first use; false when HIP refuses

// Every entry point that takes a `ctx` opens a scope over ITS options for the duration of the call on the calling thread;
// the launch helpers below it read them through pp_opt().  Nothing outlives the call: two contexts with different
// arithmetic coexist in one process, also on different threads at the same time.
struct PPOptScope {
  const int* prev;
  explicit PPOptScope(const void* ctx);
  ~PPOptScope();
};
int pp_opt(int id);
int pp_num_cus();
// X-marching total-variation value / gradient pass on a channels-last grid (pp_optim.hip); false: shape not covered
bool pp_launch_tv_march(const float* p, int X, int Y, int Z, int C, float scale, const float* g_scalar, float* grad, float* tv_out,
                        hipStream_t st);                    // compute units of the current device (queried once per process: a hardware constant)

#define PP_REQUIRE(cond, msg)                                     \
  do {                                                            \
    if (!(cond)) {                                                \
      pp_set_error("%s: %s", __func__, msg);                      \
      return PP_ERR_INVALID_ARG;                                  \
    }                                                             \
  } while (0)

#define PP_CHECK_LAUNCH()                                                           \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      pp_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
      return PP_ERR_LAUNCH;                                                         \
    }                                                                               \
  } while (0)

static inline hipStream_t pp_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int pp_div_up(int a, int b) { return (a + b - 1) / b; }

// Scene constants passed by value to kernels (kernarg), derived once on the host.
struct SceneDev {
  float mn[3], mx[3];
  int sz[3];
  float voxel, stepsize, near_, far_, bg;
  int S;
  float out_range;
  int C, Lp, Lv;
  int flat_f32;   // custom SDF sampler: form the flat voxel index in fp32 as the reference does (only differs above 2^24 voxels)
};

static inline SceneDev pp_scene_dev(const pp_scene* s) {
  SceneDev d;
  for (int i = 0; i < 3; ++i) { d.mn[i] = s->xyz_min[i]; d.mx[i] = s->xyz_max[i]; d.sz[i] = s->size[i]; }
  d.voxel = s->voxel_size; d.stepsize = s->stepsize; d.near_ = s->near_clip; d.far_ = s->far_clip; d.bg = s->bg;
  d.S = s->n_samples; d.out_range = s->out_range; d.C = s->k0_dim; d.Lp = s->pos_pe; d.Lv = s->view_pe;
  // lib/voxurf_coarse.py:632-647 computes `iz * IW * IH + iy * IW + ix` on FLOAT tensors before .long(): beyond 2^24 voxels
  // (grids above 256^3) odd flat indices are not representable and round to a neighbouring voxel.  Reproduced by default
  // for parity (SURVEY 8a parity hazards); pp_scene.sdf_index_exact = 1 gives the exact index.
  d.flat_f32 = ((long long)s->size[0] * s->size[1] * s->size[2] > (1ll << 24)) && s->sdf_index_exact == 0;
  return d;
}

#ifdef __HIPCC__
// exact-rounding primitives: the library is built with -ffp-contract=off, these document intent where the
// op order of the reference must be reproduced bit for bit (sampler / coordinate transforms).
__device__ __forceinline__ float pp_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float pp_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float pp_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float pp_div(float a, float b) { return __fdiv_rn(a, b); }

__device__ __forceinline__ float pp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// softplus(beta=10, threshold=20) as torch.nn.Softplus
__device__ __forceinline__ float pp_softplus10(float x) {
  float bx = 10.0f * x;
  return bx > 20.0f ? x : log1pf(expf(bx)) / 10.0f;
}
__device__ __forceinline__ float pp_dsoftplus10(float x) {
  float bx = 10.0f * x;
  return bx > 20.0f ? 1.0f : pp_sigmoid(bx);
}

__device__ __forceinline__ float pp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// world -> continuous voxel coordinate along one axis, op order of grid_sampler + grid_sample_3d
// (lib/voxurf_coarse.py:528, :553-555): t=(p-min)/(max-min); n=t*2-1; u=((n+1)/2)*(size-1)
__device__ __forceinline__ float pp_grid_u(float p, float mn, float mx, int size) {
  float t = pp_div(pp_sub(p, mn), pp_sub(mx, mn));
  float n = pp_sub(pp_mul(t, 2.0f), 1.0f);
  return pp_mul(pp_div(pp_add(n, 1.0f), 2.0f), (float)(size - 1));
}
#endif
