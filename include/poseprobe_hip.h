/* poseprobe_hip.h - C ABI of libposeprobe_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for PoseProbe's object-branch hot path.  It replaces the reference's pybind11
 * torch extension `render_utils_cuda` (lib/cuda/render_utils.cpp:170-184) and the eager-PyTorch op
 * chains of lib/voxurf_coarse.py / lib/dvgo_ori.py / lib/camera.py / lib/losses.py / lib/utils.py
 * listed per entry point below (file:line relative to the reference tree).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / ATen types;
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the CALLER owns every buffer
 *     (inputs, outputs, workspace) - the library never allocates and never synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are re-entrant;
 *   - sample counts that are data dependent (M) live in device memory (`count`, one int32) so that a
 *     whole train step can be enqueued / graph-captured without a host round trip; kernels are
 *     launched for the capacity and retire surplus work-groups immediately;
 *   - fp32 values, int32 indices (the Python shim converts to int64 where the reference returns it);
 *   - return value: 0 on success, negative pp_status otherwise; pp_last_error() gives the text of
 *     the last failure on the calling thread;
 *   - arguments are passed one by one rather than through per-op `pp_<op>_args` structs (SURVEY 8b sketched those): the binding
 *     derives its prototypes from this header and so checks count and type of every argument; scratch sizes come from the
 *     pp_*_workspace queries, option values from the caller-owned pp_context handed to the call - the library holds no
 *     process-wide mutable state (ABI 3; ABI 1 / 2 had pp_set_option).
 */
#ifndef POSEPROBE_HIP_H
#define POSEPROBE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PP_OK = 0,
  PP_ERR_INVALID_ARG = -1,
  PP_ERR_LAUNCH = -2,
  PP_ERR_UNSUPPORTED = -3
} pp_status;

const char* pp_last_error(void);
/* ABI history.  1: round 1.  2: pp_loss_rays / pp_loss_samples / pp_geometry_bwd_priors gained `const float* batch_norm`
 * in front of `stream` (the library kept answering 1 for it by mistake).  3 (this header): options moved from process-wide
 * pp_set_option / pp_get_option into the caller-owned pp_context, which every option-dependent entry point now takes in front
 * of `stream` (pp_rgbnet_fwd, pp_warp_fwd, pp_mlp_fwd, the four two-stage backward entry points - which also hand the bias
 * ownership from stage 1 to stage 2 explicitly -, pp_nerf_fwd / pp_nerf_bwd, pp_grid_tv_adam_step{,_sparse});
 * pp_scene gained `sdf_index_exact`; new: pp_sdf_crossing_dense_bwd, pp_context_set_option / pp_context_get_option.
 * A binding MUST compare pp_abi_version() with the PP_ABI_VERSION it was built against before calling anything else
 * (poseprobe_amd/_lib.py does): the signatures changed, so a stale caller would pass a stream where a pointer is read. */
#define PP_ABI_VERSION 3
int pp_abi_version(void);

/* Static description of the voxel scene; mirrors the attributes Voxurf derives in __init__ /
 * _set_grid_resolution (lib/voxurf_coarse.py:67-68, :319-323) and the render_kwargs
 * (lib/recon_scene.py:208-217). */
typedef struct {
  float xyz_min[3];
  float xyz_max[3];
  int32_t size[3];      /* world_size X,Y,Z */
  float voxel_size;     /* fp32 value of Voxurf.voxel_size */
  float stepsize;       /* in voxels */
  float near_clip;
  float far_clip;
  float bg;
  int32_t n_samples;    /* S = int(|world_size+1| / stepsize) + 1  (voxurf_coarse.py:700) */
  float out_range;      /* DeformedImplicitField.output_range (deform_net.py:17) */
  int32_t k0_dim;       /* 12 */
  int32_t pos_pe;       /* 5 */
  int32_t view_pe;      /* 1 */
  int32_t sdf_index_exact; /* 0: the custom SDF sampler forms the flat voxel index in fp32 like the reference (differs above 2^24
                            * voxels only, lib/voxurf_coarse.py:632-647); 1: the mathematically intended index */
} pp_scene;

/* ---------------------------------------------------------------- pose: lib/camera.py:76-99,127-188;
 * lib/recon_scene.py:62-74 (get_current_pose_pnp) + camera.pose.invert (recon_scene.py:444).
 * se3[V,6], w2c_init[V,3,4] -> w2c[V,3,4], c2w[V,3,4]; jac[V,12,6] = d c2w / d se3 (forward mode),
 * consumed by pp_pose_bwd: se3_grad[V,6] = jac^T c2w_grad.  refine_mask[V] (0 = view is not refined). */
int pp_pose_fwd(const float* se3, const float* w2c_init, const int32_t* refine_mask, int32_t n_views,
                float* w2c, float* c2w, float* jac, void* stream);
int pp_pose_bwd(const float* jac, const float* c2w_grad, int32_t n_views, float* se3_grad, void* stream);

/* ---------------------------------------------------------------- rays: lib/voxurf_coarse.py:1339-1368,
 * :1402-1407, :1518-1549 + the randperm selection (recon_scene.py:598-600), index-first: only the selected
 * pixels are generated.  ray_idx[N] indexes the flattened [V,H,W] pixel list.  normalize=1: Voxurf
 * variant (rays_d = viewdirs), 0: DVGO variant (dvgo_ori.py:562-563).  images[V,H,W,3], masks[V,H,W].
 * Also performs the slab test of sample_ray_ori (voxurf_coarse.py:702-708): t_min,t_max[N]. */
int pp_raygen_select_fwd(const pp_scene* sc, const int32_t* ray_idx, int32_t n_rays, const float* c2w,
                         const float* intr /*[V,4] fx,fy,cx,cy*/, int32_t n_views, int32_t H, int32_t W,
                         int32_t inverse_y, int32_t normalize, const float* images, const float* masks,
                         float* rays_o, float* rays_d, float* viewdirs, float* target, float* mask_px,
                         void* stream);
/* Backward of the above plus of the dense sampler's ray-level terms: consumes per-sample gradients
 * (pts_grad[M,3], step[M], viewdir_grad_s[M,3]) segmented by ray_start[N+1], optional direct grads
 * on rays (may be NULL); produces per-ray grads rays_o/rays_d/viewdirs_grad_out[N,3] (any may be NULL) and,
 * when c2w_grad != NULL (Voxurf ray variant), c2w_grad[V,3,4] (zeroed by the callee). */
int pp_raygen_select_bwd(const pp_scene* sc, const int32_t* ray_idx, int32_t n_rays, const float* c2w,
                         const float* intr, int32_t n_views, int32_t H, int32_t W, int32_t inverse_y,
                         const float* rays_o, const float* rays_d, const float* t_min,
                         const int32_t* ray_start, const float* pts_grad, const float* step,
                         const float* viewdir_grad_s, const float* rays_o_grad, const float* rays_d_grad,
                         const float* viewdirs_grad, const float* depth_grad, float* rays_o_grad_out,
                         float* rays_d_grad_out, float* viewdirs_grad_out, float* c2w_grad, void* stream);

/* ---------------------------------------------------------------- dense sampler + compaction:
 * Voxurf.sample_ray_ori (voxurf_coarse.py:697-719) followed by the boolean compaction (:936-945).
 * jitter[N] may be NULL (eval).  Outputs: t_min,t_max[N], ray_start[N+1] (exclusive prefix of the
 * per-ray in-bbox counts; ray_start[N] = M, also written to count[0]), pts[M,3], ray_id[M],
 * step_k[M] (sample index within the ray), step[M] (= stepsize*voxel_size*(k+jitter)),
 * mask_keep[N*S] (uint8, 1 = in bbox; may be NULL).  capacity = allocated rows of the per-sample outputs: when the in-bbox
 * total exceeds it, count[0] = ray_start[N] = capacity and every ray_start entry is clamped to it (rays past the capacity
 * become empty, the ray that straddles it is cut), so no consumer of ray_start ever indexes past the allocation;
 * count[0] == capacity is the caller's truncation signal.  The same holds for pp_sample_var. */
int pp_sample_dense(const pp_scene* sc, const float* rays_o, const float* rays_d, const float* jitter,
                    int32_t n_rays, int32_t capacity, float* t_min, float* t_max, int32_t* ray_start,
                    int32_t* count, float* pts, int32_t* ray_id, int32_t* step_k, float* step,
                    uint8_t* mask_keep, void* stream);

/* Variable-length sampler of the reference's CUDA extension: sample_pts_on_rays
 * (lib/cuda/render_utils_kernel.cu:12-242) as used by Voxurf.sample_ray_cuda (voxurf_coarse.py:661-695,
 * far forced to 1e9, points recomputed as rays_start + dir*step_id*stepdist, in-bbox compaction).
 * n_steps[N] are the raw per-ray counts (kernel.cu:38-55); ray_start[N+1] prefix of the kept samples. */
int pp_sample_var(const pp_scene* sc, const float* rays_o, const float* rays_d, int32_t n_rays,
                  int32_t capacity, float* t_min, float* t_max, int32_t* n_steps, int32_t* ray_start,
                  int32_t* count, float* pts, int32_t* ray_id, int32_t* step_id, void* stream);

/* ---------------------------------------------------------------- transmittance scan:
 * render_utils_cuda.alpha2weight / alpha2weight_backward (lib/cuda/render_utils_kernel.cu:577-707, bound
 * at lib/voxurf_coarse.py:1319,:1329).  ray_start[N+1] replaces the kernel's i_start/i_end bookkeeping
 * (kernel.cu:607-636).  One wavefront per ray; sequential-order products, 1e-3 early stop. */
int pp_alpha2weight_fwd(const float* alpha, const int32_t* ray_start, int32_t n_rays, float* weights,
                        float* T, float* alphainv_last, int32_t* i_end, void* stream);
int pp_alpha2weight_bwd(const float* alpha, const float* weights, const float* T, const float* alphainv_last,
                        const int32_t* ray_start, const int32_t* i_end, int32_t n_rays,
                        const float* grad_weights, const float* grad_last, float* grad_alpha, void* stream);

/* Fused scan + compositing (replaces Alphas2Weights + the segment_coo calls, voxurf_coarse.py:995,
 * :1034-1057 / inference :1179-1202): rgb_marched[N,3] (clamped), rgb_pre[N,3] (pre-clamp, for the
 * backward), cum_weights[N], depth_acc[N] = sum w*step_w (step_w[M]: train `step`, inference
 * step_id*dist), optional normal_marched[N,3] from nrm_in[M,3] (may be NULL). */
int pp_march_fwd(const float* alpha, const float* rgb, const float* step_w, const float* nrm_in,
                 const int32_t* ray_start, int32_t n_rays, float bg, float* weights, float* T,
                 float* alphainv_last, int32_t* i_end, float* rgb_marched, float* rgb_pre,
                 float* cum_weights, float* depth_acc, float* normal_marched, void* stream);
int pp_march_bwd(const float* alpha, const float* rgb, const float* step_w, const float* weights, const float* T,
                 const float* alphainv_last, const int32_t* ray_start, const int32_t* i_end, int32_t n_rays,
                 float bg, const float* rgb_pre, const float* g_rgb_marched, const float* g_cum_weights,
                 const float* g_alphainv_last, const float* g_depth_acc, const float* g_weights /*[M] or NULL*/,
                 float* grad_alpha, float* grad_rgb, void* stream);

/* ---------------------------------------------------------------- geometry: sdf mapping
 * (voxurf_coarse.py:946-949), custom trilinear lookup at deformed / undeformed points (:545-659, :967,
 * :978), spatial gradients (:968-984) and NeuS alpha (:483-519), fused.  The mapped grid is never
 * materialised: the 8 corner values are mapped in registers.
 * warp_out[M,4,4]: row 0 = (deform xyz, correction), rows 1-3 = d/dp_i of the same (already x out_range).
 * Outputs: alpha[M], gradient[M,3], sdf_final[M], sdf_deform[M], grad_deform[M,3,3]. */
int pp_geometry_fwd(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab /*[2] raw alpha,beta*/,
                    const float* pts, const float* warp_out, const float* viewdirs, const int32_t* ray_id,
                    const int32_t* count, int32_t capacity, float inv_s, float* alpha, float* gradient,
                    float* sdf_final, float* sdf_deform, float* grad_deform, void* stream);
/* Upstream grads (any may be NULL): g_alpha[M], g_gradient[M,3], g_sdf_final[M], g_sdf_deform[M],
 * g_grad_deform[M,9], g_correction[M].  Outputs: warp_out_grad[M,4,4], pts_grad[M,3] (accumulate=1: +=),
 * viewdir_grad_s[M,3] (+= if accumulate), sdf_ab_grad[2] (atomic +=; caller zeroes). */
int pp_geometry_bwd(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab, const float* pts,
                    const float* warp_out, const float* viewdirs, const int32_t* ray_id, const int32_t* count,
                    int32_t capacity, float inv_s, const float* g_alpha, const float* g_gradient,
                    const float* g_sdf_final, const float* g_sdf_deform, const float* g_grad_deform,
                    const float* g_correction, int32_t accumulate, float* warp_out_grad, float* pts_grad,
                    float* viewdir_grad_s, float* sdf_ab_grad, void* stream);

/* pp_geometry_bwd + pp_loss_samples in one kernel: the four sample-level priors of object_losses (lib/losses.py:6-23) are
 * evaluated from the recomputed forward, their values accumulated into loss_out[2..5] (as pp_loss_samples does) and
 * their gradients folded into the backward without the g_grad_deform / g_correction / g_sdf_deform round trip through
 * HBM.  g_gradient[M,3] = upstream gradient of the normal from the colour features (may be NULL).  Bit-identical to the
 * two-call sequence (except loss_out[7], the weighted total, which only pp_loss_rays / pp_loss_samples maintain). */
int pp_geometry_bwd_priors(const pp_scene* sc, const float* sdf_grid, const float* sdf_ab, const float* pts,
                           const float* warp_out, const float* viewdirs, const int32_t* ray_id, const int32_t* count,
                           int32_t capacity, float inv_s, const float* g_alpha, const float* g_gradient, float w_eikonal,
                           float w_deform, float loss_scale, int32_t accumulate, float* warp_out_grad, float* pts_grad,
                           float* viewdir_grad_s, float* sdf_ab_grad, float* loss_out, const float* batch_norm,
                           void* stream);

/* ---------------------------------------------------------------- multi-GPU: k0 gradient exchange at sample granularity
 * (no counterpart in the reference, which has no distributed path; replaces a dense 196 MB reduce-scatter by an
 * all-gather of 64 B per sample).  pp_k0_pack_samples writes packed[capacity][16] = { feat_grad[:, :k0_dim] (12 slots),
 * pts xyz, pad } and the sample count (int32 bit pattern) into packed[0][15]; pp_k0_scatter_packed replays the trilinear
 * scatter of pp_color_feat_bwd for n_shards such buffers laid out back to back ([n_shards][capacity][16]) into
 * k0_grad_cl (atomic +=).  Pass k0_grad_cl = NULL to pp_color_feat_bwd to skip its own scatter. */
int pp_k0_pack_samples(const float* pts, const float* feat_grad, const int32_t* count, int32_t capacity, int32_t k0_dim,
                       float* packed, void* stream);
int pp_k0_scatter_packed(const pp_scene* sc, const float* packed, int32_t n_shards, int32_t capacity, float* k0_grad_cl,
                         uint8_t* touched /*optional: see pp_grid_tv_adam_step_sparse*/, void* stream);
/* The scatter half of pp_color_feat_bwd on its own (same kernel), optionally marking the voxels it reaches in a map of
 * one byte per voxel (index = (x*Y + y)*Z + z; plain stores of 1, no atomics). */
int pp_k0_scatter_samples(const pp_scene* sc, const float* pts, const int32_t* count, int32_t capacity,
                          const float* feat_grad, float* k0_grad_cl, uint8_t* touched /*optional*/, void* stream);

/* Deterministic variants of the two scatters above: the (sample, corner) contributions are sorted by voxel (stable radix sort) and
 * added per voxel in ascending (shard, sample, corner) order - bit-identical results for identical inputs, from run to run and
 * between ranks that replay the same shards (float atomics retire in hardware order).  About 6 x the time of the atomic kernels:
 * an option.  `work`: device memory of pp_k0_scatter_sorted_workspace(n_shards * capacity) bytes; grids up to 2^32 - 2 voxels. */
int pp_k0_scatter_sorted_workspace(int64_t n_samples, int64_t* bytes);
int pp_k0_scatter_samples_sorted(const pp_scene* sc, const float* pts, const int32_t* count, int32_t capacity,
                                 const float* feat_grad, float* k0_grad_cl, uint8_t* touched /*optional*/, void* work,
                                 int64_t work_bytes, void* stream);
int pp_k0_scatter_packed_sorted(const pp_scene* sc, const float* packed, int32_t n_shards, int32_t capacity, float* k0_grad_cl,
                                uint8_t* touched /*optional*/, void* work, int64_t work_bytes, void* stream);

/* ---------------------------------------------------------------- colour features: DenseGrid.forward for k0
 * (lib/grid.py:47-58, zeros padding), BARF positional encoding of xyz and view (voxurf_coarse.py:721-732,
 * :1009-1025), normal (:1028-1030) -> feat[M,64] (57 used, zero padded).  k0 is stored channels-last
 * [X,Y,Z,C].  pe_w[pos_pe + view_pe] are the c2f weights for this step. */
int pp_color_feat_fwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                      const int32_t* ray_id, const float* gradient, const float* pe_w, const int32_t* count,
                      int32_t capacity, float* feat, void* stream);
/* feat_grad[M,64] -> k0_grad_cl (atomic +=), pts_grad[M,3] (=), gradient_grad[M,3] (=),
 * viewdir_grad_s[M,3] (=). */
int pp_color_feat_bwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                      const int32_t* ray_id, const float* gradient, const float* pe_w, const int32_t* count,
                      int32_t capacity, const float* feat_grad, float* k0_grad_cl, float* pts_grad,
                      float* gradient_grad, float* viewdir_grad_s, void* stream);

/* Caller-owned context = the option values of the calls it is handed to (+ one auxiliary HIP stream with its events, created
 * on first use).  NULL everywhere = the compiled-in defaults (the measured best on MI355X).  Options are plain integers
 * set by name; a call reads them from ITS context for the duration of the call only, so contexts with different arithmetic
 * coexist in one process and on concurrent threads (a context itself must not be modified while a call uses it).
 * Names (meaning and ranges: csrc/pp_common.h, csrc/pp_error.hip):
 *   arithmetic   mlp_split (bit mask: object-branch MLP kernels as 3 fp16 products per fp32 product; 0 = fp32 MFMA instructions),
 *                nerf_split, nerf_split_tn (scene branch likewise), mlp_fused, wgrad_split, nerf_bitmask, nerf_planes, nerf_chain, nerf_tn256, nerf_tn_tr
 *   scheduling   side_stream, mlp_wgs, wgrad_side_wgs, grid_chunks, nerf_gemm_wgs, nerf_tn_ch, nerf_tn_split_wgs, nerf_tn_wgs, nerf_bn, nerf_chain_nw, nerf_chain_head
 * pp_nerf_fwd and pp_nerf_bwd of one pass (and the two stages of a two-stage backward) must see the same option values.
 * Unknown names / out-of-range values are refused; pp_context_get_option(NULL, ...) reads the defaults. */
int pp_context_create(void** ctx);
int pp_context_destroy(void* ctx);
int pp_context_set_option(void* ctx, const char* name, int32_t value);
int pp_context_get_option(const void* ctx, const char* name, int32_t* value);
/* Option side_stream = 1 (2: rgbnet only): the weight-gradient kernel of pp_rgbnet_bwd / pp_mlp_bwd / pp_warp_bwd is launched
 * on the context's auxiliary stream and NOT joined before the call returns, so that the caller's next (small) kernels run
 * beside it (fork / join are event edges: the sequence stays hipGraph-capturable).  Call pp_context_join before `scratch` is
 * reused, before params_grad is read, and at most 4 forks apart: it makes `stream` wait for every deferred launch issued so
 * far.  With side_stream = 0 (default) everything is strictly sequential on `stream` and pp_context_join does nothing. */
int pp_context_join(void* ctx, void* stream);

/* ---------------------------------------------------------------- MLPs on the matrix cores (fp32 MFMA).
 * rgbnet (voxurf_coarse.py:208-216, :1032-1033): 64(57)->128->128->128->3, sigmoid.
 * Parameter block layout (floats): W0[128*64] b0[128] W1[128*128] b1[128] W2[128*128] b2[128] W3[3*128] b3[3]
 * (W0 is the reference's [128,57] weight zero-padded to 64 columns).
 * acts[3][cap][128] keeps the hidden activations for the backward; acts = NULL: forward only (inference), nothing is kept -
 * accepted with the default split-precision forward kernel (option mlp_split bit 4), refused otherwise. */
#define PP_RGBNET_PARAMS (128 * 64 + 128 + 2 * (128 * 128 + 128) + 3 * 128 + 3)
int pp_rgbnet_fwd(const float* params, const float* feat, const int32_t* count, int32_t capacity, float* acts,
                  float* rgb, void* ctx, void* stream);
int pp_rgbnet_bwd(const float* params, const float* feat, const float* acts, const float* rgb,
                  const float* rgb_grad, const int32_t* count, int32_t capacity, float* scratch /*[3][cap][128] + 49152*/,
                  float* params_grad /*atomic +=*/, float* feat_grad, void* ctx /*pp_context or NULL*/, void* stream);

/* warp MLP (DeformedImplicitField, lib/deformation/deform_net.py:12-31, modules.py:43-124): 3->128x4->4 ReLU,
 * evaluated together with its input Jacobian in forward mode (row 0 primal, rows 1-3 tangents), which
 * replaces the reference's three autograd.grad(create_graph=True) passes (voxurf_coarse.py:972-984).
 * Parameter block: W0[128*3] b0[128] W1..W3[128*128]+b[128] each, W4[4*128] b4[4].
 * acts[4][cap*4][128] (NULL: forward only, as for pp_rgbnet_fwd; option mlp_split bit 1); out[M,4,4] (x out_range). */
#define PP_WARP_PARAMS (128 * 3 + 128 + 3 * (128 * 128 + 128) + 4 * 128 + 4)
int pp_warp_fwd(const float* params, const float* pts, const int32_t* count, int32_t capacity, float out_range,
                float* acts, float* out, void* ctx, void* stream);
int pp_warp_bwd(const float* params, const float* pts, const float* acts, const float* out_grad,
                const int32_t* count, int32_t capacity, float out_range, float* scratch /*[3][cap*4][128] + 49152*/,
                float* params_grad /*atomic +=*/, float* pts_grad /* += */, void* ctx /*pp_context or NULL*/,
                void* stream);

/* Workspace queries (floats) for the `acts` and `scratch` arguments of the MLP entry points at a sample capacity. */
int pp_rgbnet_workspace(int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats);
int pp_warp_workspace(int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats);
int pp_mlp_workspace(int32_t in_ld, int32_t n_gemm, int32_t capacity, int64_t* acts_floats, int64_t* scratch_floats);

/* Two-stage forms of the two backward chains above (layer-fused kernels only; option mlp_fused = 0 returns
 * PP_ERR_UNSUPPORTED): stage 1 = data gradients + thin-layer and bias gradients, leaves the hidden layers' output gradients
 * in `scratch`; stage 2 = the hidden layers' weight gradients from `scratch` and the stored activations.
 * pp_warp_bwd == pp_warp_bwd_data then pp_warp_bwd_weights (same for rgbnet); the split lets a caller time the kernels
 * separately, interleave other work, or put stage 2 on another stream.  Which stage produces the hidden layers' bias
 * gradients depends on the kernel stage 1 ran: stage 1 reports it in *stage2_host (a HOST int32, written before the call
 * returns) and the caller hands that value to stage 2 - stage 2 never re-derives it from its own context's options. */
int pp_warp_bwd_data(const float* params, const float* pts, const float* acts, const float* out_grad,
                     const int32_t* count, int32_t capacity, float out_range, float* scratch, float* params_grad,
                     float* pts_grad, int32_t* stage2_host, void* ctx, void* stream);
int pp_warp_bwd_weights(const float* acts, const float* scratch, const int32_t* count, int32_t capacity,
                        float* params_grad, int32_t stage2, void* ctx, void* stream);
int pp_rgbnet_bwd_data(const float* params, const float* acts, const float* rgb, const float* rgb_grad,
                       const int32_t* count, int32_t capacity, float* scratch, float* params_grad, float* feat_grad,
                       int32_t* stage2_host, void* ctx, void* stream);
int pp_rgbnet_bwd_weights(const float* feat, const float* acts, const float* scratch, const int32_t* count,
                          int32_t capacity, float* params_grad, int32_t stage2, void* ctx, void* stream);

/* ---------------------------------------------------------------- losses: lib/losses.py:6-74 (object_losses),
 * forward values + gradients w.r.t. the render outputs in one pass.  loss_scale multiplies every gradient
 * (recon_scene.py:648 scales the object loss by 0.1).
 * loss_out[8] (atomic +=, caller zeroes): [0] mse, [1] entropy, [2] eikonal, [3] grad_deform, [4] sdf_correct,
 * [5] sdf_deform, [6] bce mask (unweighted scalars, as loss_scalars in the reference), [7] the WEIGHTED sum of the seven
 * (each kernel adds its share with its w_* arguments, loss_scale not applied): object_losses' `loss` without the TV term.
 * mask_sum[1] receives the batch's masked-pixel count.  g_rgb_marched is w.r.t. the CLAMPED rgb_marched
 * (pp_march_bwd applies the clamp mask).
 * batch_norm (device float[2], may be NULL): ray-sharded data parallelism.  The reference normalises the masked MSE by
 * the batch's masked-pixel count and the sample-level priors by the batch's sample count (lib/losses.py:6-29); with the
 * rays of ONE batch spread over W ranks and gradients averaged over ranks, batch_norm[0] = (sum over ranks of the
 * masked-pixel counts) / W and batch_norm[1] = (sum over ranks of the sample counts) / W make the sharded step equal the
 * union-batch step.  NULL = this call's own counts (single GPU). */
int pp_loss_rays(const float* rgb_marched, const float* alphainv_last, const float* cum_weights,
                 const float* target, const float* mask_px, float* mask_sum, int32_t n_rays, float w_main,
                 float w_entropy, float w_mask, float loss_scale, float* g_rgb_marched, float* g_alphainv_last,
                 float* g_cum_weights, float* loss_out, const float* batch_norm, void* stream);
/* g_gradient[M,3] is ACCUMULATED (+=); g_grad_deform[M,9], g_correction[M], g_sdf_deform[M] are written. */
int pp_loss_samples(const float* gradient, const float* grad_deform, const float* warp_out,
                    const float* sdf_deform, const int32_t* count, int32_t capacity, float w_eikonal,
                    float w_deform, float loss_scale, float* g_gradient, float* g_grad_deform,
                    float* g_correction, float* g_sdf_deform, float* loss_out, const float* batch_norm, void* stream);

/* ---------------------------------------------------------------- optimiser: lib/utils.py:82-198 (Adam, betas
 * (0.9,0.99)) fused with the k0 total-variation gradient (voxurf_coarse.py:443-456, :1298-1313; weight
 * tv_scale = loss_scale*weight_tv_k0/(3*numel)) and the gradient zero-fill, one streaming pass over the
 * channels-last grid.  Ping-pong parameter buffers (p_in read incl. neighbours, p_out written).
 * x-slab [x_begin,x_end) only (ZeRO-1 sharding across ranks); tv_out[1] += sum |diff| of the slab. */
int pp_grid_tv_adam_step(const float* p_in, float* p_out, float* grad, float* exp_avg, float* exp_avg_sq,
                         int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, int32_t x_begin,
                         int32_t x_end, float tv_scale, float grad_scale, float lr, float beta1, float beta2,
                         float eps, int32_t step, float* tv_out, void* ctx, void* stream);
/* Same pass for a SPARSE data gradient: `touched` (one byte per voxel, written by pp_k0_scatter_samples / _packed) tells
 * which voxels can hold a non-zero grad; for all others grad is neither read nor re-zeroed (96 of the 384 B/voxel).
 * Results are bit-identical to the dense pass as long as every non-zero grad voxel is marked.  `touched_clear` (the
 * map consumed by the PREVIOUS step, or NULL) is zeroed on the way; touched == NULL = dense behaviour. */
int pp_grid_tv_adam_step_sparse(const float* p_in, float* p_out, float* grad, float* exp_avg, float* exp_avg_sq,
                                int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, int32_t x_begin,
                                int32_t x_end, float tv_scale, float grad_scale, float lr, float beta1, float beta2,
                                float eps, int32_t step, float* tv_out, const uint8_t* touched, uint8_t* touched_clear,
                                void* ctx, void* stream);
/* Flat Adam over a packed parameter buffer with per-segment learning rates: seg_end[n_seg], seg_lr[n_seg]. */
int pp_adam_flat(float* p, float* grad, float* exp_avg, float* exp_avg_sq, int32_t n, const int32_t* seg_end,
                 const float* seg_lr, int32_t n_seg, float grad_scale, float beta1, float beta2, float eps,
                 int32_t step, int32_t zero_grad, void* stream);
/* total_variation(v) value only (voxurf_coarse.py:1298-1313) on a channels-last grid: out[1] += sum|diff|. */
int pp_grid_tv_value(const float* p, int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, float* out,
                     void* stream);

/* Generic trilinear lookup on a channels-last grid [X,Y,Z,C] (C<=16): DenseGrid.forward (lib/grid.py:47-58; border=0
 * zeros padding) and grid_sampler's F.grid_sample path (lib/voxurf_coarse.py:540, lib/dvgo_ori.py:249-261; border=1).
 * pts[n,3] world coords -> out[n,C].  Backward: grid_grad_cl (atomic +=, may be NULL), pts_grad[n,3] (=, may be NULL). */
int pp_grid_sample_fwd(const pp_scene* sc, const float* grid_cl, int32_t channels, const float* pts, int32_t n_pts,
                       int32_t border, float* out, void* stream);
int pp_grid_sample_bwd(const pp_scene* sc, const float* grid_cl, int32_t channels, const float* pts, int32_t n_pts,
                       int32_t border, const float* out_grad, float* grid_grad_cl, float* pts_grad, void* stream);
/* total_variation backward (voxurf_coarse.py:1298-1313): grad += scale * g_scalar[0] * d(sum|diff|)/dp. */
int pp_grid_tv_grad(const float* p, int32_t size_x, int32_t size_y, int32_t size_z, int32_t channels, float scale,
                    const float* g_scalar, float* grad, void* stream);

/* Surface-point query (Voxurf.query_sdf_point_wocuda / _wodeform, voxurf_coarse.py:766-795, :809-837): first sign
 * change of the per-ray SDF samples and the linear zero crossing.  Compact mode: sdf[M] + ray_start[N+1] + step_k[M]
 * (out-of-bbox slots take the reference's default 1); dense mode (ray_start = step_k = NULL): sdf[N,S].
 * Outputs: pts[N,3] = o + d*(t_min + z0/|d|), mask[N] (uint8), optional sdf_dense[N,S], zval[N]. */
int pp_sdf_first_crossing(const float* sdf, const int32_t* ray_start, const int32_t* step_k, int32_t n_rays,
                          int32_t n_samples, float dist, const float* t_min, const float* rays_o,
                          const float* rays_d, float* sdf_dense, float* pts, uint8_t* mask, float* zval,
                          void* stream);

/* Backward of the DENSE-mode query above when its SDF row is the border-padded trilinear lookup of the raw template at
 * the dense sample positions p_k = o + d (t_min + dist (k + jitter) / |d|) (query_sdf_point_wocuda_wodeform,
 * voxurf_coarse.py:797-837; the reference differentiates it by autograd and recon_scene.py:336-340 back-propagates the
 * reprojection loss through it while at most two views are active).  sdf_grid [X,Y,Z]; sdf_dense[N,S] as returned by
 * the forward; jitter[N] or NULL (eval).  Upstream: g_pts[N,3] and / or g_sdf_dense[N,S] (either may be NULL).
 * Outputs (=): g_rays_o[N,3], g_rays_d[N,3] (incl. the |d| path), g_t_min[N]; t_min's own dependence on the ray
 * (slab test, :701-705) is the caller's to chain. */
int pp_sdf_crossing_dense_bwd(const pp_scene* sc, const float* sdf_grid, const float* rays_o, const float* rays_d,
                              const float* t_min, const float* jitter, int32_t n_rays, int32_t n_samples, float dist,
                              const float* sdf_dense, const float* g_pts, const float* g_sdf_dense, float* g_rays_o,
                              float* g_rays_d, float* g_t_min, void* stream);

/* ---------------------------------------------------------------- DVGO-surface operators of the reference's
 * extensions that the live loop never calls (SURVEY.md 2a "dead"), kept for API completeness:
 * raw2alpha{,_nonuni}{,_backward} (render_utils_kernel.cu:431-574; interval_v != NULL selects the per-point form),
 * maskcache_lookup (:374-424; world[X,Y,Z] uint8), sample_ndc_pts_on_rays (:245-293), sample_bg_pts_on_rays (:301-360),
 * adam_upd / masked_adam_upd / adam_upd_with_perlr (adam_upd_kernel.cu:8-133; mode 0/1/2),
 * total_variation_add_grad{,_new} (total_variation_kernel.cu:13-134; channels-last grids, mask_cl != NULL selects
 * the masked form), cumdist_thres (ub360_utils_kernel.cu:12-48). */
int pp_raw2alpha_fwd(const float* density, float shift, float interval, const float* interval_v, int32_t n,
                     float* exp_d, float* alpha, void* stream);
int pp_raw2alpha_bwd(const float* exp_d, const float* grad_back, float interval, const float* interval_v, int32_t n,
                     float* grad, void* stream);
int pp_maskcache_lookup(const uint8_t* world, const float* xyz, int32_t size_x, int32_t size_y, int32_t size_z,
                        float scale_x, float scale_y, float scale_z, float shift_x, float shift_y, float shift_z,
                        int32_t n, uint8_t* out, void* stream);
int pp_sample_ndc(const pp_scene* sc, const float* rays_o, const float* rays_d, int32_t n_rays, int32_t n_samples,
                  float* pts, uint8_t* mask_outbbox, void* stream);
int pp_sample_bg(const float* rays_o, const float* rays_d, const float* t_max, float bg_preserve, int32_t n_rays,
                 int32_t n_samples, float* pts, void* stream);
int pp_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* perlr, int32_t n,
                int32_t step, float beta1, float beta2, float lr, float eps, int32_t mode, void* stream);
/* pp_adam_upd (mode 0) over up to 32 small tensors of one optimiser group in ONE launch: the *_host arguments are HOST arrays of
 * n_tensors device pointers / sizes (copied into the kernel's arguments; nothing is dereferenced on the host).  Same arithmetic. */
int pp_adam_upd_multi(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host,
                      float* const* exp_avg_sq_host, const int32_t* sizes_host, int32_t n_tensors, int32_t step, float beta1,
                      float beta2, float lr, float eps, void* stream);
int pp_tv_add_grad(const float* param_cl, float* grad_cl, const float* mask_cl, int32_t size_x, int32_t size_y,
                   int32_t size_z, int32_t channels, float wx, float wy, float wz, int32_t dense_mode, void* stream);
int pp_cumdist_thres(const float* dist, float thres, int32_t n_rays, int32_t n_pts, uint8_t* mask, void* stream);

/* ---------------------------------------------------------------- DirectVoxGO twin (lib/dvgo_ori.py:289-379).
 * Generic feature builder: [k0 (channels k0_skip..C) | xyz PE | view PE | optional normal], row stride ld (multiple of
 * 32), un-weighted encodings when pe_w == NULL, sel[M] (uint8, optional) = weights > fast_color_thres mask, optional
 * k0_raw[M,C] copy of the interpolated features (k0_diffuse for rgbnet_direct=False). */
int pp_feat_generic_fwd(const pp_scene* sc, const float* k0_cl, const float* pts, const float* viewdirs,
                        const int32_t* ray_id, const float* gradient, const float* pe_w, const uint8_t* sel,
                        int32_t k0_skip, int32_t ld, const int32_t* count, int32_t capacity, float* feat,
                        float* k0_raw, void* stream);
int pp_feat_generic_bwd_k0(const pp_scene* sc, const float* pts, const uint8_t* sel, int32_t k0_skip, int32_t ld,
                           const int32_t* count, int32_t capacity, const float* feat_grad,
                           const float* k0_raw_grad, float* k0_grad_cl, void* stream);
/* Generic ReLU MLP in_ld -> 128 x n_gemm -> 3 + sigmoid on the matrix cores.  Parameter block:
 * W0[128*in_ld] b0[128] | (W[128*128] b[128]) x (n_gemm-1) | Wout[3*128] bout[3].  logit_add[M,ld] (optional) is
 * added to the logits before the sigmoid (k0_diffuse, dvgo_ori.py:359).  acts[n_gemm][cap][128];
 * scratch [3][cap][128] + 16384. */
int pp_mlp_fwd(const float* params, const float* feat, int32_t in_ld, int32_t n_gemm, const int32_t* count,
               int32_t capacity, const float* logit_add, int32_t logit_add_ld, float* acts, float* out,
               void* ctx, void* stream);
int pp_mlp_bwd(const float* params, const float* feat, int32_t in_ld, int32_t n_gemm, const float* acts,
               const float* out, const float* out_grad, const int32_t* count, int32_t capacity, float* scratch,
               float* params_grad, float* feat_grad, float* logit_add_grad, int32_t logit_add_ld, void* ctx,
               void* stream);
/* cumprod_exclusive(clamp_min(1-alpha,1e-10)) compositing without early stop (dvgo_ori.py:478-489): weights[M], T[M],
 * alphainv_last[N], rgb_acc[N,3] = sum w*rgb (un-clamped, bg not added), cum_weights[N], depth_acc[N] = sum w*step_w.
 * The backward is pp_march_bwd. */
int pp_march_dvgo_fwd(const float* alpha, const float* rgb, const float* step_w, const int32_t* ray_start,
                      int32_t n_rays, float* weights, float* T, float* alphainv_last, int32_t* i_end,
                      float* rgb_acc, float* cum_weights, float* depth_acc, void* stream);

/* ---------------------------------------------------------------- scene branch: NeRF MLP + compositing
 * (lib/bg_nerf/source/models/frequency_nerf.py).  Replaces NeRF.forward_samples (:268-288 -> compute_raw_density :149-170,
 * forward :172-227, positional_encoding :239-266 with FrequencyEmbedder :42-69) and NeRF.composite (:290-343) for the
 * default architecture (default_config.py:90-105): L_3D = 10, L_view = 4 with raw coordinates, 8 x 256 feature layers,
 * skip at 4, softplus density, 283 -> 128 -> 3 colour head.
 *
 * Parameter block (floats), offsets from pp_nerf_layout(offsets[23]) = {W0,b0,...,W7,b7,wd,bd,R0,br0,R1,br1,total}:
 *   W0[256][64] (63 used) | W1..W3[256][256] | W4[256][320] (features 0..255, encoding 256..318) | W5,W6 |
 *   wd[256] W7[256][256] | bd b7[256] = the reference's last layer [257][256] / [257] stored contiguously (row 0 = density) |
 *   R0[128][288] (features 0..255, view encoding 256..282) | R1[3][128].
 * Rays: center[R,3], ray[R,3] (un-normalised), depth[R,S]; sample m = r * S + s.  bands[14] (device) = BARF's coarse-to-fine
 * weights of the 10 point bands then the 4 view bands (ones without a schedule).  count: device int32 holding R * S.
 * Workspaces in floats from pp_nerf_workspace(R * S, R, &acts, &scratch).
 * Arithmetic: fp32 operands and accumulation; the forward / data-gradient / weight-gradient matrix products are evaluated as
 * three fp16 products per fp32 product (error against fp64 equal to the fp32 matrix instructions'); options
 * nerf_split = 0 / nerf_split_tn = 0 (of the context handed to the call) select the fp32 matrix instructions instead. */
int pp_nerf_layout(int64_t* offsets);
int pp_nerf_workspace(int64_t n_samples, int64_t n_rays, int64_t* acts_floats, int64_t* scratch_floats);
int pp_nerf_fwd(const float* params, const float* center, const float* ray, const float* depth, const float* bands,
                const int32_t* count, int32_t n_rays, int32_t n_samples, float* acts, float* rgb_samples,
                float* density_samples, void* ctx, void* stream);
/* Backward of pp_nerf_fwd: params_grad is ACCUMULATED into (zero it first); g_center[R,3] and g_ray[R,3] are overwritten
 * (g_ray holds the point and view-direction paths; the compositing path is pp_nerf_composite_bwd's g_ray). */
int pp_nerf_bwd(const float* params, const float* ray, const float* depth, const int32_t* count, int32_t n_rays,
                int32_t n_samples, const float* acts, const float* rgb_samples,
                const float* g_rgb_samples, const float* g_density_samples, float* scratch, float* params_grad,
                float* g_center, float* g_ray, void* ctx, void* stream);
/* composite (:290-343): rgb[R,3] (+ 1 - opacity when white_bg), depth[R], opacity[R], weights[R,S], all_cumulated[R]
 * (= T at the second-to-last sample), rgb_var[R], depth_var[R] (the two variances are forward-only outputs). */
int pp_nerf_composite_fwd(const float* rgb_samples, const float* density_samples, const float* depth, const float* ray,
                          int32_t n_rays, int32_t n_samples, int32_t white_bg, float* rgb, float* depth_out, float* opacity,
                          float* weights, float* all_cumulated, float* rgb_var, float* depth_var, void* stream);
int pp_nerf_composite_bwd(const float* rgb_samples, const float* density_samples, const float* depth, const float* ray,
                          const float* weights, int32_t n_rays, int32_t n_samples, int32_t white_bg, const float* g_rgb,
                          const float* g_depth, const float* g_opacity, const float* g_weights, float* g_rgb_samples,
                          float* g_density_samples, float* g_ray, void* stream);

/* BARF coarse-to-fine band weights on the device (frequency_nerf.py:250-253): bands[l_3d + l_view] from the device scalar
 * `progress` - the schedule costs one launch and no host synchronisation.  width = end - start of opt.barf_c2f, formed in double
 * by the caller and rounded once (what torch does with the Python scalar). */
int pp_nerf_band_weights(const float* progress, float start, float width, int32_t l_3d, int32_t l_view, float* bands, void* stream);
/* Photometric loss of the scene branch (base_losses.py:155-156, :304-307): loss[0] = weight * huber_loss(pred, label, delta,
 * 'mean') over n floats, g_pred[n] = d loss / d pred.  Deterministic (one work-group, fixed summation order). */
int pp_nerf_huber_loss(const float* pred, const float* label, int32_t n, float delta, float weight, float* loss, float* g_pred,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POSEPROBE_HIP_H */
