/* CPU restatement (plain C, scalar, single thread) of the three live entry points of the reference's
 * native extension `render_utils_cuda`.
 *
 * TEST INFRASTRUCTURE ONLY - this file is the checker, never the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Reference followed (file:line are relative to /root/reference):
 *   alpha2weight            lib/cuda/render_utils_kernel.cu:577-651
 *   alpha2weight_backward   lib/cuda/render_utils_kernel.cu:654-707
 *   sample_pts_on_rays      lib/cuda/render_utils_kernel.cu:12-242
 *
 * Parity status: the reference ships no tests or golden vectors for these kernels and the CUDA
 * sources cannot be built here (nvcc / CUDA runtime absent) => "parity unpinned" against the
 * compiled reference; pinned only by the .cu source text.  Mixed float/double arithmetic of the
 * .cu (literals such as `1.` and `1e-10` are doubles) is kept on purpose.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile); no FMA contraction so the
 * result does not depend on the host ISA.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- segment bookkeeping: i_start / i_end per ray from a sorted ray_id (kernel.cu:607-636) ---- */
static void segment_bounds(const int64_t *ray_id, int64_t n_pts, int64_t n_rays,
                           int64_t *i_start, int64_t *i_end) {
  memset(i_start, 0, sizeof(int64_t) * (size_t)n_rays);
  memset(i_end, 0, sizeof(int64_t) * (size_t)n_rays);
  for (int64_t p = 1; p < n_pts; ++p) {
    if (ray_id[p] != ray_id[p - 1]) {
      i_start[ray_id[p]] = p;
      i_end[ray_id[p - 1]] = p;
    }
  }
  if (n_pts > 0) i_end[ray_id[n_pts - 1]] = n_pts;
}

/* weight/T must be pre-filled by the caller with 0 / 1, alphainv_last with 1 (zeros_like/ones_like
 * at kernel.cu:624-626); the early-terminated tail keeps those values. */
void pp_oracle_alpha2weight(const float *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays,
                            float *weight, float *T, float *alphainv_last,
                            int64_t *i_start, int64_t *i_end) {
  for (int64_t p = 0; p < n_pts; ++p) { weight[p] = 0.f; T[p] = 1.f; }
  for (int64_t r = 0; r < n_rays; ++r) alphainv_last[r] = 1.f;
  segment_bounds(ray_id, n_pts, n_rays, i_start, i_end);
  if (n_pts == 0) return;
  for (int64_t r = 0; r < n_rays; ++r) {
    const int64_t s = i_start[r], e_max = i_end[r];
    float t_cum = 1.f;
    int64_t i;
    for (i = s; i < e_max; ++i) {
      T[i] = t_cum;
      weight[i] = t_cum * alpha[i];
      t_cum = (float)((double)t_cum * (1. - (double)alpha[i]));
      if ((double)t_cum < 1e-3) { i += 1; break; }
    }
    i_end[r] = i;
    alphainv_last[r] = t_cum;
  }
}

void pp_oracle_alpha2weight_backward(const float *alpha, const float *weight, const float *T,
                                     const float *alphainv_last, const int64_t *i_start,
                                     const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                                     const float *grad_weights, const float *grad_last, float *grad) {
  for (int64_t p = 0; p < n_pts; ++p) grad[p] = 0.f;
  for (int64_t r = 0; r < n_rays; ++r) {
    float back = grad_last[r] * alphainv_last[r];
    for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
      const float gwT = grad_weights[i] * T[i];
      const double denom = (double)(1 - alpha[i]) + 1e-10;
      grad[i] = (float)((double)gwT - (double)back / denom);
      back += grad_weights[i] * weight[i];
    }
  }
}

/* ---- variable-length sampler (kernel.cu:12-79, 167-242) -------------------------------------- */
static inline float fmaxf2(float a, float b) { return a > b ? a : b; }
static inline float fminf2(float a, float b) { return a < b ? a : b; }

/* pass 1: per ray t_min, t_max, N_steps, start, dir ; returns total sample count */
int64_t pp_oracle_sample_rays_count(const float *rays_o, const float *rays_d, const float *xyz_min,
                                    const float *xyz_max, float near, float far, float stepdist,
                                    int64_t n_rays, float *t_min, float *t_max, int64_t *n_steps,
                                    float *rays_start, float *rays_dir) {
  int64_t total = 0;
  for (int64_t r = 0; r < n_rays; ++r) {
    const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
    float v[3], a[3], b[3];
    for (int k = 0; k < 3; ++k) {
      v[k] = (d[k] == 0) ? (float)1e-6 : d[k];
      a[k] = (xyz_max[k] - o[k]) / v[k];
      b[k] = (xyz_min[k] - o[k]) / v[k];
    }
    float lo = fmaxf2(fmaxf2(fminf2(a[0], b[0]), fminf2(a[1], b[1])), fminf2(a[2], b[2]));
    float hi = fminf2(fminf2(fmaxf2(a[0], b[0]), fmaxf2(a[1], b[1])), fmaxf2(a[2], b[2]));
    t_min[r] = fmaxf2(fminf2(lo, far), near);
    t_max[r] = fmaxf2(fminf2(hi, far), near);
    const float rnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double ns = (double)ceilf((t_max[r] - t_min[r]) * rnorm / stepdist);
    if (ns < 1.) ns = 1.;
    n_steps[r] = (int64_t)ns;
    total += n_steps[r];
    for (int k = 0; k < 3; ++k) {
      rays_start[3 * r + k] = o[k] + d[k] * t_min[r];
      rays_dir[3 * r + k] = d[k] / rnorm;
    }
  }
  return total;
}

/* pass 2: fill per-sample outputs */
void pp_oracle_sample_rays_fill(const float *rays_start, const float *rays_dir, const float *xyz_min,
                                const float *xyz_max, const int64_t *n_steps, float stepdist,
                                int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox,
                                int64_t *ray_id, int64_t *step_id) {
  int64_t p = 0;
  for (int64_t r = 0; r < n_rays; ++r) {
    for (int64_t s = 0; s < n_steps[r]; ++s, ++p) {
      const float dist = stepdist * (float)s;
      float x[3];
      int out = 0;
      for (int k = 0; k < 3; ++k) {
        x[k] = rays_start[3 * r + k] + rays_dir[3 * r + k] * dist;
        rays_pts[3 * p + k] = x[k];
        out |= (xyz_min[k] > x[k]) | (xyz_max[k] < x[k]);
      }
      mask_outbbox[p] = (uint8_t)out;
      ray_id[p] = r;
      step_id[p] = s;
    }
  }
}
