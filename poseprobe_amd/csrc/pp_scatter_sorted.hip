// Deterministic variant of the k0 gradient scatter (pp_k0_scatter_samples / pp_k0_scatter_packed, pp_color.hip).
// The atomic scatter adds the eight corner contributions of every sample in whatever order the hardware retires them; float
// addition is not associative, so the gradient grid differs in the last bits from run to run and between ranks that replay the
// same shards (the reason the "samples" multi-GPU mode re-broadcasts the grid every few hundred steps).  Here the
// (sample, corner) pairs are SORTED by voxel (stable LSD radix sort, rocPRIM) and every voxel's contributions are added by one
// lane group in ascending (shard, sample, corner) order: bit-identical results for identical inputs.
// Cost at the bench workload (~1.5 M pair slots): 0.21 ms against 0.04 ms for the atomic kernel (tools/dbg/time_scatter.py): an
// option (engine flag `deterministic_scatter`), not the default.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "pp_k0_tri.h"

namespace {

constexpr unsigned INVALID_KEY = 0xFFFFFFFFu;

struct SortedWork {                      // carved out of the caller's workspace
  unsigned *keys_in, *keys_out, *vals_in, *vals_out;
  void* tmp;
  size_t tmp_bytes;
};

size_t sort_tmp_bytes(size_t pairs) {
  size_t b = 0;
  unsigned* n = nullptr;
  rocprim::radix_sort_pairs(nullptr, b, n, n, n, n, pairs, 0, 32, hipStreamDefault, false);
  return b;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool carve(void* work, int64_t work_bytes, size_t pairs, SortedWork& w) {
  const size_t arr = align256(pairs * sizeof(unsigned));
  w.tmp_bytes = sort_tmp_bytes(pairs);
  if ((int64_t)(4 * arr + align256(w.tmp_bytes)) > work_bytes) return false;
  char* p = static_cast<char*>(work);
  w.keys_in = reinterpret_cast<unsigned*>(p); p += arr;
  w.keys_out = reinterpret_cast<unsigned*>(p); p += arr;
  w.vals_in = reinterpret_cast<unsigned*>(p); p += arr;
  w.vals_out = reinterpret_cast<unsigned*>(p); p += arr;
  w.tmp = p;
  return true;
}

// position of sample s (shard-major index) and its validity
template <bool PACKED>
__device__ __forceinline__ bool sample_pos(const float* __restrict__ src, const int32_t* __restrict__ count, int capacity, unsigned s, float (&p)[3]) {
  if (PACKED) {
    const float* shard = src + (size_t)(s / capacity) * capacity * PP_PACK_LD;
    const int m = s % capacity;
    if (m >= min(__float_as_int(shard[15]), capacity)) return false;
    const float* row = shard + (size_t)m * PP_PACK_LD;
    p[0] = row[12]; p[1] = row[13]; p[2] = row[14];
  } else {
    if ((int)s >= min(count[0], capacity)) return false;
    p[0] = src[(size_t)s * 3]; p[1] = src[(size_t)s * 3 + 1]; p[2] = src[(size_t)s * 3 + 2];
  }
  return true;
}

// pair = 8 * sample + corner  ->  key = voxel index (invalid samples / corners outside the grid: INVALID_KEY, sorted to the end)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_k0_keys(SceneDev sc, const float* __restrict__ src, const int32_t* __restrict__ count,
                                                 int capacity, unsigned pairs, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pairs) return;
  unsigned key = INVALID_KEY;
  float p[3];
  if (sample_pos<PACKED>(src, count, capacity, t >> 3, p)) {
    K0Tri tr;
    k0_setup(sc, p, tr);
    size_t off; float w;
    if (k0_corner(sc, tr, t & 7, off, w)) key = (unsigned)(off / (size_t)sc.C);
  }
  keys[t] = key;
  vals[t] = t;
}

// 16 lanes per sorted position, lane = channel; the group at the head of a voxel's run walks the run in order
template <bool PACKED>
__global__ __launch_bounds__(256) void k_k0_accumulate(SceneDev sc, const float* __restrict__ src, const float* __restrict__ feat_grad,
                                                       const int32_t* __restrict__ count, int capacity, unsigned pairs,
                                                       const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                                       float* __restrict__ k0_grad, uint8_t* __restrict__ touched) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned i = t >> 4, ch = t & 15;
  if (i >= pairs) return;
  const unsigned key = keys[i];
  if (key == INVALID_KEY || (i > 0 && keys[i - 1] == key)) return;
  float acc = 0.f;
  for (unsigned j = i; j < pairs && keys[j] == key; ++j) {
    const unsigned pair = vals[j], s = pair >> 3;
    float p[3];
    sample_pos<PACKED>(src, count, capacity, s, p);
    K0Tri tr;
    k0_setup(sc, p, tr);
    size_t off; float w;
    k0_corner(sc, tr, pair & 7, off, w);
    float g = 0.f;
    if (ch < (unsigned)sc.C) {
      if (PACKED) g = src[((size_t)(s / capacity) * capacity + s % capacity) * PP_PACK_LD + ch];
      else g = feat_grad[(size_t)s * PP_FEAT_LD + ch];
    }
    acc = pp_add(acc, pp_mul(w, g));
  }
  if (ch < (unsigned)sc.C) k0_grad[(size_t)key * sc.C + ch] += acc;
  if (touched && ch == 0) touched[key] = 1;
}

template <bool PACKED>
int run_sorted(const pp_scene* sc, const float* src, const float* feat_grad, const int32_t* count, int n_shards, int capacity,
               float* k0_grad, uint8_t* touched, void* work, int64_t work_bytes, hipStream_t st) {
  const size_t pairs = (size_t)n_shards * capacity * 8;
  SortedWork w;
  // k_k0_accumulate runs 16 lanes per pair with a 32-bit thread index: pairs * 16 must stay below 2^32
  if (pairs >= (1ull << 28) || !carve(work, work_bytes, pairs, w)) return 1;
  const SceneDev sd = pp_scene_dev(sc);
  const unsigned np = (unsigned)pairs;
  hipLaunchKernelGGL((k_k0_keys<PACKED>), dim3((np + 255) / 256), dim3(256), 0, st, sd, src, count, capacity, np, w.keys_in, w.vals_in);
  if (rocprim::radix_sort_pairs(w.tmp, w.tmp_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, pairs, 0, 32, st, false) != hipSuccess)
    return 2;
  hipLaunchKernelGGL((k_k0_accumulate<PACKED>), dim3((unsigned)((pairs * 16 + 255) / 256)), dim3(256), 0, st, sd, src, feat_grad, count,
                     capacity, np, w.keys_out, w.vals_out, k0_grad, touched);
  return 0;
}

}  // namespace

extern "C" int pp_k0_scatter_sorted_workspace(int64_t n_samples, int64_t* bytes) {
  PP_REQUIRE(bytes && n_samples > 0 && n_samples * 8 < (1ll << 28), "bad arguments (at most 2^25 - 1 sample slots)");
  const size_t pairs = (size_t)n_samples * 8;
  *bytes = (int64_t)(4 * align256(pairs * sizeof(unsigned)) + align256(sort_tmp_bytes(pairs)));
  return PP_OK;
}

extern "C" int pp_k0_scatter_samples_sorted(const pp_scene* sc, const float* pts, const int32_t* count, int32_t capacity,
                                            const float* feat_grad, float* k0_grad_cl, uint8_t* touched, void* work,
                                            int64_t work_bytes, void* stream) {
  PP_REQUIRE(sc && pts && count && feat_grad && k0_grad_cl && work, "null pointer");
  PP_REQUIRE(capacity > 0 && sc->k0_dim <= 16, "bad sizes");
  PP_REQUIRE((int64_t)sc->size[0] * sc->size[1] * sc->size[2] < 0xFFFFFFFFll, "grid too large for 32-bit voxel keys");
  const int rc = run_sorted<false>(sc, pts, feat_grad, count, 1, capacity, k0_grad_cl, touched, work, work_bytes, pp_stream(stream));
  if (rc) { pp_set_error("pp_k0_scatter_samples_sorted: %s", rc == 1 ? "workspace too small" : "sort failed"); return PP_ERR_INVALID_ARG; }
  PP_CHECK_LAUNCH();
  return PP_OK;
}

extern "C" int pp_k0_scatter_packed_sorted(const pp_scene* sc, const float* packed, int32_t n_shards, int32_t capacity,
                                           float* k0_grad_cl, uint8_t* touched, void* work, int64_t work_bytes, void* stream) {
  PP_REQUIRE(sc && packed && k0_grad_cl && work, "null pointer");
  PP_REQUIRE(capacity > 0 && n_shards > 0 && sc->k0_dim <= 12, "bad sizes");
  PP_REQUIRE((int64_t)sc->size[0] * sc->size[1] * sc->size[2] < 0xFFFFFFFFll, "grid too large for 32-bit voxel keys");
  const int rc = run_sorted<true>(sc, packed, nullptr, nullptr, n_shards, capacity, k0_grad_cl, touched, work, work_bytes, pp_stream(stream));
  if (rc) { pp_set_error("pp_k0_scatter_packed_sorted: %s", rc == 1 ? "workspace too small" : "sort failed"); return PP_ERR_INVALID_ARG; }
  PP_CHECK_LAUNCH();
  return PP_OK;
}
