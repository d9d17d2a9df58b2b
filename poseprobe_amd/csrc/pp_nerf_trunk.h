// The eight 256-wide feature layers of the scene-branch NeRF (lib/bg_nerf/source/models/frequency_nerf.py:152-170: 63 -> 256 x 8,
// skip connection at layer 4) as ONE kernel, and their data-gradient chain as one more: a work-group carries a 128-sample tile
// through all eight stages, so a stage's input never comes back from HBM - every stage's output is still WRITTEN (the backward
// pass / the weight-gradient products need it), which halves the activation traffic of the layer-by-layer path
// (pp_gemm_planes.h: 1 KB in + 1 KB out per sample and layer) and removes its per-K-chunk barriers.  Same arithmetic: three
// fp16 products per fp32 product, fp32 accumulation (pp_gemm_split.h).
//
// Data flow per tile
//   * the activation tile lives in LDS as the split-precision image the matrix instructions read: 8 K-chunks of
//     [128 rows][32 hi | 32 lo halfs], 16-byte slots XOR-swizzled (pl_slot_off), 128 KB;
//   * what enters a chain from HBM - the 64 encoded-point columns (input of layer 0, skip input of layer 4), the 128 columns of
//     the colour head's hidden gradient in the backward chain - goes through a separate one-chunk image E (16 KB), fetched into
//     registers well ahead of its use;
//   * wavefront w of eight owns output columns 32 w .. 32 w + 31 of all 128 rows.  The matrix instructions run TRANSPOSED,
//     D[n][row] = sum_k W[n][k] X[row][k]: the weights are the first operand and come straight from L2 into registers
//     (k_pack_trunk lays them out in the order of use, so a wavefront's stream is linear: 4 KB per step, double-buffered one
//     step ahead), the activations are the second operand, read from the LDS image.  A lane then holds 4 x 4 CONSECUTIVE
//     output columns of one row per 32 x 32 block: the stage's output goes into chunk w of the NEXT stage's image as 8-byte
//     LDS writes - wavefront w's columns are exactly K-chunk w of the next stage - and from there to HBM while the next stage
//     runs (see the note at the resident chunks); the last stage's output is stored from the accumulators;
//   * no barrier inside a stage (the image is read-only while a stage runs, weights are private to a wavefront): two per
//     stage around the in-place rewrite of the image, two per streamed chunk.
// Scales.  The layer-by-layer path scales a GEMM's input by the power of two derived from the maximum of the WHOLE tensor,
// which a fused kernel cannot know before it has finished; it uses the maximum of the TILE (>= as many significant bits),
// reduced through LDS between the two barriers of the epilogue, and still records the tensor maxima for the kernels downstream.
// ReLU masks: one bit per activation, 32 bytes per sample and layer either way.  MROW = false writes the layout the
// layer-by-layer data-gradient kernels read (pp_gemm.h gemm_epilogue: 16 rows of a column per 16-bit word; a lane holds 16
// columns of one row here, so the 32 x 32 bit block of a wavefront is transposed with v_cmp - the 64-lane mask of one
// accumulator register = two columns x 32 rows -, v_writelane and a nibble shuffle); MROW = true writes the lane's own 16
// columns as one word, [row][wavefront][lane half], which is what the fused backward chain reads back in the same lane.
// The density head (row 0 of the reference's last feature layer applied to layer 6's output) is folded into layer 6's
// epilogue: per-wavefront partial dot products through LDS, summed in a fixed order; its gradient enters the backward chain
// as the rank-1 term d raw[row] * wd[k] in the epilogue of the stage that produces d(layer 6).
#pragma once
#include "pp_gemm_planes.h"

#define TR_IMG_BYTES (8 * PL_A_BYTES)
#define TR_STEPS 60                     // K-chunks of a chain: forward 2 + 8 + 8 + 8 + (2 + 8) + 8 + 8 + 8, backward 4 + 7 x 8
#ifndef TR_DBG
#define TR_DBG 0      // experiments only, bit mask: 1 no matrix instructions, 2 no output stores, 4 no mask words, 8 no weight fetches in the loop, 16 no activation reads
#endif
#ifndef TR_NT
#define TR_NT 1       // output stores non-temporal: the 1 GB of activations a pass writes should not push the 2 MB of weights out of L2 (508 -> 434 us)
#endif
#define TR_WSTEP 32768                  // bytes of one step's weights: 256 columns x 32 k x (hi | lo) halfs

#ifdef TR_TIMERS      // phase timers (experiments): wave 0 of every work-group sums s_memtime deltas per phase
__device__ unsigned long long g_tr_t[16];
extern "C" int pp_debug_read_trunk_timers(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tr_t), sizeof(g_tr_t)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tr_t), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#define TR_TICK(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tsum[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define TR_TICK(i) do {} while (0)
#endif

struct TrunkArgs {
  const float* in;                      // what the chain reads from HBM: forward [M][64] encoded points, backward [M][128] d(hidden of the colour head)
  int in_ld;
  float* out[9];                        // stage outputs, fp32 [M][ld] (forward stage 8 = the colour head's hidden layer, [M][128])
  int ld[9];
  const float* bias[9];                 // forward
  uint32_t* bits[8];                    // forward, MROW = false: ReLU masks in the layer-by-layer layout (pairs of its 16-bit words)
  uint16_t* bitsr[8];                   // MROW: [row][8 wavefronts][2 lane halves] words; forward writes stage s's, backward reads the mask of stage s's OUTPUT
  const unsigned char* wstream;         // k_pack_trunk's image
  const float* wd;                      // density head weights (forward: NULL = not folded; backward: the rank-1 term)
  const float* bd;
  float* raw;
  float* density;
  const float* draw;                    // backward: d raw of row r at draw[r * draw_ld]
  int draw_ld;
  float* mx;                            // operand-maximum slots
  int mx_in;                            // slot of `in`
  int mx_w[9], mx_out[8];               // per stage: slot of the weights, slot recording the output
  int head;                             // forward: 1 = stage 8, the colour head's 288 -> 128 layer on layer 7's tile + the view columns (read at in2)
  const float* in2;                     // forward with head: [M][in2_ld] view-direction encoding, 32 columns
  int in2_ld;
};

__host__ __device__ __forceinline__ void tr_step_layer(int g, int& l, int& kc) {      // forward: step -> layer, K-chunk
  l = g < 2 ? 0 : g < 10 ? 1 : g < 18 ? 2 : g < 26 ? 3 : g < 36 ? 4 : g < 44 ? 5 : g < 52 ? 6 : 7;
  const int start = l == 0 ? 0 : l < 5 ? 2 + 8 * (l - 1) : 36 + 8 * (l - 5);
  kc = g - start;
  if (l == 4) kc = kc < 2 ? 8 + kc : kc - 2;          // the two skip chunks (input columns 256 .. 319) run first
}

// weights in the order of use: step g, wavefront w, 16-wide half ks, plane (hi | lo), lane -> 8 halfs:
// forward   W_l[32 w + (lane & 31)][32 kc + 16 ks + 8 (lane >> 5) .. + 7]            (src[l] = W_l, [256][ld])
// backward  stage s = 0: R0[i][o], stages 1 .. 7: W_{8 - s}[i][o] with o = 32 w + (lane & 31), i = 32 kc + 16 ks + 8 (lane >> 5) .. + 7
//           (src[s] = the stage's matrix with rows i, [.][ld]: the transposed read happens here, once per pass)
// scaled by the matrix' power of two (slot mx_w[stage])
struct TrunkPackJobs { const float* src[9]; int ld[9]; int mx_w[9]; int nsteps; int nw; };
template <bool BWD>
static __global__ __launch_bounds__(256) void k_pack_trunk(TrunkPackJobs J, const float* __restrict__ mx, unsigned char* __restrict__ dst) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= J.nsteps * 1024) return;
  const int lane = e & 63, ks = (e >> 6) & 1, w = (e >> 7) & 7, g = e >> 10;
  int l, kc;
  if (BWD) { l = g < 4 ? 0 : 1 + ((g - 4) >> 3); kc = g < 4 ? g : (g - 4) & 7; }
  else if (g < TR_STEPS) tr_step_layer(g, l, kc);
  else { l = 8; kc = g == TR_STEPS ? 8 : g - TR_STEPS - 2; }          // head: the view chunk (input columns 256 .. 287), a zero step, chunks 0 .. 7
  const float s = pp_split_scale(mx[J.mx_w[l]]);
  // column block this slot of the stream feeds: the wavefront's own everywhere except in the head stage, whose 128 columns are
  // blocks 0 .. 3: with four wavefronts (two slots each) wavefront w takes block w in its FIRST slot, with eight wavefront w < 4 block w
  int blk = w;
  bool zero = false;
  if (!BWD && l == 8) {
    if (J.nw == 4) { blk = w >> 1; zero = (w & 1) != 0; } else zero = w >= 4;
    if (g == TR_STEPS + 1) zero = true;
  }
  const int o = 32 * blk + (lane & 31), i0 = kc * 32 + ks * 16 + (lane >> 5) * 8;
  float v[8];
  if (zero) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 0.f;
  } else if (BWD) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = J.src[l][(size_t)(i0 + u) * J.ld[l] + o];
  } else {
    const float* p = J.src[l] + (size_t)o * J.ld[l] + i0;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  pp_half8 h, lo;
  pp_split8(v, s, h, lo);
  unsigned char* q = dst + (size_t)g * TR_WSTEP + w * 4096 + ks * 2048 + lane * 16;
  *reinterpret_cast<pp_half8*>(q) = h;
  *reinterpret_cast<pp_half8*>(q + 1024) = lo;
}

// one K-chunk: TM x TU x 6 matrix instructions per wavefront on the chunk image `img` (TM blocks of 32 rows, `rb` bytes apart)
// and the weight registers wb = TU column blocks x 2 halves x {hi, lo}
template <int TM, int TU, int NU = TU>
__device__ __forceinline__ void tr_step(const unsigned char* img, const pp_half8 (&wb)[TU * 4], f32x16 (&acc)[TM][TU], int l31, int lh) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    pp_half8 ah[TM], al[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = t * 32 + l31;
      if (TR_DBG & 16) { ah[t] = wb[t & 3]; al[t] = wb[3 - (t & 3)]; continue; }
      ah[t] = *reinterpret_cast<const pp_half8*>(img + pl_slot_off(row, ks * 2 + lh));
      al[t] = *reinterpret_cast<const pp_half8*>(img + pl_slot_off(row, 4 + ks * 2 + lh));
    }
    if (TR_DBG & 1) {
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[t][0][ks] += (float)ah[t][0] + (float)al[t][1] + (float)wb[ks * 2][2] + (float)wb[ks * 2 + 1][3];
      continue;
    }
    // small terms first; the three products of one block are TM x TU instructions apart
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[u * 4 + ks * 2 + 1], ah[t], acc[t][u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[u * 4 + ks * 2], al[t], acc[t][u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[u * 4 + ks * 2], ah[t], acc[t][u], 0, 0, 0);
  }
}

// lane `lane_` (a literal) of v_ <- the scalar s_
// (s_nop: the scalar comes straight from a vector compare; the compiler's hazard recogniser does not look into inline assembly,
// and without the wait states the lane receives the PREVIOUS compare's mask - found by the gradient parity test)
#define TR_WRITELANE(v_, s_, lane_) asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(v_) : "s"(s_), "n"(lane_))
#define TR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   // LDS-only: the loads in flight stay in flight

// BWD = false: stages = layers 0 .. 7 (bias, ReLU, masks out, density head at layer 6), streamed chunks: 2 in front of layers 0 and 4
// BWD = true:  stage 0 = d(layer 7) from d(hidden) through R0 (4 streamed chunks), stage s = d(layer 7 - s) through W_{8 - s};
//              epilogue = the mask of the stage's output (+ the density term at stage 1)
// NW = 8: eight wavefronts on a 128-row tile, one work-group per CU (wavefront w: column block w, all four 32-row blocks);
// NW = 4: four wavefronts on a 64-row tile, TWO work-groups per CU (wavefront w: column blocks 2 w and 2 w + 1, two row blocks).
//         Every output element costs ~10 vector instructions of epilogue (scale, bias, ReLU, mask bit, split into hi / lo) against
//         768 multiply-adds on the matrix pipe, so with both wavefronts of a SIMD in the same work-group - in the epilogue at
//         the same time, behind the same barriers - the matrix pipe idles for 45 % of a stage (phase timers: 12.4 k ticks of
//         steps, 11.5 k of epilogue + conversion per stage and tile).  Two independent work-groups per CU drift apart, so one
//         wavefront's epilogue runs beside the other's matrix instructions; the price is that each streams the weights for half
//         as many rows (2 x the L2 -> CU weight traffic).
template <bool BWD, bool MROW, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_nerf_trunk(TrunkArgs T, const int32_t* __restrict__ count, int rcap) {
  constexpr int TM = NW / 2;            // 32-row blocks of a tile
  constexpr int TU = 8 / NW;            // 32-column blocks of a wavefront
  constexpr int TR = 32 * TM;           // rows of a tile
  constexpr int CH = TR * 128;          // bytes of one chunk image
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * CH + CH + 8 * TR * 4 + 128 + 3 * 1024 + 64];
  unsigned char* const Img = smem;
  unsigned char* const E = smem + 8 * CH;
  float* const dpart = reinterpret_cast<float*>(E + CH);                      // [8 column blocks][TR rows]
  float* const tmax = dpart + 8 * TR;                                         // [8] tile maxima, [8] running maxima, [8] weight scales
  float* const lmax = tmax + 8;
  float* const swl = lmax + 8;
  float* const bl = swl + 16;                                                 // [2][256] bias of this / the next stage, [256] density weights, density bias
  float* const wdl_ = bl + 512;                                               // (epilogue operands: no global load, no vmcnt wait there)
  const int R = min(count[0], rcap);
  const int ntiles = (R + TR - 1) / TR;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const float in_max = T.mx[T.mx_in];
  const float sE = pp_split_scale(in_max);
  const bool head = !BWD && T.head;                     // forward: a ninth stage, the colour head's hidden layer
  const int NS = head ? 9 : 8, nsteps = head ? TR_STEPS + 10 : TR_STEPS;
  if (tid < 8) { tmax[tid] = 0.f; lmax[tid] = 0.f; }
  if (tid < NS) swl[tid] = pp_split_scale(T.mx[T.mx_w[tid]]);
  if (tid == 0) wdl_[256] = (!BWD && T.bd) ? T.bd[0] : 0.f;
  for (int i = tid; i < 256; i += NT) { wdl_[i] = T.wd ? T.wd[i] : 0.f; bl[i] = BWD ? 0.f : T.bias[0][i]; }

  // weight stream: this lane's 16 bytes of the TU x four 1 KB pieces of a step; gs = the next step to fetch
  const unsigned char* const wbase = T.wstream + (TU * w) * 4096 + lane * 16;
  int gs = 0;
  pp_half8 wb0[TU * 4], wb1[TU * 4];
#define TR_WLOAD(wb)                                                                                      \
  do {                                                                                                    \
    const unsigned char* p_ = wbase + (size_t)gs * TR_WSTEP;                                              \
    if (!(TR_DBG & 8) || first_)                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < TU * 4; ++i_) (wb)[i_] = *reinterpret_cast<const pp_half8*>(p_ + i_ * 1024); \
    gs = gs + 1 == nsteps ? 0 : gs + 1;                                                                   \
  } while (0)
  bool first_ = true;
  TR_WLOAD(wb0);
  TR_WLOAD(wb1);
  first_ = false;

  // streamed input of a tile: thread -> row tid / 4, eight columns at 8 (tid & 3) of a 32-wide chunk; pe holds two chunks
  float4 pe[4];
#define TR_ELOAD1(slot_, tile_, chunk_)                                                                   \
  do {                                                                                                    \
    const int row_ = min((tile_) * TR + (tid >> 2), R - 1);                                               \
    const float* p_ = T.in + (size_t)row_ * T.in_ld + (chunk_) * 32 + (tid & 3) * 8;                      \
    pe[2 * (slot_)] = *reinterpret_cast<const float4*>(p_);                                               \
    pe[2 * (slot_) + 1] = *reinterpret_cast<const float4*>(p_ + 4);                                       \
  } while (0)
#define TR_ELOAD(tile_) do { TR_ELOAD1(0, tile_, 0); TR_ELOAD1(1, tile_, 1); } while (0)
#define TR_ECONV(c_, s_)                                                                                  \
  do {                                                                                                    \
    const float v_[8] = {pe[2 * (c_)].x, pe[2 * (c_)].y, pe[2 * (c_)].z, pe[2 * (c_)].w,                  \
                         pe[2 * (c_) + 1].x, pe[2 * (c_) + 1].y, pe[2 * (c_) + 1].z, pe[2 * (c_) + 1].w}; \
    pp_half8 h_, l_;                                                                                      \
    pp_split8(v_, (s_), h_, l_);                                                                          \
    *reinterpret_cast<pp_half8*>(E + pl_slot_off(tid >> 2, tid & 3)) = h_;                                \
    *reinterpret_cast<pp_half8*>(E + pl_slot_off(tid >> 2, 4 + (tid & 3))) = l_;                          \
  } while (0)
  // head stage: the tile's 32 view-encoding columns, fetched a stage ahead
  float4 pv[2];
#define TR_VLOAD(tile_)                                                                                   \
  do {                                                                                                    \
    const int row_ = min((tile_) * TR + (tid >> 2), R - 1);                                               \
    const float* p_ = T.in2 + (size_t)row_ * T.in2_ld + (tid & 3) * 8;                                    \
    pv[0] = *reinterpret_cast<const float4*>(p_);                                                         \
    pv[1] = *reinterpret_cast<const float4*>(p_ + 4);                                                     \
  } while (0)
  TR_ELOAD((int)blockIdx.x);
  TR_BARRIER();
  int bp = 0;                                           // which half of bl holds the current stage's bias
#ifdef TR_TIMERS
  unsigned long long tsum[8] = {0}, tprev = __builtin_readcyclecounter();
#endif

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * TR;
    float sA = sE;                                      // scale of the current stage's input
    for (int l = 0; l < NS; ++l) {
      f32x16 acc[TM][TU];
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
      // forward: the next stage's bias on its way to LDS (fetched here, written behind barrier A, read a stage later)
      float bnext = 0.f;
      if (!BWD && tid < (l == 7 && head ? 128 : 256)) bnext = T.bias[l + 1 == NS ? 0 : l + 1][tid];
      TR_TICK(6);
      if (!BWD && l == 8) {                             // head: the view columns through E, then a zero-weight step (keeps the two weight register sets in step)
        const float v_[8] = {pv[0].x, pv[0].y, pv[0].z, pv[0].w, pv[1].x, pv[1].y, pv[1].z, pv[1].w};
        pp_half8 h_, l_;
        pp_split8(v_, sA, h_, l_);
        *reinterpret_cast<pp_half8*>(E + pl_slot_off(tid >> 2, tid & 3)) = h_;
        *reinterpret_cast<pp_half8*>(E + pl_slot_off(tid >> 2, 4 + (tid & 3))) = l_;
        TR_BARRIER();
        tr_step<TM, TU>(E, wb0, acc, l31, lh);
        TR_WLOAD(wb0);
        tr_step<TM, TU>(E, wb1, acc, l31, lh);
        TR_WLOAD(wb1);
        TR_BARRIER();
      }
      if (l == 0 || (!BWD && l == 4)) {                 // the streamed columns, one chunk at a time through E
        TR_ECONV(0, sA);
        if (BWD) TR_ELOAD1(0, tile, 2);
        TR_BARRIER();
        tr_step<TM, TU>(E, wb0, acc, l31, lh);
        TR_WLOAD(wb0);
        TR_BARRIER();
        TR_ECONV(1, sA);
        if (BWD) TR_ELOAD1(1, tile, 3);
        TR_BARRIER();
        tr_step<TM, TU>(E, wb1, acc, l31, lh);
        TR_WLOAD(wb1);
        TR_BARRIER();
        if (BWD) {
          TR_ECONV(0, sA);
          TR_BARRIER();
          tr_step<TM, TU>(E, wb0, acc, l31, lh);
          TR_WLOAD(wb0);
          TR_BARRIER();
          TR_ECONV(1, sA);
          TR_BARRIER();
          tr_step<TM, TU>(E, wb1, acc, l31, lh);
          TR_WLOAD(wb1);
          TR_BARRIER();
        }
      }
      TR_TICK(0);
      if (!BWD && l == 3) TR_ELOAD(tile);               // for layer 4 of this tile
      if (l == 7) TR_ELOAD(tile + (int)gridDim.x);      // for stage 0 of the next one (rows are clamped)
      if (!BWD && head && l == 6) TR_VLOAD(tile);       // for the head stage of this one
      // backward: the epilogue's per-row operands, fetched before the resident chunks so that their latency is long over
      // (d raw of the tile's rows waits in LDS - dpart is otherwise unused in the backward chain - from stage 0 to the epilogue of stage 1)
      unsigned mword[TM][TU];
      if (BWD) {
        const uint16_t* __restrict__ br = T.bitsr[l];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int row = r0 + t * 32 + l31;          // (the mask planes are padded to whole 128-row tiles)
            mword[t][u] = br[((size_t)row * 8 + TU * w + u) * 2 + lh];
          }
        if (l == 0 && tid < TR) dpart[tid] = T.draw[(size_t)min(r0 + tid, R - 1) * T.draw_ld];
      }
      if (l > 0) {
        // eight resident chunks, fully unrolled (a rolled loop makes the weight registers loop-carried: the compiler then loads
        // into temporaries and copies them at the latch behind a vmcnt(0), i.e. no prefetch at all).  Behind each chunk's
        // matrix instructions the wavefront also sends 16 rows of that chunk - the PREVIOUS stage's output - to HBM from the
        // image: x = (hi + lo) / s, exactly the operand the matrix instructions consume (22 significant bits), as 128-byte
        // rows; the write traffic is thereby spread evenly over the kernel instead of arriving in one burst per stage that
        // every later weight fetch would have to wait behind (vmcnt retires in order): 712 -> 553 us.
        float* __restrict__ outp = T.out[l - 1];
        const int ldp = T.ld[l - 1];
        const float invs = 1.0f / sA;
        const int drow = 16 * w + (lane >> 2), dc8 = lane & 3;
        const bool dok = r0 + drow < R;
        float* const dptr = outp + (size_t)(r0 + drow) * ldp + 8 * dc8;
#define TR_DRAIN(kc)                                                                                      \
        do {                                                                                              \
            if (!(TR_DBG & 2)) {                                                                        \
              const pp_half8 h = *reinterpret_cast<const pp_half8*>(Img + kc * CH + pl_slot_off(drow, dc8));\
              const pp_half8 lo = *reinterpret_cast<const pp_half8*>(Img + kc * CH + pl_slot_off(drow, 4 + dc8));\
              typedef float tr_f4 __attribute__((ext_vector_type(4)));                                  \
              tr_f4 a, b;                                                                               \
              a.x = ((float)h[0] + (float)lo[0]) * invs; a.y = ((float)h[1] + (float)lo[1]) * invs;     \
              a.z = ((float)h[2] + (float)lo[2]) * invs; a.w = ((float)h[3] + (float)lo[3]) * invs;     \
              b.x = ((float)h[4] + (float)lo[4]) * invs; b.y = ((float)h[5] + (float)lo[5]) * invs;     \
              b.z = ((float)h[6] + (float)lo[6]) * invs; b.w = ((float)h[7] + (float)lo[7]) * invs;     \
              if (dok) {                                                                                \
                if (TR_NT) {                                                                            \
                  __builtin_nontemporal_store(a, reinterpret_cast<tr_f4*>(dptr + 32 * kc));             \
                  __builtin_nontemporal_store(b, reinterpret_cast<tr_f4*>(dptr + 32 * kc + 4));         \
                } else {                                                                                \
                  *reinterpret_cast<tr_f4*>(dptr + 32 * kc) = a;                                        \
                  *reinterpret_cast<tr_f4*>(dptr + 32 * kc + 4) = b;                                    \
                }                                                                                       \
              }                                                                                         \
            }                                                                                           \
        } while (0)
        // (the head stage runs like every other one - its 128 columns sit in the wavefronts' first slots, the other slots multiply
        // zero weights: a narrower variant of the step for that stage alone, as a second unrolled loop or as a test inside this
        // one, cost 260-350 bytes of register spills per lane)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc & 1) { tr_step<TM, TU>(Img + kc * CH, wb1, acc, l31, lh); TR_WLOAD(wb1); }
          else { tr_step<TM, TU>(Img + kc * CH, wb0, acc, l31, lh); TR_WLOAD(wb0); }
          TR_DRAIN(kc);
        }
#undef TR_DRAIN
      }
      TR_TICK(1);
      // ---- epilogue.  forward: bias, ReLU, masks; backward: mask (+ the density term); then the tile maximum
      const float inv = 1.0f / (sA * swl[l]);
      float* __restrict__ out = T.out[l];
      const int ld = T.ld[l];
      float vmax = 0.f;
      const bool hstage = !BWD && l == 8;               // head stage: block w of four, first slot only
      const bool last = l == NS - 1;
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        // (head stage: the slots without a block still run the arithmetic on zeros - skipping it would make the 64 accumulators
        // conditionally updated values, which the register allocator pays for with spills - and store nothing)
        const bool live = !hstage || (u == 0 && (NW == 4 || w < 4));
        const int cb = hstage ? (w & 3) : TU * w + u;   // 32-column block of the stage's output = K-chunk of the next stage
        const float* const bias = bl + bp * 256 + 32 * cb + 4 * lh;
        const float* const wdl = wdl_ + 32 * cb + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>((BWD ? wdl : bias) + 8 * q);
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            float4 v;
            if (BWD) {
              // mask: bit -> all ones / zero by a signed bit-field extract, then AND on the value's bits (2 instructions per value)
              const float d = l == 1 ? dpart[t * 32 + l31] : 0.f;
              const int mw = (int)mword[t][u];
              v.x = __builtin_fmaf(d, b.x, acc[t][u][4 * q] * inv);
              v.y = __builtin_fmaf(d, b.y, acc[t][u][4 * q + 1] * inv);
              v.z = __builtin_fmaf(d, b.z, acc[t][u][4 * q + 2] * inv);
              v.w = __builtin_fmaf(d, b.w, acc[t][u][4 * q + 3] * inv);
              v.x = __uint_as_float(__float_as_uint(v.x) & (unsigned)__builtin_amdgcn_sbfe(mw, 4 * q, 1));
              v.y = __uint_as_float(__float_as_uint(v.y) & (unsigned)__builtin_amdgcn_sbfe(mw, 4 * q + 1, 1));
              v.z = __uint_as_float(__float_as_uint(v.z) & (unsigned)__builtin_amdgcn_sbfe(mw, 4 * q + 2, 1));
              v.w = __uint_as_float(__float_as_uint(v.w) & (unsigned)__builtin_amdgcn_sbfe(mw, 4 * q + 3, 1));
              vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            } else {
              // (the product with the power of two is exact, so the fused multiply-add rounds exactly as multiply, add would)
              v.x = fmaxf(__builtin_fmaf(acc[t][u][4 * q], inv, b.x), 0.f);
              v.y = fmaxf(__builtin_fmaf(acc[t][u][4 * q + 1], inv, b.y), 0.f);
              v.z = fmaxf(__builtin_fmaf(acc[t][u][4 * q + 2], inv, b.z), 0.f);
              v.w = fmaxf(__builtin_fmaf(acc[t][u][4 * q + 3], inv, b.w), 0.f);
              vmax = fmaxf(vmax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
            acc[t][u][4 * q] = v.x; acc[t][u][4 * q + 1] = v.y; acc[t][u][4 * q + 2] = v.z; acc[t][u][4 * q + 3] = v.w;
            const int row = r0 + t * 32 + l31;
            if (last && live && row < R && (!(TR_DBG & 2) || v.x == 123.456f)) *reinterpret_cast<float4*>(out + (size_t)row * ld + 32 * cb + 8 * q + 4 * lh) = v;   // (the other stages leave through the image)
          }
        }
      }
      if (!BWD && !hstage && !(TR_DBG & 4)) {
        if (MROW) {
          uint16_t* __restrict__ br = T.bitsr[l];
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TU; ++u) {
              // bit j <-> accumulator register j of the lane.  The values are ReLU outputs (+0 or positive): bits + 0x7FFFFFFF
              // carries into bit 31 exactly for the positive ones, and v_alignbit shifts that bit into the word - 2 instructions per value
              unsigned m = 0u;
#pragma unroll
              for (int j = 15; j >= 0; --j) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[t][u][j]) + 0x7FFFFFFFu, 31);
              const int row = r0 + t * 32 + l31;
              if (row < R) br[((size_t)row * 8 + TU * w + u) * 2 + lh] = (uint16_t)m;
            }
        } else {
          uint32_t* __restrict__ bits = T.bits[l];
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TU; ++u) {
              // lane c of rm <- the 32 row bits of column c: the 64-lane compare mask of accumulator register j is rows 0..31 of
              // column (j & 3) + 8 (j >> 2) in its low word and of that column + 4 in its high word
              unsigned rm = 0u;
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const unsigned long long bal = __ballot(acc[t][u][j] > 0.f);
                const unsigned blo = (unsigned)bal, bhi = (unsigned)(bal >> 32);
                TR_WRITELANE(rm, blo, (j & 3) + 8 * (j >> 2));
                TR_WRITELANE(rm, bhi, (j & 3) + 8 * (j >> 2) + 4);
              }
              // word of row half h' of a column = nibbles h', h' + 2, h' + 4, h' + 6 of its row bits; all 32 columns at once
              unsigned e = rm & 0x0F0F0F0Fu, o = (rm >> 4) & 0x0F0F0F0Fu;
              e = (e | (e >> 4)) & 0x00FF00FFu;
              o = (o | (o >> 4)) & 0x00FF00FFu;
              e = (e | (e >> 8)) & 0x0000FFFFu;
              o = (o | (o >> 8)) & 0x0000FFFFu;
              if (lane < 32) bits[((size_t)(r0 >> 5) + t) * 256 + 32 * (TU * w + u) + lane] = e | (o << 16);
            }
        }
      }
      // maximum of each row of 16 lanes by four DPP moves (no LDS round trips, unlike __shfl_xor), then four lanes per
      // wavefront into the LDS slot (all 64 lanes on one address measured far slower: +120 k ticks per work-group)
      {
        int vi = (int)__float_as_uint(vmax);              // non-negative floats order as integers
        vi = max(vi, __builtin_amdgcn_mov_dpp(vi, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
        vi = max(vi, __builtin_amdgcn_mov_dpp(vi, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
        vi = max(vi, __builtin_amdgcn_mov_dpp(vi, 0x141, 0xF, 0xF, true));    // row_half_mirror
        vi = max(vi, __builtin_amdgcn_mov_dpp(vi, 0x140, 0xF, 0xF, true));    // row_mirror
        if ((lane & 15) == 0 && !hstage) atomicMax(reinterpret_cast<unsigned int*>(tmax + l), (unsigned)vi);
      }
      if (tid == 0) tmax[(l + 7) & 7] = 0.f;            // the slot of the stage before: read long ago, next written a tile from now
      const bool dens = !BWD && l == 6 && T.wd;
      if (dens) {
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const float* const wdl = wdl_ + 32 * (TU * w + u) + 4 * lh;
          float p[TM];
#pragma unroll
          for (int t = 0; t < TM; ++t) p[t] = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 c = *reinterpret_cast<const float4*>(wdl + 8 * q);
#pragma unroll
            for (int t = 0; t < TM; ++t)
              p[t] += acc[t][u][4 * q] * c.x + acc[t][u][4 * q + 1] * c.y + acc[t][u][4 * q + 2] * c.z + acc[t][u][4 * q + 3] * c.w;
          }
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            p[t] += __shfl_xor(p[t], 32, 64);
            if (lh == 0) dpart[(TU * w + u) * TR + t * 32 + l31] = p[t];
          }
        }
      }
      TR_TICK(2);
      TR_BARRIER();                                     // A: every wavefront is done with the image; maximum and partials complete
      TR_TICK(3);
      const float tm = hstage ? 0.f : tmax[l];
      if (tid == 0 && !hstage) lmax[l] = fmaxf(lmax[l], tm);
      if (!BWD && tid < 256) bl[(bp ^ 1) * 256 + tid] = bnext;
      bp ^= 1;
      if (dens && tid < TR) {
        float s = wdl_[256];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += dpart[u * TR + tid];
        if (r0 + tid < R) { T.raw[r0 + tid] = s; T.density[r0 + tid] = s > 20.f ? s : log1pf(expf(s)); }
      }
      if (!last) {
        // layer 4 reads the encoded points, the head the view encoding (bounded by 1), at the same scale as the tile
        sA = pp_split_scale((!BWD && l == 3) ? fmaxf(tm, in_max) : (!BWD && l == 7) ? fmaxf(tm, 1.f) : tm);
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          unsigned char* const chunk = Img + (TU * w + u) * CH;
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              pp_half4 h, lo;
              pp_split4(make_float4(acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]), sA, h, lo);
              const int row = t * 32 + l31;
              *reinterpret_cast<pp_half4*>(chunk + pl_slot_off(row, q) + 8 * lh) = h;
              *reinterpret_cast<pp_half4*>(chunk + pl_slot_off(row, 4 + q) + 8 * lh) = lo;
            }
        }
      }
      TR_TICK(4);
      TR_BARRIER();                                     // B: the next stage's image is complete
      TR_TICK(5);
    }
  }
  if (tid < 8) pp_record_max_lane(T.mx + T.mx_out[tid], lmax[tid]);      // (skips the atomic when the slot already holds as much: 512 work-groups end together)
#ifdef TR_TIMERS
  if (tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_tr_t[(BWD ? 8 : 0) + i], tsum[i]);
#endif
#undef TR_WLOAD
#undef TR_ELOAD
#undef TR_ELOAD1
#undef TR_VLOAD
#undef TR_ECONV
}
