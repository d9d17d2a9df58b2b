// Split-precision NT GEMM, second generation, for the 256-wide layers of the scene branch:
//   C[r][0..255] = epi( sum_k A[r][k] W[n][k] ),  fp32 in memory, three fp16 products per fp32 product (pp_gemm_split.h).
//
// What the first generation (k_gemm128s) spends its time on is not the matrix pipe (busy 16-20 % of the CU-busy cycles,
// profiles/r01_scene_pmc_split.json) but operand preparation: every work-group re-splits a 128 x 32 weight tile per K-chunk,
// and the two column blocks of a 256-wide layer each fetch and re-split the same activation tile.  Here
//   * the WEIGHTS are split once per pass by k_pack_planes into the exact byte image the kernel wants in LDS - per 32-wide
//     K-chunk a [256 columns][128 B] block, row = 32 hi halfs | 32 lo halfs, 16-byte slots XOR-swizzled - so a chunk's 32 KB
//     arrive by plain linear LDS-direct loads (global_load_lds_dwordx4): no registers, no conversion, no ds_write;
//   * one work-group of EIGHT wavefronts (2 x 4, 64 x 64 outputs each = the register footprint of the old kernel) owns a
//     128 x 256 tile: the activation tile is fetched and split ONCE for all 256 columns (a quarter of the old per-wavefront
//     conversion work), weights are double-buffered, ONE barrier per K-chunk;
//   * LDS images are swizzled so that every ds_read_b128 fragment read is conflict-free (MI355X_MICROARCH.md, LDS: a
//     ds_read_b128 is served in 4 groups of 16 lanes over 64 banks; rows are 128 B = half the bank row, slot s of row r sits
//     at slot s ^ ((r >> 1) & 7), so the 16 rows of a lane group cover all 64 banks exactly once).
// Per K-chunk and CU: 1536 matrix-pipe cycles per SIMD against 512 LDS-read + ~330 LDS-write cycles and 16 KB of HBM reads, i.e.
// the kernel is bound by HBM (1 KB in + 1 KB out per row and layer: 54 us per layer at 131 k rows and 5 TB/s), not by
// operand preparation.
#pragma once
#include "pp_gemm_split.h"

#define PL_A_BYTES (128 * 128)      // one activation chunk image: 128 rows x (32 hi | 32 lo halfs)
#define PL_B_BYTES (256 * 128)      // one weight chunk image: 256 columns x (32 hi | 32 lo halfs)

__device__ __forceinline__ int pl_slot_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------------- weight images
// job q: src[256 rows][ld] fp32 (row n = output column n of the GEMM, K_q columns used, K_q % 32 == 0) ->
// dst[K_q / 32][256][64 halfs] in the swizzled LDS image order, scaled by the power of two derived from *mx_q.
struct PlanePackJobs { const float* src[16]; _Float16* dst[16]; int ld[16], K[16], mx_slot[16]; int n; };

static __global__ __launch_bounds__(256) void k_pack_planes(PlanePackJobs J, const float* __restrict__ mx) {
  const int q = blockIdx.y;
  const float* __restrict__ src = J.src[q];
  unsigned char* __restrict__ dst = reinterpret_cast<unsigned char*>(J.dst[q]);
  const int ld = J.ld[q], nk = J.K[q] >> 5;
  const float s = pp_split_scale(mx[J.mx_slot[q]]);
  const int total = nk * 256 * 8;                       // float4 pieces: chunk, row, c4
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c4 = e & 7, row = (e >> 3) & 255, kc = e >> 11;
    const float4 x = *reinterpret_cast<const float4*>(src + (size_t)row * ld + kc * 32 + c4 * 4);
    pp_half4 h, l;
    pp_split4(x, s, h, l);
    unsigned char* img = dst + (size_t)kc * PL_B_BYTES;
    *reinterpret_cast<pp_half4*>(img + pl_slot_off(row, c4 >> 1) + (c4 & 1) * 8) = h;
    *reinterpret_cast<pp_half4*>(img + pl_slot_off(row, 4 + (c4 >> 1)) + (c4 & 1) * 8) = l;
  }
}

// ---------------------------------------------------------------------------------------------------- the GEMM
// grid = min(row tiles, CUs) persistent work-groups of 512 threads; Nout = 256 exactly; K % 32 == 0, K >= 64.
// EPI_RELU: C = relu(acc + bias), ReLU bits written to bits16, max|C| recorded;  EPI_MASK: C = bit ? acc : 0 (bits16 of the
// forward activation), max|C| recorded.  Layout of bits16: pp_gemm.h gemm_epilogue (row group of 32, 256 columns, 2 halves).
//
// PIPELINE.  A work-group's work is ONE linear sequence of steps g = (tile, 32-wide K-chunk).  Every global read is an
// LDS-direct load (no register destination, explicit counted waits):
//   * activations (HBM): 16 KB of fp32 per step into a 3-slot staging ring, issued four steps ahead of their use; a
//     wavefront converts exactly the 2 x 1 KB it fetched itself (its own vmcnt wait is the only ordering an LDS-direct load
//     has) into the hi / lo image of the step;
//   * weights (L2): each column half of the next step's image (16 KB) into that half's other buffer.
//
// PING-PONG.  Phase timers (s_memtime) on a single-phase version showed the matrix pipe busy for a third of a step: all eight
// wavefronts sat behind the same barrier, so the two wavefronts of a SIMD converted, waited and issued loads together and then
// competed for the pipe together.  Here the work-group is two halves of four wavefronts (one per SIMD each; half h owns the
// output columns 128 h .. 128 h + 127 of the 128-row tile): in phase A of a step half 0 runs its 24 MFMAs per wavefront on
// the images of step g while half 1 does the memory work for step g + 1 (waits for / converts its rows of the activation
// chunk, issues the loads of its weight half and of the activations four steps on); phase B swaps the roles; one LDS-only
// barrier per phase.  Every SIMD thus always has one wavefront on the matrix pipe and one doing vector / LDS / memory work.
//
// vmcnt bookkeeping: a wavefront's memory operations complete in order (loads, stores, LDS-direct loads alike), so "these
// loads have landed" is "at most n operations are outstanding", n = operations this wavefront issued after them.  The
// counts live in scalar registers and are mapped onto the immediates of pl_wait_vm (a smaller immediate only waits longer).
#define PL_S_BYTES (128 * 128)      // one staging slot: 128 rows x 32 fp32
#define PL_DEPTH 3
#define PL_BH_BYTES (128 * 128)     // one column half of a weight chunk image
#ifndef PL_PINGPONG
#define PL_PINGPONG 0  // 1: two-phase ping-pong schedule (measured SLOWER: 113 vs 96 us at 131 k rows, kept for experiments)
#endif
#ifndef PL_DBG
#define PL_DBG 0      // experiments only: 1 = no MFMAs, 2 = activations always from the work-group's first tile, 3 = no C stores
#endif

#define PL_VM(n) (0x0F70 | ((n) & 15) | (((n) >> 4) << 14))     // s_waitcnt immediate: vmcnt(n) only (gfx9 encoding)
__device__ __forceinline__ void pl_wait_vm(int n) {     // waits until at most n' <= n operations are outstanding
  if (n >= 63) __builtin_amdgcn_s_waitcnt(PL_VM(63));
  else if (n >= 48) __builtin_amdgcn_s_waitcnt(PL_VM(48));
  else if (n >= 32) __builtin_amdgcn_s_waitcnt(PL_VM(32));
  else if (n >= 24) __builtin_amdgcn_s_waitcnt(PL_VM(24));
  else if (n >= 16) __builtin_amdgcn_s_waitcnt(PL_VM(16));
  else if (n >= 14) __builtin_amdgcn_s_waitcnt(PL_VM(14));
  else if (n >= 12) __builtin_amdgcn_s_waitcnt(PL_VM(12));
  else if (n >= 10) __builtin_amdgcn_s_waitcnt(PL_VM(10));
  else if (n >= 8) __builtin_amdgcn_s_waitcnt(PL_VM(8));
  else if (n >= 6) __builtin_amdgcn_s_waitcnt(PL_VM(6));
  else if (n >= 4) __builtin_amdgcn_s_waitcnt(PL_VM(4));
  else if (n >= 2) __builtin_amdgcn_s_waitcnt(PL_VM(2));
  else __builtin_amdgcn_s_waitcnt(PL_VM(0));
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void k_gemm256p(const float* __restrict__ A, int lda, const _Float16* __restrict__ Wimg, int K,
                                                     const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                     const int32_t* __restrict__ count, int rcap, const float* __restrict__ a_max,
                                                     const float* __restrict__ w_max, float* __restrict__ c_max,
                                                     uint16_t* __restrict__ bits16) {
  constexpr int BM = 128;
  // ONE shared object (a second one would make the compiler serialise the LDS-direct loads against unrelated LDS traffic)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * PL_A_BYTES + 4 * PL_BH_BYTES + PL_DEPTH * PL_S_BYTES + 8 * 512];
  unsigned char* const As = smem;                                        // [2][PL_A_BYTES]       hi / lo image of the activation chunk
  unsigned char* const Bs = smem + 2 * PL_A_BYTES;                       // [2 halves][2][PL_BH_BYTES] weight images
  unsigned char* const St = Bs + 4 * PL_BH_BYTES;                        // [PL_DEPTH][PL_S_BYTES] fp32 staging ring
  unsigned char* const Mk = St + PL_DEPTH * PL_S_BYTES;                  // [8 wavefronts][2][256 B] mask words of the tile
  const int R = min(count[0], rcap);
  const int ntiles = (R + BM - 1) / BM;
  if ((int)blockIdx.x >= ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  // the wavefront index in a SCALAR register: everything that depends on it (the role of the wavefront in a phase, the
  // operation counters, the wait immediates) is then uniform control flow instead of exec-masked vector code
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wid >> 2, w4 = wid & 3;              // half: column half and phase of the matrix work
  const int wr = w4 >> 1, wq = w4 & 1;                  // 64 x 64 outputs per wavefront: rows 64 wr, columns 128 half + 64 wq
  const int l31 = lane & 31, lh = lane >> 5;
  const float sA = pp_split_scale(a_max[0]), sW = pp_split_scale(w_max[0]);
  const float inv = 1.0f / (sA * sW);
  const int nk = K >> 5;
  // a half FETCHES the other half's weights (see the pipeline note at dma_b) and COMPUTES on its own
  const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(Wimg) + (PL_PINGPONG ? half ^ 1 : half) * PL_BH_BYTES;
  unsigned char* const Bmine = Bs + half * 2 * PL_BH_BYTES;
  unsigned char* const Bother = Bs + (PL_PINGPONG ? half ^ 1 : half) * 2 * PL_BH_BYTES;   // where this wavefront's weight loads go
  const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  const int nsteps = my_tiles * nk;
  // loop-invariant epilogue operands are fetched before the pipeline starts
  float bcols[2] = {0.f, 0.f};
  if (EPI == EPI_RELU && bias) { bcols[0] = bias[half * 128 + wq * 64 + l31]; bcols[1] = bias[half * 128 + wq * 64 + 32 + l31]; }
  __builtin_amdgcn_s_waitcnt(PL_VM(0));                 // the counting starts at zero

  // ---- memory-operation bookkeeping (scalar): operations issued after each staging slot / weight image / mask fetch
  // The staging slots are a FIFO (fetched and converted in the same order): f0 / f1 / f2 = operations issued after the
  // oldest / middle / newest outstanding slot.  Shifting three scalars instead of indexing counters by slot number keeps
  // them in scalar registers (an indexed triple is turned into a scratch-memory array, whose accesses are themselves
  // vector-memory operations).
  int f0 = 0, f1 = 0, f2 = 0, after_b = 0, after_m = 0;
  // (a macro, not a lambda: a lambda called from the lambdas below would be captured by reference as an object, and the
  // counters behind two levels of references end up in scratch memory instead of scalar registers)
#define PL_ISSUED(n) do { f0 += (n); f1 += (n); f2 += (n); after_b += (n); after_m += (n); } while (0)

  // activation fetch of one step: wavefront w owns the 1 KB blocks 2 w and 2 w + 1 of the 16 KB chunk (8 rows x 128 B each);
  // rows past the end re-read the last row (never stored)
  int ptile = blockIdx.x, pk = 0, pslot = 0, nissued = 0;
  auto prefetch_a = [&]() {                             // fetch the next step's chunk into the newest FIFO position
    if (nissued >= nsteps) { f2 = 1 << 20; return; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = wid * 2 + i;
      const int row = min((PL_DBG == 2 ? (int)blockIdx.x : ptile) * BM + blk * 8 + (lane >> 3), R - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)row * lda + pk * 32 + (lane & 7) * 4),
                                       (__attribute__((address_space(3))) void*)(St + pslot * PL_S_BYTES + blk * 1024), 16, 0, 0);
    }
    PL_ISSUED(2);
    f2 = 0;
    ++nissued;
    if (++pk == nk) { pk = 0; ptile += gridDim.x; }
    if (++pslot == PL_DEPTH) pslot = 0;
  };
  // this wavefront's 2 KB of a staging slot (rows 16 w .. 16 w + 15) -> hi / lo image.  The two staging reads are inline
  // assembly on purpose: a ds_read the compiler can see makes its wait-count pass put a vmcnt(0) in front of it (it cannot
  // tell the slot that has landed from the ones still being filled), which would drain the whole prefetch ring every step.
  auto convert_a = [&](int slot, unsigned char* img) {  // consumes the oldest FIFO position
    pl_wait_vm(f0);
    f0 = f1; f1 = f2;
    asm volatile("" ::: "memory");
    const unsigned s0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(St + slot * PL_S_BYTES + (wid * 2) * 1024 + lane * 16);
    float4 x[2];
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]) : "v"(s0) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = wid * 2 + i;
      const int row = blk * 8 + (lane >> 3), c4 = lane & 7;
      pp_half4 h, l;
      pp_split4(x[i], sA, h, l);
      *reinterpret_cast<pp_half4*>(img + pl_slot_off(row, c4 >> 1) + (c4 & 1) * 8) = h;
      *reinterpret_cast<pp_half4*>(img + pl_slot_off(row, 4 + (c4 >> 1)) + (c4 & 1) * 8) = l;
    }
  };
  // 16 KB of a weight chunk image - the OTHER half's columns: linear copy, 4 x 1 KB per wavefront.  A half's memory phase
  // directly precedes its own matrix phase of the next step but lies a full phase before the other half's, so the loads a
  // half issues feed the other half: they have one and a half phases to land (L2 latency ~1000 cycles) and are waited for
  // by their issuer at the end of its next matrix phase, in front of the barrier that opens the consumer's matrix phase.
  auto dma_b = [&](int kc, unsigned char* img) {
    const unsigned char* src = Wb + (size_t)kc * PL_B_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = (i * 4 + w4) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + blk + lane * 16),
                                       (__attribute__((address_space(3))) void*)(img + blk), 16, 0, 0);
    }
    PL_ISSUED(4);
    after_b = 0;
  };
  // ReLU bits of a tile (EPI_MASK): lane (l31, lh) fetches the aligned dword that holds both row halves' words of column
  // u = lh of its two column blocks, for both 32-row groups t -> Mk[wavefront][t][lane]; no VGPR destination (an
  // asynchronous load into a register would have to own that register until it lands, which the compiler cannot know)
  auto fetch_masks = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint16_t* mp = bits16 + ((((size_t)(tile * BM + wr * 64 + t * 32) >> 5) * 256 + half * 128 + wq * 64 + lh * 32 + l31) * 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mp,
                                       (__attribute__((address_space(3))) void*)(Mk + (wid * 2 + t) * 256), 4, 0, 0);
    }
    PL_ISSUED(2);
    after_m = 0;
  };
  auto read_masks = [&](unsigned (&m)[2][2]) {
    pl_wait_vm(after_m);
    asm volatile("" ::: "memory");
    const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(Mk + (wid * 2) * 256 + l31 * 4);
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:128\n\tds_read_b32 %2, %4 offset:256\n\tds_read_b32 %3, %4 offset:384\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=&v"(m[0][0]), "=&v"(m[0][1]), "=&v"(m[1][0]), "=&v"(m[1][1]) : "v"(a0) : "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) m[t][u] = (m[t][u] >> (16 * lh)) & 0xFFFFu;
  };

  // ---- prologue: weights of step 0, slots of steps 0 .. 2, image of step 0, then the slot of step 3
  dma_b(0, Bother);
  prefetch_a(); f0 = f2;                                // FIFO filling: the first fetch becomes the oldest position,
  prefetch_a(); f1 = f2;                                // the second the middle one, the third stays the newest
  prefetch_a();
  convert_a(0, As);                                     // its wait covers the weights issued before it
  prefetch_a();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  float vmax = 0.f;
#if PL_PINGPONG
  int g = 0, cslot = 1;                                 // cslot: staging slot of step g + 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * BM;
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
    for (int kc = 0; kc < nk; ++kc, ++g) {
      const int par = g & 1;
      unsigned char* const Acur = As + par * PL_A_BYTES;
      unsigned char* const Bcur = Bmine + par * PL_BH_BYTES;
#pragma unroll
      for (int phase = 0; phase < 2; ++phase) {
        if (phase == half) {
          // ---- matrix phase: 24 MFMAs on the images of step g
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            pp_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int row = wr * 64 + t * 32 + l31;
              ah[t] = *reinterpret_cast<const pp_half8*>(Acur + pl_slot_off(row, ks * 2 + lh));
              al[t] = *reinterpret_cast<const pp_half8*>(Acur + pl_slot_off(row, 4 + ks * 2 + lh));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int colh = wq * 64 + u * 32 + l31;  // column within the half (the swizzle only looks at its low bits)
              bh[u] = *reinterpret_cast<const pp_half8*>(Bcur + pl_slot_off(colh, ks * 2 + lh));
              bl[u] = *reinterpret_cast<const pp_half8*>(Bcur + pl_slot_off(colh, 4 + ks * 2 + lh));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int u = 0; u < 2; ++u) {             // small terms first
                if (PL_DBG == 1) { acc[t][u][0] += (float)ah[t][0] + (float)bl[u][0] + (float)al[t][1] + (float)bh[u][1]; continue; }
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
              }
          }
          if (kc == nk - 1) {
            // ---- end of a tile: outputs straight from the accumulators (same arithmetic and bit layout as k_gemm128s)
            unsigned mw[2][2] = {{0u, 0u}, {0u, 0u}};
            if (EPI == EPI_MASK) read_masks(mw);
            const bool full = r0 + BM <= R;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int col = half * 128 + wq * 64 + u * 32 + l31;
                const int rbase = r0 + wr * 64 + t * 32 + 4 * lh;
                unsigned mbits = mw[t][u];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                  const int row = rbase + (reg & 3) + 8 * (reg >> 2);
                  float val = acc[t][u][reg] * inv;
                  if (EPI == EPI_RELU) {
                    val = fmaxf(val + bcols[u], 0.f);
                    mbits |= (val > 0.f ? 1u : 0u) << reg;
                  } else if (EPI == EPI_MASK) {
                    val = ((mbits >> reg) & 1u) ? val : 0.f;
                  }
                  if (full || row < R) {
                    if (PL_DBG != 3 || val == 123.456f) C[(size_t)row * ldc + col] = val;
                    vmax = fmaxf(vmax, fabsf(val));
                  }
                }
                if (EPI == EPI_RELU) bits16[(((size_t)(rbase - 4 * lh) >> 5) * 256 + col) * 2 + lh] = (uint16_t)mbits;
              }
            if (full) PL_ISSUED(EPI == EPI_RELU ? 68 : 64);    // a partial tile may skip stores: count none (stricter waits only)
          }
          pl_wait_vm(after_b);                          // the weight loads of this wavefront's last memory phase have landed
        } else {
          // ---- memory phase, for step g + 1: the other half's weight image, this wavefront's rows of the activation chunk,
          // the fetch four steps on, the tile's mask words
          if (g + 1 < nsteps) {
            dma_b(kc + 1 < nk ? kc + 1 : 0, Bother + (par ^ 1) * PL_BH_BYTES);
            convert_a(cslot, As + (par ^ 1) * PL_A_BYTES);
          }
          prefetch_a();
          if (EPI == EPI_MASK && kc == 0) fetch_masks(tile);
        }
        // LDS-only barrier: __syncthreads() would add a vmcnt(0) (its memory fence) and drain the prefetch ring
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (++cslot == PL_DEPTH) cslot = 0;
    }
  }
#else
  // ---- single-phase schedule: all eight wavefronts convert, meet at ONE barrier per step, issue the next loads and run
  // their MFMAs together.  (The prologue above has already converted step 0 and fetched slot 3.)
  int g = 0, cslot = 1;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * BM;
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
    for (int kc = 0; kc < nk; ++kc, ++g) {
      const int par = g & 1;
      unsigned char* const Acur = As + par * PL_A_BYTES;
      unsigned char* const Bcur = Bmine + par * PL_BH_BYTES;
      if (g > 0) {
        convert_a(cslot == 0 ? PL_DEPTH - 1 : cslot - 1, Acur);   // (slot number only selects the LDS address)
        pl_wait_vm(after_b);                            // this step's weights (issued a step ago) have landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (g + 1 < nsteps) dma_b(kc + 1 < nk ? kc + 1 : 0, Bmine + (par ^ 1) * PL_BH_BYTES);
      if (g > 0) prefetch_a();
      if (EPI == EPI_MASK && kc == 0) fetch_masks(tile);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        pp_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = wr * 64 + t * 32 + l31;
          ah[t] = *reinterpret_cast<const pp_half8*>(Acur + pl_slot_off(row, ks * 2 + lh));
          al[t] = *reinterpret_cast<const pp_half8*>(Acur + pl_slot_off(row, 4 + ks * 2 + lh));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int colh = wq * 64 + u * 32 + l31;
          bh[u] = *reinterpret_cast<const pp_half8*>(Bcur + pl_slot_off(colh, ks * 2 + lh));
          bl[u] = *reinterpret_cast<const pp_half8*>(Bcur + pl_slot_off(colh, 4 + ks * 2 + lh));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {                 // small terms first
            if (PL_DBG == 1) { acc[t][u][0] += (float)ah[t][0] + (float)bl[u][0] + (float)al[t][1] + (float)bh[u][1]; continue; }
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
          }
      }
      if (++cslot == PL_DEPTH) cslot = 0;
      if (kc == nk - 1) {
        // ---- end of a tile: outputs straight from the accumulators (same arithmetic and bit layout as k_gemm128s)
        unsigned mw[2][2] = {{0u, 0u}, {0u, 0u}};
        if (EPI == EPI_MASK) read_masks(mw);
        const bool full = r0 + BM <= R;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int col = half * 128 + wq * 64 + u * 32 + l31;
            const int rbase = r0 + wr * 64 + t * 32 + 4 * lh;
            unsigned mbits = mw[t][u];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int row = rbase + (reg & 3) + 8 * (reg >> 2);
              float val = acc[t][u][reg] * inv;
              if (EPI == EPI_RELU) {
                val = fmaxf(val + bcols[u], 0.f);
                mbits |= (val > 0.f ? 1u : 0u) << reg;
              } else if (EPI == EPI_MASK) {
                val = ((mbits >> reg) & 1u) ? val : 0.f;
              }
              if (full || row < R) {
                if (PL_DBG != 3 || val == 123.456f) C[(size_t)row * ldc + col] = val;
                vmax = fmaxf(vmax, fabsf(val));
              }
            }
            if (EPI == EPI_RELU) bits16[(((size_t)(rbase - 4 * lh) >> 5) * 256 + col) * 2 + lh] = (uint16_t)mbits;
          }
        if (full) PL_ISSUED(EPI == EPI_RELU ? 68 : 64);    // a partial tile may skip stores: count none (stricter waits only)
      }
    }
  }
#endif
  pp_record_max(c_max, vmax);
}
