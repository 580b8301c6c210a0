// Sparse convolution on bf16 features for gfx950 (MI355X): forward / dgrad on v_mfma_f32_16x16x32_bf16
// with fp32 accumulation (the weight gradient on bf16 rows is k_wgrad_bf16 in conv.hip, next to the
// reduction it shares with the fp32 kernel).
//
// The reference has no reduced-precision path (AT_DISPATCH_FLOATING_TYPES: float / double only,
// src/convolution_gpu.cu:137-155); BASELINE config 3 (MinkUNet34C, bf16) and the north_star's
// "MFMA bf16/fp32 tiles" ask for one.  Semantics: features and weights are rounded to bf16 (RNE),
// products are exact, sums are fp32 in the plan's fixed order, the result is rounded to bf16 once.
//
// Same target-stationary structure and the SAME tile plan as k_conv_tile_f32 (conv.hip) — a workgroup
// owns tile_rows target rows x NC output columns with an fp32 accumulator tile in LDS, walks the
// single-offset batches of its tile, gathers rows one batch ahead into registers, stages them in LDS —
// but a gathered row is half as many bytes, one MFMA covers 32 source channels (8 per lane: ONE
// ds_read_b128 per operand), and a 16-group costs 2 MFMAs per 64 channels instead of 16.  The matrix pipe
// is nearly idle here; the kernel is bound by the LDS traffic of the operand reads + the accumulator
// read-add-write and by the gather, so source-channel chunks are as wide as possible (up to 128: one pass
// over the plan per 128 channels).
#include "conv_common.hpp"

namespace me {

// ME_BF16_TIMING (tagged tuning builds only: scripts/ablate_bf16_tile.sh): s_memtime cycles of wave 0 of every workgroup
// per phase of k_conv_tile_bf16, summed over all launches since the last reset (me_debug_bf16_timing).  Slots 0-9: deep
// pipeline, 10-19: plain loop — barrier A, stage write (with its wait for the gathers), barrier B, load issue, multiply,
// weight refill + next descriptors, prologue, epilogue, batches, workgroups.
__device__ unsigned long long d_bf16_timing[20];

// stage row stride in elements: KC + 16 (32 bytes of padding: the 16 (row, piece) accesses of every lane group in
// which the LDS serves a ds_read_b128 — {0-3,12-15,20-27}, ... — then fall on 16 distinct 16-byte bank slots; with 16
// bytes of padding, the round-1 layout, they were 2-way conflicting)
__host__ __device__ constexpr int conv_bf16_lds_bytes(int nc, int kc, int tile_rows, int batch_groups, bool two_buffers = false) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + (two_buffers ? 2 : 1) * batch_groups * 16 * ((kc + 16) * 2 + 4);
}

// Timing ablations (round 3; INVALID results, never defined by the build scripts: scripts/ablate_bf16_tile.sh compiles
// tagged libraries with them): ME_ABL_NO_AREAD (no LDS operand reads), ME_ABL_NO_ACC (accumulators written without the
// read), ME_ABL_NO_MFMA, ME_ABL_NO_STAGE (no stage writes), ME_ABL_NO_GATHER (every gather hits 64 cached rows),
// ME_ABL_NO_WLOAD (every batch loads offset 0's weights).  Result on MinkUNet34C's layers
// (profiles/r03_ablation_bf16_tile_kernel.log): all LDS traffic together is 9 % of the forward + dgrad time, MFMAs +
// LDS + stage writes + memory misses together 22 %; the rest is the per-batch skeleton (two barriers, descriptor /
// index / address arithmetic of ~300 instructions per wave and batch, tile prologue and epilogue).
// R groups of one offset: per 32-channel step one ds_read_b128 per group and one MFMA, issued
// "transposed" (A = weights, B = gathered rows) so a lane ends with 4 consecutive output columns of one
// target row; then all accumulator reads, then all writes (distinct rows within a batch).
template <int R, int KS, int A_LD, int ACC_LD>
__device__ __forceinline__ void mma_groups_bf16(const __bf16 *__restrict__ a0p, const bf16x8 (&wreg)[KS],
                                                const int32_t *__restrict__ dstp, float *__restrict__ accp) {
  int d[R];
#pragma unroll
  for (int r = 0; r < R; ++r) d[r] = (int)__umul24((unsigned)dstp[r * 16], (unsigned)ACC_LD);  // see mma_groups
  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    bf16x8 a[R];
#pragma unroll
#ifdef ME_ABL_NO_AREAD   /* timing ablation (INVALID results): no LDS operand reads */
    for (int r = 0; r < R; ++r) a[r] = wreg[(s + r) % KS];
#else
    for (int r = 0; r < R; ++r) a[r] = *reinterpret_cast<const bf16x8 *>(a0p + r * 16 * A_LD + s * 32);
#endif
#pragma unroll
#ifdef ME_ABL_NO_MFMA
    for (int r = 0; r < R; ++r) acc[r][0] += (float)a[r][0] + (float)wreg[s][0];
#else
    for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[s], a[r], acc[r], 0, 0, 0);
#endif
  }
#ifdef ME_ABL_NO_ACC     /* timing ablation (INVALID results): accumulators written without the read */
#pragma unroll
  for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4 *>(accp + d[r]) = acc[r];
#else
  f32x4 old[R];
#pragma unroll
  for (int r = 0; r < R; ++r) old[r] = *reinterpret_cast<const f32x4 *>(accp + d[r]);
#pragma unroll
  for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4 *>(accp + d[r]) = old[r] + acc[r];
#endif
}

// n single-group batches of DIFFERENT offsets staged together (batch fusion): group r multiplies with its own
// offset's weights w[r].  All index / operand reads first, then the MFMAs (independent accumulators), then the
// accumulator updates in batch order — a later group may hit a row of an earlier one, and LDS operations of a wave
// execute in order, so issuing read r + 1 behind write r is enough.  Same sums in the same order as n calls of
// mma_groups_bf16<1>: bit-identical, without n dependent LDS round-trip chains.
template <int M, int KS, int A_LD, int ACC_LD>
__device__ __forceinline__ void mma_singles_bf16(const __bf16 *__restrict__ a0p, const bf16x8 (&w)[M][KS], int n,
                                                 const int32_t *__restrict__ dstp, float *__restrict__ accp) {
  int d[M];
  bf16x8 a[M][KS];
  f32x4 acc[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    d[r] = (int)__umul24((unsigned)dstp[r * 16], (unsigned)ACC_LD);
#pragma unroll
    for (int s = 0; s < KS; ++s) a[r][s] = *reinterpret_cast<const bf16x8 *>(a0p + r * 16 * A_LD + s * 32);
    acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int r = 0; r < M; ++r)
      if (r < n) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[r][s], a[r][s], acc[r], 0, 0, 0);   // wave-uniform
  }
#pragma unroll
  for (int r = 0; r < M; ++r) {
    if (r < n) {
      const f32x4 old = *reinterpret_cast<const f32x4 *>(accp + d[r]);
      *reinterpret_cast<f32x4 *>(accp + d[r]) = old + acc[r];
    }
  }
}

// Packed weights: the register image of the MFMA A operand.  For offset k, source-channel chunk c,
// 16-column block cb and 32-channel step v, lane (q = lane >> 4, i16 = lane & 15) finds its eight weights
//   W[k][c*KC + v*32 + q*8 + j][cb*16 + i16],  j = 0..7
// as ONE 16-byte element at ((((k*nchunks + c)*ncb + cb)*(KC/32) + v)*64 + lane), zero beyond the real
// channel counts.  W_F32: the weights are given in fp32 (master weights) and rounded here.
template <int KC, bool W_F32>
__global__ __launch_bounds__(256) void k_pack_weights_bf16(const void *__restrict__ w_, int c_src, int c_dst,
                                                          int transposed, int nchunks, int ncb,
                                                          bf16x8 *__restrict__ wp, int64_t total) {
  constexpr int KS = KC / 32;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e % 64);
  int64_t r = e / 64;
  const int v = (int)(r % KS);
  r /= KS;
  const int cb = (int)(r % ncb);
  r /= ncb;
  const int c = (int)(r % nchunks);
  const int64_t k = r / nchunks;
  const int q = lane >> 4, i16 = lane & 15;
  const int col = cb * 16 + i16;
  bf16x8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = c * KC + v * 32 + q * 8 + j;
    __bf16 val = (__bf16)0.f;
    if (ch < c_src && col < c_dst) {
      // plain: w is [K, c_src, c_dst]; transposed (dgrad): w is the forward kernel [K, c_dst, c_src]
      const int64_t idx = transposed ? (k * c_dst + col) * c_src + ch : (k * c_src + ch) * c_dst + col;
      if (W_F32) val = (__bf16) reinterpret_cast<const float *>(w_)[idx];
      else val = reinterpret_cast<const __bf16 *>(w_)[idx];
    }
    out[j] = val;
  }
  wp[e] = out;
}

// See k_conv_tile_f32 (conv.hip) for the pipeline; differences are the element type and the MFMA shape.
// EXACT: c_src is a multiple of KC (rows need no channel guards).
// SMALL: 32-bit gather offsets with a 24-bit row multiply (host-checked: < 2^24 rows, source matrix < 4 GiB).
// waves per SIMD the register budget is sized for: 3 (168 registers) for four-wave workgroups; an eight-wave
// workgroup (128 columns) is sized for 4 (128 registers: two of them per CU) unless its weight slices need more
// (KC = 256: 64 registers of weights alone)
__host__ __device__ constexpr int conv_bf16_waves_per_simd(int nc, int kc) {
  return (nc == 128 && kc <= 128) ? 4 : (nc == 128 ? 2 : 3);   // (96 columns: six waves, 3 per SIMD = two workgroups per CU)
}

// DEEP (round 3): the gathers and weights of the batch after next are requested while a batch multiplies — two
// register sets, the loop unrolled by two.  For launches that leave ONE workgroup per CU (a 256-channel chunk's 64
// weight registers per lane; a coarse level's 128 - 256 tiles): nothing else hides the L2 latency of the next
// batch's 32 - 64 KB weight slice there, and a batch of one or two groups lasted one memory round trip (2 us on the
// 5k-voxel 256 -> 256 layers for ~0.2 us of matrix work).  Same sums in the same order: bit-identical to the plain loop.
//
// SPLITK (round 4): the coarse MinkUNet levels (256 -> 256 channels on 5k voxels: 128 tiles of 39
// rows) stream the WHOLE packed weight tensor once per tile — 27 x 64 KB per workgroup, 453 MB of L2 -> CU traffic for a
// 3.5 MB tensor, and one L2 round trip per batch of one or two 16-row groups.  With SPLITK the launch has gridDim.z = G
// offset groups: workgroup (tile, slab, g) walks only the batches of offsets [g K / G, (g + 1) K / G) of a G-times
// taller tile (same number of workgroups, 1 / G of the weight bytes and fuller batches per workgroup), and stores its
// fp32 accumulator tile to `partial[g]`; k_conv_splitk_reduce adds the G partial tiles in group order, rounds to bf16
// once and writes the rows (and the batch-norm statistics of the tile).  The sum of a target row is
// ((offsets of group 0) + (group 1)) + ... in fp32: a fixed order — reproducible, not the unsplit kernel's order.
//
// TWOBUF (round 4, with DEEP): two stage buffers and ONE barrier per batch.  Step n: barrier -> request the rows of batch
// n + 2 into the register set batch n just left -> multiply batch n from stage buffer n & 1 -> refill its weights ->
// write batch n + 1 (the OTHER register set, in flight for a whole step) into stage buffer (n + 1) & 1.  The stage write
// of a fast wave overlaps the multiplies of the slow ones instead of sitting between two barriers.  Same sums, same
// order: bit-identical.
template <int NC, int KC, bool EXACT, bool SMALL, bool FUSE = false, bool DEEP = false, bool SPLITK = false,
          bool TWOBUF = false>
__global__ __launch_bounds__(NC * 4, DEEP ? 2 : conv_bf16_waves_per_simd(NC, KC)) void k_conv_tile_bf16(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, __bf16 *__restrict__ dst, int64_t n_tgt, int tile_rows, int batch_groups,
    int fuse, float *__restrict__ stat_mean, float *__restrict__ stat_m2, float *__restrict__ partial, int volume) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  constexpr int WAVES = NC / 16;
  constexpr int NT = WAVES * 64;
  constexpr int A_LD = KC + 16;        // bf16 elements
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int KS = KC / 32;          // MFMA steps per chunk (= 16-byte weight registers per lane)
  constexpr int F8 = KC / 8;           // 16-byte pieces per gathered row
  constexpr int ITER = (ME_MAX_BATCH_GROUPS * 16 * F8 + NT - 1) / NT;
  static_assert(KC % 32 == 0, "KC must be a multiple of 32");
  static_assert(ME_MAX_BATCH_GROUPS == 4, "mma_groups runs cover at most 4 groups");

  const int cap_rows = batch_groups * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!TWOBUF || DEEP, "the second stage buffer belongs to the deep pipeline");
  constexpr int NBUF = TWOBUF ? 2 : 1;
  float *s_acc = reinterpret_cast<float *>(smem);                           // [(tile_rows + 1) x ACC_LD]
  __bf16 *s_a = reinterpret_cast<__bf16 *>(s_acc + (tile_rows + 1) * ACC_LD);  // [NBUF][cap_rows x A_LD]
  int32_t *s_dst = reinterpret_cast<int32_t *>(s_a + NBUF * cap_rows * A_LD);  // [NBUF][cap_rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15;
  const int q = lane >> 4;
  // workgroups take the tiles in the plan's dispatch order (heaviest first: me_plan_build), stored behind the
  // n_tiles + 1 batch pointers; gridDim.x == n_tiles
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];
  const int col_base = blockIdx.y * NC;
  const bool vec_ok = (c_src % 8) == 0;
  const int nchunks = (c_src + KC - 1) / KC;
  const int ncb = (c_dst + 15) / 16;
  const int cb = min(col_base / 16 + wave, ncb - 1);

#ifdef ME_BF16_TIMING
  unsigned long long tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define ME_TICK(slot)                                                  \
  do {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    tm[slot] += now_ - t_prev;                                         \
    t_prev = now_;                                                     \
  } while (0)
#define ME_COUNT(slot) (tm[slot] += 1)
#else
#define ME_TICK(slot) do {} while (0)
#define ME_COUNT(slot) do {} while (0)
#endif
  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the rows this tile's positions stand for (tiles of a position-space map): requested now, parked in the free
  // stage buffer after the main loop (as in k_conv_tile_f32)
  constexpr int ORD = (ME_MAX_TILE_ROWS + NT - 1) / NT;
  int32_t my_ord[ORD];
#pragma unroll
  for (int j = 0; j < ORD; ++j) {
    const int r = j * NT + tid;
    my_ord[j] = (order != nullptr && r < tile_rows && (int64_t)tile * tile_rows + r < n_tgt)
                    ? order[(int64_t)tile * tile_rows + r] : 0;
  }

  int b0 = tile_bptr[tile];
  int nb = tile_bptr[tile + 1] - b0;
  if constexpr (SPLITK) {
    // the tile's batches are listed by ascending offset (k_plan_fill): this workgroup's offsets are a sub-range
    const int k_lo = (int)(((int64_t)blockIdx.z * volume) / gridDim.z);
    const int k_hi = (int)(((int64_t)(blockIdx.z + 1) * volume) / gridDim.z);
    int32_t *s_rng = reinterpret_cast<int32_t *>(smem + conv_bf16_lds_bytes(NC, KC, tile_rows, batch_groups, TWOBUF));
    if (tid == 0) {
      s_rng[0] = nb;
      s_rng[1] = nb;
    }
    __syncthreads();
    for (int b = tid; b < nb; b += NT) {
      const int k = (int)((uint32_t)batch_desc[2 * (int64_t)(b0 + b) + 1] >> 8);
      const int kp = b > 0 ? (int)((uint32_t)batch_desc[2 * (int64_t)(b0 + b - 1) + 1] >> 8) : -1;
      if (k >= k_lo && kp < k_lo) s_rng[0] = b;
      if (k >= k_hi && kp < k_hi) s_rng[1] = b;
    }
    __syncthreads();
    const int r_lo = s_rng[0], r_hi = s_rng[1];
    b0 += r_lo;
    nb = r_hi - r_lo;
  }

  // Batch fusion (round 2).  On sparse maps most (tile, offset) items hold ONE 16-row group — MinkUNet's stride-1
  // level: 7 - 11 pairs per item; config 5: 11 — and a batch of one group pays the two barriers, the stage write,
  // the gather latency and a chain of dependent LDS round trips for 16 rows.  Consecutive batches of a tile are
  // contiguous in the plan, so up to MAXSUB single-group batches are staged together ("super-batch"), each group
  // multiplied with its own offset's weights, and the accumulator tile updated in batch order (mma_singles_bf16):
  // results are bit-identical.  MAXSUB is 4 while the weights of four offsets fit the register budget
  // (KC <= 64), 2 beyond.  FUSE is a template parameter: the descriptor look-ahead, the extra weight registers and
  // the sub-batch loop cost a dense layer (whose batches are full) 20 - 40 %, so the host picks the instantiation by
  // the map's density (me_conv_target_bf16 / me_conv_target_bf16_fused); fuse = 0 at run time: tests.
  constexpr int MAXSUB = !FUSE ? 1 : (KS <= 2 ? 4 : 2);
  struct Super {
    int chunk, g0, ng, nsub;     // source-channel chunk, first group, groups in total, sub-batches (0: none)
    int k[MAXSUB], sg[MAXSUB];   // offset and groups of each sub-batch
  };
  int cur_chunk = 0, cur_r = 0;  // the next batch to hand out
  auto next_super = [&]() {
    Super sb;
    const bool valid = cur_chunk < nchunks && nb > 0;
    const int r = valid ? cur_r : max(nb - 1, 0);
    sb.chunk = valid ? cur_chunk : max(nchunks - 1, 0);
    const int avail = valid ? nb - cur_r : 1;
    i32x2 d[MAXSUB];
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j)   // (descriptors behind the tile's last batch are readable: me_plan_max_groups)
      d[j] = *reinterpret_cast<const i32x2 *>(batch_desc + 2 * (int64_t)(b0 + r + j));
    sb.g0 = d[0].x;
    sb.ng = 0;
    sb.nsub = 0;
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j) {
      const int g = d[j].y & 255;
      // (only single-group batches are fused: group r <-> sub-batch r, see mma_singles_bf16)
      const bool take = j == 0 || (fuse && sb.nsub == j && j < avail && sb.ng == j && g == 1);
      sb.k[j] = take ? (int)((uint32_t)d[j].y >> 8) : sb.k[j > 0 ? j - 1 : 0];
      sb.sg[j] = take ? g : 0;
      if (take) {
        sb.ng += g;
        sb.nsub = j + 1;
      }
    }
    if (valid) {
      cur_r += sb.nsub;
      if (cur_r >= nb) {
        cur_r = 0;
        ++cur_chunk;
      }
    } else {
      sb.nsub = 0;
    }
    return sb;
  };

  bf16x8 stage[ITER];
  int32_t dstv = tile_rows;
  int32_t sidx[ITER];
  bf16x8 wreg[MAXSUB][KS], wnxt[MAXSUB][KS];

  // As in k_conv_tile_f32: the 64-entry index window of a batch is read to its end unconditionally (the plan is
  // followed by 64 valid entries), and padding slots (index -1) gather row 0 without being zeroed — their
  // products land in the dummy accumulator row.
  auto load_sidx = [&](int g0) {
    const char *pb = reinterpret_cast<const char *>(plan_src + (int64_t)g0 * 16);
#pragma unroll
    for (int j = 0; j < ITER; ++j)
      sidx[j] = *reinterpret_cast<const int32_t *>(
          pb + (unsigned)(min((j * NT + tid) / F8, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));   // (eight-wave workgroups
                                                                     // have more threads than a 64-row window has pieces)
  };
  const char *srcb = reinterpret_cast<const char *>(src);
  const unsigned row_bytes = (unsigned)c_src * 2u;
  // (rows_used: the 16 * groups rows the batch really holds.  The index window is read to its end — 64 entries, the
  // next batches' rows behind the batch's own — and round 3 gathered them all: on the coarse levels, whose batches
  // hold one or two groups, half of the gathered bytes were rows nobody multiplies, through a vector-memory path that
  // IS the bound there (round 4: ~30 B / clk / CU of weights + rows per batch).  Slots beyond the batch gather row 0.)
  auto gather = [&](int chunk, int g0, bf16x8 (&stage)[ITER], int32_t &dstv, int rows_used = ME_MAX_BATCH_GROUPS * 16) {
    const int c0 = chunk * KC;
    dstv = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)g0 * 16) +
                                             (unsigned)(min(tid, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int ch = c0 + ((j * NT + tid) % F8) * 8;
#ifdef ME_ABL_NO_GATHER
      const int sr = max(sidx[j], 0) & 63;
#else
      const int sr = (j * NT + tid) / F8 < rows_used ? max(sidx[j], 0) : 0;
#endif
      if (SMALL && (EXACT || vec_ok)) {
        const unsigned off = __umul24((unsigned)sr, row_bytes) + (unsigned)(EXACT ? ch : min(ch, c_src - 8)) * 2u;
        stage[j] = *reinterpret_cast<const bf16x8 *>(srcb + off);
        continue;
      }
      const __bf16 *rowp = src + (int64_t)sr * c_src;
      if (EXACT) {
        stage[j] = *reinterpret_cast<const bf16x8 *>(rowp + ch);
      } else if (vec_ok) {
        // a piece is either wholly inside the row or wholly beyond it (zeroed at the stage write)
        stage[j] = *reinterpret_cast<const bf16x8 *>(rowp + min(ch, c_src - 8));
      } else {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = rowp[min(ch + e, c_src - 1)];
        stage[j] = t;
      }
    }
  };
  auto write_stage = [&](int chunk, const bf16x8 (&stage)[ITER], int32_t dstv, int bufi = 0) {
    __bf16 *sa_b = s_a + bufi * cap_rows * A_LD;       // (bufi != 0 only with TWOBUF)
    int32_t *sd_b = s_dst + bufi * cap_rows;
    const int c0 = chunk * KC;
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int idx = j * NT + tid;
      const int r = idx / F8;
      const int ch = c0 + (idx % F8) * 8;
      bf16x8 t = stage[j];
      if (!EXACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (ch + e >= c_src) t[e] = (__bf16)0.f;
      }
#ifdef ME_ABL_NO_STAGE
      if (r < cap_rows && t[0] == (__bf16)1234.5f) *reinterpret_cast<bf16x8 *>(&sa_b[r * A_LD + (idx % F8) * 8]) = t;
#else
      if (r < cap_rows) *reinterpret_cast<bf16x8 *>(&sa_b[r * A_LD + (idx % F8) * 8]) = t;
#endif
    }
    if (tid < cap_rows) sd_b[tid] = dstv;
  };
  // weights of every sub-batch of a super-batch (wave-uniform branches: a dense batch loads one slice)
  auto load_w = [&](const Super &sb, bf16x8 (&wnxt)[MAXSUB][KS]) {
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j) {
      if (j == 0 || j < sb.nsub) {
#ifdef ME_ABL_NO_WLOAD
        const bf16x8 *p = wp + ((((int64_t)0 * nchunks + 0) * ncb + cb) * KS) * 64 + lane;
#else
        const bf16x8 *p = wp + ((((int64_t)sb.k[j] * nchunks + sb.chunk) * ncb + cb) * KS) * 64 + lane;
#endif
#pragma unroll
        for (int v = 0; v < KS; ++v) wnxt[j][v] = p[v * 64];
      }
    }
  };

  auto multiply = [&](const Super &sb, const bf16x8 (&wreg)[MAXSUB][KS], int bufi = 0) {
    const __bf16 *a0p = &s_a[bufi * cap_rows * A_LD + i16 * A_LD + q * 8];
    const int32_t *dstp = &s_dst[bufi * cap_rows + i16];
    float *accp = &s_acc[wave * 16 + q * 4];
    if (MAXSUB > 1 && sb.nsub > 1) {   // wave-uniform
      mma_singles_bf16<MAXSUB, KS, A_LD, ACC_LD>(a0p, wreg, sb.nsub, dstp, accp);
    } else {
      const int g = sb.sg[0];
      if (g == 4) {
        mma_groups_bf16<4, KS, A_LD, ACC_LD>(a0p, wreg[0], dstp, accp);
      } else if (g == 3) {
        mma_groups_bf16<3, KS, A_LD, ACC_LD>(a0p, wreg[0], dstp, accp);
      } else if (g == 2) {
        mma_groups_bf16<2, KS, A_LD, ACC_LD>(a0p, wreg[0], dstp, accp);
      } else {
        mma_groups_bf16<1, KS, A_LD, ACC_LD>(a0p, wreg[0], dstp, accp);
      }
    }
  };

  Super sA = next_super();
  if constexpr (DEEP) {
    static_assert(EXACT && SMALL, "the deep pipeline: whole chunks, 32-bit offsets");
    if (sA.nsub > 0) {
      bf16x8 stage1[ITER], wnxt1[MAXSUB][KS];
      int32_t dstv1 = tile_rows;
      Super sB = next_super();
      Super sC = next_super();
      Super sD = next_super();
      load_sidx(sA.g0);
      load_w(sA, wnxt);
      gather(sA.chunk, sA.g0, stage, dstv, sA.ng * 16);
      __builtin_amdgcn_sched_barrier(0);   // set 0 strictly older than set 1: the loop's counted waits merge with this path
      load_sidx(sB.g0);
      load_w(sB, wnxt1);
      gather(sB.chunk, sB.g0, stage1, dstv1, sB.ng * 16);
      load_sidx(sC.g0);
      // one step: stage and multiply batch A from register set (st, dv, wn); refill the set with batch C.  The
      // index window of batch D is requested BEFORE the refill: the gather of the next step then waits for loads
      // older than this refill (vmcnt counts in order), not for the refill itself
      ME_TICK(6);
      auto step = [&](bf16x8 (&st)[ITER], int32_t &dv, bf16x8 (&wn)[MAXSUB][KS]) {
        __syncthreads();
        ME_TICK(0);
        write_stage(sA.chunk, st, dv);
        ME_TICK(1);
        __syncthreads();
        ME_TICK(2);
        int32_t sidx_c[ITER];
#pragma unroll
        for (int j = 0; j < ITER; ++j) sidx_c[j] = sidx[j];
        load_sidx(sD.g0);
        {   // gather(sC) from the indices loaded one step ago
          const int c0 = sC.chunk * KC;
          dv = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)sC.g0 * 16) +
                                                  (unsigned)(min(tid, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));
#pragma unroll
          for (int j = 0; j < ITER; ++j) {
            const int ch = c0 + ((j * NT + tid) % F8) * 8;
#ifdef ME_ABL_NO_GATHER
            const unsigned off = __umul24((unsigned)(max(sidx_c[j], 0) & 63), row_bytes) + (unsigned)ch * 2u;
#else
            const int sr = (j * NT + tid) / F8 < sC.ng * 16 ? max(sidx_c[j], 0) : 0;   // (see gather)
            const unsigned off = __umul24((unsigned)sr, row_bytes) + (unsigned)ch * 2u;
#endif
            st[j] = *reinterpret_cast<const bf16x8 *>(srcb + off);
          }
        }
        // the set's weights are multiplied in place and replaced behind the MFMAs (a copy to a third set would
        // make the refill wait at the loop's back edge)
        ME_TICK(3);
        multiply(sA, wn);
        ME_TICK(4);
        load_w(sC, wn);
        sA = sB;
        sB = sC;
        sC = sD;
        sD = next_super();
        ME_TICK(5);
        ME_COUNT(8);
      };
      if constexpr (TWOBUF) {
        write_stage(sA.chunk, stage, dstv, 0);     // batch A into stage buffer 0 (waits for its rows)
        // one step: (st_p, dv_p, wn_p) is the register set of batch A — already staged in buffer bp —, (st_q, dv_q) holds
        // the rows of batch B, in flight since the previous step
        auto step2 = [&](bf16x8 (&st_p)[ITER], int32_t &dv_p, bf16x8 (&wn_p)[MAXSUB][KS], const bf16x8 (&st_q)[ITER],
                         const int32_t &dv_q, int bp) {
          __syncthreads();
          int32_t sidx_c[ITER];
#pragma unroll
          for (int j = 0; j < ITER; ++j) sidx_c[j] = sidx[j];
          load_sidx(sD.g0);
          {   // rows of batch C into the set batch A has left
            const int c0 = sC.chunk * KC;
            dv_p = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)sC.g0 * 16) +
                                                      (unsigned)(min(tid, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));
#pragma unroll
            for (int j = 0; j < ITER; ++j) {
              const int ch = c0 + ((j * NT + tid) % F8) * 8;
              const int sr = (j * NT + tid) / F8 < sC.ng * 16 ? max(sidx_c[j], 0) : 0;   // (see gather)
              const unsigned off = __umul24((unsigned)sr, row_bytes) + (unsigned)ch * 2u;
              st_p[j] = *reinterpret_cast<const bf16x8 *>(srcb + off);
            }
          }
          multiply(sA, wn_p, bp);
          load_w(sC, wn_p);
          if (sB.nsub > 0) write_stage(sB.chunk, st_q, dv_q, bp ^ 1);   // (wave-uniform)
          sA = sB;
          sB = sC;
          sC = sD;
          sD = next_super();
        };
        while (true) {
          step2(stage, dstv, wnxt, stage1, dstv1, 0);
          if (sA.nsub == 0) break;
          step2(stage1, dstv1, wnxt1, stage, dstv, 1);
          if (sA.nsub == 0) break;
        }
      } else {
      while (true) {
        step(stage, dstv, wnxt);
        if (sA.nsub == 0) break;
        step(stage1, dstv1, wnxt1);
        if (sA.nsub == 0) break;
      }
      }
    }
  } else if (sA.nsub > 0) {
    Super sB = next_super();
    Super sC = next_super();
    load_w(sA, wnxt);
    load_sidx(sA.g0);
    gather(sA.chunk, sA.g0, stage, dstv, sA.ng * 16);
    load_sidx(sB.g0);

    ME_TICK(6);
    while (sA.nsub > 0) {
      __syncthreads();
      ME_TICK(0);
      write_stage(sA.chunk, stage, dstv);
#pragma unroll
      for (int j = 0; j < MAXSUB; ++j) {
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) wreg[j][sx] = wnxt[j][sx];
      }
      ME_TICK(1);
      __syncthreads();
      ME_TICK(2);
      load_w(sB, wnxt);
      gather(sB.chunk, sB.g0, stage, dstv, sB.ng * 16);
      load_sidx(sC.g0);
      ME_TICK(3);
      multiply(sA, wreg);
      ME_TICK(4);
      sA = sB;
      sB = sC;
      sC = next_super();
      ME_TICK(5);
      ME_COUNT(8);
    }
  }
  __syncthreads();

  if constexpr (SPLITK) {
    // the fp32 accumulator tile of this offset group, rows in tile (position) order; k_conv_splitk_reduce finishes
    const int64_t rows_all = (int64_t)gridDim.x * tile_rows;
    float *pt = partial + ((int64_t)blockIdx.z * rows_all + (int64_t)tile * tile_rows) * c_dst + col_base;
    for (int x = tid; x < tile_rows * NC / 4; x += NT) {
      const int row = x / (NC / 4);
      const int c4 = x % (NC / 4);
      if (col_base + c4 * 4 < c_dst)
        *reinterpret_cast<f32x4 *>(pt + (int64_t)row * c_dst + c4 * 4) =
            *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
    }
    return;
  }
  // every target row of the tile is written exactly once, rounded to bf16 (RNE)
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const bool vec_out = (c_dst % 4) == 0;
  int32_t *s_ord = reinterpret_cast<int32_t *>(s_a);   // (the stage buffer is free now: >= 16 rows x 32 channels x 2 B)
  if (order != nullptr) {
#pragma unroll
    for (int j = 0; j < ORD; ++j)
      if (j * NT + tid < tile_rows) s_ord[j * NT + tid] = my_ord[j];
    __syncthreads();
  }
  // Batch-norm statistics of the tile (round 3): when the host passes stat_mean / stat_m2, the
  // store loop also accumulates, per output column, the sum and the sum of squares of the ROUNDED values it stores
  // (shifted by the tile's first row, as k_bn_partial shifts by its chunk's first row), and the tile's (mean, M2) go
  // to slot `tile` of the partials: the batch norm that follows merges the tiles (k_bn_final_tiles) instead of
  // reading the matrix again.  Fixed order (thread's rows ascending, xor-shuffle tree, waves ascending): reproducible.
  constexpr int G4 = NC / 4;                       // threads per tile row = four-column groups
  constexpr bool kStats = NC == 32 || NC == 64 || NC == 128;   // (G4 a power of two: the shuffle tree)
  const bool do_stats = kStats && stat_mean != nullptr;        // uniform
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (do_stats) {
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&s_acc[(tid % G4) * 4]);   // row 0 of the tile
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[t] = (float)(__bf16)v0[t];
  }
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / (NC / 4);
    const int c4 = x % (NC / 4);
    const int cc = col_base + c4 * 4;
    if (row < rows_here && (cc < c_dst || do_stats)) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
      const bf16x4 vb = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      if (do_stats) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float d = (float)vb[t] - sh[t];
          st1[t] += d;
          st2[t] = fmaf(d, d, st2[t]);
        }
      }
      if (cc < c_dst) {
        const int64_t grow = order ? (int64_t)s_ord[row] : row0 + row;
        __bf16 *o = dst + grow * c_dst + cc;
        if (vec_out) {
          *reinterpret_cast<bf16x4 *>(o) = vb;
        } else {
          o[0] = vb[0];
          if (cc + 1 < c_dst) o[1] = vb[1];
          if (cc + 2 < c_dst) o[2] = vb[2];
          if (cc + 3 < c_dst) o[3] = vb[3];
        }
      }
    }
  }
  if constexpr (kStats) {
    if (do_stats) {
      // lanes l, l + G4, l + 2 G4, ... of a wave hold the same four columns
#pragma unroll
      for (int off = G4; off < 64; off <<= 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          st1[t] += __shfl_xor(st1[t], off, 64);
          st2[t] += __shfl_xor(st2[t], off, 64);
        }
      }
      __syncthreads();                              // every wave is done with the accumulator tile
      float *s_st = s_acc + ACC_LD;                 // [WAVES][G4][8] behind row 0 (NC * NC / 8 floats <= 16 rows)
      if (lane < G4) {
        float *w = s_st + (wave * G4 + lane) * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          w[t] = st1[t];
          w[4 + t] = st2[t];
        }
      }
      __syncthreads();
      if (tid < NC && col_base + tid < c_dst) {
        const int c4 = tid >> 2, t = tid & 3;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
          a += s_st[(w * G4 + c4) * 8 + t];
          b += s_st[(w * G4 + c4) * 8 + 4 + t];
        }
        const float shift = (float)(__bf16)s_acc[tid];
        const float cnt = (float)rows_here, m = a / cnt;
        stat_mean[(int64_t)tile * c_dst + col_base + tid] = shift + m;
        stat_m2[(int64_t)tile * c_dst + col_base + tid] = fmaxf(b - a * m, 0.f);
      }
    }
  }
#ifdef ME_BF16_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  ME_TICK(7);
  ME_COUNT(9);
  if (tid == 0) {
#pragma unroll
    for (int sl = 0; sl < 10; ++sl) atomicAdd(&d_bf16_timing[(DEEP ? 0 : 10) + sl], tm[sl]);
  }
#endif
#undef ME_TICK
#undef ME_COUNT
}

#ifdef ME_DEBUG_VARIANTS   // measured 1.4 - 2.6x SLOWER than k_conv_tile_bf16 (profiles/r04_offsync_schedule_sweep.log): tuning build only
// =================================================================================================
// k_conv_off_bf16 (round 4): the same plan, the same sums — another schedule ("offset-synchronous")
// =================================================================================================
// k_conv_tile_bf16 splits a batch by COLUMNS: every wave multiplies all staged rows of the batch by its own 16 columns,
// so a batch is barrier -> stage write -> barrier -> eight short multiplies, and a tile is a chain of 27+ such batches
// (~3,200 cycles each whatever their size: r03 phase counters) with the matrix pipe busy 14 % of the time.  Here the
// waves of a workgroup split an item — the groups of ONE offset of the tile — by ROWS first:
//   * the offset's weight slice (KC x NC, the packed MFMA image, one contiguous block) is copied to LDS once per item
//     and workgroup, double-buffered, its loads requested one item ahead;
//   * wave (gw, cw) takes the groups g0 + gw, g0 + gw + GW, ... of the item and the column blocks [cw CBW, (cw + 1) CBW):
//     it gathers its 16 rows STRAIGHT into the MFMA B operand (lane (row, q) loads its 8 channels per k-step: 16 bytes,
//     four lanes cover a row's 64-byte piece) — no stage buffer, no stage write, no barrier between gather and
//     multiply — reads the A operand (weights) from LDS, runs its KS x CBW MFMAs and adds the 16 x (16 CBW) block into
//     the LDS accumulator tile;
//   * ONE barrier per non-empty item: it publishes the weight slice and separates the accumulator updates of
//     consecutive offsets (rows of one offset are distinct, so waves never collide inside an item).
// Rows of the next group (and their indices one group further) are in flight while a group multiplies; waves drift
// apart inside an item, so one wave's gather wait is another's multiply.
// Same additions in the same order as k_conv_tile_bf16 (chunk-major, offsets ascending, one zero-initialised MFMA chain
// over the k-steps per group, then one add into the tile): results are BIT-IDENTICAL to it
// (tests/test_gpu_bf16.py::test_offset_synchronous_kernel_is_bit_identical).
// Requirements (else the column-split kernel runs): c_src % KC == 0, c_dst % NC == 0, 32-bit gather offsets.
__host__ __device__ constexpr int conv_off_lds_bytes(int nc, int kc, int tile_rows, int volume) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + 2 * kc * nc * 2 + (volume + 2) * 4;
}

template <int NC, int KC, int GW, int CBW>
__global__ __launch_bounds__(GW * (NC / 16 / CBW) * 64, 4) void k_conv_off_bf16(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, __bf16 *__restrict__ dst, int64_t n_tgt, int tile_rows, int volume,
    float *__restrict__ stat_mean, float *__restrict__ stat_m2) {
  constexpr int NCB = NC / 16;            // column blocks of the slab
  constexpr int CW = NCB / CBW;           // column parts
  constexpr int WAVES = GW * CW;
  constexpr int NT = WAVES * 64;
  constexpr int KS = KC / 32;
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int WELEMS = KS * NCB * 64;   // 16-byte elements of one weight slice
  constexpr int WPT = (WELEMS + NT - 1) / NT;
  static_assert(NCB % CBW == 0 && KC % 32 == 0, "shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);                                        // [(tile_rows + 1) x ACC_LD]
  bf16x8 *s_w = reinterpret_cast<bf16x8 *>(s_acc + (tile_rows + 1) * ACC_LD);            // [2][WELEMS]
  int32_t *s_g = reinterpret_cast<int32_t *>(s_w + 2 * WELEMS);                          // [volume + 1] first group of item k

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gw = wave / CW, cw = wave % CW;
  const int i16 = lane & 15, q = lane >> 4;
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];
  const int col_base = blockIdx.y * NC;
  const int nchunks = c_src / KC;
  const int ncb = (c_dst + 15) / 16;

  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  // first group of every item of the tile, from the tile's batch descriptors (batches are listed by ascending offset,
  // an item's groups are contiguous): s_g[k] .. s_g[k + 1]
  const int b0 = tile_bptr[tile], nb = tile_bptr[tile + 1] - b0;
  for (int k = tid; k <= volume; k += NT) s_g[k] = -1;
  __syncthreads();
  for (int b = tid; b < nb; b += NT) {
    const int2 d = *reinterpret_cast<const int2 *>(batch_desc + 2 * (int64_t)(b0 + b));
    const int k = (int)((uint32_t)d.y >> 8);
    const int kp = b > 0 ? (int)((uint32_t)batch_desc[2 * (int64_t)(b0 + b - 1) + 1] >> 8) : -1;
    if (k != kp) s_g[k] = d.x;
    if (b == nb - 1) s_g[volume] = d.x + (d.y & 255);
  }
  __syncthreads();
  if (tid == 0) {
    if (nb == 0) s_g[volume] = 0;
    for (int k = volume - 1; k >= 0; --k)
      if (s_g[k] < 0) s_g[k] = s_g[k + 1];
  }
  __syncthreads();

  const char *srcb = reinterpret_cast<const char *>(src);
  const unsigned row_bytes = (unsigned)c_src * 2u;
  const int32_t *dsts = plan_dst + i16;

  // ---- this wave's stream of groups: (offset, group) in plan order, its row indices two groups ahead, its rows one ----
  struct Cursor {
    int k, g;         // offset, group (k == volume: the stream has ended)
  };
  auto sg = [&](int k) { return __builtin_amdgcn_readfirstlane(s_g[k]); };   // (wave-uniform: keep it scalar)
  auto first_from = [&](int k) {   // first group of this wave at or after offset k
    Cursor c;
    c.k = k;
    c.g = 0;
    while (c.k < volume) {
      const int g = sg(c.k) + gw;
      if (g < sg(c.k + 1)) {
        c.g = g;
        break;
      }
      ++c.k;
    }
    return c;
  };
  auto advance = [&](Cursor c) {
    if (c.k >= volume) return c;
    c.g += GW;
    if (c.g < sg(c.k + 1)) return c;
    return first_from(c.k + 1);
  };
  auto load_index = [&](const Cursor &c, int32_t &sidx, int32_t &didx) {
    const int g = c.k < volume ? c.g : 0;                 // (a finished stream keeps loading group 0 of the plan: valid memory)
    sidx = plan_src[(int64_t)g * 16 + i16];
    didx = dsts[(int64_t)g * 16];
  };
  auto load_rows = [&](int32_t sidx, int chunk, bf16x8 (&b)[KS]) {
    const unsigned off = __umul24((unsigned)max(sidx, 0), row_bytes) + (unsigned)(chunk * KC + q * 8) * 2u;
#pragma unroll
    for (int v = 0; v < KS; ++v) b[v] = *reinterpret_cast<const bf16x8 *>(srcb + off + v * 64);
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // weights of the first non-empty item of this pass
    int kw = 0;
    while (kw < volume && sg(kw) == sg(kw + 1)) ++kw;
    bf16x8 wq[WPT];
    auto load_w = [&](int k) {
      const bf16x8 *p = wp + ((((int64_t)min(k, volume - 1) * nchunks + chunk) * ncb + col_base / 16) * KS) * 64;
#pragma unroll
      for (int j = 0; j < WPT; ++j) wq[j] = p[min(j * NT + tid, WELEMS - 1)];
    };
    load_w(kw);
    Cursor cur = first_from(0), nxt = advance(cur), nn = advance(nxt);
    int32_t s_cur, d_cur, s_nxt, d_nxt, s_nn, d_nn;
    bf16x8 b_cur[KS], b_nxt[KS];
    load_index(cur, s_cur, d_cur);
    load_index(nxt, s_nxt, d_nxt);
    load_index(nn, s_nn, d_nn);
    load_rows(s_cur, chunk, b_cur);
    load_rows(s_nxt, chunk, b_nxt);
    int buf = 0;
    for (int k = kw; k < volume;) {
      // publish this item's weights (requested one item ago); the barrier also separates the accumulator updates of
      // consecutive offsets
      bf16x8 *wdst = s_w + buf * WELEMS;
#pragma unroll
      for (int j = 0; j < WPT; ++j)
        if (j * NT + tid < WELEMS) wdst[j * NT + tid] = wq[j];
      int kn = k + 1;
      while (kn < volume && sg(kn) == sg(kn + 1)) ++kn;
      __syncthreads();
      load_w(kn);                                           // next non-empty item (clamped past the end: unused)
      const bf16x8 *wa = s_w + buf * WELEMS + (cw * CBW * KS) * 64 + lane;
      while (cur.k == k) {
        f32x4 acc[CBW];
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < KS; ++v) {
#pragma unroll
          for (int c = 0; c < CBW; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[(c * KS + v) * 64], b_cur[v], acc[c], 0, 0, 0);
        }
        float *ap = s_acc + (int)__umul24((unsigned)d_cur, (unsigned)ACC_LD) + cw * CBW * 16 + q * 4;
        f32x4 old[CBW];
#pragma unroll
        for (int c = 0; c < CBW; ++c) old[c] = *reinterpret_cast<const f32x4 *>(ap + c * 16);
#pragma unroll
        for (int c = 0; c < CBW; ++c) *reinterpret_cast<f32x4 *>(ap + c * 16) = old[c] + acc[c];
        // shift the stream: next -> current (its rows have been in flight for a whole group), request the rows of the
        // group after it and the indices one further
        cur = nxt;
        nxt = nn;
        nn = advance(nn);
        s_cur = s_nxt;
        d_cur = d_nxt;
#pragma unroll
        for (int v = 0; v < KS; ++v) b_cur[v] = b_nxt[v];
        s_nxt = s_nn;
        d_nxt = d_nn;
        load_rows(s_nxt, chunk, b_nxt);
        load_index(nn, s_nn, d_nn);
      }
      buf ^= 1;
      k = kn;
    }
    __syncthreads();     // (the next pass rewrites the weight buffers; the last one is followed by the epilogue)
  }

  // ---- epilogue: as k_conv_tile_bf16 (every target row written once, rounded to bf16; optional tile statistics) ----
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  constexpr int G4 = NC / 4;
  constexpr bool kStats = NC == 32 || NC == 64 || NC == 128;
  const bool do_stats = kStats && stat_mean != nullptr;
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (do_stats) {
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&s_acc[(tid % G4) * 4]);
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[t] = (float)(__bf16)v0[t];
  }
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / G4;
    const int c4 = x % G4;
    const int cc = col_base + c4 * 4;
    if (row < rows_here) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
      const bf16x4 vb = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      if (do_stats) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float d = (float)vb[t] - sh[t];
          st1[t] += d;
          st2[t] = fmaf(d, d, st2[t]);
        }
      }
      const int64_t grow = order ? (int64_t)order[row0 + row] : row0 + row;
      *reinterpret_cast<bf16x4 *>(dst + grow * c_dst + cc) = vb;
    }
  }
  if constexpr (kStats) {
    if (do_stats) {
      static_assert(!kStats || NT % G4 == 0, "a thread keeps its four columns over the store loop");
#pragma unroll
      for (int off = G4; off < 64; off <<= 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          st1[t] += __shfl_xor(st1[t], off, 64);
          st2[t] += __shfl_xor(st2[t], off, 64);
        }
      }
      __syncthreads();
      float *s_st = s_acc + ACC_LD;                 // [WAVES][G4][8] behind row 0
      if (lane < G4) {
        float *w = s_st + (wave * G4 + lane) * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          w[t] = st1[t];
          w[4 + t] = st2[t];
        }
      }
      __syncthreads();
      if (tid < NC && col_base + tid < c_dst) {
        const int c4 = tid >> 2, t = tid & 3;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
          a += s_st[(w * G4 + c4) * 8 + t];
          b += s_st[(w * G4 + c4) * 8 + 4 + t];
        }
        const float shift = (float)(__bf16)s_acc[tid];
        const float cnt = (float)rows_here, m = a / cnt;
        stat_mean[(int64_t)tile * c_dst + col_base + tid] = shift + m;
        stat_m2[(int64_t)tile * c_dst + col_base + tid] = fmaxf(b - a * m, 0.f);
      }
    }
  }
}

#endif  // ME_DEBUG_VARIANTS (k_conv_off_bf16)

// Second phase of a SPLITK launch: out[row] = bf16(sum over offset groups g of partial[g][row]) — fp32 adds in group
// order, one rounding — for one (tile, column slab) per workgroup, and the tile's batch-norm statistics exactly as the
// unsplit kernel's store loop forms them (shifted by the tile's first row; thread's rows ascending, xor-shuffle tree,
// waves ascending).  HBM / L2 streaming: G x tile x NC x 4 bytes in, tile x NC x 2 out.
template <int NC, int G>
__global__ __launch_bounds__(NC * 4) void k_conv_splitk_reduce(const float *__restrict__ partial, int c_dst,
                                                              const int32_t *__restrict__ order,
                                                              __bf16 *__restrict__ dst, int64_t n_tgt, int tile_rows,
                                                              float *__restrict__ stat_mean,
                                                              float *__restrict__ stat_m2) {
  constexpr int WAVES = NC / 16, NT = WAVES * 64, G4 = NC / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_st = reinterpret_cast<float *>(smem);                 // [WAVES][G4][8]
  float *s_shift = s_st + WAVES * G4 * 8;                        // [NC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int col_base = blockIdx.y * NC;
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const int64_t gstride = (int64_t)gridDim.x * tile_rows * c_dst;   // one offset group's partial matrix
  const bool do_stats = stat_mean != nullptr;                       // uniform
  auto row_sum = [&](int row, int cc) {
    const float *p = partial + (row0 + row) * c_dst + cc;
    f32x4 v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = *reinterpret_cast<const f32x4 *>(p + g * gstride);
    f32x4 a = v[0];
#pragma unroll
    for (int g = 1; g < G; ++g) a += v[g];
    return a;
  };
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  const int my_c4 = tid % G4;
  const bool col_ok = col_base + my_c4 * 4 < c_dst;
  if (do_stats && col_ok) {
    const f32x4 v0 = row_sum(0, col_base + my_c4 * 4);             // row 0 of the tile
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[t] = (float)(__bf16)v0[t];
    if (tid < G4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) s_shift[tid * 4 + t] = sh[t];
    }
  }
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / G4;            // (x % G4 == tid % G4: NT is a multiple of G4)
    const int cc = col_base + my_c4 * 4;
    if (row < rows_here && col_ok) {
      const f32x4 v = row_sum(row, cc);
      const bf16x4 vb = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      if (do_stats) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float d = (float)vb[t] - sh[t];
          st1[t] += d;
          st2[t] = fmaf(d, d, st2[t]);
        }
      }
      const int64_t grow = order ? (int64_t)order[row0 + row] : row0 + row;
      *reinterpret_cast<bf16x4 *>(dst + grow * c_dst + cc) = vb;
    }
  }
  if (do_stats) {
#pragma unroll
    for (int off = G4; off < 64; off <<= 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st1[t] += __shfl_xor(st1[t], off, 64);
        st2[t] += __shfl_xor(st2[t], off, 64);
      }
    }
    if (lane < G4) {
      float *w = s_st + (wave * G4 + lane) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        w[t] = st1[t];
        w[4 + t] = st2[t];
      }
    }
    __syncthreads();
    if (tid < NC && col_base + tid < c_dst) {
      const int c4 = tid >> 2, t = tid & 3;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        a += s_st[(w * G4 + c4) * 8 + t];
        b += s_st[(w * G4 + c4) * 8 + 4 + t];
      }
      const float shift = s_shift[tid];
      const float cnt = (float)rows_here, m = a / cnt;
      stat_mean[(int64_t)tile * c_dst + col_base + tid] = shift + m;
      stat_m2[(int64_t)tile * c_dst + col_base + tid] = fmaxf(b - a * m, 0.f);
    }
  }
}

#ifdef ME_DEBUG_VARIANTS   // slower than the plan kernel on every MinkUNet layer but one (docs/HISTORY.md 9.6): tuning build only
// =================================================================================================
// output-stationary ("gather") convolution on bf16 features: k_conv_gather_bf16 (round 2)
// =================================================================================================
// In bf16 one MFMA (16 rows x 16 columns x 32 channels) costs ~17 cycles against 32 x 16 for the same block in
// fp32: the plan kernel above is bound by everything EXCEPT the matrix pipe (LDS accumulator read-add-write,
// staging, two barriers per batch: 8 % of the bf16 peak on config 2).  With the matrix pipe this cheap the
// classical trade flips: keep the TARGET rows' accumulators in registers for the whole tile and multiply EVERY
// kernel offset against them, absent neighbours as zero rows —
//   * no tile plan, no LDS accumulator, no scatter: a wave owns 32 target rows (two MFMA row blocks) x up to 128
//     columns (64 fp32 accumulator registers), walks the K offsets in order and stores its rows once (rounded to
//     bf16 once): the sum of a target row is the sum over k = 0 .. K-1 in fp32, a fixed order;
//   * the B operand (gathered rows) needs no LDS either: lane (row i16, channel group q) loads its 8 consecutive
//     channels of row nbr[k][target i16] straight into the MFMA register (16-byte loads, exec-masked where the
//     neighbour is absent); a row block whose 16 neighbours are all absent skips its MFMAs (wave-uniform);
//   * the A operand (weights) is the same for every workgroup: the packed register image of (offset, 64-channel
//     chunk, column slab) — 16 KiB for 64 x 128 — streams global -> LDS by LDS-DMA into two buffers, one barrier per
//     stage, and every MFMA reads its A operand with one conflict-free ds_read_b128;
//   * wasted matrix work on absent neighbours: K N / P (3.2x at the dense headline density, ~17x at config 5,
//     minus the skipped all-absent blocks) — affordable at 1/16 of the fp32 price; the gather traffic is the
//     plan kernel's (absent rows are not loaded).
// Works on the neighbour table directly (row space or position space + order), forward and dgrad alike.
// Requirements (else the plan kernel runs): c_src a multiple of 32, c_dst a multiple of 32 (slabs of CB = 2, 4, 6
// or 8 column blocks), rows 16-byte aligned.
// a row of zeros for absent neighbours / channel blocks beyond the row: their loads stay unconditional (the load
// count per stage is fixed, so hipcc can wait for "all but the newest stage" with a counted s_waitcnt)
constexpr int kZeroRowElems = 8192;
__device__ __attribute__((aligned(16))) __bf16 d_zero_row[kZeroRowElems];

// Pipeline (round-2 measurement: with the rows of only the NEXT stage in flight a stage lasted one memory latency,
// 2.4 us for 540 cycles of MFMAs): at stage s the rows of stage s + 2, the neighbour indices of stage s + 3 and the
// weights of stage s + 2 (into registers; written to the LDS buffer one stage later) are requested; every load is
// unconditional and plain (no LDS-DMA: hipcc drains everything at the next use once a DMA is in flight), so its
// counted waits leave two stages of gathers in flight across the one barrier per stage.
template <int CB>
__global__ __launch_bounds__(256, 2) void k_conv_gather_bf16(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ tbl, const int32_t *__restrict__ order, __bf16 *__restrict__ dst, int64_t n_tgt,
    int volume) {
  constexpr int R = 2;                      // 16-row blocks per wave
  constexpr int ROWS = 4 * R * 16;          // target rows per workgroup
  constexpr int STAGE = CB * 2 * 64;        // bf16x8 elements of one weight stage (64 channels x CB * 16 columns)
  constexpr int WPT = CB / 2;               // weight elements (16 bytes) per thread and stage
  constexpr int DEPTH = 3;                  // row ring: stage s multiplies slot s % 3, stage s + 2 is in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8 *s_w = reinterpret_cast<bf16x8 *>(smem);   // [2][CB][2][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int nchunks = (c_src + 63) >> 6;    // 64-channel chunks (the last one may hold 32 channels)
  const int ncb = c_dst >> 4;
  const int slab = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * ROWS + wave * (R * 16);
  const int n_stage = volume * nchunks;

  // target position of this lane in its two row blocks (clamped: loads stay unconditional)
  int64_t pos[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    pos[r] = p0 + r * 16 + i16;
    live[r] = pos[r] < n_tgt;
    if (!live[r]) pos[r] = n_tgt - 1;
  }
  auto load_idx = [&](int st, int32_t (&idx)[R]) {
    const int k = min(st / nchunks, volume - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int32_t v = tbl[(int64_t)k * n_tgt + pos[r]];
      idx[r] = live[r] ? v : -1;
    }
  };
  auto load_w = [&](int st, bf16x8 (&w)[WPT]) {
    st = min(st, n_stage - 1);
    const int k = st / nchunks, c = st % nchunks;
    const bf16x8 *g = wp + ((int64_t)(k * nchunks + c) * ncb + slab * CB) * 2 * 64 + tid;
#pragma unroll
    for (int j = 0; j < WPT; ++j) w[j] = g[j * 256];
  };
  auto store_w = [&](const bf16x8 (&w)[WPT], int buf) {
#pragma unroll
    for (int j = 0; j < WPT; ++j) s_w[buf * STAGE + j * 256 + tid] = w[j];
  };
  // B operands of a stage: 8 channels of the neighbour row per (row block, 32-channel block); absent neighbours and
  // channel blocks beyond c_src read the zero row
  auto load_b = [&](const int32_t (&idx)[R], int st, bf16x8 (&b)[R][2]) {
    const int c = min(st, n_stage - 1) % nchunks;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int ch = c * 64 + kb * 32 + q * 8;
        const bool real = idx[r] >= 0 && ch < c_src;
        const __bf16 *p = real ? src + (int64_t)idx[r] * c_src + ch : d_zero_row + ch;
        b[r][kb] = *reinterpret_cast<const bf16x8 *>(p);
      }
    }
  };

  f32x4 acc[R][CB];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[r][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  int32_t idx[DEPTH][R];
  bf16x8 b[DEPTH][R][2];
  bf16x8 wst[WPT];
  // prologue: indices of stages 0 .. 2, rows of stages 0 and 1, weights of stage 0 in LDS, of stage 1 in registers
  load_idx(0, idx[0]);
  load_idx(1, idx[1]);
  load_idx(2, idx[2]);
  load_w(0, wst);
  load_b(idx[0], 0, b[0]);
  store_w(wst, 0);
  load_w(1, wst);
  load_b(idx[1], 1, b[1]);

  auto stage = [&](int s, int32_t (&idx_c)[R], bf16x8 (&b_c)[R][2], int32_t (&idx_2)[R], bf16x8 (&b_2)[R][2]) {
    // idx_c / b_c: this stage's ring slot; idx_2: indices of stage s + 2 (loaded two stages ago), b_2: its ring slot
    __syncthreads();                 // weights of stage s are visible; buffer (s + 1) & 1 is free
    store_w(wst, (s + 1) & 1);       // weights of stage s + 1 (requested a stage ago)
    load_w(s + 2, wst);
    load_b(idx_2, s + 2, b_2);
    int32_t idx_t[R];                // indices of stage s + 3 take this stage's slot after its last use
    load_idx(s + 3, idx_t);
    const bf16x8 *w = s_w + (s & 1) * STAGE + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (__ballot(idx_c[r] >= 0) == 0ull) continue;     // all 16 neighbours of this row block absent
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(cb * 2 + kb) * 64], b_c[r][kb], acc[r][cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) idx_c[r] = idx_t[r];
  };
  int s = 0;
  for (; s + 2 < n_stage; s += 3) {
    stage(s, idx[0], b[0], idx[2], b[2]);
    stage(s + 1, idx[1], b[1], idx[0], b[0]);
    stage(s + 2, idx[2], b[2], idx[1], b[1]);
  }
  if (s < n_stage) stage(s, idx[0], b[0], idx[2], b[2]);
  if (s + 1 < n_stage) stage(s + 1, idx[1], b[1], idx[0], b[0]);

  // every target row is written once, rounded to bf16 (RNE); the lane holds columns cb * 16 + q * 4 .. + 3 of row i16
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!live[r]) continue;
    const int64_t grow = order ? (int64_t)order[pos[r]] : pos[r];
    __bf16 *o = dst + grow * c_dst + slab * CB * 16 + q * 4;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const f32x4 v = acc[r][cb];
      *reinterpret_cast<bf16x4 *>(o + cb * 16) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
  }
}

#endif  // ME_DEBUG_VARIANTS (k_conv_gather_bf16)

#ifdef ME_DEBUG_VARIANTS
// column blocks per slab of the gather kernel for c_dst columns: the largest of 8, 6, 4, 2 that divides c_dst / 16;
// 0: not eligible
static int conv_gather_cb(int c_src, int c_dst) {
  if (c_src < 32 || c_src % 32 != 0 || c_src + 64 > kZeroRowElems || c_dst < 32 || c_dst % 32 != 0) return 0;
  const int ncb = c_dst / 16;
  for (int cb = 8; cb >= 2; cb -= 2)
    if (ncb % cb == 0) return cb;
  return 0;
}
#endif

extern int g_conv_variant;  // conv.hip: variant 6 = 64-bit gather addresses, 7 = no batch fusion

struct ConvVariantBf16 {
  int nc, slabs, kc;
};

int g_bf16_nc = 0, g_bf16_kc = 0;   // me_debug_set_bf16_shape: tuning overrides of the slab width / chunk depth (0 = policy)
int g_bf16_deep = -1;               // me_debug_set_bf16_deep: -1 policy, 0 never, 1 wherever instantiated
int g_bf16_twobuf = -1;             // me_debug_set_bf16_twobuf: -1 policy, 0 never, 1 wherever the deep pipeline runs
int g_bf16_offsync = 0;             // me_debug_set_bf16_offsync: 0 column-split kernel (k_conv_tile_bf16), 1 offset-synchronous
                                    // kernel where eligible, 2 / 3: the same with its other wave shapes
int g_bf16_splitk = -1;             // me_debug_set_bf16_splitk: -1 policy, 0 / 1 never, G >= 2: G offset groups where eligible
int g_bf16_splitk_same_tiles = 0;   // me_debug_set_bf16_splitk_mode: 1 = forced groups keep the unsplit tile height (G x the workgroups)
int g_bf16_ws_fuse = 0;   // me_debug_set_bf16_ws_fuse: 1 = multi-offset batches (sparse maps) on the wave-specialised kernel too (tuning build)
int g_bf16_ws = -1;   // me_debug_set_bf16_ws: -1 policy, 0 never, 1 wherever the wave-specialised kernel is instantiated
constexpr bool kTwoBufDefault = false;
constexpr int kSplitKMaxTileRows = 48;   // policy: split launches whose unsplit tiles are at most this tall

static ConvVariantBf16 conv_variant_bf16(int c_src, int c_dst) {
  ConvVariantBf16 v;
  // 64 columns per workgroup whenever there are more than 32: the matrix pipe is nearly idle in bf16, so a
  // half-empty last slab costs nothing, while every extra slab re-gathers all source rows.  128 columns (eight
  // waves) where they tile the output exactly: the per-batch chain of a tile (index window -> gather -> stage ->
  // barrier -> operands -> MFMA -> accumulate, ~1 us) is then paid once per 128 columns, and a 256-channel layer on
  // a 5k - 20k voxel map (MinkUNet's deepest levels: one workgroup per CU, nothing to overlap with) walks half as
  // many batches (round 3, profiles/r03_*layers*)
  // widest chunk that tiles the source channels exactly (MinkUNet's 96 / 192-channel layers sit on its largest
  // maps); otherwise the smallest chunk that covers them, or 128
  if (c_src % 128 == 0) v.kc = 128;
  else if (c_src % 96 == 0) v.kc = 96;
  else v.kc = c_src <= 32 ? 32 : (c_src <= 64 ? 64 : 128);
  // 64 columns per workgroup whenever there are more than 32 ...; 128 columns (eight waves) where they tile the output
  // exactly and the chunk is not 96 channels deep.  Measured per layer inside a MinkUNet34C step (round 3,
  // profiles/r03_layers_minkunet34c_bf16_*.log; the kernel is bound by its per-batch chain, not by the gathers, so the
  // slab width moves a layer by a few per cent only): 256 -> 256 on 21k voxels 103.6 us (64 columns) / 95.4 (128) /
  // 91.5 (128 columns, 256-channel chunks); 64 -> 128 on 100k voxels 68 -> 70 us forward but the step 397 -> 420
  // Mpoints/s; 192 -> 128 (96-channel chunks) 147 us with 64 columns, 161 with 128: excluded; 96 columns in one six-wave
  // workgroup (instantiated, override only) 96 -> 96 107 us against 103 with a 64- and a 32-column slab: not used.
  v.nc = c_dst <= 32 ? 32 : 64;
  if (c_dst % 128 == 0 && v.kc != 96) v.nc = 128;
  // 256-channel chunks halve the batches of a 256-channel layer (64 registers of weights per lane: one eight-wave
  // workgroup per CU) — only where the launch is at most two slabs wide: with three (256 -> 384 input gradient, 21k
  // voxels) the single resident workgroup per CU runs two rounds, 246 us against 158
  if (v.nc == 128 && c_src % 256 == 0 && c_dst <= 256 && g_bf16_offsync == 0) v.kc = 256;   // (the offset-synchronous
                                                                // kernel keeps the slice in LDS twice: 128-channel chunks)
#ifdef ME_DEBUG_VARIANTS
  constexpr bool kHas96 = true;    // six-wave workgroups: measured, not used by the policy, instantiated for the tuning build only
#else
  constexpr bool kHas96 = false;
#endif
  if (g_bf16_nc == 32 || g_bf16_nc == 64 || (g_bf16_nc == 96 && kHas96) || g_bf16_nc == 128)
    v.nc = (g_bf16_nc == 32 && c_dst > 32) ? 64 : g_bf16_nc;
  if (v.nc > 64 && c_dst <= 64) v.nc = 64;
  if (v.nc != 128 && v.kc == 256) v.kc = 128;
  if (g_bf16_kc == 32 || g_bf16_kc == 64 || g_bf16_kc == 96 || g_bf16_kc == 128 || (g_bf16_kc == 256 && v.nc == 128))
    v.kc = g_bf16_kc;
  v.slabs = (int)ceil_div(c_dst, v.nc);
  return v;
}

// Tile height for the wave-specialised kernel (conv_bf16_ws.hip).  Measured per layer inside a MinkUNet34C step and in
// sweeps over forced heights (profiles/r04_ws_tile_rows.log, r04_layers_minkunet34c_bf16_ws_*.log): a launch costs its
// BATCHES — 1,000 - 2,000 cycles each, almost whatever their fill — so the tallest tile wins as long as an off-centre
// (tile, offset) item stays within one batch of four groups (64 pairs on average: T <= 64 / p) and the LDS of the
// resident workgroups holds it; with one or two rounds of tiles the height is cut so that the rounds are whole (uniform
// scenes: 2.03 rounds take as long as 3); two workgroups per CU where the registers allow (<= 128 per lane: the
// 64-column shapes, 128 columns up to 64-channel chunks), one when the tiles would get shorter than 64 rows.
static int plan_tile_rows_ws(const ConvVariantBf16 &v, int64_t n_tgt, int64_t volume, double p) {
  const int cus = device_cu_count();
  int occ = (v.nc == 64 || v.kc <= 64) ? 2 : 1;
  for (;; --occ) {
    int t_max = ME_MAX_TILE_ROWS;
    while (t_max > ME_GROUP_ROWS && (int64_t)conv_bf16_ws_lds_bytes(v.nc, v.kc, t_max) * occ > kLdsBudget) --t_max;
    int64_t t_cap = t_max;
    if (volume > 1 && p > 0.0) t_cap = std::min<int64_t>(t_max, std::max<int64_t>(64, (int64_t)(64.0 / p)));
    const int64_t slots = (int64_t)cus * occ;
    const int64_t rounds = ceil_div(n_tgt * v.slabs, slots * t_cap);
    int64_t t = rounds <= 2 ? ceil_div(n_tgt * v.slabs, slots * rounds) : t_cap;
    if (t < ME_GROUP_ROWS) t = ME_GROUP_ROWS;
    if (occ > 1 && rounds == 1 && t < 64) continue;   // a launch of short tiles: one taller workgroup per CU
    return (int)t;
  }
}

// two stage buffers + one barrier per batch for the deep-pipeline (eight-wave) launches
static bool conv_bf16_twobuf(int nc) { return nc == 128 && (g_bf16_twobuf >= 0 ? g_bf16_twobuf != 0 : kTwoBufDefault); }

template <int NC, int KC>
static int launch_conv_tile_bf16(const __bf16 *src, int c_src, const bf16x8 *wp, int c_dst, int slabs,
                                 const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                                 const int32_t *tile_bptr, const int32_t *order, __bf16 *dst, int64_t n_tgt,
                                 int tile_rows, int batch_groups, hipStream_t stream, bool small, bool fuse = false,
                                 float *stat_mean = nullptr, float *stat_m2 = nullptr, int split_k = 1,
                                 float *partial = nullptr, int volume = 0) {
  const bool exact = (c_src % KC) == 0;
  // (the second stage buffer of TWOBUF only where the deep pipeline will run and the tile leaves room for it)
  const bool deep_ok = NC == 128 && exact && small && (g_bf16_deep >= 0 ? g_bf16_deep != 0 : true);
  const bool twobuf = deep_ok && conv_bf16_twobuf(NC) &&
                      conv_bf16_lds_bytes(NC, KC, tile_rows, batch_groups, true) + 16 <= kLdsBudget;
  const int lds = conv_bf16_lds_bytes(NC, KC, tile_rows, batch_groups, twobuf) + (split_k > 1 ? 16 : 0);
  ME_CHECK(lds <= kLdsBudget, "tile_rows / batch_groups too large for the LDS of one workgroup");
  typedef void (*kernel_t)(const __bf16 *, int, const bf16x8 *, int, const int32_t *, const int32_t *, const int32_t *,
                           const int32_t *, const int32_t *, __bf16 *, int64_t, int, int, int, float *, float *, float *,
                           int);
  // two instantiations per shape and mode (round 6; four before): the fast one (exact chunk AND 32-bit gather offsets) and
  // the general one, which also serves the two mixed cases — same sums in the same order
  const bool fast = small && exact;
  kernel_t fn = fast ? &k_conv_tile_bf16<NC, KC, true, true> : &k_conv_tile_bf16<NC, KC, false, false>;
  if constexpr (NC <= 96 && KC <= 128) {   // batch fusion: four- and six-wave workgroups (sparse maps of narrow layers)
    if (fuse) fn = fast ? &k_conv_tile_bf16<NC, KC, true, true, true> : &k_conv_tile_bf16<NC, KC, false, false, true>;
  } else {
    fuse = false;
  }
  // deep pipeline (see k_conv_tile_bf16): where one workgroup per CU is all a launch has — a 256-channel chunk (its
  // registers and LDS allow no second one), or at most ~1.25 workgroups per CU in the grid
  bool deep = false;
  // (measured on four-wave workgroups too — profiles/r03_layers_minkunet34c_bf16_deep_all_widths.log: -5 % on the
  // denser 96-channel layers, +10 % on the sparse ones, whose three workgroups per CU it reduces to two — and
  // instantiated for them in the tuning build only)
#ifdef ME_DEBUG_VARIANTS
  constexpr bool kDeepNarrow = NC != 96;
#else
  constexpr bool kDeepNarrow = false;
#endif
  if constexpr (NC == 128 || kDeepNarrow) {
    deep = exact && small && (g_bf16_deep >= 0 ? g_bf16_deep != 0 : NC == 128);
    if (deep) {
      fn = &k_conv_tile_bf16<NC, KC, true, true, false, true>;
      if constexpr (NC <= 96 && KC <= 128) {
        if (fuse) fn = &k_conv_tile_bf16<NC, KC, true, true, true, true>;
      }
      if constexpr (NC == 128) {
        if (twobuf) fn = &k_conv_tile_bf16<NC, KC, true, true, false, true, false, true>;
      }
    }
  }
  // split-K (see k_conv_tile_bf16): eight-wave deep-pipeline launches only; the statistics move to the reduce kernel
  bool splitk = false;
  if constexpr (NC == 128 && (KC == 64 || KC == 128 || KC == 256)) {
    // (a plan made for split-K is an ordinary plan with taller tiles: a launch that turns out not to be eligible —
    // 64-bit gather addresses, deep pipeline switched off — runs it unsplit)
    if (split_k > 1 && deep && !fuse && volume >= split_k && c_dst % 16 == 0 && split_k <= 8) {
      ME_CHECK(partial != nullptr, "split-K needs its workspace (me_conv_splitk_workspace_bytes)");
      fn = twobuf ? &k_conv_tile_bf16<NC, KC, true, true, false, true, true, true>
                  : &k_conv_tile_bf16<NC, KC, true, true, false, true, true>;
      splitk = true;
    }
  }
  if (!splitk) split_k = 1;
  static bool attr_set[64] = {};  // per instantiation
  const int which = (twobuf && deep ? 32 : 0) + (splitk ? 16 : 0) + (deep ? 8 : 0) + (fuse ? 4 : 0) + (small ? 2 : 0) + (exact ? 1 : 0);
  if (lds > 48 * 1024 && !attr_set[which]) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kLdsBudget));
    attr_set[which] = true;
  }
  const unsigned n_tiles = (unsigned)ceil_div(n_tgt, tile_rows);
  const dim3 grid(n_tiles, (unsigned)slabs, (unsigned)(splitk ? split_k : 1));
  hipLaunchKernelGGL(fn, grid, dim3(NC * 4), (size_t)lds, stream, src, c_src, wp, c_dst, plan_src, plan_dst, batch_desc,
                     tile_bptr, order, dst, n_tgt, tile_rows, batch_groups, g_conv_variant == 7 ? 0 : 1,
                     splitk ? nullptr : stat_mean, splitk ? nullptr : stat_m2, partial, volume);
  ME_LAUNCH_CHECK();
  if constexpr (NC == 128) {
    if (splitk) {
      typedef void (*reduce_t)(const float *, int, const int32_t *, __bf16 *, int64_t, int, float *, float *);
      reduce_t rf = nullptr;
      switch (split_k) {
        case 2: rf = &k_conv_splitk_reduce<128, 2>; break;
        case 3: rf = &k_conv_splitk_reduce<128, 3>; break;
        case 4: rf = &k_conv_splitk_reduce<128, 4>; break;
        case 5: rf = &k_conv_splitk_reduce<128, 5>; break;
        case 6: rf = &k_conv_splitk_reduce<128, 6>; break;
        case 7: rf = &k_conv_splitk_reduce<128, 7>; break;
        default: rf = &k_conv_splitk_reduce<128, 8>; break;
      }
      constexpr int kReduceLds = (128 / 16) * (128 / 4) * 8 * 4 + 128 * 4;
      hipLaunchKernelGGL(rf, dim3(n_tiles, (unsigned)slabs), dim3(512), (size_t)kReduceLds, stream, partial, c_dst, order,
                         dst, n_tgt, tile_rows, stat_mean, stat_m2);
      ME_LAUNCH_CHECK();
    }
  }
  return 0;
}

#ifdef ME_DEBUG_VARIANTS
// k_conv_off_bf16 for a launch, or -1 when the shape / plan is not eligible (the column-split kernel runs then)
template <int NC, int KC, int GW, int CBW>
static int launch_conv_off_bf16(const __bf16 *src, int c_src, const bf16x8 *wp, int c_dst, const int32_t *plan_src,
                                const int32_t *plan_dst, const int32_t *batch_desc, const int32_t *tile_bptr,
                                const int32_t *order, __bf16 *dst, int64_t n_tgt, int tile_rows, int volume,
                                hipStream_t stream, float *stat_mean, float *stat_m2) {
  const int lds = conv_off_lds_bytes(NC, KC, tile_rows, volume);
  if (lds > kLdsBudget) return -1;
  auto fn = &k_conv_off_bf16<NC, KC, GW, CBW>;
  static bool attr_set = false;
  if (lds > 48 * 1024 && !attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_set = true;
  }
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)(c_dst / NC));
  hipLaunchKernelGGL(fn, grid, dim3(GW * (NC / 16 / CBW) * 64), (size_t)lds, stream, src, c_src, wp, c_dst, plan_src, plan_dst,
                     batch_desc, tile_bptr, order, dst, n_tgt, tile_rows, volume, stat_mean, stat_m2);
  ME_LAUNCH_CHECK();
  return 0;
}

static int conv_off_dispatch(const __bf16 *src, int64_t n_src, int c_src, const bf16x8 *wp, int volume, int c_dst,
                             const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                             const int32_t *tile_bptr, const int32_t *order, __bf16 *dst, int64_t n_tgt, int tile_rows,
                             hipStream_t stream, float *stat_mean, float *stat_m2) {
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  const bool small = n_src > 0 && n_src < (1ll << 24) && n_src * c_src * 2 < (1ll << 32);
  if (!small || c_src % v.kc != 0 || volume > 255) return -1;
  const int nc = c_dst % 128 == 0 ? 128 : (c_dst % 96 == 0 ? 96 : (c_dst % 64 == 0 ? 64 : (c_dst % 32 == 0 ? 32 : 0)));
  if (nc == 0) return -1;
  if (stat_mean != nullptr && nc == 96) return -1;     // (no statistics epilogue for 96-column slabs)
  const int shape = g_bf16_offsync;                     // 1: more group waves, 2: more column waves
#define ME_OFF(NCV, KCV, GWV, CBWV)                                                                                   \
  return launch_conv_off_bf16<NCV, KCV, GWV, CBWV>(src, c_src, wp, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, \
                                                   dst, n_tgt, tile_rows, volume, stream, stat_mean, stat_m2)
#define ME_OFF_KC(NCV, GA, CA, GB, CB)                                      \
  do {                                                                      \
    if (v.kc == 128) { if (shape == 2) ME_OFF(NCV, 128, GB, CB); ME_OFF(NCV, 128, GA, CA); } \
    if (v.kc == 96) { if (shape == 2) ME_OFF(NCV, 96, GB, CB); ME_OFF(NCV, 96, GA, CA); }   \
    if (v.kc == 64) { if (shape == 2) ME_OFF(NCV, 64, GB, CB); ME_OFF(NCV, 64, GA, CA); }   \
    if (v.kc == 32) { if (shape == 2) ME_OFF(NCV, 32, GB, CB); ME_OFF(NCV, 32, GA, CA); }   \
  } while (0)
  if (nc == 128) ME_OFF_KC(128, 4, 4, 2, 2);    // 8 waves: 4 group waves x 2 column parts | 2 x 4
  if (nc == 96) ME_OFF_KC(96, 4, 3, 2, 3);      // 8 waves: 4 x 2 | 4 waves: 2 x 2
  if (nc == 64) ME_OFF_KC(64, 4, 2, 8, 4);      // 8 waves: 4 x 2 | 8 x 1
  if (nc == 32) ME_OFF_KC(32, 4, 2, 8, 2);      // 4 waves: 4 x 1 | 8 x 1
#undef ME_OFF_KC
#undef ME_OFF
  return -1;
}

#endif  // ME_DEBUG_VARIANTS

}  // namespace me

using namespace me;

extern "C" {

int me_conv_plan_config_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                             int32_t *tile_rows, int32_t *batch_groups) {
  ME_CHECK(tile_rows != nullptr && batch_groups != nullptr, "output pointers must not be null");
  *tile_rows = 128;
  *batch_groups = ME_MAX_BATCH_GROUPS;
  if (n_tgt <= 0 || volume <= 0 || c_src <= 0 || c_dst <= 0) return 0;
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  PlanShape s;
  s.nc = v.nc;
  s.slabs = v.slabs;
  s.chunks = (int)ceil_div(c_src, v.kc);
  // the wave-specialised kernel: one eight-wave workgroup per CU, two unpadded stage buffers — unless the map is so
  // sparse that every tile height leaves fewer than 24 pairs per (tile, offset) item (the hosts' batch-fusion rule:
  // those launches stay with k_conv_tile_bf16)
  const double p_side = volume > 1 ? (double)(n_pairs > n_tgt ? n_pairs - n_tgt : 0) / ((double)(volume - 1) * (double)n_tgt) : 1.0;
  if (g_bf16_ws != 0 && conv_bf16_ws_shape(v.nc, v.kc) && c_src % v.kc == 0 && p_side * ME_MAX_TILE_ROWS >= 24.0) {
    const int t_ws = plan_tile_rows_ws(v, n_tgt, volume, p_side);
    // The hosts fuse offsets when an off-centre (tile, offset) item holds fewer than 24 pairs AT THE TILE HEIGHT THEY GET,
    // and a fused launch runs on k_conv_tile_bf16: a short wave-specialised tile (a small map: few rounds) would send it
    // there on a geometry tuned for the other kernel.  Same rule here: pairs per item at t_ws (the hosts' figure
    // when both sides of the map have n_tgt rows; strided maps differ by the rows of the smaller side).
    const int64_t tiles_ws = ceil_div(n_tgt, t_ws);
    const double per_item = volume > 1 ? (double)(n_pairs > n_tgt ? n_pairs - n_tgt : 0) / (double)std::max<int64_t>(1, (volume - 1) * tiles_ws)
                                       : 1e9;
    if (per_item >= 24.0) {
      *tile_rows = t_ws;
      return 0;
    }
  }
  s.group_cycles = 64.0 + (v.kc / 32) * 24.0;  // LDS-bound: accumulator read-add-write + operand reads
  s.stage_row_bytes = (v.kc + 16) * 2 + 4;
  s.wave_slots = 4 * conv_bf16_waves_per_simd(v.nc, v.kc);
  *tile_rows = plan_tile_rows(s, n_tgt, volume, n_pairs);
  return 0;
}

// Split-K policy (see k_conv_tile_bf16): a launch whose tiles are short because the map is small — 128 tiles of 39 rows
// for 256 -> 256 channels on 4,977 voxels — spends its time streaming the packed weights (27 x 64 KB per workgroup at
// the ~42 B / clk / CU all CUs get out of the L2s together: scripts/ubench/l2_stream.hip, 17.5 us per walk).  G offset
// groups on G-times taller tiles keep the workgroup count and divide the weight bytes of a workgroup by G.
int me_conv_plan_config_bf16_ex(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                                int32_t *tile_rows, int32_t *batch_groups, int32_t *split_k) {
  ME_CHECK(split_k != nullptr, "output pointers must not be null");
  *split_k = 1;
  const int rc = me_conv_plan_config_bf16(n_tgt, volume, n_pairs, c_src, c_dst, tile_rows, batch_groups);
  if (rc != 0 || n_tgt <= 0 || volume < 2 || g_bf16_splitk == 0 || g_bf16_splitk == 1 || g_bf16_deep == 0) return rc;
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  if (v.nc != 128 || (v.kc != 64 && v.kc != 128 && v.kc != 256) || c_src % v.kc != 0 || c_dst % 16 != 0) return 0;
  const int t0 = *tile_rows;
  const int64_t tiles0 = ceil_div(n_tgt, t0);
  const int gmax = (int)std::min<int64_t>(volume, 8);
  int g = 1;
  if (g_bf16_splitk >= 2) {
    g = std::min(g_bf16_splitk, gmax);
  } else if (v.kc == 256 && t0 <= kSplitKMaxTileRows && tiles0 * v.slabs <= (int64_t)device_cu_count() * 3 / 2) {
    // Measured on the 4,977-voxel level of the MinkUNet34C scene (profiles/r04_splitk_sweep.log, us per launch):
    //   256 -> 256 (T 39): unsplit 44.0, G = 2 36.3, G = 4 38.6;   256 -> 128 (T 20): 38.4, G = 2 29.2, G = 4 25.2;
    //   128 -> 256 (128-channel chunks): 26.8 unsplit, 27.6 / 30.7 split — half the weight bytes per batch, nothing to
    //   gain; every split of the 21k-voxel level lost 10 - 40 % (its tiles are tall already, the reduce pass is not free).
    // Twice the workgroups per CU on the SAME tiles changed nothing (44.5): the launch is bound by the weight bytes the
    // L2s deliver, not by latency.  So: 256-channel chunks only, tiles of ~80 rows, and never more workgroups than CUs
    // (a second round costs more than the split saves: G = 3 and 6 measured 46 - 51 us).
    g = std::min(gmax, std::max(2, std::min(4, (80 + t0 / 2) / t0)));
    while (g > 1 && ceil_div(tiles0, g) * v.slabs * g > (int64_t)device_cu_count()) --g;
  }
  if (g < 2) return 0;
  if (g_bf16_splitk >= 2 && g_bf16_splitk_same_tiles) {   // tuning: more workgroups (occupancy) instead of taller tiles
    *split_k = g;
    return 0;
  }
  // the tallest tile the LDS holds next to a full stage buffer
  const int cap = std::min<int>(ME_MAX_TILE_ROWS,
                                (kLdsBudget - 16 - ME_MAX_BATCH_GROUPS * 16 * ((v.kc + 16) * 2 + 4)) / ((v.nc + kAccPad) * 4) - 1);
  const int64_t tiles = std::max<int64_t>(ceil_div(tiles0, g), ceil_div(n_tgt, cap));
  const int t = (int)std::max<int64_t>(ME_GROUP_ROWS, ceil_div(n_tgt, tiles));
  if (t <= t0) return 0;
  *tile_rows = t;
  *batch_groups = ME_MAX_BATCH_GROUPS;
  *split_k = g;
  return 0;
}

int64_t me_conv_splitk_workspace_bytes(int64_t n_tgt, int32_t tile_rows, int32_t c_dst, int32_t split_k) {
  if (split_k <= 1 || n_tgt <= 0 || tile_rows <= 0 || c_dst <= 0) return 0;
  return (int64_t)split_k * ceil_div(n_tgt, tile_rows) * tile_rows * c_dst * 4;
}

void me_debug_set_bf16_twobuf(int mode) { g_bf16_twobuf = mode; }
void me_debug_set_bf16_ws(int mode) { g_bf16_ws = mode; }
void me_debug_set_bf16_ws_fuse(int mode) { g_bf16_ws_fuse = mode; }
void me_debug_set_bf16_offsync(int mode) {
#ifdef ME_DEBUG_VARIANTS
  g_bf16_offsync = mode;
#else
  (void)mode;   // (the kernel is not in the default build: the switch stays off, me_debug_variants_compiled() says so)
#endif
}
void me_debug_set_bf16_splitk(int g) { g_bf16_splitk = g; }
void me_debug_set_bf16_splitk_mode(int same_tiles) { g_bf16_splitk_same_tiles = same_tiles; }

int64_t me_conv_packed_weight_elems_bf16(int64_t volume, int32_t c_src, int32_t c_dst) {
  if (volume <= 0 || c_src <= 0 || c_dst <= 0) return 0;
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  return volume * align_up(c_src, v.kc) * align_up(c_dst, 16);
}

void me_debug_set_bf16_shape(int nc, int kc) {
  g_bf16_nc = nc;
  g_bf16_kc = kc;
}

void me_debug_set_bf16_deep(int deep) { g_bf16_deep = deep; }

int me_debug_bf16_timing(uint64_t *out20, int32_t reset) {
  if (out20 != nullptr) {
    unsigned long long h[20];
    ME_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(d_bf16_timing), sizeof(h)));
    for (int i = 0; i < 20; ++i) out20[i] = h[i];
  }
  if (reset) {
    unsigned long long z[20] = {};
    ME_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_bf16_timing), z, sizeof(z)));
  }
  return 0;
}

int32_t me_conv_pack_chunk_bf16(int32_t c_src, int32_t c_dst) {
  return (c_src > 0 && c_dst > 0) ? conv_variant_bf16(c_src, c_dst).kc : 0;
}

int me_conv_pack_weights_bf16(const void *w, int32_t w_is_f32, int64_t volume, int32_t c_src, int32_t c_dst,
                              int32_t transposed, uint16_t *wp, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && c_src > 0 && c_dst > 0, "invalid weight shape");
  ME_CHECK((uintptr_t)wp % 16 == 0, "packed weights must be 16-byte aligned");
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  const int nchunks = (int)ceil_div(c_src, v.kc), ncb = (int)ceil_div(c_dst, 16);
  const int64_t total = volume * nchunks * ncb * (v.kc / 32) * 64;  // 16-byte elements
  bf16x8 *wp8 = reinterpret_cast<bf16x8 *>(wp);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
#define ME_PACK(KCV)                                                                                            \
  do {                                                                                                          \
    if (w_is_f32)                                                                                               \
      hipLaunchKernelGGL((k_pack_weights_bf16<KCV, true>), grid, block, 0, stream, w, c_src, c_dst, transposed, \
                         nchunks, ncb, wp8, total);                                                             \
    else                                                                                                        \
      hipLaunchKernelGGL((k_pack_weights_bf16<KCV, false>), grid, block, 0, stream, w, c_src, c_dst, transposed, \
                         nchunks, ncb, wp8, total);                                                             \
  } while (0)
  if (v.kc == 256) ME_PACK(256);
  else if (v.kc == 128) ME_PACK(128);
  else if (v.kc == 96) ME_PACK(96);
  else if (v.kc == 64) ME_PACK(64);
  else ME_PACK(32);
#undef ME_PACK
  ME_LAUNCH_CHECK();
  return 0;
}

static int conv_target_bf16(const uint16_t *src_, int64_t n_src, int32_t c_src, const uint16_t *wp_, int64_t volume,
                            int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                            const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, uint16_t *dst_,
                            int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream_, bool fuse,
                            float *stat_mean = nullptr, float *stat_m2 = nullptr, int split_k = 1,
                            float *partial = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK((stat_mean == nullptr) == (stat_m2 == nullptr), "both statistics buffers or none");
  // 32-bit byte offsets with a 24-bit row multiply need a source matrix below 4 GiB
  const bool small = n_src > 0 && n_src < (1ll << 24) && n_src * c_src * 2 < (1ll << 32) && g_conv_variant != 6;
  ME_CHECK(c_src > 0 && c_dst > 0, "channel counts must be positive");
  ME_CHECK(tile_rows >= ME_GROUP_ROWS && tile_rows <= ME_MAX_TILE_ROWS, "tile_rows out of range");
  ME_CHECK(batch_groups >= 1 && batch_groups <= ME_MAX_BATCH_GROUPS, "batch_groups out of range");
  ME_CHECK((uintptr_t)src_ % 16 == 0 && (uintptr_t)dst_ % 16 == 0 && (uintptr_t)wp_ % 16 == 0,
           "feature and weight pointers must be 16-byte aligned");
  if (n_tgt == 0) return 0;
  const __bf16 *src = reinterpret_cast<const __bf16 *>(src_);
  const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wp_);
  __bf16 *dst = reinterpret_cast<__bf16 *>(dst_);
#ifdef ME_DEBUG_VARIANTS
  if (g_bf16_offsync != 0 && split_k <= 1) {
    const int rc = conv_off_dispatch(src, n_src, c_src, wp, (int)volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr,
                                     order, dst, n_tgt, tile_rows, stream, stat_mean, stat_m2);
    if (rc != -1) return rc;
  }
#endif
  const ConvVariantBf16 v = conv_variant_bf16(c_src, c_dst);
  // the wave-specialised kernel (conv_bf16_ws.hip) wherever it is instantiated and the launch needs none of the things
  // it does not do: split-K, 64-bit gather offsets.  Bit-identical output.
  if (g_bf16_ws != 0 && small && split_k <= 1 && g_conv_variant != 7 && (!fuse || g_bf16_ws_fuse != 0)) {
    const int rc = launch_conv_bf16_ws(v.nc, v.kc, src, c_src, wp, c_dst, v.slabs, plan_src, plan_dst, batch_desc,
                                       tile_bptr, order, dst, n_tgt, tile_rows, stream, stat_mean, stat_m2, fuse);
    if (rc != -1) return rc;
  }
#define ME_CONV_CASE(NCV, KCV)                                                                                   \
  return launch_conv_tile_bf16<NCV, KCV>(src, c_src, wp, c_dst, v.slabs, plan_src, plan_dst, batch_desc, tile_bptr, \
                                         order, dst, n_tgt, tile_rows, batch_groups, stream, small, fuse, stat_mean,   \
                                         stat_m2, split_k, partial, (int)volume)
  if (v.nc == 32) {
    if (v.kc == 128) ME_CONV_CASE(32, 128);
    if (v.kc == 96) ME_CONV_CASE(32, 96);
    if (v.kc == 64) ME_CONV_CASE(32, 64);
    ME_CONV_CASE(32, 32);
#ifdef ME_DEBUG_VARIANTS
  } else if (v.nc == 96) {
    if (v.kc == 128) ME_CONV_CASE(96, 128);
    if (v.kc == 96) ME_CONV_CASE(96, 96);
    if (v.kc == 64) ME_CONV_CASE(96, 64);
    ME_CONV_CASE(96, 32);
#endif
  } else if (v.nc == 128) {
    if (v.kc == 256) ME_CONV_CASE(128, 256);
    if (v.kc == 128) ME_CONV_CASE(128, 128);
    if (v.kc == 96) ME_CONV_CASE(128, 96);
    if (v.kc == 64) ME_CONV_CASE(128, 64);
    ME_CONV_CASE(128, 32);
  } else {
    if (v.kc == 128) ME_CONV_CASE(64, 128);
    if (v.kc == 96) ME_CONV_CASE(64, 96);
    if (v.kc == 64) ME_CONV_CASE(64, 64);
    ME_CONV_CASE(64, 32);
  }
#undef ME_CONV_CASE
}

int me_conv_target_bf16(const uint16_t *src, int64_t n_src, int32_t c_src, const uint16_t *wp, int64_t volume,
                        int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                        const int32_t *tile_bptr, const int32_t *order, uint16_t *dst, int64_t n_tgt, int32_t tile_rows,
                        int32_t batch_groups, void *stream) {
  return conv_target_bf16(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                          n_tgt, tile_rows, batch_groups, stream, false);
}

int me_conv_target_bf16_fused(const uint16_t *src, int64_t n_src, int32_t c_src, const uint16_t *wp, int64_t volume,
                              int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                              const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, uint16_t *dst,
                              int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream) {
  return conv_target_bf16(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                          n_tgt, tile_rows, batch_groups, stream, true);
}

int32_t me_conv_stats_supported_bf16(int32_t c_src, int32_t c_dst) {
  if (c_src <= 0 || c_dst <= 0) return 0;
  const int nc = conv_variant_bf16(c_src, c_dst).nc;
  return (nc == 32 || nc == 64 || nc == 128) ? 1 : 0;
}

int me_conv_target_bf16_stats(const uint16_t *src, int64_t n_src, int32_t c_src, const uint16_t *wp, int64_t volume,
                              int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                              const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, uint16_t *dst,
                              int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, int32_t fused, float *part_mean,
                              float *part_m2, void *stream) {
  ME_CHECK(part_mean != nullptr && part_m2 != nullptr, "the statistics buffers must be given");
  ME_CHECK(me_conv_stats_supported_bf16(c_src, c_dst), "no statistics epilogue for this tile shape");
  return conv_target_bf16(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                          n_tgt, tile_rows, batch_groups, stream, fused != 0, part_mean, part_m2);
}

}  // extern "C"


extern "C" {

int me_conv_target_bf16_ex(const uint16_t *src, int64_t n_src, int32_t c_src, const uint16_t *wp, int64_t volume,
                           int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                           const int32_t *tile_bptr, const int32_t *order, uint16_t *dst, int64_t n_tgt,
                           int32_t tile_rows, int32_t batch_groups, int32_t fused, int32_t split_k, void *workspace,
                           float *part_mean, float *part_m2, void *stream) {
  ME_CHECK(split_k >= 1 && split_k <= 8, "split_k out of range");
  ME_CHECK(split_k == 1 || workspace != nullptr, "split-K needs its workspace (me_conv_splitk_workspace_bytes)");
  ME_CHECK(part_mean == nullptr || me_conv_stats_supported_bf16(c_src, c_dst), "no statistics epilogue for this tile shape");
  return conv_target_bf16(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                          n_tgt, tile_rows, batch_groups, stream, fused != 0, part_mean, part_m2, split_k,
                          reinterpret_cast<float *>(workspace));
}

int32_t me_conv_gather_supported_bf16(int32_t c_src, int32_t c_dst) {
#ifdef ME_DEBUG_VARIANTS
  return conv_gather_cb(c_src, c_dst) > 0 ? 1 : 0;
#else
  (void)c_src;
  (void)c_dst;
  return 0;     // the output-stationary kernel is in the tuning build only (scripts/build_debug.sh): use me_conv_target_bf16
#endif
}

int64_t me_conv_gather_weight_elems_bf16(int64_t volume, int32_t c_src, int32_t c_dst) {
  if (volume <= 0 || c_src <= 0 || c_dst <= 0) return 0;
  return volume * align_up(c_src, 64) * align_up(c_dst, 16);
}

int me_conv_gather_pack_weights_bf16(const void *w, int32_t w_is_f32, int64_t volume, int32_t c_src, int32_t c_dst,
                                     int32_t transposed, uint16_t *wp, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && c_src > 0 && c_dst > 0, "invalid weight shape");
  ME_CHECK((uintptr_t)wp % 16 == 0, "packed weights must be 16-byte aligned");
  const int nchunks = (int)ceil_div(c_src, 64), ncb = (int)ceil_div(c_dst, 16);
  const int64_t total = volume * nchunks * ncb * 2 * 64;  // 16-byte elements
  bf16x8 *wp8 = reinterpret_cast<bf16x8 *>(wp);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (w_is_f32)
    hipLaunchKernelGGL((k_pack_weights_bf16<64, true>), grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks,
                       ncb, wp8, total);
  else
    hipLaunchKernelGGL((k_pack_weights_bf16<64, false>), grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks,
                       ncb, wp8, total);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_conv_gather_bf16(const uint16_t *src_, int64_t n_src, int32_t c_src, const uint16_t *wp_, int64_t volume,
                        int32_t c_dst, const int32_t *tbl, const int32_t *order, uint16_t *dst_, int64_t n_tgt,
                        void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)n_src;
#ifndef ME_DEBUG_VARIANTS
  (void)src_; (void)c_src; (void)wp_; (void)volume; (void)c_dst; (void)tbl; (void)order; (void)dst_; (void)n_tgt; (void)stream;
  ME_CHECK(false, "the output-stationary kernel is in the tuning build only (me_conv_gather_supported_bf16 == 0)");
  return -1;
#else
  const int cb = conv_gather_cb(c_src, c_dst);
  ME_CHECK(cb > 0, "channel counts not eligible for the gather kernel (me_conv_gather_supported_bf16)");
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK((uintptr_t)src_ % 16 == 0 && (uintptr_t)dst_ % 8 == 0 && (uintptr_t)wp_ % 16 == 0,
           "feature and weight pointers must be 16-byte aligned");
  if (n_tgt == 0) return 0;
  const __bf16 *src = reinterpret_cast<const __bf16 *>(src_);
  const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wp_);
  __bf16 *dst = reinterpret_cast<__bf16 *>(dst_);
  const dim3 grid((unsigned)ceil_div(n_tgt, 128), (unsigned)(c_dst / (cb * 16)));
#define ME_GATHER(CBV)                                                                                        \
  hipLaunchKernelGGL(k_conv_gather_bf16<CBV>, grid, dim3(256), (size_t)(2 * CBV * 2 * 64 * 16), stream, src, c_src, \
                     wp, c_dst, tbl, order, dst, n_tgt, (int)volume)
  if (cb == 8) ME_GATHER(8);
  else if (cb == 6) ME_GATHER(6);
  else if (cb == 4) ME_GATHER(4);
  else ME_GATHER(2);
#undef ME_GATHER
  ME_LAUNCH_CHECK();
  return 0;
#endif
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv_bf16(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_pack_weights_bf16<128, true>));
}
