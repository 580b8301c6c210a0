// Sparse convolution feature kernels for gfx950 (MI355X): forward / dgrad / wgrad in exact fp32 on
// the matrix cores (v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32).
//
// Replaces ConvolutionForwardKernelGPU / ConvolutionBackwardKernelGPU
// (src/convolution_kernel.cu:320-496, 553-757).  The reference launches one gather-GEMM-scatter per
// kernel offset and scatters with one global atomicAdd per output element
// (src/convolution_kernel.cu:114-180).  Here the TARGET rows are stationary instead:
//
//   * a workgroup owns `tile_rows` target rows x NC output columns; its fp32 accumulator tile
//     lives in LDS (e.g. 131 x 64 x 4 B = 33 KiB of the CU's 160 KiB; the tile height is chosen so
//     that tiles x column slabs fill the chip's resident-workgroup slots evenly);
//   * the tile plan (coords.hip) lists, per tile, the valid (offset k, source row) entries grouped
//     by k in groups of 16 rows = one MFMA M-tile, so no matrix-core work is spent on absent
//     neighbours beyond the padding of the last group of each (tile, k);
//   * source rows are gathered with 16-byte loads into a padded LDS tile (register-staged, issued
//     one batch ahead of the MFMAs that consume it), each wave keeps its 16-column slice of W_k
//     in registers for the whole run of groups of one offset, and adds its 16x16 result block into
//     the LDS accumulator at the target rows (wave-private columns -> no atomics, fixed summation
//     order -> bitwise reproducible);
//   * every target row is written exactly once with coalesced 16-byte stores: no zero-fill pass,
//     no global atomics.
// dgrad is the same kernel with source = grad_out, W = per-offset transposed kernel and the plan
// of the transposed neighbour table.  wgrad reduces over the per-offset pair lists in chunks.
#include "conv_common.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace me {

// Stage buffer layout.  A staged row is KC floats = KC/4 16-byte pieces.  The MFMA operand read is a
// ds_read_b128 by lane (i16 = row of the group, q): the LDS serves it in four groups of 16 lanes that are NOT
// contiguous ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS), so with padded rows and lane q reading the
// pieces 4q .. 4q+3 half of every group collided (two passes per read; the ablation in
// profiles/r01_ablation_conv_v6c.log puts the operand reads at 78 of 192 us).  Instead lane q reads the pieces
// q, 4 + q, 8 + q, ... (the weights are packed in the same channel order) and piece p of row r is stored at
// slot p ^ swz(r) of an UNPADDED row: every 16-lane service group then touches 16 distinct 16-byte bank slots.
// (KC = 96: 24 pieces per row, an XOR swizzle would leave the row; rows padded to 100 floats = 25 pieces instead,
// an odd stride, so the 16 rows of a group start in 16 different bank slots)
__host__ __device__ constexpr int stage_ld(int kc) {  // floats per staged row
  return kc == 96 ? 100 : (kc >= 32 ? kc : kc + 4);
}
__device__ __forceinline__ int stage_swz(int kc, int row) {
  return kc == 96 ? 0 : (kc >= 64 ? (row & 15) : (kc == 32 ? ((row >> 1) & 7) : 0));
}

// LDS bytes of one workgroup of k_conv_tile_f32<NC, KC>: accumulator tile (+ one dummy row for padding
// slots), one stage buffer of gathered rows and the target-row indices of the batch
__host__ __device__ constexpr int conv_lds_bytes(int nc, int kc, int tile_rows, int batch_groups) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + batch_groups * 16 * (stage_ld(kc) * 4 + 4);
}

// =================================================================================================
// target-stationary convolution (forward and dgrad)
// =================================================================================================
// R groups of one offset at once: R independent accumulators share every weight register, which
// covers the 40-cycle dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (issue interval 32).
// The MFMA is issued "transposed" — A operand = weights (M = 16 output columns), B operand = gathered
// rows (N = 16 rows) — so a lane ends up with 4 CONSECUTIVE output columns of ONE target row and the
// accumulate into LDS is one 16-byte read + one 16-byte write per group.  All reads first, then all
// writes (rows of one offset are distinct, padding slots share a dummy row), so the R LDS round
// trips overlap instead of forming a chain.
// a0p: this lane's row of the first group (s_a + i16 * A_LD); pofs[s4]: float offset of the piece it reads at
// quad-step s4 (swizzled, see stage_swz).
template <int R, int KQ, int A_LD, int ACC_LD, int VAR = 0>
__device__ __forceinline__ void mma_groups(const float *__restrict__ a0p, const int (&pofs)[KQ / 4],
                                           const float (&wreg)[KQ], const int32_t *__restrict__ dstp,
                                           float *__restrict__ accp) {
  // accumulator row of this lane's entry as a float offset (24-bit multiply: a 64-bit mad here picked up an
  // undefined high half from a register with a load in flight and waited for the whole gather)
  int d[R];
#pragma unroll
  for (int r = 0; r < R; ++r) d[r] = (int)__umul24((unsigned)dstp[r * 16], (unsigned)ACC_LD);
  // The accumulators START from the LDS tile (the old sum is the C operand of the first MFMA) and are written back
  // after the last k-step: no zero-fill, no v_add_f32 — on gfx950 every VALU instruction costs matrix time
  // (docs/HISTORY.md 3.1a).  The sum of a target row is ((old + x_0 w_0) + x_1 w_1) + ... in plan order: still a fixed
  // order.  (VAR & 64: timing ablation without the LDS accumulator traffic.)
  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r)
    acc[r] = (VAR & 64) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(accp + d[r]);
  // ALL operand reads of the R groups are issued before the first MFMA (LDS returns in order, so the MFMAs of
  // quad-step 0 start as soon as its reads land while the rest stream in).  hipcc's own schedule read two
  // pieces, waited, multiplied, read the next two, waited ...: the LDS latency was exposed KQ/4 times per
  // pair of groups — 78 of 192 us on config 2 (profiles/r01_ablation_conv_v6c.log).
  f32x4 a[KQ / 4][R];
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      a[s4][r] = *reinterpret_cast<const f32x4 *>(a0p + r * 16 * A_LD + pofs[s4]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s4 * 4 + j], a[s4][r][j], acc[r], 0, 0, 0);
    }
  }
  if (VAR & 64) {  // ablation: no write of the LDS accumulator (results kept alive through one never-taken write)
    f32x4 s = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) s += acc[r];
    if (s.x == 12345.678f) *reinterpret_cast<f32x4 *>(accp + d[0]) = s;
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4 *>(accp + d[r]) = acc[r];
}

// n single-group batches of DIFFERENT offsets staged together (multi-offset batch, round 3): group r multiplies with its
// own offset's weights w[r].  On a sparse map a (tile, offset) item holds one half-empty 16-row group, and a batch of
// one group pays the whole chain (index window -> gather -> stage -> two barriers -> operands -> 8 .. 16 MFMAs ->
// accumulate) for 256 .. 512 cycles of matrix work: config 5 (K = 81, 4.6 pairs per voxel) ran 81 such batches per tile
// at 18 % of the fp32 MFMA peak.  Four groups per batch give the matrix pipe 4x the work per chain.  Two groups of a
// batch may hit the SAME target row (different offsets), so the accumulators start from zero, all MFMAs run on
// independent chains, and the tile is updated group by group in batch order (LDS operations of a wave execute in order:
// a read issued behind the previous group's write sees it).  The sum of a row is old + (x_0 w_0 + x_1 w_1 + ...) per
// group in plan order: fixed order, bitwise reproducible (not bit-identical to the unfused kernel, whose accumulators
// start from the tile).
template <int M, int KQ, int A_LD, int ACC_LD>
__device__ __forceinline__ void mma_singles_f32(const float *__restrict__ a0p, const int (&pofs)[KQ / 4],
                                                const float (&w)[M][KQ], int n, const int32_t *__restrict__ dstp,
                                                float *__restrict__ accp) {
  int d[M];
  f32x4 a[KQ / 4][M];
  f32x4 acc[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    d[r] = (int)__umul24((unsigned)dstp[r * 16], (unsigned)ACC_LD);
    acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) {
#pragma unroll
    for (int r = 0; r < M; ++r) a[s4][r] = *reinterpret_cast<const f32x4 *>(a0p + r * 16 * A_LD + pofs[s4]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < M; ++r)
        if (r < n) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r][s4 * 4 + j], a[s4][r][j], acc[r], 0, 0, 0);   // wave-uniform
    }
  }
#pragma unroll
  for (int r = 0; r < M; ++r) {
    if (r < n) {
      const f32x4 old = *reinterpret_cast<const f32x4 *>(accp + d[r]);
      *reinterpret_cast<f32x4 *>(accp + d[r]) = old + acc[r];
    }
  }
}

// Packed weights: the exact register image of the kernel.  For offset k, source-channel chunk c,
// 16-column block cb and k-step quad v, lane (q = lane >> 4, i16 = lane & 15) finds its four weights
//   W[k][c*KC + (4*v + q)*4 + j][cb*16 + i16],  j = 0..3      (piece 4*v + q of the staged row)
// as ONE 16-byte element at  ((((k*nchunks + c)*ncb + cb)*(KQ/4) + v)*64 + lane)  — zero beyond the
// real channel counts, so the kernel needs no guards and a wave reads 1 KiB contiguous per load.
template <int KC>
__global__ __launch_bounds__(256) void k_pack_weights(const float *__restrict__ w, int c_src, int c_dst,
                                                     int transposed, int nchunks, int ncb,
                                                     f32x4 *__restrict__ wp, int64_t total) {
  constexpr int KQ = KC / 4;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e % 64);
  int64_t r = e / 64;
  const int v = (int)(r % (KQ / 4));
  r /= (KQ / 4);
  const int cb = (int)(r % ncb);
  r /= ncb;
  const int c = (int)(r % nchunks);
  const int64_t k = r / nchunks;
  const int q = lane >> 4, i16 = lane & 15;
  const int col = cb * 16 + i16;
  f32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = c * KC + (4 * v + q) * 4 + j;
    float val = 0.f;
    if (ch < c_src && col < c_dst) {
      // plain: w is [K, c_src, c_dst]; transposed (dgrad): w is the forward kernel [K, c_dst, c_src]
      val = transposed ? w[(k * c_dst + col) * c_src + ch] : w[(k * c_src + ch) * c_dst + col];
    }
    out[j] = val;
  }
  wp[e] = out;
}

// One workgroup of NC/16 waves owns `tile_rows` target rows x NC output columns (one column slab); its
// fp32 accumulator tile lives in LDS (e.g. 131 x 68 x 4 B = 36 KiB: three workgroups share a CU and
// hide each other's barriers and memory latencies).  The work of a tile is a list of BATCHES (plan,
// coords.hip): up to `batch_groups` (<= 4) groups of 16 (source row, target row) entries of ONE kernel
// offset.  Per batch, software-pipelined over three batches:
//   * batch b+2: its plan indices are fetched;
//   * batch b+1: this wave's slice of W_k (register image, 16-byte loads) and the source rows (16-byte
//     gather, global -> registers) are requested right after the barrier that publishes batch b, and are
//     not touched again before the top of the next iteration — a whole batch of MFMAs later;
//   * batch b: after a barrier the rows gathered during batch b-1 are written to the LDS stage buffer
//     (padding slots zeroed here, not at the load), after a second barrier every wave multiplies its 16
//     output columns of all groups on the matrix cores (mma_groups) and adds the 16x16 blocks into the
//     LDS accumulator.  The target rows of a batch are distinct: plain read-add-write, no atomics, fixed
//     summation order (bitwise reproducible).
// EVERY vector load of the loop is unconditional (indices clamped, redundant loads of the last batch at
// the end of a tile): hipcc's s_waitcnt insertion counts loads per basic block, and a load behind a
// branch, a select next to its load or a register spill turned into s_waitcnt vmcnt(0) in the middle of
// the iteration — the whole gather latency exposed per batch (measured, profiles/r01_tune_conv_v6b*).
// After the last batch every target row is written exactly once with coalesced 16-byte stores (rows
// without entries get zeros: no zero-fill pass, no global atomics).
// VAR bits (timing ablations only; 0 is the shipped configuration): 16: no gather traffic (constant rows);
// 2: no per-batch weight loads; 4: no barriers (timing only); 64: no read-add-write of the LDS accumulator;
// 128: no stage writes; 256: s_memtime phase counters; 2048 + n: n KiB of LDS padding (caps the resident
// workgroups).  Restructurings that were measured and removed again (logs under profiles/): two stage buffers
// with one barrier per batch, gathering two batches ahead, staggered start of the co-resident workgroups,
// 128-column workgroups.
// EXACT: c_src is a multiple of KC (and of 4): the gather needs no channel guards.
// phase cycle counters of the VAR & 256 instrumentation build (summed over wave 0 of every workgroup):
// 0 barrier A, 1 stage write (incl. the wait for the gathered rows), 2 barrier B, 3 load issue, 4 multiply,
// 5 prologue (accumulator clear + first loads), 6 epilogue, 7 batches
__device__ unsigned long long d_conv_timing[8];

// SMALL: source rows < 2^24, row bytes < 2^24, source matrix < 4 GiB (host-checked): gather addresses are the scalar
// base + one v_mad_u32_u24 instead of 64-bit per-lane arithmetic.  On gfx950 an fp32 MFMA occupies the SIMD
// exclusively: while v_mfma_f32_16x16x4_f32 executes, no VALU and no vector-memory instruction of ANY wave of that
// SIMD issues, only LDS and scalar instructions do (scripts/ubench/coissue*.hip, profiles/r01_ubench_coissue*.log:
// a v_add_u32 costs ~5 cycles and a global_load_dwordx4 ~17 cycles of matrix time, in the same wave or in a
// neighbour).  Occupancy hides latencies but not instruction issue, so every VALU / VMEM instruction of the loop is
// paid for in matrix time.
// (the fused instantiation of the two-wave, 64-channel shape — config 5's input gradient — needs 200 registers for the
// weight slices of two offsets in use and two in flight: two waves per SIMD instead of spilling)
template <int NC, int KC, bool EXACT, int VAR, bool SMALL = false, bool FUSE = false>
__global__ __launch_bounds__(NC * 4, ((NC >= 96 || KC == 96 || (FUSE && NC == 32 && KC == 64)) ? 2 : 3)) void k_conv_tile_f32(
    const float *__restrict__ src, int c_src, const f32x4 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, float *__restrict__ dst, int64_t n_tgt, int tile_rows, int batch_groups) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  constexpr int WAVES = NC / 16;
  constexpr int NT = WAVES * 64;
  constexpr int A_LD = stage_ld(KC);   // floats per staged row (unpadded + swizzled for KC >= 32)
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int KQ = KC / 4;           // MFMA k-steps per chunk (= weight registers per lane)
  constexpr int F4 = KC / 4;           // 16-byte pieces per gathered row
  constexpr int ITER = (ME_MAX_BATCH_GROUPS * 16 * F4 + NT - 1) / NT;
  static_assert(KC % 16 == 0, "KC must be a multiple of 16");
  static_assert(ME_MAX_BATCH_GROUPS == 4, "mma_groups runs cover at most 4 groups");

  const int cap_rows = batch_groups * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);                      // [(tile_rows + 1) x ACC_LD]
  float *s_a = s_acc + (tile_rows + 1) * ACC_LD;                       // [cap_rows x A_LD]
  int32_t *s_dst = reinterpret_cast<int32_t *>(s_a + cap_rows * A_LD);  // [cap_rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15;  // gathered row (MFMA B column) / weight column (MFMA A row) of this lane
  const int q = lane >> 4;    // MFMA k index of this lane; after the MFMA: output columns q*4 .. q*4+3
  // workgroups take the tiles in the plan's dispatch order (heaviest first: me_plan_build), stored behind the
  // n_tiles + 1 batch pointers; gridDim.x == n_tiles
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];
  const int col_base = blockIdx.y * NC;
  const bool vec_ok = (c_src % 4) == 0;
  const int nchunks = (c_src + KC - 1) / KC;
  const int ncb = (c_dst + 15) / 16;
  // this wave's 16-column block; a block beyond c_dst (last slab) multiplies the last real block again
  // and its columns are simply not stored
  const int cb = min(col_base / 16 + wave, ncb - 1);
  // float offsets of the pieces this lane reads as MFMA operand: piece 4*s4 + q at its swizzled slot (the row
  // of a group is i16 mod 16, so the swizzle is a lane constant)
  int pofs[KQ / 4];
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) pofs[s4] = ((4 * s4 + q) ^ stage_swz(KC, i16)) * 4;

  unsigned long long tm[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = (VAR & 256) ? __builtin_amdgcn_s_memtime() : 0ull;
  auto tick = [&](int slot) {
    if (VAR & 256) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tm[slot] += now - t_prev;
      t_prev = now;
    }
  };
  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the rows this tile's positions stand for (tiles of a position-space map): requested now, parked in the free
  // stage buffer after the main loop — a load of order[] inside the store loop put one memory latency in front of
  // every store (spatial tiles measured 5 - 13 % slower than row tiles because of it)
  constexpr int ORD = (ME_MAX_TILE_ROWS + NT - 1) / NT;
  int32_t my_ord[ORD];
#pragma unroll
  for (int j = 0; j < ORD; ++j) {
    const int r = j * NT + tid;
    my_ord[j] = (order != nullptr && r < tile_rows && (int64_t)tile * tile_rows + r < n_tgt)
                    ? order[(int64_t)tile * tile_rows + r] : 0;
  }

  const int b0 = tile_bptr[tile];
  const int nb = tile_bptr[tile + 1] - b0;
  const int n_it = nb * nchunks;  // iterations: chunk-major, then the batches of the tile

  // batch `it` (clamped to the last one): chunk, first group, groups, offset — scalar loads (the address is
  // wave-uniform), fetched three iterations ahead
  auto locate = [&](int it, int &chunk, int &g0, int &ng, int &k) {
    int r = min(it, n_it - 1);
    chunk = 0;
    while (r >= nb) {
      r -= nb;
      ++chunk;
    }
    const i32x2 d = *reinterpret_cast<const i32x2 *>(batch_desc + 2 * (int64_t)(b0 + r));
    g0 = d.x;
    ng = d.y & 255;
    k = (int)((uint32_t)d.y >> 8);
  };

  // software pipeline registers
  struct GatherSet {
    f32x4 stage[ITER];    // gathered rows of a batch on their way to the LDS stage buffer
    int32_t dstv;         // this thread's target-row entry of the batch
  };
  // Padding slots (plan index -1) gather row 0 and are NOT zeroed: their target row is the dummy accumulator row,
  // which is never stored.
  GatherSet G0;
  G0.dstv = tile_rows;
  int32_t sidx[ITER];   // plan indices of the batch that is gathered next (rows beyond the batch repeat its
                        // last row: they are staged into slots nobody reads)
  float wreg[KQ], wnxt[KQ];

  // (the index window of a batch is read to its end unconditionally: the plan is followed by 64 valid entries,
  // k_plan_fill; rows beyond the batch are staged into slots nobody reads)
  auto load_sidx = [&](int g0, int ng) {
    (void)ng;
    const char *pb = reinterpret_cast<const char *>(plan_src + (int64_t)g0 * 16);
#pragma unroll
    for (int j = 0; j < ITER; ++j)
      sidx[j] = *reinterpret_cast<const int32_t *>(
          pb + (unsigned)(min((j * NT + tid) / F4, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));
  };
  // issue the gather of the batch whose indices sit in sidx (global -> registers); nothing here consumes
  // the result
  const char *srcb = reinterpret_cast<const char *>(src);
  const unsigned row_bytes = (unsigned)c_src * 4u;
  auto gather = [&](GatherSet &G, int chunk, int g0, int ng) {
    f32x4 (&stage)[ITER] = G.stage;
    const int c0 = chunk * KC;
    (void)ng;
    G.dstv = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)g0 * 16) +
                                               (unsigned)(min(tid, ME_MAX_BATCH_GROUPS * 16 - 1) * 4));
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int ch = c0 + ((j * NT + tid) % F4) * 4;
      const int sr = max(sidx[j], 0);
      if (VAR & 16) {  // timing ablation only: no gather traffic
        stage[j] = f32x4{1.f, 1.f, 1.f, 1.f};
      } else if (SMALL && (EXACT || vec_ok)) {
        // channels beyond c_src (not EXACT) are zeroed at the stage write
        const unsigned off = __umul24((unsigned)sr, row_bytes) + (unsigned)(EXACT ? ch : min(ch, c_src - 4)) * 4u;
        stage[j] = *reinterpret_cast<const f32x4 *>(srcb + off);
      } else if (EXACT) {
        stage[j] = *reinterpret_cast<const f32x4 *>(src + (int64_t)sr * c_src + ch);
      } else {
        // clamped scalar / vector loads; channels beyond c_src are zeroed at the stage write
        const float *rowp = src + (int64_t)sr * c_src;
        if (vec_ok) {
          stage[j] = *reinterpret_cast<const f32x4 *>(rowp + min(ch, c_src - 4));
        } else {
          stage[j] = f32x4{rowp[min(ch, c_src - 1)], rowp[min(ch + 1, c_src - 1)], rowp[min(ch + 2, c_src - 1)],
                           rowp[min(ch + 3, c_src - 1)]};
        }
      }
    }
  };
  auto write_stage = [&](GatherSet &G, int chunk) {
    f32x4 (&stage)[ITER] = G.stage;
    const int32_t dstv = G.dstv;
    const int c0 = chunk * KC;
    float *s_ab = s_a;
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int idx = j * NT + tid;
      const int r = idx / F4;
      const int ch = c0 + (idx % F4) * 4;
      f32x4 t = stage[j];
      if (!EXACT) {
        if (ch >= c_src) t.x = 0.f;
        if (ch + 1 >= c_src) t.y = 0.f;
        if (ch + 2 >= c_src) t.z = 0.f;
        if (ch + 3 >= c_src) t.w = 0.f;
      }
      const int slot = (idx % F4) ^ stage_swz(KC, r);
      if (VAR & 128) {  // ablation: no stage writes (the registers are kept alive through a never-taken store)
        if (t.x == 12345.678f) *reinterpret_cast<f32x4 *>(&s_ab[r * A_LD + slot * 4]) = t;
      } else if (r < cap_rows) {
        *reinterpret_cast<f32x4 *>(&s_ab[r * A_LD + slot * 4]) = t;
      }
    }
    if (tid < cap_rows) s_dst[tid] = dstv;
  };
  bool w_loaded = false;
  auto load_w = [&](int chunk, int k) {
    if ((VAR & 2) && w_loaded) return;  // ablation: the weights of the first batch are reused
    w_loaded = true;
    const f32x4 *p = wp + ((((int64_t)k * nchunks + chunk) * ncb + cb) * (KQ / 4)) * 64 + lane;
#pragma unroll
    for (int v = 0; v < KQ / 4; ++v) {
      const f32x4 t = p[v * 64];
      wnxt[v * 4 + 0] = t.x;
      wnxt[v * 4 + 1] = t.y;
      wnxt[v * 4 + 2] = t.z;
      wnxt[v * 4 + 3] = t.w;
    }
  };

  if constexpr (FUSE) {
    // ---- multi-offset batches (see mma_singles_f32): runs of single-group batches of consecutive offsets — contiguous
    // in the plan — are staged and multiplied together, up to MAXSUB of them (the weights of MAXSUB offsets have to fit
    // the registers next to the slices in flight: 4 up to 32 source channels per chunk, 2 beyond) ----
    constexpr int MAXSUB = KQ <= 8 ? 4 : 2;
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
        // (only single-group batches are fused: group r <-> sub-batch r)
        const bool take = j == 0 || (sb.nsub == j && j < avail && sb.ng == j && g == 1);
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
    float wf[MAXSUB][KQ], wfn[MAXSUB][KQ];
    auto load_ws = [&](const Super &sb) {
#pragma unroll
      for (int j = 0; j < MAXSUB; ++j) {
        if (j == 0 || j < sb.nsub) {   // wave-uniform: a full batch loads one slice
          const f32x4 *p = wp + ((((int64_t)sb.k[j] * nchunks + sb.chunk) * ncb + cb) * (KQ / 4)) * 64 + lane;
#pragma unroll
          for (int v = 0; v < KQ / 4; ++v) {
            const f32x4 t = p[v * 64];
            wfn[j][v * 4 + 0] = t.x;
            wfn[j][v * 4 + 1] = t.y;
            wfn[j][v * 4 + 2] = t.z;
            wfn[j][v * 4 + 3] = t.w;
          }
        }
      }
    };
    Super sA = next_super();
    if (sA.nsub > 0) {
      Super sB = next_super();
      Super sC = next_super();
      load_ws(sA);
      load_sidx(sA.g0, 0);
      gather(G0, sA.chunk, sA.g0, 0);
      load_sidx(sB.g0, 0);
      while (sA.nsub > 0) {
        __syncthreads();
        write_stage(G0, sA.chunk);
#pragma unroll
        for (int j = 0; j < MAXSUB; ++j) {
#pragma unroll
          for (int sx = 0; sx < KQ; ++sx) wf[j][sx] = wfn[j][sx];
        }
        __syncthreads();
        load_ws(sB);
        gather(G0, sB.chunk, sB.g0, 0);
        load_sidx(sC.g0, 0);
        {
          const float *a0p = &s_a[i16 * A_LD];
          const int32_t *dstp = &s_dst[i16];
          float *accp = &s_acc[wave * 16 + q * 4];
          if (sA.nsub > 1) {   // wave-uniform
            mma_singles_f32<MAXSUB, KQ, A_LD, ACC_LD>(a0p, pofs, wf, sA.nsub, dstp, accp);
          } else {
            const int g = sA.sg[0];
            if (g == 4) {
              mma_groups<2, KQ, A_LD, ACC_LD, 0>(a0p, pofs, wf[0], dstp, accp);
              mma_groups<2, KQ, A_LD, ACC_LD, 0>(a0p + 32 * A_LD, pofs, wf[0], dstp + 32, accp);
            } else if (g == 3) {
              mma_groups<2, KQ, A_LD, ACC_LD, 0>(a0p, pofs, wf[0], dstp, accp);
              mma_groups<1, KQ, A_LD, ACC_LD, 0>(a0p + 32 * A_LD, pofs, wf[0], dstp + 32, accp);
            } else if (g == 2) {
              mma_groups<2, KQ, A_LD, ACC_LD, 0>(a0p, pofs, wf[0], dstp, accp);
            } else {
              mma_groups<1, KQ, A_LD, ACC_LD, 0>(a0p, pofs, wf[0], dstp, accp);
            }
          }
        }
        sA = sB;
        sB = sC;
        sC = next_super();
      }
    }
  } else if (n_it > 0) {
    // batch cursors: A = it, B = it + 1, C = it + 2 (clamped to the last batch: the tail of a tile repeats
    // its last loads instead of branching around them)
    int chA, gA, nA, kA, chB, gB, nB, kB, chC, gC, nC, kC;
    locate(0, chA, gA, nA, kA);
    locate(1, chB, gB, nB, kB);
    locate(2, chC, gC, nC, kC);
    load_w(chA, kA);
    load_sidx(gA, nA);
    gather(G0, chA, gA, nA);
    load_sidx(gB, nB);

    auto multiply = [&](int n_groups) {
      // After the MFMA this lane holds columns wave*16 + q*4 .. +3 of target row s_dst[g*16 + i16]; columns are
      // private to this wave -> plain LDS read-add-write in a fixed order; padding slots land in the dummy row
      // `tile_rows`.
      const float *a0p = &s_a[i16 * A_LD];
      const int32_t *dstp = &s_dst[i16];
      float *accp = &s_acc[wave * 16 + q * 4];
      if (n_groups == 4) {
        mma_groups<2, KQ, A_LD, ACC_LD, VAR>(a0p, pofs, wreg, dstp, accp);
        mma_groups<2, KQ, A_LD, ACC_LD, VAR>(a0p + 32 * A_LD, pofs, wreg, dstp + 32, accp);
      } else if (n_groups == 3) {
        mma_groups<2, KQ, A_LD, ACC_LD, VAR>(a0p, pofs, wreg, dstp, accp);
        mma_groups<1, KQ, A_LD, ACC_LD, VAR>(a0p + 32 * A_LD, pofs, wreg, dstp + 32, accp);
      } else if (n_groups == 2) {
        mma_groups<2, KQ, A_LD, ACC_LD, VAR>(a0p, pofs, wreg, dstp, accp);
      } else {
        mma_groups<1, KQ, A_LD, ACC_LD, VAR>(a0p, pofs, wreg, dstp, accp);
      }
    };
    tick(5);
    for (int it = 0; it < n_it; ++it) {
      if (!(VAR & 4)) __syncthreads();  // everybody is done reading the previous batch from s_a / s_dst
      tick(0);
      write_stage(G0, chA);  // rows of batch it (gathered during the previous iteration)
#pragma unroll
      for (int sx = 0; sx < KQ; ++sx) wreg[sx] = wnxt[sx];  // its weights were requested before its rows
      if (VAR & 256) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      tick(1);
      if (!(VAR & 4)) __syncthreads();  // (VAR & 4: timing-only ablation without barriers; results are garbage)
      tick(2);
      // next batch: its weights, its rows, and the indices of the one after
      load_w(chB, kB);
      gather(G0, chB, gB, nB);
      load_sidx(gC, nC);
      tick(3);
      multiply(nA);
      if (VAR & 256) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tick(4);
      chA = chB; gA = gB; nA = nB; kA = kB;
      chB = chC; gB = gC; nB = nC; kB = kC;
      locate(it + 3, chC, gC, nC, kC);
    }
  }
  __syncthreads();
  tick(0);

  // every target row of the tile is written exactly once; local row r is target row order[row0 + r]
  // (row0 + r without an order)
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  int32_t *s_ord = reinterpret_cast<int32_t *>(s_a);   // (the stage buffer is free now; cap_rows * A_LD >= 256 ints)
  if (order != nullptr) {
#pragma unroll
    for (int j = 0; j < ORD; ++j)
      if (j * NT + tid < tile_rows) s_ord[j * NT + tid] = my_ord[j];
    __syncthreads();
  }
  const bool vec_out = (c_dst % 4) == 0;
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / (NC / 4);
    const int c4 = x % (NC / 4);
    const int cc = col_base + c4 * 4;
    if (row < rows_here && cc < c_dst) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
      const int64_t grow = order ? (int64_t)s_ord[row] : row0 + row;
      float *o = dst + grow * c_dst + cc;
      if (vec_out) {
        *reinterpret_cast<f32x4 *>(o) = v;
      } else {
        o[0] = v.x;
        if (cc + 1 < c_dst) o[1] = v.y;
        if (cc + 2 < c_dst) o[2] = v.z;
        if (cc + 3 < c_dst) o[3] = v.w;
      }
    }
  }
  if (VAR & 256) {
    tick(6);
    if (tid == 0) {
#pragma unroll
      for (int s = 0; s < 7; ++s) atomicAdd(&d_conv_timing[s], tm[s]);
      atomicAdd(&d_conv_timing[7], (unsigned long long)n_it);
    }
  }
}


// =================================================================================================
// target-stationary convolution, LDS-DMA variant (round 2): k_conv_tile_dma_f32
// =================================================================================================
// The same tile plan and the same arithmetic as k_conv_tile_f32 (single-offset batches of <= 4 groups of 16
// (source row, target row) entries, W_k slice in registers, fp32 accumulator tile in LDS, wave-private columns,
// fixed summation order), with the staging redesigned around what round 1 measured (docs/HISTORY.md 3.1a: an fp32 MFMA
// owns its SIMD — every VALU / VMEM instruction of ANY resident wave is paid for in matrix time — and the
// per-batch barrier pair + register staging + LDS round trips left the matrix pipe 50 % busy):
//   * gathered rows go global -> LDS directly (global_load_lds_dwordx4, 1 KiB = 4 rows per wave instruction; the
//     XOR swizzle of the stage image is applied to the per-lane SOURCE address, the LDS image of an instruction is
//     lane-linear): no staging registers, no ds_write pass, no second barrier;
//   * two stage buffers: the DMA of batch b+1 is issued right after the ONE barrier of batch b and lands while
//     batch b is multiplied; the barrier's s_waitcnt vmcnt(0) is the only wait on vector memory in the loop;
//   * the MFMA accumulators are INITIALISED from the LDS accumulator tile (old value as the C operand of the
//     first MFMA) and written back after the last k-step: no zero-fill, no v_add_f32 — the sum of a target row is
//     ((old + x_0 w_0) + x_1 w_1) + ... in plan order, still a fixed order (bitwise reproducible);
//   * all operand reads of up to four groups are in flight before the first MFMA and the four accumulator chains
//     are interleaved k-step-major (dependent MFMAs 128 cycles apart): one LDS latency per batch instead of two;
//   * plan indices reach the lanes through ds_bpermute_b32 from ONE coalesced load per wave (no LDS index
//     array), weights alternate between two register sets (no copy);
//   * NC = 128: eight waves own ALL output columns of a 64 -> 128 layer, so every gathered row is fetched once
//     instead of once per 64-column slab (half the gather instructions and half the L2-side traffic per MFMA).
// Requirements (host-checked, else k_conv_tile_f32 runs): c_src a multiple of 64, c_dst a multiple of NC, source
// matrix below 2^24 rows and 4 GiB.
// R groups x CBW 16-column blocks x (KCH * 16) k-steps.  `w[c][s]`: weight register of column block c, k-step s;
// stage image: chunk h of the batch at a0p + h * st_chunk (floats), group r at + r * 16 * 64.
template <int R, int CBW, int KCH, int VAR = 0, typename Mid>
__device__ __forceinline__ void mma_groups_dma(const float *__restrict__ a0p, int st_chunk, const int (&pofs)[4],
                                               const float (&w)[CBW][16 * KCH], const int *d,
                                               float *__restrict__ accp, Mid &&mid) {
  // Read order = use order (the LDS returns in order): accumulators and the operands of quad-steps 0 and 1 first;
  // the operands of quad-steps 2p + 2 and 2p + 3 are requested from INSIDE the MFMA stream at the end of block 2p
  // (LDS instructions co-issue with the matrix pipe) and first used a whole block later, so the s_waitcnt
  // lgkmcnt(0) hipcc puts behind a sched_barrier never finds a young read outstanding.
  constexpr int S = 4 * KCH;  // quad-steps (4 MFMA k-steps each)
  f32x4 acc[R][CBW];
  f32x4 a[S][R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      if (VAR & 64) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};   // timing ablation: no accumulator read
      else acc[r][c] = *reinterpret_cast<const f32x4 *>(accp + d[r] + c * 16);
    }
  }
  auto read_step = [&](int s) {
    const float *p = a0p + (s >> 2) * st_chunk + pofs[s & 3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (VAR & 32) {   // timing ablation: no operand reads (opaque register values, so the MFMAs are not folded)
        f32x4 t = f32x4{1.f + r, 2.f + s, 3.f, 4.f};
        asm volatile("" : "+v"(t));
        a[s][r] = t;
      } else {
        a[s][r] = *reinterpret_cast<const f32x4 *>(p + r * 16 * 64);
      }
    }
  };
  read_step(0);
  read_step(1);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc2[CBW];   // R == 1 only: second chain (odd k-steps)
#pragma unroll
  for (int c = 0; c < CBW; ++c) acc2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto block = [&](int s) {
    if constexpr (R == 1 && CBW == 1) {
      // one group, one column block: two interleaved chains instead of back-to-back dependent MFMAs (40-cycle
      // latency against a 32-cycle issue interval)
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][s * 4 + 0], a[s][0][0], acc[0][0], 0, 0, 0);
      acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][s * 4 + 1], a[s][0][1], acc2[0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][s * 4 + 2], a[s][0][2], acc[0][0], 0, 0, 0);
      acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][s * 4 + 3], a[s][0][3], acc2[0], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int c = 0; c < CBW; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][s * 4 + j], a[s][r][j], acc[r][c], 0, 0, 0);
        }
      }
    }
  };
#pragma unroll
  for (int s = 0; s < S; s += 2) {
    block(s);
    if (s + 2 < S) {
      read_step(s + 2);
      read_step(s + 3);
    }
    if (s == 0) mid();   // scalar / LDS-crossbar work of the NEXT iterations, issued inside the MFMA stream
    __builtin_amdgcn_sched_barrier(0);
    block(s + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (R == 1 && CBW == 1) acc[0][0] += acc2[0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      if (VAR & 64) {   // timing ablation: no accumulator write (kept alive through a never-taken store)
        if (acc[r][c].x == 12345.678f) *reinterpret_cast<f32x4 *>(accp + d[r] + c * 16) = acc[r][c];
      } else {
        *reinterpret_cast<f32x4 *>(accp + d[r] + c * 16) = acc[r][c];
      }
    }
  }
}

__host__ __device__ constexpr int conv_dma_lds_bytes(int nc, int kch, int tile_rows, int batch_groups) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + 2 * kch * batch_groups * 16 * 64 * 4;
}

// NC output columns per workgroup; CBW 16-column blocks per wave (waves = NC / (16 * CBW)); KCH 64-channel source
// chunks staged and multiplied per batch.  Measured on config 2 (profiles/r02_tune_conv_dma.log): waves of one SIMD
// do NOT overlap each other's non-matrix work (the VALU / VMEM instructions of a wave starve while its neighbour's
// fp32 MFMAs run, docs/HISTORY.md 3.1a), so what counts is MFMAs per barrier interval — <128, 2, 1> (forward 64 -> 128)
// and <64, 1, 2> (dgrad 128 -> 64) run ONE four-wave workgroup per CU with 128 MFMAs per wave and batch.
template <int NC, int CBW, int KCH, int VAR>
__global__ __launch_bounds__(NC / CBW * 4, (CBW * KCH >= 2 ? 1 : 2)) void k_conv_tile_dma_f32(
    const float *__restrict__ src, int c_src, const f32x4 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, float *__restrict__ dst, int64_t n_tgt, int tile_rows, int batch_groups) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  constexpr int WAVES = NC / (16 * CBW);
  constexpr int NT = WAVES * 64;
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int GSTEP = WAVES / 4;        // groups covered by one "DMA slot" of the workgroup (1 or 2)
  constexpr int JMAX = 4 / GSTEP;         // DMA slots per wave and batch (4 or 2)
  static_assert(WAVES == 4 || WAVES == 8, "four or eight waves");
  static_assert(ME_MAX_BATCH_GROUPS == 4, "batches hold at most 4 groups");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);              // [(tile_rows + 1) x ACC_LD]
  float *s_st = s_acc + (tile_rows + 1) * ACC_LD;              // [2][KCH][batch_groups * 16 x 64], swizzled rows
  const int st_chunk = batch_groups * 16 * 64;                 // floats of one chunk of one stage buffer
  const int st_floats = st_chunk * KCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  // workgroups take the tiles in the plan's dispatch order (heaviest first: me_plan_build), stored behind the
  // n_tiles + 1 batch pointers; gridDim.x == n_tiles
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];
  const int nchunks = c_src >> 6;          // 64-channel chunks of the source rows (a multiple of KCH)
  const int nsuper = nchunks / KCH;        // chunk groups = passes over the tile's batches
  const int ncb = c_dst >> 4;
  const int cb0 = (blockIdx.y * WAVES + wave) * CBW;           // this wave's first 16-column block
  int pofs[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) pofs[s4] = ((4 * s4 + q) ^ i16) * 4;

  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the rows this tile's positions stand for: requested now, parked in the free stage buffer after the main loop
  int32_t my_ord = 0;
  if (order != nullptr && tid < tile_rows && (int64_t)tile * tile_rows + tid < n_tgt)
    my_ord = order[(int64_t)tile * tile_rows + tid];     // (tile_rows <= 256 = NT)

  const int b0 = tile_bptr[tile];
  const int nb = tile_bptr[tile + 1] - b0;
  const int n_it = (VAR & 512) ? 0 : nb * nsuper;  // iterations: chunk-group-major, then the batches of the tile
  auto locate = [&](int it, int &chunk, int &g0, int &ng, int &k) {                // (VAR & 512: prologue + epilogue only)
    int r = min(it, n_it - 1);
    chunk = 0;
    while (r >= nb) {
      r -= nb;
      chunk += KCH;
    }
    const i32x2 dsc = *reinterpret_cast<const i32x2 *>(batch_desc + 2 * (int64_t)(b0 + r));
    g0 = dsc.x;
    ng = dsc.y & 255;
    k = (int)((uint32_t)dsc.y >> 8);
  };

  // DMA geometry of this wave: slot j covers the 4 rows (wave & 3) * 4 .. + 3 of group j * GSTEP + (wave >> 2);
  // lane l moves piece (l & 15) ^ swz(row) of row (l >> 4) — the LDS image is lane-linear, the swizzle is on the
  // source address (row & 15 = (wave & 3) * 4 + (l >> 4) in every group)
  const int rb = (wave & 3) * 4;
  const int gsel = wave >> 2;
  const unsigned piece_off = (unsigned)((lane & 15) ^ (rb + (lane >> 4))) * 16u;
  const char *srcb = reinterpret_cast<const char *>(src);
  const unsigned row_bytes = (unsigned)c_src * 4u;
  const int sel_src = (rb + (lane >> 4)) * 4;   // ds_bpermute byte selector of this lane's row inside a group
  // plan indices of this lane's rows in the JMAX DMA slots (ds_bpermute from the wave's 64-entry window)
  auto dma_rows = [&](int sv, int (&idx)[JMAX]) {
#pragma unroll
    for (int j = 0; j < JMAX; ++j) idx[j] = __builtin_amdgcn_ds_bpermute(sel_src + (j * GSTEP + gsel) * 64, sv);
  };
  auto dma = [&](const int (&idx)[JMAX], int chunk, int ng, float *st) {
    const unsigned cofs = piece_off + (unsigned)chunk * 256u;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int g = j * GSTEP + gsel;
      if (g < ng) {                                             // wave-uniform
        const unsigned off = __umul24((unsigned)max(idx[j], 0), row_bytes) + cofs;
#pragma unroll
        for (int h = 0; h < KCH; ++h) {
          if (!(VAR & 16))
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(srcb + off + h * 256),
                (__attribute__((address_space(3))) void *)(st + h * st_chunk + (g * 16 + rb) * 64), 16, 0, 0);
        }
      }
    }
  };
  auto load_w = [&](float (&w)[CBW][16 * KCH], int chunk, int k) {
#pragma unroll
    for (int h = 0; h < KCH; ++h) {
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        // (32-bit element offset: the packed image of a layer is far below 2^31 pieces)
        const unsigned e = (unsigned)(((k * nchunks + chunk + h) * ncb + cb0 + c) * 256 + lane);
        const f32x4 *p = wp + e;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const f32x4 t = p[v * 64];
          w[c][h * 16 + v * 4 + 0] = t.x;
          w[c][h * 16 + v * 4 + 1] = t.y;
          w[c][h * 16 + v * 4 + 2] = t.z;
          w[c][h * 16 + v * 4 + 3] = t.w;
        }
      }
    }
  };
  // (the 64-entry index window of a batch is read to its end unconditionally: the plan is followed by 64 valid
  // entries, k_plan_fill)
  auto load_idx = [&](const int32_t *plan, int g0) { return plan[(unsigned)(g0 * 16 + lane)]; };

  float wA[CBW][16 * KCH], wB[CBW][16 * KCH];
  if (n_it > 0) {
    int chA, gA, nA, kA, chB, gB, nB, kB, chC, gC, nC, kC;
    locate(0, chA, gA, nA, kA);
    locate(1, chB, gB, nB, kB);
    locate(2, chC, gC, nC, kC);
    int svB, dvA, dvB;
    {
      const int svA = load_idx(plan_src, gA);
      dvA = load_idx(plan_dst, gA);
      svB = load_idx(plan_src, gB);
      dvB = load_idx(plan_dst, gB);
      int idxA[JMAX];
      dma_rows(svA, idxA);
      __syncthreads();                       // accumulator tile cleared (the DMA does not touch it; ordering only)
      dma(idxA, chA, nA, s_st);
      load_w(wA, chA, kA);
    }
    float *accp = &s_acc[wave * CBW * 16 + q * 4];

    // accumulator rows of this lane's entries of a batch (float offsets; padding slots -> the dummy row `tile_rows`)
    auto dst_rows = [&](int dv, int (&d)[4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d[r] = (int)__umul24((unsigned)__builtin_amdgcn_ds_bpermute((r * 16 + i16) * 4, dv), (unsigned)ACC_LD);
    };
    int dA[4];
    dst_rows(dvA, dA);

    auto iteration = [&](int it, float (&wc)[CBW][16 * KCH], float (&wn)[CBW][16 * KCH]) {
      // plan indices of the NEXT batch's rows reach the lanes by ds_bpermute before the barrier (their load was
      // issued an iteration ago; the crossbar latency overlaps the wait for the other waves)
      int idxB[JMAX];
      dma_rows(svB, idxB);
      // batch `it` has landed in stage[it & 1] (this wave's DMA: vmcnt(0); everybody's: the barrier), and every
      // wave is done reading stage[(it + 1) & 1] (batch it - 1)
      if (VAR & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // timing ablation: no barrier
      else __syncthreads();
      float *st_cur = s_st + (it & 1) * st_floats;
      float *st_nxt = s_st + ((it + 1) & 1) * st_floats;
      int svC = svB, dvC = dvB;
      if (it + 1 < n_it) {                   // wave-uniform
        dma(idxB, chB, nB, st_nxt);
        if (!(VAR & 2)) load_w(wn, chB, kB);   // (VAR & 2: timing ablation, the first batch's weights are reused)
        svC = load_idx(plan_src, gC);
        dvC = load_idx(plan_dst, gC);
      }
      const float *a0p = st_cur + i16 * 64;
      // inside the MFMA stream (LDS crossbar and scalar instructions co-issue with the matrix pipe): the
      // accumulator rows of the NEXT batch (its index load was issued an iteration ago) and the descriptor of the
      // batch three iterations ahead (scalar load: its latency used to be exposed at the loop's end)
      int dN[4];
      int chD, gD, nD, kD;
      auto mid = [&]() {
        dst_rows(dvB, dN);
        locate(it + 3, chD, gD, nD, kD);
      };
      if (nA == 4) mma_groups_dma<4, CBW, KCH, VAR>(a0p, st_chunk, pofs, (VAR & 2) ? wA : wc, dA, accp, mid);
      else if (nA == 3) mma_groups_dma<3, CBW, KCH, VAR>(a0p, st_chunk, pofs, (VAR & 2) ? wA : wc, dA, accp, mid);
      else if (nA == 2) mma_groups_dma<2, CBW, KCH, VAR>(a0p, st_chunk, pofs, (VAR & 2) ? wA : wc, dA, accp, mid);
      else mma_groups_dma<1, CBW, KCH, VAR>(a0p, st_chunk, pofs, (VAR & 2) ? wA : wc, dA, accp, mid);
      chA = chB; gA = gB; nA = nB; kA = kB;
      chB = chC; gB = gC; nB = nC; kB = kC;
      chC = chD; gC = gD; nC = nD; kC = kD;
#pragma unroll
      for (int r = 0; r < 4; ++r) dA[r] = dN[r];
      svB = svC;
      dvB = dvC;
    };
    int it = 0;
    for (; it + 1 < n_it; it += 2) {
      iteration(it, wA, wB);
      iteration(it + 1, wB, wA);
    }
    if (it < n_it) iteration(it, wA, wB);
  }
  __syncthreads();

  // every target row of the tile is written exactly once; local row r is target row order[row0 + r]
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const int col_base = blockIdx.y * NC;
  int32_t *s_ord = reinterpret_cast<int32_t *>(s_st);   // (free now; >= 1024 ints)
  if (order != nullptr) {
    if (tid < tile_rows) s_ord[tid] = my_ord;
    __syncthreads();
  }
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / (NC / 4);
    const int c4 = x % (NC / 4);
    if (row < rows_here) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
      const int64_t grow = order ? (int64_t)s_ord[row] : row0 + row;
      *reinterpret_cast<f32x4 *>(dst + grow * c_dst + col_base + c4 * 4) = v;
    }
  }
}


__global__ __launch_bounds__(256) void k_transpose_kernel(const float *__restrict__ w, int64_t volume,
                                                         int c_in, int c_out,
                                                         float *__restrict__ wt) {
  // wt[k][j][i] = w[k][i][j]; 32x32 LDS tile transpose
  __shared__ float s[32][33];
  const int k = blockIdx.z;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *wk = w + (int64_t)k * c_in * c_out;
  float *wtk = wt + (int64_t)k * c_in * c_out;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    s[r][tx] = (i < c_in && j < c_out) ? wk[(int64_t)i * c_out + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + r, i = i0 + tx;
    if (i < c_in && j < c_out) wtk[(int64_t)j * c_in + i] = s[tx][r];
  }
}

// =================================================================================================
// wgrad: grad_w[k] = X_g^T (c_in x n_k) . dY_g (n_k x c_out), reduction over the pairs of offset k
// =================================================================================================
// Barrier-free and LDS-free.  The concatenated pair list (sorted by offset k) is cut into R equal
// RANGES of pairs, one per workgroup, regardless of the offset boundaries, so every SIMD gets the
// same number of MFMAs; a range that crosses an offset boundary simply flushes its accumulators and
// starts the next offset ("segment").  A wave owns a 64 x (16*NB) block of grad_w[k] and feeds
// v_mfma_f32_16x16x4_f32 straight from global memory: one step = 4 pairs; lane (i16 = lane & 15,
// q = lane >> 4) loads ONE 16-byte piece of x[in[e + q]] (channels ci0 + 4*i16 .. +3) and one
// 4*NB-byte piece of dy[out[e + q]] — 16 lanes x 16 B = the whole 256-byte x row, so every gathered
// byte is loaded exactly once per wave — and that feeds 4*NB MFMAs on independent accumulators
// (MFMA row i of block m <-> input channel ci0 + 4*i + m; column j of block n <-> output channel
// co0 + NB*j + n).  Pair indices arrive 64 at a time with one coalesced load and reach the lanes
// through ds_bpermute; operands are prefetched DEPTH steps ahead in a statically indexed register
// ring.  The waves of a workgroup sit side by side along the output channels (same x rows -> L1).
// Flushes go to workspace slot (range + k) — unique, because (range, k) only ever moves up a
// staircase — as the raw register image (coalesced); k_wgrad_reduce sums the slots of each offset
// in range order (fixed summation order -> bitwise reproducible) and undoes the channel interleave.
constexpr int kWgMB = 4;  // 16-row MFMA blocks of input channels per wave (64 channels, one dwordx4 per lane)

// Which range a workgroup takes (round 2).  All ranges are resident at once and every range walks its pairs in
// ascending output row, so the ranges that started at the same FRACTION of their offset's list read the same dy rows
// at the same time — one range per offset, ~27 of them.  Workgroups go to the eight XCDs round-robin by index, so in
// launch order those 27 land on eight different L2s and every dy row is fetched from the Infinity Cache / HBM once
// per pair (TCC hit rate 19 %).  The table sends ranges of the same fraction to the same XCD; it travels as a kernel
// argument (no device buffer, no copy).  n = 0: identity.
constexpr int kWgMaxOrder = 1024;
struct WgRangeOrder {
  int n;
  uint16_t v[kWgMaxOrder];
};

// largest k in [0, volume) with koffs[k] <= e
__device__ __forceinline__ int wgrad_locate_offset(const int64_t *__restrict__ koffs, int volume, int64_t e) {
  int lo = 0, hi = volume;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (koffs[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__host__ __device__ __forceinline__ int64_t wgrad_range_begin(int64_t r, int64_t n_pairs, int64_t n_ranges) {
  return r * n_pairs / n_ranges;
}

// the range r with range_begin(r) <= e < range_begin(r + 1)
__device__ __forceinline__ int64_t wgrad_range_of_pair(int64_t e, int64_t n_pairs, int64_t n_ranges) {
  int64_t r = e * n_ranges / n_pairs;
  while (r > 0 && wgrad_range_begin(r, n_pairs, n_ranges) > e) --r;
  while (r + 1 < n_ranges && wgrad_range_begin(r + 1, n_pairs, n_ranges) <= e) ++r;
  return r;
}

// One lane's V consecutive channels of a gathered row, always as unconditional loads (loads inside
// exec-masked branches make hipcc fall back to s_waitcnt vmcnt(0), which serialises the prefetch
// ring).  `ch` is already clamped into the row (channels beyond the real count only feed outputs that
// k_wgrad_reduce ignores).  CHECK: the pair may be a padding pair (row < 0) -> zeros.
// VEC: c is a multiple of V (one V*4-byte load); otherwise V scalar loads with clamped channels.
// SMALL: rows < 2^24, row bytes < 2^24 and the matrix below 4 GiB (checked by the host): the address is the scalar
// base plus ONE v_mad_u32_u24 — an fp32 MFMA blocks the VALU of its SIMD, so the 64-bit multiply-add, the 64-bit
// add and the register copies of the general address cost matrix time (see k_conv_tile_f32).
template <typename T, int V, bool VEC, bool CHECK, bool SMALL = false>
__device__ __forceinline__ void load_piece(const T *__restrict__ base, int32_t row, int c, int ch,
                                           float (&out)[V]) {
  typedef T tvec __attribute__((ext_vector_type(V)));
  const bool ok = !CHECK || row >= 0;
  if constexpr (VEC && SMALL && V > 1) {
    const uint32_t off = __umul24((uint32_t)(CHECK ? max(row, 0) : row), (uint32_t)c * (uint32_t)sizeof(T)) +
                         (uint32_t)ch * (uint32_t)sizeof(T);
    const tvec t = *reinterpret_cast<const tvec *>(reinterpret_cast<const char *>(base) + off);
#pragma unroll
    for (int m = 0; m < V; ++m) out[m] = ok ? (float)t[m] : 0.f;
    return;
  }
  const uint32_t r0 = (uint32_t)(CHECK ? max(row, 0) : row) * (uint32_t)c;
  if constexpr (VEC) {
    const T *p = base + (r0 + (uint32_t)ch);
    if constexpr (V == 1) {
      const T t = *p;
      out[0] = ok ? (float)t : 0.f;
    } else {
      const tvec t = *reinterpret_cast<const tvec *>(p);  // one V * sizeof(T)-byte load
#pragma unroll
      for (int m = 0; m < V; ++m) out[m] = ok ? (float)t[m] : 0.f;
    }
  } else {
#pragma unroll
    for (int m = 0; m < V; ++m) {
      const T t = base[r0 + (uint32_t)(ch + m < c ? ch + m : 0)];
      out[m] = ok ? (float)t : 0.f;
    }
  }
}

// T: element type of the gathered rows (float, or __bf16 converted on load: the products of bf16 values are
// exact in fp32, so the bf16 path computes the same sums as an fp32 convolution of the rounded inputs).
template <typename T, int NB, int DEPTH, bool VEC, bool SMALL = false>
__global__ __launch_bounds__(256, (NB == 4 ? 2 : 3)) void k_wgrad_f32(const T *__restrict__ x, int c_in,
                                                  const T *__restrict__ dy, int c_out,
                                                  const int32_t *__restrict__ in_pairs,
                                                  const int32_t *__restrict__ out_pairs,
                                                  const int64_t *__restrict__ koffs, int volume,
                                                  int64_t n_pairs, int n_ranges, int n_cob,
                                                  float *__restrict__ partial, const WgRangeOrder order) {
  constexpr int MB = kWgMB;
  static_assert(DEPTH <= 8 && 16 % DEPTH == 0, "ring depth must divide the 16 steps of a 64-pair block");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int range = order.n > 0 ? (int)order.v[blockIdx.x] : (int)blockIdx.x;
  const int ci0 = blockIdx.y * (16 * MB);
  const int cob = blockIdx.z * (blockDim.x >> 6) + wave;  // block of 16*NB output channels
  if (cob >= n_cob) return;  // whole wave idle (there are no barriers in this kernel)
  const int co0 = cob * (16 * NB);
  const int64_t e_lo = wgrad_range_begin(range, n_pairs, n_ranges);
  const int64_t e_hi = wgrad_range_begin(range + 1, n_pairs, n_ranges);
  if (e_lo >= e_hi) return;
  // this lane's channels, clamped into the row (see load_piece)
  const int cha = (ci0 + MB * i16 < c_in) ? ci0 + MB * i16 : 0;
  const int chb = (co0 + NB * i16 < c_out) ? co0 + NB * i16 : 0;
  // register image of one (slot, cin block, cout block): MB*NB*4 registers x 64 lanes
  constexpr int kImage = MB * NB * 4 * 64;
  const int64_t image_stride = (int64_t)gridDim.y * n_cob * kImage;  // floats per slot
  float *const image0 = partial + ((int64_t)blockIdx.y * n_cob + cob) * kImage + lane;

  f32x4 acc[MB][NB];
  float ra[DEPTH][MB], rb[DEPTH][NB];

  int k = wgrad_locate_offset(koffs, volume, e_lo);
  int64_t e0 = e_lo;
  while (e0 < e_hi) {
    const int64_t kend = koffs[k + 1];
    if (kend <= e0) {  // offsets without pairs
      ++k;
      continue;
    }
    const int64_t e1 = min(e_hi, kend);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pair indices of the current block of 64 pairs (lane l holds pair eb + l) and of the next one;
    // pairs beyond the segment are -1 (padding)
    int32_t pin, pout, pin_n, pout_n;
    auto load_pairs = [&](int64_t eb, int32_t &pi, int32_t &po) {
      const int64_t e = eb + lane;
      const int64_t ec = min(e, e1 - 1);  // unconditional load from a valid address
      const int32_t vi = in_pairs[ec], vo = out_pairs[ec];
      pi = (e < e1) ? vi : -1;
      po = (e < e1) ? vo : -1;
    };
    // hand the indices of step `pos` (counted from the current block; pos >= 16: next block) to the lanes
    int32_t ri_nx, ro_nx;
    auto permute = [&](int pos) {
      const int src_lane = 4 * (pos & 15) + q;
      ri_nx = __shfl(pos < 16 ? pin : pin_n, src_lane, 64);
      ro_nx = __shfl(pos < 16 ? pout : pout_n, src_lane, 64);
    };
    auto mma_step = [&](const float (&a)[MB], const float (&b)[NB]) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    };
    load_pairs(e0, pin, pout);
    load_pairs(e0 + 64, pin_n, pout_n);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      permute(d);
      load_piece<T, MB, VEC, true, SMALL>(x, ri_nx, c_in, cha, ra[d]);
      load_piece<T, NB, VEC, true, SMALL>(dy, ro_nx, c_out, chb, rb[d]);
    }
    permute(DEPTH);
    // one block = 16 steps of 4 pairs.  At step st: multiply ring slot st % DEPTH, refill it with step
    // st + DEPTH (indices permuted one step earlier), permute the indices of step st + DEPTH + 1.
    auto block = [&](auto check, int nsteps) {
      constexpr bool CHECK = decltype(check)::value;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        if (!CHECK || st < nsteps) {  // wave-uniform
          const int d = st % DEPTH;
          mma_step(ra[d], rb[d]);
          load_piece<T, MB, VEC, CHECK, SMALL>(x, ri_nx, c_in, cha, ra[d]);
          load_piece<T, NB, VEC, CHECK, SMALL>(dy, ro_nx, c_out, chb, rb[d]);
          permute(st + DEPTH + 1);
          // keep hipcc's scheduler from sinking the refill loads down to their use DEPTH steps later
          // (it would trade the whole prefetch distance for a few registers)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      pin = pin_n;
      pout = pout_n;
    };
    int64_t eb = e0;
    // blocks whose own pairs and the next block's pairs are all real: no padding checks
    for (; eb + 128 <= e1; eb += 64) {
      block(std::false_type{}, 16);
      load_pairs(eb + 128, pin_n, pout_n);
    }
    for (; eb < e1; eb += 64) {
      block(std::true_type{}, (int)min((int64_t)16, (e1 - eb + 3) >> 2));
      load_pairs(eb + 128, pin_n, pout_n);
    }
    float *img = image0 + (int64_t)(range + k) * image_stride;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[((m * NB + n) * 4 + r) * 64] = acc[m][n][r];
    e0 = e1;
  }
}

// =================================================================================================
// wgrad on bf16 rows: grad_w[k] = X_g^T . dY_g on v_mfma_f32_16x16x32_bf16 (reduction dimension = pairs)
// =================================================================================================
// Both MFMA operands need, per lane, 8 CONSECUTIVE PAIRS of ONE channel — the transpose of how rows lie
// in memory — so the rows go through LDS: a workgroup (4 waves) stages a step of SP = 32*KSTEPS pairs,
// x rows as [SP][64 channels] and dy rows as [SP][64*NB channels] (every gathered byte is loaded ONCE per
// workgroup with 16-byte loads and shared by the four waves), and reads the operands back with
// ds_read_b64_tr_b16: the 16 lanes of a group point at 4 rows x 4 chunks of 4 channels and each lane
// receives ONE channel of the 4 rows (probed: scripts/ubench/tr_probe.hip).  Row stride = channels + 16
// elements: the 8 rows a 32-lane pass touches fall into 8 disjoint 32-byte bank windows.
// Wave w owns the 64 x (16*NB) block of grad_w[k] at output channels (4*blockIdx.z + w)*16*NB; the four
// waves share the x tile.  Work split, slots and the ordered reduction are those of k_wgrad_f32 (conv.hip):
// equal pair ranges regardless of offset boundaries, register-image partials at slot (range + k).
// Pipeline (one barrier per step, two LDS buffers): rows of step s+1 are in flight (global -> registers)
// and the pair indices of step s+2 are being fetched while step s is multiplied.
constexpr int kWgStepLd = 16;  // padding elements per staged row

// (Round 3 also built a variant with TWO steps of gathered rows in flight — second register set, one counted wait per step,
// bit-identical — and measured it 3 - 6 % slower, profiles/r03_wgrad_bf16_two_steps_in_flight.log: the loop is not waiting
// for more requests in flight.  Removed again.)
// MB: 16-channel blocks of INPUT channels per workgroup — 4 (64 channels) or 8 (128).  The kernel is bound by its gathers
// (config 2: 326 MB fetched for 45 MB of rows, TCC hit rate 13 %, profiles/r03_pmc_traffic_bf16.log): every x slice is
// gathered once per block of output channels and every dy slice once per block of input channels, so a 128 x 128 block
// moves 2/3 of the bytes of a 64 x 128 one per multiply-add (64 accumulator registers instead of 32).
template <int NB, int KSTEPS, int MB = 4>
__global__ __launch_bounds__(256, 2) void k_wgrad_bf16(const __bf16 *__restrict__ x, int c_in,
                                                      const __bf16 *__restrict__ dy, int c_out,
                                                      const int32_t *__restrict__ in_pairs,
                                                      const int32_t *__restrict__ out_pairs,
                                                      const int64_t *__restrict__ koffs, int volume,
                                                      int64_t n_pairs, int n_ranges, int n_cob,
                                                      float *__restrict__ partial) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  static_assert(MB == 4 || MB == 8, "input-channel blocks per workgroup");
  constexpr int CI = 16 * MB;            // input channels per workgroup
  constexpr int XQ = CI / 8;             // 16-byte pieces per staged x row
  constexpr int SP = 32 * KSTEPS;        // pairs per step (<= 64: one index register per wave)
  constexpr int COB = 64 * NB;           // output channels per workgroup
  constexpr int XLD = CI + kWgStepLd;    // elements
  constexpr int DLD = COB + kWgStepLd;
  constexpr int XP = SP * XQ / 256;      // 16-byte x pieces per thread and step
  constexpr int DP = SP * (COB / 8) / 256;
  static_assert(SP <= 64 && XP >= 1, "step size");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16 *s_x = reinterpret_cast<__bf16 *>(smem);            // [2][SP][XLD]
  __bf16 *s_d = s_x + 2 * SP * XLD;                           // [2][SP][DLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int range = blockIdx.x;
  const int ci0 = blockIdx.y * CI;
  const int cog = blockIdx.z * COB;                           // first output channel of the workgroup
  const int cob = blockIdx.z * 4 + wave;                      // this wave's block of 16*NB output channels
  const int64_t e_lo = n_pairs * range / n_ranges;
  const int64_t e_hi = n_pairs * (range + 1) / n_ranges;
  if (e_lo >= e_hi) return;                                   // whole workgroup

  constexpr int kImage = MB * NB * 4 * 64;
  const int64_t image_stride = (int64_t)gridDim.y * n_cob * kImage;
  float *const image0 = partial + ((int64_t)blockIdx.y * n_cob + min(cob, n_cob - 1)) * kImage + lane;

  // step cursor: offset k, first pair e, number of pairs cnt (0 = past the end of the range)
  auto first_offset = [&](int64_t e) {
    int lo = 0, hi = volume;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (koffs[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  };
  auto step_count = [&](int k, int64_t e) -> int {
    if (e >= e_hi) return 0;
    const int64_t kend = min(koffs[k + 1], e_hi);
    return (int)min((int64_t)SP, kend - e);
  };
  auto advance = [&](int &k, int64_t &e, int cnt) {
    e += cnt;
    if (e < e_hi) {
      while (koffs[k + 1] <= e) ++k;
    }
  };

  // this thread's pieces: x piece j is row (j*256 + tid) >> 3, channels ci0 + ((j*256 + tid) & 7) * 8
  int32_t pin = 0, pout = 0;            // pair indices of the step whose rows are loaded next (lane l: pair e + l)
  bf16x8 rx[XP], rd[DP];
  auto load_idx = [&](int64_t e) {
    const int64_t ec = min(e + lane, n_pairs - 1);            // unconditional load from a valid address
    const int32_t *pi = in_pairs + ec, *po = out_pairs + ec;
    asm volatile("global_load_dword %0, %1, off" : "=v"(pin) : "v"(pi) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(pout) : "v"(po) : "memory");
  };
  // The row loads are inline asm: hipcc sinks ordinary loads whose first use is in the next iteration below
  // the MFMAs (into the last block before the back edge), which serialises gather and multiply.  asm volatile
  // keeps them where they are written; the matching s_waitcnt is issued by hand (wait_rows) right before the
  // registers are stored to LDS one iteration later.
  auto load_rows = [&]() {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / XQ;
      const int ch = ci0 + (idx % XQ) * 8;
      const int32_t r = __shfl(pin, row, 64);
      const __bf16 *p = x + (int64_t)r * c_in + (ch < c_in ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rx[j]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / (COB / 8);
      const int ch = cog + (idx % (COB / 8)) * 8;
      const int32_t r = __shfl(pout, row, 64);
      const __bf16 *p = dy + (int64_t)r * c_out + (ch < c_out ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rd[j]) : "v"(p) : "memory");
    }
  };
  auto wait_rows = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
#pragma unroll
    for (int j = 0; j < XP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rx[j]));
#pragma unroll
    for (int j = 0; j < DP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rd[j]));
  };
  auto write_lds = [&](int buf, int cnt) {
    const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / XQ;
      const int ch = ci0 + (idx % XQ) * 8;
      const bool ok = row < cnt && ch < c_in;                // pairs beyond the step / channels beyond c_in: zeros
      *reinterpret_cast<bf16x8 *>(s_x + (buf * SP + row) * XLD + (idx % XQ) * 8) = ok ? rx[j] : zero;
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / (COB / 8);
      const int pc = idx % (COB / 8);
      const bool ok = row < cnt && cog + pc * 8 < c_out;
      *reinterpret_cast<bf16x8 *>(s_d + (buf * SP + row) * DLD + pc * 8) = ok ? rd[j] : zero;
    }
  };

  f32x4 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto flush = [&](int k) {
    if (cob < n_cob) {
      float *img = image0 + (int64_t)(range + k) * image_stride;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) img[((m * NB + n) * 4 + r) * 64] = acc[m][n][r];
    }
  };
  // operand fragments: lane (i16, q) of a 16-lane group addresses row 4*q + (i16 >> 2) of each 16-row half and
  // the 4-channel chunk (i16 & 3); it receives channel i16 of rows 4*q .. 4*q + 3
  const int frag_row = 4 * q + (i16 >> 2);
  const int frag_col = 4 * (i16 & 3);
  auto multiply = [&](int buf) {
    const __bf16 *bx = s_x + (buf * SP + frag_row) * XLD + frag_col;
    const __bf16 *bd = s_d + (buf * SP + frag_row) * DLD + wave * 16 * NB + frag_col;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 a[MB], b[NB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + (ks * 32) * XLD + 16 * m));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + (ks * 32 + 16) * XLD + 16 * m));
        a[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + (ks * 32) * DLD + 16 * n));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + (ks * 32 + 16) * DLD + 16 * n));
        b[n] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  };

  // cursors: A = step being multiplied, B = step whose rows are in flight, C = step whose indices are in flight
  int kA = first_offset(e_lo);
  int64_t eA = e_lo;
  int cA = step_count(kA, eA);
  int kB = kA;
  int64_t eB = eA;
  advance(kB, eB, cA);
  int cB = step_count(kB, eB);
  int kC = kB;
  int64_t eC = eB;
  advance(kC, eC, cB);
  int cC = step_count(kC, eC);

  load_idx(eA);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
  load_rows();          // rows of step A
  load_idx(eB);         // indices of step B
  zero_acc();
  int buf = 0;
  int pending = -1;     // offset whose accumulators must be flushed before the next step is multiplied
  while (cA > 0) {
    // (the flush sits at the top of the loop so that the block that issues the row loads ends with the
    // MFMAs: hipcc otherwise sinks the loads below the multiply into the block after the flush branch)
    if (pending >= 0) {
      flush(pending);
      zero_acc();
    }
    wait_rows();
    write_lds(buf, cA);   // rows of step A (requested one step ago)
    load_rows();          // rows of step B (its indices arrived with A's rows)
    load_idx(eC);         // indices of step C
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();      // step A visible; everybody is done reading the other buffer
    multiply(buf);
    pending = (cB == 0 || kB != kA) ? kA : -1;  // last step of offset kA inside this range
    buf ^= 1;
    kA = kB; eA = eB; cA = cB;
    kB = kC; eB = eC; cB = cC;
    advance(kC, eC, cC);
    cC = step_count(kC, eC);
  }
  // The last iteration requested rows and indices of steps beyond the range (asm loads the compiler knows nothing
  // about): they must have LANDED, with their destination registers still reserved, before the final flush — hipcc
  // otherwise recycles those registers for the flush's store addresses and the late data turns them into wild
  // pointers (round 3: GPU memory faults of the 128-channel variant on busy chips; round 1's <4, 1> the same way).
  wait_rows();
  if (pending >= 0) flush(pending);
}

#ifdef ME_DEBUG_VARIANTS   // bit-identical and NOT faster (profiles/r04_wgrad_ws_sweep.log): tuning build only
// =================================================================================================
// k_wgrad_bf16_ws (round 4): the same weight gradient, wave-specialised
// =================================================================================================
// MEASURED: with four row sets (one workgroup per CU) 10 - 50 % slower than k_wgrad_bf16 on every MinkUNet34C layer, with
// two sets (two workgroups per CU) within +-5 % of it; the step 11.48 -> 12.12 / 11.72 ms.  Unlike the forward kernels,
// this loop is not waiting on its own latency chain: it gathers two rows per pair (5 - 6 TB/s through the L2s).
// k_wgrad_bf16 runs its four waves in lock-step — wait for the rows, stage them, request the next, barrier, multiply:
// 3,500 cycles per 64-pair step for ~250 cycles of MFMAs per SIMD (a MinkUNet34C step spends 2.5 of its 11.6 ms here).
// As for the forward kernels (conv_bf16_ws.hip): waves 4-7 ONLY produce (pair indices DEPTH + 2 steps ahead, rows DEPTH
// steps ahead in DEPTH register sets, stage writes into the free one of the two LDS buffers), waves 0-3 ONLY multiply
// (transposed operand reads + MFMAs of k_wgrad_bf16, flushes at offset boundaries), one barrier per step.  Same pair
// ranges, same slots, same order of additions inside a range: the partial images — and grad_w after k_wgrad_reduce — are
// bit-identical to k_wgrad_bf16's.  Producer loads are plain C++ (no MFMAs in those waves for hipcc to sink them under).
template <int NB, int KSTEPS, int MB, int DEPTH>
__global__ __launch_bounds__(512, 1) void k_wgrad_bf16_ws(const __bf16 *__restrict__ x, int c_in,
                                                         const __bf16 *__restrict__ dy, int c_out,
                                                         const int32_t *__restrict__ in_pairs,
                                                         const int32_t *__restrict__ out_pairs,
                                                         const int64_t *__restrict__ koffs, int volume,
                                                         int64_t n_pairs, int n_ranges, int n_cob,
                                                         float *__restrict__ partial) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  static_assert(MB == 4 || MB == 8, "input-channel blocks per workgroup");
  static_assert(DEPTH == 2 || DEPTH == 4, "row register sets (the loop is unrolled by four)");
  constexpr int CI = 16 * MB;
  constexpr int XQ = CI / 8;
  constexpr int SP = 32 * KSTEPS;
  constexpr int COB = 64 * NB;
  constexpr int XLD = CI + kWgStepLd;
  constexpr int DLD = COB + kWgStepLd;
  constexpr int XP = SP * XQ / 256;      // 16-byte x pieces per producer thread and step
  constexpr int DP = SP * (COB / 8) / 256;
  static_assert(SP <= 64 && XP >= 1, "step size");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16 *s_x = reinterpret_cast<__bf16 *>(smem);            // [2][SP][XLD]
  __bf16 *s_d = s_x + 2 * SP * XLD;                           // [2][SP][DLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int range = blockIdx.x;
  const int ci0 = blockIdx.y * CI;
  const int cog = blockIdx.z * COB;
  const int64_t e_lo = n_pairs * range / n_ranges;
  const int64_t e_hi = n_pairs * (range + 1) / n_ranges;
  if (e_lo >= e_hi) return;                                   // whole workgroup

  // step cursor: offset k, first pair e, number of pairs cnt (0 = past the end of the range)
  struct Cur {
    int k;
    int64_t e;
    int cnt;
  };
  auto step_count = [&](int k, int64_t e) -> int {
    if (e >= e_hi) return 0;
    const int64_t kend = min(koffs[k + 1], e_hi);
    return (int)min((int64_t)SP, kend - e);
  };
  auto first_cur = [&]() {
    int lo = 0, hi = volume;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (koffs[mid] <= e_lo) lo = mid; else hi = mid;
    }
    Cur c;
    c.k = lo;
    c.e = e_lo;
    c.cnt = step_count(c.k, c.e);
    return c;
  };
  auto next_cur = [&](const Cur &c) {
    Cur n = c;
    n.e += c.cnt;
    if (n.e < e_hi) {
      while (koffs[n.k + 1] <= n.e) ++n.k;
    }
    n.cnt = step_count(n.k, n.e);
    return n;
  };

  if (wave >= 4) {
    // ------------------------------------------------ producer waves ------------------------------------------------
    const int ptid = tid - 256;
    int32_t pin[4], pout[4];              // pair indices of four steps (lane l: pair e + l)
    bf16x8 rx[DEPTH][XP], rd[DEPTH][DP];
    auto load_idx = [&](const Cur &c, int32_t &pi, int32_t &po) {
      const int64_t ec = min(c.e + lane, n_pairs - 1);        // unconditional load from a valid address
      pi = in_pairs[ec];
      po = out_pairs[ec];
    };
    auto load_rows = [&](int32_t pi, int32_t po, bf16x8 (&ax)[XP], bf16x8 (&ad)[DP]) {
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        const int idx = j * 256 + ptid;
        const int row = idx / XQ;
        const int ch = ci0 + (idx % XQ) * 8;
        const int32_t r = __shfl(pi, row, 64);
        ax[j] = *reinterpret_cast<const bf16x8 *>(x + (int64_t)r * c_in + (ch < c_in ? ch : 0));
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const int idx = j * 256 + ptid;
        const int row = idx / (COB / 8);
        const int ch = cog + (idx % (COB / 8)) * 8;
        const int32_t r = __shfl(po, row, 64);
        ad[j] = *reinterpret_cast<const bf16x8 *>(dy + (int64_t)r * c_out + (ch < c_out ? ch : 0));
      }
    };
    auto write_lds = [&](int buf, int cnt, const bf16x8 (&ax)[XP], const bf16x8 (&ad)[DP]) {
      const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        const int idx = j * 256 + ptid;
        const int row = idx / XQ;
        const int ch = ci0 + (idx % XQ) * 8;
        const bool ok = row < cnt && ch < c_in;                // pairs beyond the step / channels beyond c_in: zeros
        *reinterpret_cast<bf16x8 *>(s_x + (buf * SP + row) * XLD + (idx % XQ) * 8) = ok ? ax[j] : zero;
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const int idx = j * 256 + ptid;
        const int row = idx / (COB / 8);
        const int pc = idx % (COB / 8);
        const bool ok = row < cnt && cog + pc * 8 < c_out;
        *reinterpret_cast<bf16x8 *>(s_d + (buf * SP + row) * DLD + pc * 8) = ok ? ad[j] : zero;
      }
    };
    // cursors: cm = the step the multipliers are at, cw = the step staged next (cm + 1), ci = the step whose indices are
    // requested next (cw + DEPTH + 2)
    Cur cm = first_cur();
    Cur ci = cm;
    // prologue: indices of steps 0 .. DEPTH + 1, rows of steps 0 .. DEPTH - 1
    Cur pro[DEPTH + 2];
#pragma unroll
    for (int j = 0; j < DEPTH + 2; ++j) {
      pro[j] = ci;
      ci = next_cur(ci);
    }
#pragma unroll
    for (int j = 0; j < 4 && j < DEPTH + 2; ++j) load_idx(pro[j], pin[j], pout[j]);
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) load_rows(pin[j], pout[j], rx[j], rd[j]);
    if constexpr (DEPTH == 4) {           // (indices of steps 4 and 5 go into the sets of steps 0 and 1, whose rows are requested)
      load_idx(pro[4], pin[0], pout[0]);
      load_idx(pro[5], pin[1], pout[1]);
    }
    // produce(x), x = 4 n + X: stage step x (row set x % DEPTH) into buffer x & 1, request the rows of step x + DEPTH with
    // the indices in set (x + DEPTH) % 4 and the indices of step x + DEPTH + 2 into set (x + DEPTH + 2) % 4
    Cur cw = cm;
    auto produce = [&](auto x_) {
      constexpr int X = decltype(x_)::value;
      constexpr int RS = X % DEPTH, IA = (X + DEPTH) % 4, IB = (X + DEPTH + 2) % 4;
      write_lds(X & 1, cw.cnt, rx[RS], rd[RS]);
      load_rows(pin[IA], pout[IA], rx[RS], rd[RS]);
      load_idx(ci, pin[IB], pout[IB]);
      ci = next_cur(ci);
    };
    produce(std::integral_constant<int, 0>{});   // step 0 -> buffer 0
    __syncthreads();
#define ME_WG_STEP(XV)                               \
  if (cm.cnt == 0) break;                            \
  cw = next_cur(cm);                                 \
  produce(std::integral_constant<int, XV>{});        \
  __syncthreads();                                   \
  cm = cw;
    for (;;) {
      ME_WG_STEP(1)
      ME_WG_STEP(2)
      ME_WG_STEP(3)
      ME_WG_STEP(0)
    }
#undef ME_WG_STEP
  } else {
    // ----------------------------------------------- multiplier waves -----------------------------------------------
    __builtin_amdgcn_s_setprio(2);
    const int i16 = lane & 15, q = lane >> 4;
    const int cob = blockIdx.z * 4 + wave;                      // this wave's block of 16*NB output channels
    constexpr int kImage = MB * NB * 4 * 64;
    const int64_t image_stride = (int64_t)gridDim.y * n_cob * kImage;
    float *const image0 = partial + ((int64_t)blockIdx.y * n_cob + min(cob, n_cob - 1)) * kImage + lane;
    f32x4 acc[MB][NB];
    auto zero_acc = [&]() {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto flush = [&](int k) {
      if (cob < n_cob) {
        float *img = image0 + (int64_t)(range + k) * image_stride;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[((m * NB + n) * 4 + r) * 64] = acc[m][n][r];
      }
    };
    const int frag_row = 4 * q + (i16 >> 2);
    const int frag_col = 4 * (i16 & 3);
    auto multiply = [&](int buf) {
      const __bf16 *bx = s_x + (buf * SP + frag_row) * XLD + frag_col;
      const __bf16 *bd = s_d + (buf * SP + frag_row) * DLD + wave * 16 * NB + frag_col;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        bf16x8 a[MB], b[NB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + (ks * 32) * XLD + 16 * m));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + (ks * 32 + 16) * XLD + 16 * m));
          a[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + (ks * 32) * DLD + 16 * n));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + (ks * 32 + 16) * DLD + 16 * n));
          b[n] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    };
    Cur cA = first_cur();
    Cur cB = next_cur(cA);
    zero_acc();
    __syncthreads();                      // step 0 is staged
    int buf = 0;
    while (cA.cnt > 0) {
      multiply(buf);
      if (cB.cnt == 0 || cB.k != cA.k) {  // last step of offset cA.k inside this range
        flush(cA.k);
        zero_acc();
      }
      __syncthreads();
      buf ^= 1;
      cA = cB;
      cB = next_cur(cB);
    }
    __builtin_amdgcn_s_setprio(0);
  }
}

#endif  // ME_DEBUG_VARIANTS (k_wgrad_bf16_ws)

// =================================================================================================
// wgrad of fp32 rows on the bf16 matrix pipe (round 2): k_wgrad_f32x3
// =================================================================================================
// k_wgrad_f32 runs at 0.68 of the fp32 MFMA peak (107 TF at config 2) — with 5 TB/s of gathers it LOOKED bandwidth
// bound, but sending ranges that read the same dy rows to the same XCD (WgRangeOrder) cut its HBM-side traffic by
// 20 % and its duration by nothing: the fp32 MFMA, which blocks the SIMD for the loop's address arithmetic and
// loads, is the limit.  Here the fp32 rows are split exactly into three bf16 terms (conv_f32x3.hip) while they are
// written to LDS and the products are rebuilt from six v_mfma_f32_16x16x32_bf16: structure of k_wgrad_bf16 (rows
// staged once per workgroup, operands read back transposed with ds_read_b64_tr_b16, equal pair ranges, register
// image partials, ordered reduction), one LDS buffer of three planes per operand and two barriers per 32-pair step
// so that three workgroups fit a CU and cover each other's split / store phases.
template <int NB>
__global__ __launch_bounds__(256, 3) void k_wgrad_f32x3(const float *__restrict__ x, int c_in,
                                                       const float *__restrict__ dy, int c_out,
                                                       const int32_t *__restrict__ in_pairs,
                                                       const int32_t *__restrict__ out_pairs,
                                                       const int64_t *__restrict__ koffs, int volume,
                                                       int64_t n_pairs, int n_ranges, int n_cob,
                                                       float *__restrict__ partial, const WgRangeOrder order) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  constexpr int MB = 4;
  constexpr int SP = 32;                 // pairs per step = the K of one MFMA
  constexpr int COB = 64 * NB;           // output channels per workgroup
  constexpr int XLD = 64 + kWgStepLd;    // elements
  constexpr int DLD = COB + kWgStepLd;
  constexpr int XPL = SP * XLD, DPL = SP * DLD;   // elements of one plane
  constexpr int XP = SP * 8 / 256;       // 8-channel x pieces (32 bytes of fp32) per thread and step
  constexpr int DP = SP * (COB / 8) / 256;
  static_assert(XP == 1 && DP >= 1, "step size");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16 *s_x = reinterpret_cast<__bf16 *>(smem);            // [3][SP][XLD]
  __bf16 *s_d = s_x + 3 * XPL;                                // [3][SP][DLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int range = order.n > 0 ? (int)order.v[blockIdx.x] : (int)blockIdx.x;
  const int ci0 = blockIdx.y * 64;
  const int cog = blockIdx.z * COB;                           // first output channel of the workgroup
  const int cob = blockIdx.z * 4 + wave;                      // this wave's block of 16*NB output channels
  const int64_t e_lo = n_pairs * range / n_ranges;
  const int64_t e_hi = n_pairs * (range + 1) / n_ranges;
  if (e_lo >= e_hi) return;                                   // whole workgroup

  constexpr int kImage = MB * NB * 4 * 64;
  const int64_t image_stride = (int64_t)gridDim.y * n_cob * kImage;
  float *const image0 = partial + ((int64_t)blockIdx.y * n_cob + min(cob, n_cob - 1)) * kImage + lane;

  auto first_offset = [&](int64_t e) {
    int lo = 0, hi = volume;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (koffs[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  };
  auto step_count = [&](int k, int64_t e) -> int {
    if (e >= e_hi) return 0;
    const int64_t kend = min(koffs[k + 1], e_hi);
    return (int)min((int64_t)SP, kend - e);
  };
  auto advance = [&](int &k, int64_t &e, int cnt) {
    e += cnt;
    if (e < e_hi) {
      while (koffs[k + 1] <= e) ++k;
    }
  };

  int32_t pin = 0, pout = 0;            // pair indices of the step whose rows are loaded next (lane l: pair e + l)
  f32x4 rx[XP][2], rd[DP][2];
  auto load_idx = [&](int64_t e) {
    const int64_t ec = min(e + lane, n_pairs - 1);            // unconditional load from a valid address
    const int32_t *pi = in_pairs + ec, *po = out_pairs + ec;
    asm volatile("global_load_dword %0, %1, off" : "=v"(pin) : "v"(pi) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(pout) : "v"(po) : "memory");
  };
  // (inline asm as in k_wgrad_bf16: hipcc sinks ordinary loads whose first use is in the next iteration below the
  // MFMAs; the matching s_waitcnt is wait_rows)
  auto load_rows = [&]() {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx >> 3;
      const int ch = ci0 + (idx & 7) * 8;
      const int32_t r = __shfl(pin, row, 64);
      const float *p = x + (int64_t)r * c_in + (ch < c_in ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rx[j][0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rx[j][1]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / (COB / 8);
      const int ch = cog + (idx % (COB / 8)) * 8;
      const int32_t r = __shfl(pout, row, 64);
      const float *p = dy + (int64_t)r * c_out + (ch < c_out ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rd[j][0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rd[j][1]) : "v"(p) : "memory");
    }
  };
  auto wait_rows = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
#pragma unroll
    for (int j = 0; j < XP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rx[j][0]), "+v"(rx[j][1]));
#pragma unroll
    for (int j = 0; j < DP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rd[j][0]), "+v"(rd[j][1]));
  };
  auto write_lds = [&](int cnt) {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    // non-finite rows (conv_common.hpp): x is the row side of the rule, dy the weight side; FIX = the exact, rare pass
    auto store_step = [&](auto fix) {
      constexpr bool FIX = decltype(fix)::value;
      uint32_t flag = 0u;
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        const int idx = j * 256 + tid;
        const int row = idx >> 3;
        const int ch = ci0 + (idx & 7) * 8;
        const bool ok = row < cnt && ch < c_in;                // pairs beyond the step / channels beyond c_in: zeros
        u32x4 p1, p2, p3;
        if (FIX) split3_fix<false>(rx[j][0], rx[j][1], p1, p2, p3);
        else split3_raw(rx[j][0], rx[j][1], p1, p2, p3);
        if (ok) flag = split3_flag(flag, p3);
        __bf16 *o = s_x + row * XLD + (idx & 7) * 8;
        *reinterpret_cast<u32x4 *>(o) = ok ? p1 : zero;
        *reinterpret_cast<u32x4 *>(o + XPL) = ok ? p2 : zero;
        *reinterpret_cast<u32x4 *>(o + 2 * XPL) = ok ? p3 : zero;
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const int idx = j * 256 + tid;
        const int row = idx / (COB / 8);
        const int pc = idx % (COB / 8);
        const bool ok = row < cnt && cog + pc * 8 < c_out;
        u32x4 p1, p2, p3;
        if (FIX) split3_fix<true>(rd[j][0], rd[j][1], p1, p2, p3);
        else split3_raw(rd[j][0], rd[j][1], p1, p2, p3);
        if (ok) flag = split3_flag(flag, p3);
        __bf16 *o = s_d + row * DLD + pc * 8;
        *reinterpret_cast<u32x4 *>(o) = ok ? p1 : zero;
        *reinterpret_cast<u32x4 *>(o + DPL) = ok ? p2 : zero;
        *reinterpret_cast<u32x4 *>(o + 2 * DPL) = ok ? p3 : zero;
      }
      return flag;
    };
    const uint32_t flag = store_step(std::false_type{});
    if (__builtin_expect(__any(split3_suspect(flag)), 0)) store_step(std::true_type{});
  };

  f32x4 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto flush = [&](int k) {
    if (cob < n_cob) {
      float *img = image0 + (int64_t)(range + k) * image_stride;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) img[((m * NB + n) * 4 + r) * 64] = acc[m][n][r];
    }
  };
  // operand fragments as in k_wgrad_bf16: lane (i16, q) addresses row 4*q + (i16 >> 2) of each 16-row half and the
  // 4-channel chunk (i16 & 3); it receives channel i16 of rows 4*q .. 4*q + 3
  const int frag_row = 4 * q + (i16 >> 2);
  const int frag_col = 4 * (i16 & 3);
  auto multiply = [&]() {
    const __bf16 *bx = s_x + frag_row * XLD + frag_col;
    const __bf16 *bd = s_d + frag_row * DLD + wave * 16 * NB + frag_col;
    bf16x8 a[3][MB], b[3][NB];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + pl * XPL + 16 * m));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bx + pl * XPL + 16 * XLD + 16 * m));
        a[pl][m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + pl * DPL + 16 * n));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bd + pl * DPL + 16 * DLD + 16 * n));
        b[pl][n] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
    // (x plane, dy plane) by ascending magnitude: 2^-16 terms, 2^-8 terms, leading term
    constexpr int XPI[6] = {2, 0, 1, 1, 0, 0};
    constexpr int DPI[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[XPI[t]][m], b[DPI[t]][n], acc[m][n], 0, 0, 0);
  };

  // cursors: A = step being multiplied, B = step whose rows are in flight, C = step whose indices are in flight
  int kA = first_offset(e_lo);
  int64_t eA = e_lo;
  int cA = step_count(kA, eA);
  int kB = kA;
  int64_t eB = eA;
  advance(kB, eB, cA);
  int cB = step_count(kB, eB);
  int kC = kB;
  int64_t eC = eB;
  advance(kC, eC, cB);
  int cC = step_count(kC, eC);

  load_idx(eA);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
  load_rows();          // rows of step A
  load_idx(eB);         // indices of step B
  zero_acc();
  int pending = -1;     // offset whose accumulators must be flushed before the next step is multiplied
  while (cA > 0) {
    if (pending >= 0) {
      flush(pending);
      zero_acc();
    }
    wait_rows();
    __syncthreads();      // everybody is done reading the previous step
    write_lds(cA);        // rows of step A (requested one step ago), split into the three planes
    load_rows();          // rows of step B (its indices arrived with A's rows)
    load_idx(eC);         // indices of step C
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();      // step A visible
    multiply();
    pending = (cB == 0 || kB != kA) ? kA : -1;  // last step of offset kA inside this range
    kA = kB; eA = eB; cA = cB;
    kB = kC; eB = eC; cB = cC;
    advance(kC, eC, cC);
    cC = step_count(kC, eC);
  }
  // The last iteration requested rows and indices of steps beyond the range (asm loads the compiler knows nothing
  // about): they must have LANDED, with their destination registers still reserved, before the final flush — hipcc
  // otherwise recycles those registers for the flush's store addresses and the late data turns them into wild
  // pointers (round 3: GPU memory faults of the 128-channel variant on busy chips; round 1's <4, 1> the same way).
  wait_rows();
  if (pending >= 0) flush(pending);
}

#ifdef ME_DEBUG_VARIANTS   // (measured at the speed of k_wgrad_f32, never the default: tuning build only)
// The same staged design in fp32 (v_mfma_f32_16x16x4_f32, exact fp32): rows are gathered ONCE per workgroup with
// 16-byte loads and shared by the four waves through LDS, so the matrix pipe is fed by one conflict-free
// ds_read_b32 per operand instead of one global gather per wave — k_wgrad_f32 above re-gathers every x row in
// each of its waves and tops out at 54 % of the fp32 MFMA peak.  Used when both channel counts are multiples
// of 4; register images in the layout of k_wgrad_bf16 (k_wgrad_reduce<NB, true>).
template <int NB, int KSTEPS>
__global__ __launch_bounds__(256, 2) void k_wgrad_lds_f32(const float *__restrict__ x, int c_in,
                                                         const float *__restrict__ dy, int c_out,
                                                      const int32_t *__restrict__ in_pairs,
                                                      const int32_t *__restrict__ out_pairs,
                                                      const int64_t *__restrict__ koffs, int volume,
                                                      int64_t n_pairs, int n_ranges, int n_cob,
                                                      float *__restrict__ partial) {
  constexpr int MB = 4;
  constexpr int SP = 16 * KSTEPS;        // pairs per step (<= 64: one index register per wave)
  constexpr int COB = 64 * NB;           // output channels per workgroup
  constexpr int XLD = 64 + kWgStepLd;    // elements
  constexpr int DLD = COB + kWgStepLd;
  constexpr int XP = SP * 16 / 256;      // 16-byte x pieces per thread and step
  constexpr int DP = SP * (COB / 4) / 256;
  static_assert(SP <= 64 && XP >= 1, "step size");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_x = reinterpret_cast<float *>(smem);              // [2][SP][XLD]
  float *s_d = s_x + 2 * SP * XLD;                            // [2][SP][DLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int range = blockIdx.x;
  const int ci0 = blockIdx.y * 64;
  const int cog = blockIdx.z * COB;                           // first output channel of the workgroup
  const int cob = blockIdx.z * 4 + wave;                      // this wave's block of 16*NB output channels
  const int64_t e_lo = n_pairs * range / n_ranges;
  const int64_t e_hi = n_pairs * (range + 1) / n_ranges;
  if (e_lo >= e_hi) return;                                   // whole workgroup

  constexpr int kImage = MB * NB * 4 * 64;
  const int64_t image_stride = (int64_t)gridDim.y * n_cob * kImage;
  float *const image0 = partial + ((int64_t)blockIdx.y * n_cob + min(cob, n_cob - 1)) * kImage + lane;

  // step cursor: offset k, first pair e, number of pairs cnt (0 = past the end of the range)
  auto first_offset = [&](int64_t e) {
    int lo = 0, hi = volume;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (koffs[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  };
  auto step_count = [&](int k, int64_t e) -> int {
    if (e >= e_hi) return 0;
    const int64_t kend = min(koffs[k + 1], e_hi);
    return (int)min((int64_t)SP, kend - e);
  };
  auto advance = [&](int &k, int64_t &e, int cnt) {
    e += cnt;
    if (e < e_hi) {
      while (koffs[k + 1] <= e) ++k;
    }
  };

  // this thread's pieces: x piece j is row (j*256 + tid) >> 3, channels ci0 + ((j*256 + tid) & 7) * 8
  int32_t pin = 0, pout = 0;            // pair indices of the step whose rows are loaded next (lane l: pair e + l)
  f32x4 rx[XP], rd[DP];
  auto load_idx = [&](int64_t e) {
    const int64_t ec = min(e + lane, n_pairs - 1);            // unconditional load from a valid address
    const int32_t *pi = in_pairs + ec, *po = out_pairs + ec;
    asm volatile("global_load_dword %0, %1, off" : "=v"(pin) : "v"(pi) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(pout) : "v"(po) : "memory");
  };
  // The row loads are inline asm: hipcc sinks ordinary loads whose first use is in the next iteration below
  // the MFMAs (into the last block before the back edge), which serialises gather and multiply.  asm volatile
  // keeps them where they are written; the matching s_waitcnt is issued by hand (wait_rows) right before the
  // registers are stored to LDS one iteration later.
  auto load_rows = [&]() {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx >> 4;
      const int ch = ci0 + (idx & 15) * 4;
      const int32_t r = __shfl(pin, row, 64);
      const float *p = x + (int64_t)r * c_in + (ch < c_in ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rx[j]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / (COB / 4);
      const int ch = cog + (idx % (COB / 4)) * 4;
      const int32_t r = __shfl(pout, row, 64);
      const float *p = dy + (int64_t)r * c_out + (ch < c_out ? ch : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rd[j]) : "v"(p) : "memory");
    }
  };
  auto wait_rows = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
#pragma unroll
    for (int j = 0; j < XP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rx[j]));
#pragma unroll
    for (int j = 0; j < DP; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rd[j]));
  };
  auto write_lds = [&](int buf, int cnt) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx >> 4;
      const int ch = ci0 + (idx & 15) * 4;
      const bool ok = row < cnt && ch < c_in;                // pairs beyond the step / channels beyond c_in: zeros
      *reinterpret_cast<f32x4 *>(s_x + (buf * SP + row) * XLD + (idx & 15) * 4) = ok ? rx[j] : zero;
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const int idx = j * 256 + tid;
      const int row = idx / (COB / 4);
      const int pc = idx % (COB / 4);
      const bool ok = row < cnt && cog + pc * 4 < c_out;
      *reinterpret_cast<f32x4 *>(s_d + (buf * SP + row) * DLD + pc * 4) = ok ? rd[j] : zero;
    }
  };

  f32x4 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto flush = [&](int k) {
    if (cob < n_cob) {
      float *img = image0 + (int64_t)(range + k) * image_stride;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) img[((m * NB + n) * 4 + r) * 64] = acc[m][n][r];
    }
  };
  // v_mfma_f32_16x16x4_f32: lane (i16, q) supplies x[pair 4*s + q][16*m + i16] and dy[pair 4*s + q][16*n + i16]:
  // one ds_read_b32 per operand, 16 consecutive floats per 16-lane group, row stride = 16 mod 32 banks
  auto multiply = [&](int buf) {
    const float *bx = s_x + (buf * SP + q) * XLD + i16;
    const float *bd = s_d + (buf * SP + q) * DLD + wave * 16 * NB + i16;
#pragma unroll
    for (int s = 0; s < SP / 4; ++s) {
      float a[MB], b[NB];
#pragma unroll
      for (int m = 0; m < MB; ++m) a[m] = bx[(4 * s) * XLD + 16 * m];
#pragma unroll
      for (int n = 0; n < NB; ++n) b[n] = bd[(4 * s) * DLD + 16 * n];
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  };

  // cursors: A = step being multiplied, B = step whose rows are in flight, C = step whose indices are in flight
  int kA = first_offset(e_lo);
  int64_t eA = e_lo;
  int cA = step_count(kA, eA);
  int kB = kA;
  int64_t eB = eA;
  advance(kB, eB, cA);
  int cB = step_count(kB, eB);
  int kC = kB;
  int64_t eC = eB;
  advance(kC, eC, cB);
  int cC = step_count(kC, eC);

  load_idx(eA);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin), "+v"(pout));
  load_rows();          // rows of step A
  load_idx(eB);         // indices of step B
  zero_acc();
  int buf = 0;
  int pending = -1;     // offset whose accumulators must be flushed before the next step is multiplied
  while (cA > 0) {
    // (the flush sits at the top of the loop so that the block that issues the row loads ends with the
    // MFMAs: hipcc otherwise sinks the loads below the multiply into the block after the flush branch)
    if (pending >= 0) {
      flush(pending);
      zero_acc();
    }
    wait_rows();
    write_lds(buf, cA);   // rows of step A (requested one step ago)
    load_rows();          // rows of step B (its indices arrived with A's rows)
    load_idx(eC);         // indices of step C
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();      // step A visible; everybody is done reading the other buffer
    multiply(buf);
    pending = (cB == 0 || kB != kA) ? kA : -1;  // last step of offset kA inside this range
    buf ^= 1;
    kA = kB; eA = eB; cA = cB;
    kB = kC; eB = eC; cB = cC;
    advance(kC, eC, cC);
    cC = step_count(kC, eC);
  }
  // The last iteration requested rows and indices of steps beyond the range (asm loads the compiler knows nothing
  // about): they must have LANDED, with their destination registers still reserved, before the final flush — hipcc
  // otherwise recycles those registers for the flush's store addresses and the late data turns them into wild
  // pointers (round 3: GPU memory faults of the 128-channel variant on busy chips; round 1's <4, 1> the same way).
  wait_rows();
  if (pending >= 0) flush(pending);
}
#endif  // ME_DEBUG_VARIANTS

// grad_w[k] = sum of the register images of the ranges that touch offset k, written back through the
// channel interleave of k_wgrad_f32.  A block sums 64 consecutive image elements; its four waves take
// the slots s = first + phase, + 4, ... (four independent loads in flight per thread: the loop is
// latency-bound otherwise) and the four partial sums are combined in phase order, so the summation
// order is fixed (bitwise reproducible).

// grad_w[k] = sum of the register images of the ranges that touch offset k, written back through the
// channel interleave of k_wgrad_f32.  A block sums 64 consecutive image elements; its four waves take
// the slots s = first + phase, + 4, ... (four independent loads in flight per thread: the loop is
// latency-bound otherwise) and the four partial sums are combined in phase order, so the summation
// order is fixed (bitwise reproducible).
// TR: the images come from k_wgrad_bf16 (MFMA row i of block m <-> input channel 16*m + i, column j of block
// n <-> output channel 16*n + j) instead of k_wgrad_f32's interleave.
constexpr int kWgReducePhases = 4;   // waves per block of k_wgrad_reduce (16 measured slower: most offsets own ~10
                                     // slots, and 4x the waves cost more than the long chain of the centre offset saves)

template <int NB, bool TR, int MB = kWgMB>
__global__ __launch_bounds__(64 * kWgReducePhases) void k_wgrad_reduce(const float *__restrict__ partial,
                                                     const int64_t *__restrict__ koffs, int volume,
                                                     int64_t n_pairs, int n_ranges, int n_cib, int n_cob,
                                                     int c_in, int c_out, float *__restrict__ grad_w) {
  // Round 3: a thread owns FOUR consecutive floats of the register image (one 16-byte load per slot) and a block 256 of
  // them: a quarter of the blocks, waves and load instructions of the one-float version, which spent its time
  // dispatching 110k four-instruction waves on a 256 x 256 layer (33 us for 23 MB of partials, longer than the
  // weight-gradient kernel it follows: profiles/r03_pmc_bf16_layers.log).  The sums are the same sums in the same
  // order (per element: slots of a phase in ascending order, then the phases in order): bit-identical results.
  static_assert(MB == 4 || MB == 8, "input-channel blocks per workgroup");
  constexpr int kImage = MB * NB * 4 * 64;
  constexpr int PH = kWgReducePhases;
  __shared__ int64_t s_r[2];
  __shared__ f32x4 s_part[PH][64];
  const int k = blockIdx.y;
  const int64_t b = koffs[k], e = koffs[k + 1];
  if (threadIdx.x == 0) {
    s_r[0] = 0;
    s_r[1] = -1;
    if (e > b) {
      s_r[0] = wgrad_range_of_pair(b, n_pairs, n_ranges) + k;
      s_r[1] = wgrad_range_of_pair(e - 1, n_pairs, n_ranges) + k;
    }
  }
  __syncthreads();
  // the waves take the slots first + phase, + PH, ...; eight independent loads in flight per thread (the centre
  // offset of a sparse map owns half of all slots: its block sets the kernel's duration, and the loop is latency-bound)
  const int j = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int64_t idx = ((int64_t)blockIdx.x * 64 + j) * 4;      // first of this thread's four floats (kImage % 256 == 0)
  const int64_t per_slot = (int64_t)n_cib * n_cob * kImage;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (idx < per_slot) {
    const int64_t first = s_r[0], last = s_r[1];
    const float *p = partial + idx;
    int64_t slot = first + phase;
    for (; slot + 7 * PH <= last; slot += 8 * PH) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (slot + u * PH) * per_slot);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; slot <= last; slot += PH) s += *reinterpret_cast<const f32x4 *>(p + slot * per_slot);
  }
  s_part[phase][j] = s;
  __syncthreads();
  if (phase != 0 || idx >= per_slot) return;
  s = s_part[0][j];
#pragma unroll
  for (int ph = 1; ph < PH; ++ph) s += s_part[ph][j];   // fixed order: bitwise reproducible
  // the four floats are four consecutive LANES of one register of the image (idx % 4 == 0, 64 lanes per register)
  const int lane0 = (int)(idx % 64);
  const int reg = (int)((idx / 64) % (MB * NB * 4));
  const int cob = (int)((idx / kImage) % n_cob);
  const int cib = (int)(idx / ((int64_t)kImage * n_cob));
  const int r = reg % 4, n = (reg / 4) % NB, m = reg / (4 * NB);
  const int q = lane0 >> 4;
  const int ci = cib * (16 * MB) + (TR ? 16 * m + 4 * q + r : MB * (4 * q + r) + m);
  if (ci >= c_in) return;
  float *row = grad_w + ((int64_t)k * c_in + ci) * c_out;
  if (TR) {
    // co = cob * 16 * NB + 16 * n + i16 with i16 = lane & 15: four consecutive output channels
    const int co = cob * (16 * NB) + 16 * n + (lane0 & 15);
    if ((c_out % 4) == 0 && co + 3 < c_out) {
      *reinterpret_cast<f32x4 *>(row + co) = s;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (co + t < c_out) row[co + t] = s[t];
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int co = cob * (16 * NB) + NB * ((lane0 + t) & 15) + n;
      if (co < c_out) row[co] = s[t];
    }
  }
}

// =================================================================================================
// naive cross-check kernels (VALU + global atomics on the pair lists)
// =================================================================================================
__device__ __forceinline__ int locate_offset(const int64_t *__restrict__ koffs, int volume, int64_t e) {
  int lo = 0, hi = volume;  // largest k with koffs[k] <= e
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (koffs[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_naive_forward(const float *__restrict__ in_feat, int c_in,
                                                      const float *__restrict__ w, int c_out,
                                                      const int32_t *__restrict__ in_pairs,
                                                      const int32_t *__restrict__ out_pairs,
                                                      const int64_t *__restrict__ koffs, int volume,
                                                      int64_t n_pairs, float *out_feat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_pairs * c_out) return;
  const int64_t e = t / c_out;
  const int j = (int)(t % c_out);
  const int k = locate_offset(koffs, volume, e);
  const float *xr = in_feat + (int64_t)in_pairs[e] * c_in;
  const float *wk = w + (int64_t)k * c_in * c_out + j;
  float s = 0.f;
  for (int i = 0; i < c_in; ++i) s = fmaf(xr[i], wk[(int64_t)i * c_out], s);
  atomicAdd(&out_feat[(int64_t)out_pairs[e] * c_out + j], s);
}

__global__ __launch_bounds__(256) void k_naive_dgrad(const float *__restrict__ grad_out, int c_out,
                                                    const float *__restrict__ w, int c_in,
                                                    const int32_t *__restrict__ in_pairs,
                                                    const int32_t *__restrict__ out_pairs,
                                                    const int64_t *__restrict__ koffs, int volume,
                                                    int64_t n_pairs, float *grad_in) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_pairs * c_in) return;
  const int64_t e = t / c_in;
  const int i = (int)(t % c_in);
  const int k = locate_offset(koffs, volume, e);
  const float *gr = grad_out + (int64_t)out_pairs[e] * c_out;
  const float *wk = w + ((int64_t)k * c_in + i) * c_out;
  float s = 0.f;
  for (int j = 0; j < c_out; ++j) s = fmaf(gr[j], wk[j], s);
  atomicAdd(&grad_in[(int64_t)in_pairs[e] * c_in + i], s);
}

__global__ __launch_bounds__(256) void k_naive_wgrad(const float *__restrict__ in_feat, int c_in,
                                                    const float *__restrict__ grad_out, int c_out,
                                                    const int32_t *__restrict__ in_pairs,
                                                    const int32_t *__restrict__ out_pairs,
                                                    const int64_t *__restrict__ koffs, int volume,
                                                    float *__restrict__ grad_w) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t cc = (int64_t)c_in * c_out;
  if (t >= (int64_t)volume * cc) return;
  const int k = (int)(t / cc);
  const int i = (int)((t % cc) / c_out), j = (int)(t % c_out);
  float s = 0.f;
  for (int64_t e = koffs[k]; e < koffs[k + 1]; ++e)
    s = fmaf(in_feat[(int64_t)in_pairs[e] * c_in + i], grad_out[(int64_t)out_pairs[e] * c_out + j], s);
  grad_w[t] = s;
}

// kernel variant for a (c_src, c_dst) problem
struct ConvVariant {
  int nc;     // output columns per workgroup (64 or 32); waves per workgroup = nc / 16
  int slabs;  // column slabs (grid.y)
  int kc;     // source-channel chunk (64, 32 or 16)
};

extern int g_conv_variant;

static ConvVariant conv_variant(int c_src, int c_dst) {
  ConvVariant v;
  const bool allow96 = g_conv_variant != 9;  // (variant 9: the 32-wide passes that 96 channels used to take)
  // columns per workgroup: 64 unless the last 64-column slab would be at most half full; 96-channel layers
  // (MinkUNet's decoder) get ONE 96-column slab of six waves instead of three 32-column passes over the same rows
  const int rem = c_dst % 64;
  v.nc = (allow96 && c_dst % 96 == 0 && rem != 0) ? 96 : ((rem != 0 && rem <= 32) ? 32 : 64);
  v.slabs = (int)ceil_div(c_dst, v.nc);
  // source-channel chunk: the largest of {96, 64, 32, 16} that adds at most 16 channels of padding
  if (allow96 && c_src % 96 == 0 && c_src % 64 != 0) v.kc = 96;
  else if (c_src % 64 == 0) v.kc = 64;
  else if (c_src % 32 == 0) v.kc = 32;
  else if (c_src <= 16) v.kc = 16;
  else if (c_src <= 32) v.kc = 32;
  else v.kc = (align_up(c_src, 64) - c_src <= 16) ? 64 : ((align_up(c_src, 32) - c_src <= 16) ? 32 : 16);
  return v;
}

// Tile height for a (target rows, kernel shape) problem, shared by the fp32 and bf16 kernels: tiles x column
// slabs just below a multiple of the GPU's resident-workgroup slots, the accumulator tile + stage buffer within
// the LDS of `occ` resident workgroups, density steering the trade-off against the 16-row group padding.
// two barriers, the stage write and the exposed part of the loads of one batch (phase timing,
// profiles/r01_phase_timing_conv_v7.log: ~2150 cycles); with 400 the model preferred many small tiles on
// sparse maps, where T = 256 measured 12-21 % faster at config 5
constexpr double kBatchOverheadCycles = 2000.0;

int plan_tile_rows(const PlanShape &s, int64_t n_tgt, int64_t volume, int64_t n_pairs) {
  const int waves = s.nc / 16;
  const int cus = device_cu_count();
  // occupancy p of a neighbour offset (centre excluded) -> expected 16-row groups of a (tile, k) item
  const double p = volume > 1 ? (double)(n_pairs > n_tgt ? n_pairs - n_tgt : 0) / ((double)(volume - 1) * n_tgt)
                              : 0.0;
  const int cap = ME_MAX_BATCH_GROUPS;
  double best_cost = 1e300;
  int best_t = 128;
  // resident workgroups per CU: 3 waves per SIMD by registers (__launch_bounds__(NC * 4, 3)), then the LDS
  int occ_max = (s.wave_slots > 0 ? s.wave_slots : 12) / waves;
  if (s.max_occ > 0 && s.max_occ < occ_max) occ_max = s.max_occ;
  if (occ_max < 1) occ_max = 1;
  for (int occ = occ_max; occ >= 1; --occ) {
    const int64_t slots = (int64_t)cus * occ;
    for (int rounds = 1; rounds <= 64; ++rounds) {
      int64_t t = ceil_div(n_tgt * s.slabs, slots * rounds);
      if (t < ME_GROUP_ROWS) t = ME_GROUP_ROWS;
      // (the tallest tile is a candidate of its own: on sparse maps the per-offset cost of a tile hardly grows
      // with its height, so fewer, taller tiles win even when they do not fill the last round)
      if (t > ME_MAX_TILE_ROWS) t = ME_MAX_TILE_ROWS;
      const int64_t lds = (t + 1) * (s.nc + kAccPad) * 4 + (int64_t)cap * 16 * s.stage_row_bytes;
      if (lds * occ > kLdsBudget) continue;
      const double m = (double)t * p;  // expected entries of an off-centre item
      const double g_side = m < 6.0 ? (m <= 0 ? 0.0 : (1.0 - exp(-m)) * (1.0 + m / 16.0)) : m / 16.0 + 0.5;
      const double groups = (double)ceil_div(t, 16) + (double)(volume - 1) * g_side;
      const double batches = (double)ceil_div(ceil_div(t, 16), cap) +
                             (double)(volume - 1) * (m <= 0 ? 0.0 : (m < 3.0 ? 1.0 - exp(-m) : ceil(g_side / cap)));
      const int64_t items = ceil_div(n_tgt, t) * s.slabs;
      // workgroups are dispatched as slots free up: half way between whole rounds and perfect packing
      const double real_rounds = 0.5 * (double)ceil_div(items, slots) + 0.5 * (double)items / (double)slots;
      // cycles of one tile if its waves had their SIMDs alone: per-group work + per-batch barrier / pipeline
      // overhead + store / pipeline fill per tile; occ * waves / 4 waves share a SIMD
      const double tile_cycles =
          s.chunks * (groups * s.group_cycles + batches * kBatchOverheadCycles) + (double)t * s.nc * 0.4 + 3000.0;
      const double cost = real_rounds * tile_cycles * (double)(occ * waves) / 4.0 *
                          (1.0 + 0.15 * 12.0 / (occ * waves));
      if (cost < best_cost) {
        best_cost = cost;
        best_t = (int)t;
      }
      if (t == ME_GROUP_ROWS) break;
    }
  }
  return best_t;
}

int g_conv_variant = 0;  // me_debug_set_conv_variant
unsigned long long *g_conv_timing = nullptr;  // me_debug_conv_timing: 8 counters (VAR & 256 builds)

template <int NC, int KC, int VAR>
static int launch_conv_tile(const float *src, int c_src, const float *wp, int c_dst, int slabs,
                            const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                            const int32_t *tile_bptr, const int32_t *order, float *dst, int64_t n_tgt,
                            int tile_rows, int batch_groups, hipStream_t stream, bool small = false, bool fuse = false) {
  // variants 2048 + n (experiment): n KiB of unused LDS, to cap the resident workgroups per CU
  const int lds = conv_lds_bytes(NC, KC, tile_rows, batch_groups) +
                  (g_conv_variant >= 2048 && g_conv_variant < 3000 ? (g_conv_variant - 2048) * 1024 : 0);
  ME_CHECK(lds <= kLdsBudget, "tile_rows / batch_groups too large for the LDS of one workgroup");
  const bool exact = (c_src % KC) == 0;
  // the timing ablations (VAR != 0) exist with 64-bit addresses only
  constexpr bool kHasSmall = VAR == 0;
  small = small && kHasSmall;
  typedef void (*kernel_t)(const float *, int, const f32x4 *, int, const int32_t *, const int32_t *, const int32_t *,
                           const int32_t *, const int32_t *, float *, int64_t, int, int);
  // two instantiations per shape (round 6; four before): the fast one (exact chunk AND 32-bit gather offsets) and the
  // general one, which also serves the two mixed cases — same sums in the same order
  kernel_t fn = (small && exact) ? &k_conv_tile_f32<NC, KC, true, VAR, kHasSmall> : &k_conv_tile_f32<NC, KC, false, VAR, false>;
  // multi-offset batches (sparse maps; host policy): narrow shapes with the small-address path only — wide layers
  // run the split kernels, and the weights of four offsets have to fit the registers
  constexpr bool kHasFuse = VAR == 0 && NC <= 64 && KC <= 64;
  if constexpr (kHasFuse) {
    if (fuse && small) fn = exact ? &k_conv_tile_f32<NC, KC, true, 0, true, true> : &k_conv_tile_f32<NC, KC, false, 0, true, true>;
    else fuse = false;
  } else {
    fuse = false;
  }
  static bool attr_set[6] = {false, false, false, false, false, false};  // per instantiation
  const int which = fuse ? 4 + (exact ? 1 : 0) : (small ? 2 : 0) + (exact ? 1 : 0);
  if (lds > 48 * 1024 && !attr_set[which]) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kLdsBudget));
    attr_set[which] = true;
  }
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)slabs);
  hipLaunchKernelGGL(fn, grid, dim3(NC * 4), (size_t)lds, stream, src, c_src, reinterpret_cast<const f32x4 *>(wp), c_dst,
                     plan_src, plan_dst, batch_desc, tile_bptr, order, dst, n_tgt, tile_rows, batch_groups);
  ME_LAUNCH_CHECK();
  return 0;
}

template <int NC, int CBW, int KCH>
static int launch_conv_tile_dma(const float *src, int c_src, const float *wp, int c_dst, const int32_t *plan_src,
                                const int32_t *plan_dst, const int32_t *batch_desc, const int32_t *tile_bptr,
                                const int32_t *order, float *dst, int64_t n_tgt, int tile_rows, int batch_groups,
                                hipStream_t stream) {
  const int lds = conv_dma_lds_bytes(NC, KCH, tile_rows, batch_groups);
  ME_CHECK(lds <= kLdsBudget, "tile_rows / batch_groups too large for the LDS of one workgroup (LDS-DMA kernel)");
  typedef void (*kernel_t)(const float *, int, const f32x4 *, int, const int32_t *, const int32_t *, const int32_t *,
                           const int32_t *, const int32_t *, float *, int64_t, int, int);
  // timing ablations (variant = family * 100 + code; instantiated for the <128, 2, 1> shape only): 16 no gather,
  // 2 weights of the first batch reused, 4 no barrier, 32 no operand reads, 64 no accumulator read / write,
  // 99 all of them, 98 prologue + epilogue only
  kernel_t fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 0>;
  const int code = g_conv_variant % 100;
  if (code == 16) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 16>;
  if constexpr (NC == 128 && CBW == 2) {
    if (code == 2) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 2>;
    if (code == 4) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 4>;
    if (code == 32) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 32>;
    if (code == 64) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 64>;
    if (code == 99) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 16 + 2 + 4 + 32 + 64>;
    if (code == 98) fn = &k_conv_tile_dma_f32<NC, CBW, KCH, 512>;
  }
  ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                             kLdsBudget));
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)(c_dst / NC));
  hipLaunchKernelGGL(fn, grid, dim3(NC / CBW * 4), (size_t)lds, stream, src, c_src,
                     reinterpret_cast<const f32x4 *>(wp), c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                     n_tgt, tile_rows, batch_groups);
  ME_LAUNCH_CHECK();
  return 0;
}

// LDS-DMA instantiation for a (c_src, c_dst) problem: columns per workgroup, column blocks per wave, chunks per batch
struct ConvDmaShape {
  int nc, cbw, kch;   // nc == 0: not eligible
};
static ConvDmaShape conv_dma_shape(int c_src, int c_dst) {
  ConvDmaShape z{0, 0, 0};
  if (g_conv_variant < 3000 || g_conv_variant >= 3400) return z;     // (experiment switch; see me_conv_target_f32)
  if (c_src % 64 != 0 || c_dst % 64 != 0) return z;
  const int family = g_conv_variant / 100;   // 30: eight-wave / two-per-CU shapes of the first experiment; 31: long batches
  if (family == 30) {
    if (g_conv_variant == 3064) return ConvDmaShape{64, 1, 1};
    return c_dst % 128 == 0 ? ConvDmaShape{128, 1, 1} : ConvDmaShape{64, 1, 1};
  }
  // families 31+: one four-wave workgroup per CU, as many MFMAs per batch as the shape allows
  if (c_dst % 128 == 0) return ConvDmaShape{128, 2, 1};
  if (c_src % 128 == 0) return ConvDmaShape{64, 1, 2};
  return ConvDmaShape{64, 1, 1};
}

int g_wgrad_depth = 0;         // me_debug_set_wgrad_config: 0 = default
int g_wgrad_wgs_per_cu = 0;
int g_wgrad_order = 0;         // me_debug_set_wgrad_order: 0 = XCD-aware range order, -1 = launch order

// launch geometry of the wgrad kernels for a (pairs, channels) problem
struct WgradGeom {
  int nb;        // 16-column MFMA blocks of output channels per wave (1, 2 or 4)
  int n_cib;     // blocks of 64 input channels   (grid.y)
  int n_cob;     // blocks of 16*nb output channels (waves along grid.z x waves per workgroup)
  int waves;     // waves per workgroup (side by side along the output channels)
  int gz;        // grid.z
  int64_t ranges;  // grid.x: equal ranges of the pair list
  int64_t slot_floats;
};

static WgradGeom wgrad_geom(int64_t n_pairs, int64_t volume, int c_in, int c_out) {
  WgradGeom g;
  g.nb = c_out <= 64 ? 1 : (c_out <= 128 ? 2 : 4);
  if (g_wgrad_depth >= 100) g.nb = g_wgrad_depth / 100;  // tuning: depth = 100 * nb + ring depth
  g.n_cib = (int)ceil_div(c_in, 16 * kWgMB);
  g.n_cob = (int)ceil_div(c_out, 16 * g.nb);
  g.waves = g.n_cob < 4 ? g.n_cob : 4;
  g.gz = (int)ceil_div(g.n_cob, g.waves);
  const int wpc = g_wgrad_wgs_per_cu > 0 ? g_wgrad_wgs_per_cu : 3;
  int64_t r = ceil_div((int64_t)device_cu_count() * wpc, (int64_t)g.n_cib * g.gz);
  if (r > n_pairs / 64) r = n_pairs / 64;  // at least 64 pairs per range (and no empty ranges)
  if (r < 1) r = 1;
  g.ranges = r;
  g.slot_floats = (int64_t)g.n_cib * g.n_cob * (kWgMB * g.nb * 4 * 64);
  (void)volume;
  return g;
}

// geometry of the LDS-staged kernels (k_wgrad_bf16, k_wgrad_lds_f32): always four waves side by side along the
// output channels
int g_wgrad_ws = 0;   // me_debug_set_wgrad_ws: 0 = k_wgrad_bf16 (default), 1 / 2 = k_wgrad_bf16_ws with four / two row register sets (tuning build)
int g_wgrad_mb = 0;   // me_debug_set_wgrad_mb: 0 = policy, 4 / 8 = input-channel blocks per workgroup of k_wgrad_bf16

// 128 x 128 blocks where at least two of them stand side by side along the input channels (c_in >= 192): measured per
// layer inside a MinkUNet34C step (gpurun_out/r04r): 384 -> 256 201 -> 151 us, 192 -> 128 157 -> 107, 256 -> 256 on 21k
// voxels 80 -> 71; with ONE block (c_in = 128) the doubled range count costs more than the bytes save (32 -> 34 us).
static int wgrad_bf16_mb(int c_in, int c_out) {
  if (g_wgrad_mb == 4 || g_wgrad_mb == 8) return (g_wgrad_mb == 8 && c_out > 64) ? 8 : 4;
  return (c_in >= 192 && c_out > 64) ? 8 : 4;
}

static WgradGeom wgrad_geom_staged(int64_t n_pairs, int c_in, int c_out, int wpc_default = 2, int mb = kWgMB) {
  WgradGeom g;
  // 64 or 128 output channels per workgroup; wider layers take several workgroup columns (grid.z) that
  // re-gather the x rows (the <4, 1> instantiation — 256 channels, 32-pair steps — faulted on the GPU and is
  // not used; its cause was not found in round 1)
  g.nb = c_out <= 64 ? 1 : 2;
  g.n_cib = (int)ceil_div(c_in, 16 * mb);
  g.n_cob = (int)ceil_div(c_out, 16 * g.nb);
  g.waves = 4;
  g.gz = (int)ceil_div(g.n_cob, 4);
  // (three input-channel blocks side by side — 384 -> 256 on 21k voxels — leave 86 ranges x 6 = 516 workgroups at two per CU:
  // a third of a round idle; at three per CU 148 -> 110 us, profiles/r05_wgrad_ranges_sweep.log; every other MinkUNet34C
  // shape is fastest at two)
  const int wpc = g_wgrad_wgs_per_cu > 0 ? g_wgrad_wgs_per_cu : (g.n_cib == 3 ? 3 : wpc_default);
  int64_t r = ceil_div((int64_t)device_cu_count() * wpc, (int64_t)g.n_cib * g.gz);
  if (r > n_pairs / 64) r = n_pairs / 64;
  if (r < 1) r = 1;
  g.ranges = r;
  g.slot_floats = (int64_t)g.n_cib * g.n_cob * (mb * g.nb * 4 * 64);
  return g;
}

template <int NB, int KSTEPS, int MB = kWgMB>
static int launch_wgrad_bf16(const WgradGeom &g, const __bf16 *x, int c_in, const __bf16 *dy, int c_out,
                             const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                             int volume, int64_t n_pairs, float *partial, hipStream_t stream) {
  constexpr int SP = 32 * KSTEPS;
  const int lds = 2 * SP * ((16 * MB + kWgStepLd) + (64 * NB + kWgStepLd)) * 2;
  auto fn = &k_wgrad_bf16<NB, KSTEPS, MB>;
  static bool attr_set = false;   // per instantiation
  if (lds > 32 * 1024 && !attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kLdsBudget));
    attr_set = true;
  }
  const dim3 grid((unsigned)g.ranges, (unsigned)g.n_cib, (unsigned)g.gz);
#ifdef ME_DEBUG_VARIANTS
  if (g_wgrad_ws != 0) {   // wave-specialised (k_wgrad_bf16_ws): same geometry, slots and sums
    auto ws = g_wgrad_ws == 2 ? &k_wgrad_bf16_ws<NB, KSTEPS, MB, 2> : &k_wgrad_bf16_ws<NB, KSTEPS, MB, (MB == 8 ? 2 : 4)>;
    static bool ws_attr[2] = {false, false};
    if (lds > 32 * 1024 && !ws_attr[g_wgrad_ws == 2]) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ws), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
      ws_attr[g_wgrad_ws == 2] = true;
    }
    hipLaunchKernelGGL(ws, grid, dim3(512), (size_t)lds, stream, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev,
                       volume, n_pairs, (int)g.ranges, g.n_cob, partial);
    ME_LAUNCH_CHECK();
    return 0;
  }
#endif
  hipLaunchKernelGGL(fn, grid, dim3(256), (size_t)lds, stream, x, c_in, dy, c_out, in_pairs,
                     out_pairs, k_offsets_dev, volume, n_pairs, (int)g.ranges, g.n_cob, partial);
  ME_LAUNCH_CHECK();
  return 0;
}

#ifdef ME_DEBUG_VARIANTS
template <int NB, int KSTEPS>
static int launch_wgrad_lds_f32(const WgradGeom &g, const float *x, int c_in, const float *dy, int c_out,
                                const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                                int volume, int64_t n_pairs, float *partial, hipStream_t stream) {
  constexpr int SP = 16 * KSTEPS;
  const int lds = 2 * SP * ((64 + kWgStepLd) + (64 * NB + kWgStepLd)) * 4;
  static bool attr_set = false;
  if (lds > 32 * 1024 && !attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_lds_f32<NB, KSTEPS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_set = true;
  }
  const dim3 grid((unsigned)g.ranges, (unsigned)g.n_cib, (unsigned)g.gz);
  hipLaunchKernelGGL((k_wgrad_lds_f32<NB, KSTEPS>), grid, dim3(256), (size_t)lds, stream, x, c_in, dy, c_out,
                     in_pairs, out_pairs, k_offsets_dev, volume, n_pairs, (int)g.ranges, g.n_cob, partial);
  ME_LAUNCH_CHECK();
  return 0;
}
#endif

template <int NB>
static int launch_wgrad_f32x3(const WgradGeom &g, const float *x, int c_in, const float *dy, int c_out,
                              const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                              int volume, int64_t n_pairs, float *partial, const WgRangeOrder &order,
                              hipStream_t stream) {
  const int lds = 3 * 32 * ((64 + kWgStepLd) + (64 * NB + kWgStepLd)) * 2;
  const dim3 grid((unsigned)g.ranges, (unsigned)g.n_cib, (unsigned)g.gz);
  hipLaunchKernelGGL((k_wgrad_f32x3<NB>), grid, dim3(256), (size_t)lds, stream, x, c_in, dy, c_out, in_pairs,
                     out_pairs, k_offsets_dev, volume, n_pairs, (int)g.ranges, g.n_cob, partial, order);
  ME_LAUNCH_CHECK();
  return 0;
}

// fp32 rows on the bf16 matrix pipe: whole 32-byte pieces and at least 32 x 32 channels (measured on dense and sparse
// maps, profiles/r02_tune_wgrad_f32x3.log: 32->32 on par, 32->64 1.4x, 64->128 1.3x, 96->96 1.4x, 256->256 1.1 - 1.5x;
// narrower layers would multiply mostly padding: the x tile is 64 channels wide);
// me_debug_set_wgrad_config(-3, 0) = never, (-4, 0) = wherever the rows allow
static bool wgrad_use_split(int c_in, int c_out) {
  if (g_wgrad_depth == -3 || (c_in % 8) != 0 || (c_out % 8) != 0) return false;
  if (g_wgrad_depth == -4) return true;
  return g_wgrad_depth == 0 && c_in >= 32 && c_out >= 32;
}

}  // namespace me

using namespace me;

// see WgRangeOrder.  Ranges sorted by the fraction of their offset's pair list at which they start, cut into groups
// of 32 (one range of every offset + the extra ranges of the centre offset), group g -> XCD g % 8; XCD x owns the
// workgroups x, x + 8, ...  Only for grids with one workgroup per range (the table indexes blockIdx.x).
static WgRangeOrder wgrad_range_order(const int64_t *k_offsets, int64_t volume, int64_t n_pairs, int64_t ranges,
                                      int64_t wgs_per_range) {
  WgRangeOrder o;
  o.n = 0;
  if (g_wgrad_order < 0 || ranges < 64 || ranges > kWgMaxOrder || wgs_per_range != 1 || volume < 2) return o;
  struct Item {
    double f;
    int r;
  };
  std::vector<Item> items((size_t)ranges);
  int k = 0;
  for (int64_t r = 0; r < ranges; ++r) {
    const int64_t e = wgrad_range_begin(r, n_pairs, ranges);
    while (k + 1 < volume && k_offsets[k + 1] <= e) ++k;
    const int64_t len = k_offsets[k + 1] - k_offsets[k];
    items[(size_t)r] = {len > 0 ? (double)(e - k_offsets[k]) / (double)len : 0.0, (int)r};
  }
  std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.f < b.f; });
  constexpr int kXcds = 8, kGroup = 32;
  std::vector<std::vector<int>> per_xcd(kXcds);
  for (int64_t i = 0; i < ranges; ++i) per_xcd[(size_t)((i / kGroup) % kXcds)].push_back(items[(size_t)i].r);
  std::vector<int> assigned((size_t)ranges, -1), spill;
  for (int x = 0; x < kXcds; ++x) {
    for (size_t i = 0; i < per_xcd[(size_t)x].size(); ++i) {
      const int64_t b = (int64_t)i * kXcds + x;
      if (b < ranges) assigned[(size_t)b] = per_xcd[(size_t)x][i];
      else spill.push_back(per_xcd[(size_t)x][i]);
    }
  }
  size_t sp = 0;
  for (int64_t b = 0; b < ranges; ++b)
    if (assigned[(size_t)b] < 0) assigned[(size_t)b] = spill[sp++];
  o.n = (int)ranges;
  for (int64_t b = 0; b < ranges; ++b) o.v[b] = (uint16_t)assigned[(size_t)b];
  return o;
}

template <typename T>
static int wgrad_launch(const T *x, int64_t n_in, int32_t c_in, const T *dy, int64_t n_out, int32_t c_out,
                        const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets,
                        const int64_t *k_offsets_dev, int64_t volume, float *grad_w, void *workspace,
                        int64_t workspace_bytes, hipStream_t stream) {
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(c_in > 0 && c_out > 0, "channel counts must be positive");
  ME_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0, "feature pointers must be 16-byte aligned");
  ME_CHECK(workspace_bytes >= me_conv_wgrad_workspace_bytes(k_offsets, volume, c_in, c_out),
           "workspace too small");
  const int64_t n_pairs = k_offsets[volume];
  const WgradGeom g = wgrad_geom(n_pairs, volume, c_in, c_out);
  float *partial = reinterpret_cast<float *>(workspace);
  const int depth = (g_wgrad_depth > 0 && g_wgrad_depth % 100 > 0) ? g_wgrad_depth % 100 : (g.nb == 4 ? 8 : 4);  // measured: profiles/r01_tune_wgrad_v2.log
  if (n_pairs > 0) {
    const dim3 grid((unsigned)g.ranges, (unsigned)g.n_cib, (unsigned)g.gz);
    const dim3 block(64 * g.waves);
    const bool vec = (c_in % kWgMB) == 0 && (c_out % g.nb) == 0;
    // 32-bit byte offsets with a 24-bit row multiply (load_piece<SMALL>)
    const int64_t lim = 1ll << 32;
    const bool small = n_in > 0 && n_out > 0 && n_in < (1 << 24) && n_out < (1 << 24) &&
                       n_in * c_in * (int64_t)sizeof(T) < lim && n_out * c_out * (int64_t)sizeof(T) < lim &&
                       g_conv_variant != 6;  // (variant 6: the 64-bit address path, tests/test_gpu_conv.py)
    const WgRangeOrder order = wgrad_range_order(k_offsets, volume, n_pairs, g.ranges, (int64_t)g.n_cib * g.gz);
#define ME_WGRAD_LAUNCH(NBV, DV)                                                                                 \
  do {                                                                                                           \
    if (vec && small)                                                                                            \
      hipLaunchKernelGGL((k_wgrad_f32<T, NBV, DV, true, true>), grid, block, 0, stream, x, c_in, dy, c_out,      \
                         in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs, (int)g.ranges, g.n_cob,       \
                         partial, order);                                                                        \
    else if (vec)                                                                                                \
      hipLaunchKernelGGL((k_wgrad_f32<T, NBV, DV, true>), grid, block, 0, stream, x, c_in, dy, c_out, in_pairs,  \
                         out_pairs, k_offsets_dev, (int)volume, n_pairs, (int)g.ranges, g.n_cob, partial, order); \
    else                                                                                                         \
      hipLaunchKernelGGL((k_wgrad_f32<T, NBV, 8, false>), grid, block, 0, stream, x, c_in, dy, c_out, in_pairs,  \
                         out_pairs, k_offsets_dev, (int)volume, n_pairs, (int)g.ranges, g.n_cob, partial, order); \
  } while (0)
    if (g.nb == 1) { if (depth == 4) ME_WGRAD_LAUNCH(1, 4); else ME_WGRAD_LAUNCH(1, 8); }
    else if (g.nb == 2) { if (depth == 4) ME_WGRAD_LAUNCH(2, 4); else ME_WGRAD_LAUNCH(2, 8); }
    else { if (depth == 4) ME_WGRAD_LAUNCH(4, 4); else ME_WGRAD_LAUNCH(4, 8); }
#undef ME_WGRAD_LAUNCH
    ME_LAUNCH_CHECK();
  }
  const dim3 rgrid((unsigned)ceil_div(g.slot_floats, 256), (unsigned)volume);
#define ME_WGRAD_REDUCE(NBV)                                                                               \
  hipLaunchKernelGGL((k_wgrad_reduce<NBV, false>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev, (int)volume, \
                     n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w)
  if (g.nb == 1) ME_WGRAD_REDUCE(1);
  else if (g.nb == 2) ME_WGRAD_REDUCE(2);
  else ME_WGRAD_REDUCE(4);
#undef ME_WGRAD_REDUCE
  ME_LAUNCH_CHECK();
  return 0;
}

extern "C" {

int me_conv_plan_config(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                        int32_t *tile_rows, int32_t *batch_groups) {
  ME_CHECK(tile_rows != nullptr && batch_groups != nullptr, "output pointers must not be null");
  *tile_rows = 128;
  *batch_groups = ME_MAX_BATCH_GROUPS;
  if (n_tgt <= 0 || volume <= 0 || c_src <= 0 || c_dst <= 0) return 0;
  const ConvVariant v = conv_variant(c_src, c_dst);
  PlanShape s;
  s.nc = v.nc;
  s.slabs = v.slabs;
  s.chunks = (int)ceil_div(c_src, v.kc);
  s.group_cycles = (v.kc / 4) * 32.0;  // fp32 MFMAs of one 16-row group and chunk
  s.stage_row_bytes = stage_ld(v.kc) * 4 + 4;
  *tile_rows = plan_tile_rows(s, n_tgt, volume, n_pairs);
  return 0;
}

int64_t me_conv_packed_weight_elems(int64_t volume, int32_t c_src, int32_t c_dst) {
  if (volume <= 0 || c_src <= 0 || c_dst <= 0) return 0;
  const ConvVariant v = conv_variant(c_src, c_dst);
  return volume * align_up(c_src, v.kc) * align_up(c_dst, 16);
}

int me_conv_pack_weights_f32(const float *w, int64_t volume, int32_t c_src, int32_t c_dst, int32_t transposed,
                             float *wp, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && c_src > 0 && c_dst > 0, "invalid weight shape");
  ME_CHECK((uintptr_t)wp % 16 == 0, "packed weights must be 16-byte aligned");
  const ConvVariant v = conv_variant(c_src, c_dst);
  const int nchunks = (int)ceil_div(c_src, v.kc), ncb = (int)ceil_div(c_dst, 16);
  const int64_t total = volume * nchunks * ncb * (v.kc / 16) * 64;  // 16-byte elements
  f32x4 *wp4 = reinterpret_cast<f32x4 *>(wp);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (v.kc == 96)
    hipLaunchKernelGGL(k_pack_weights<96>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  else if (v.kc == 64)
    hipLaunchKernelGGL(k_pack_weights<64>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  else if (v.kc == 32)
    hipLaunchKernelGGL(k_pack_weights<32>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  else
    hipLaunchKernelGGL(k_pack_weights<16>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  ME_LAUNCH_CHECK();
  return 0;
}

static int conv_target_f32(const float *src, int64_t n_src, int32_t c_src, const float *wp, int64_t volume,
                           int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                           const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, float *dst,
                           int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream_, bool fuse) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)volume;
  ME_CHECK(c_src > 0 && c_dst > 0, "channel counts must be positive");
  ME_CHECK(tile_rows >= ME_GROUP_ROWS && tile_rows <= ME_MAX_TILE_ROWS, "tile_rows out of range");
  ME_CHECK(batch_groups >= 1 && batch_groups <= ME_MAX_BATCH_GROUPS, "batch_groups out of range");
  ME_CHECK((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)wp % 16 == 0,
           "feature and weight pointers must be 16-byte aligned");
  if (n_tgt == 0) return 0;
  const ConvVariant v = conv_variant(c_src, c_dst);
#define ME_CONV_ARGS                                                                                         \
  src, c_src, wp, c_dst, v.slabs, plan_src, plan_dst, batch_desc, tile_bptr, order, dst, n_tgt, tile_rows, \
      batch_groups, stream
  // 32-bit byte offsets with a 24-bit row multiply need a source matrix below 4 GiB (variant 6: 64-bit addresses)
  const bool small = n_src > 0 && n_src < (1ll << 24) && n_src * c_src * 4 < (1ll << 32) && g_conv_variant != 6;
#ifdef ME_DEBUG_VARIANTS   // experiments and timing ablations (some with invalid results): tuning builds only
  if (small) {
    const ConvDmaShape ds = conv_dma_shape(c_src, c_dst);
#define ME_DMA_ARGS src, c_src, wp, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst, n_tgt, tile_rows, \
                    batch_groups, stream
    if (ds.nc == 128 && ds.cbw == 1) return launch_conv_tile_dma<128, 1, 1>(ME_DMA_ARGS);
    if (ds.nc == 128 && ds.cbw == 2) return launch_conv_tile_dma<128, 2, 1>(ME_DMA_ARGS);
    if (ds.nc == 64 && ds.kch == 2) return launch_conv_tile_dma<64, 1, 2>(ME_DMA_ARGS);
    if (ds.nc == 64) return launch_conv_tile_dma<64, 1, 1>(ME_DMA_ARGS);
#undef ME_DMA_ARGS
  }
  if (g_conv_variant >= 2048 && g_conv_variant < 3000 && v.nc == 64 && v.kc == 64)
    return launch_conv_tile<64, 64, 0>(ME_CONV_ARGS, small);
  if (g_conv_variant == 1000 && v.kc == 64 && c_dst >= 128 && c_dst % 128 == 0)  // experiment: 128-column workgroups
    return launch_conv_tile<128, 64, 0>(src, c_src, wp, c_dst, c_dst / 128, plan_src, plan_dst, batch_desc, tile_bptr,
                                        order, dst, n_tgt, tile_rows, batch_groups, stream);
  if (g_conv_variant != 0 && v.nc == 64 && v.kc == 64) {  // ablation builds exist for the headline shape only
    switch (g_conv_variant) {
      case 16: return launch_conv_tile<64, 64, 16>(ME_CONV_ARGS);
      case 18: return launch_conv_tile<64, 64, 18>(ME_CONV_ARGS);
      case 210: return launch_conv_tile<64, 64, 210>(ME_CONV_ARGS);
      case 214: return launch_conv_tile<64, 64, 214>(ME_CONV_ARGS);
      case 4: return launch_conv_tile<64, 64, 4>(ME_CONV_ARGS);
      case 80: return launch_conv_tile<64, 64, 80>(ME_CONV_ARGS);
      case 256: return launch_conv_tile<64, 64, 256>(ME_CONV_ARGS);
      case 272: return launch_conv_tile<64, 64, 272>(ME_CONV_ARGS);
      default: break;
    }
  }
#endif
  // sparse maps (multi-offset batches): on the bf16 pipe with exactly split operands where that kernel is instantiated
  // (conv_f32x3_fused.hip: the fp32 MFMA of k_conv_tile_f32 blocks its SIMD for every other wave)
  if (fuse && small && g_conv_variant == 0) {
    const int rc = launch_conv_f32x3_fused(v.nc, v.kc, src, n_src, c_src, wp, c_dst, v.slabs, plan_src, plan_dst, batch_desc,
                                           tile_bptr, order, dst, n_tgt, tile_rows, stream);
    if (rc != -1) return rc;
  }
#define ME_CONV_CASE(NCV, KCV) return launch_conv_tile<NCV, KCV, 0>(ME_CONV_ARGS, small, fuse)
  if (v.nc == 96) {
    if (v.kc == 96) ME_CONV_CASE(96, 96);
    if (v.kc == 64) ME_CONV_CASE(96, 64);
    if (v.kc == 32) ME_CONV_CASE(96, 32);
    ME_CONV_CASE(96, 16);
  } else if (v.nc == 32) {
    if (v.kc == 96) ME_CONV_CASE(32, 96);
    if (v.kc == 64) ME_CONV_CASE(32, 64);
    if (v.kc == 32) ME_CONV_CASE(32, 32);
    ME_CONV_CASE(32, 16);
  } else {
    if (v.kc == 96) ME_CONV_CASE(64, 96);
    if (v.kc == 64) ME_CONV_CASE(64, 64);
    if (v.kc == 32) ME_CONV_CASE(64, 32);
    ME_CONV_CASE(64, 16);
  }
#undef ME_CONV_CASE
#undef ME_CONV_ARGS
}

int me_conv_target_f32(const float *src, int64_t n_src, int32_t c_src, const float *wp, int64_t volume,
                       int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                       const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, float *dst,
                       int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream) {
  return conv_target_f32(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                         n_tgt, tile_rows, batch_groups, stream, false);
}

int me_conv_target_f32_fused(const float *src, int64_t n_src, int32_t c_src, const float *wp, int64_t volume,
                             int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                             const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order, float *dst,
                             int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream) {
  return conv_target_f32(src, n_src, c_src, wp, volume, c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst,
                         n_tgt, tile_rows, batch_groups, stream, true);
}

int32_t me_debug_variants_compiled(void) {
#ifdef ME_DEBUG_VARIANTS
  return 1;
#else
  return 0;
#endif
}

// The default build holds the shipped kernels and the alternates whose results are VALID (the bit-identity tests
// compare them): 6 = 64-bit gather addresses, 7 = no batch fusion, 9 = 32-wide passes for 96 channels, 30 / 31 =
// ping-pong / four-multiplier split kernels.  Everything else (40 / 41 / 43: ring and persistent split kernels, phase counters, timing ablations with invalid
// results, the LDS-DMA family) exists only in a -DME_DEBUG_VARIANTS build (scripts/) and is refused otherwise.
int me_debug_set_conv_variant(int variant) {
#ifndef ME_DEBUG_VARIANTS
  const bool ok = variant == 0 || variant == 6 || variant == 7 || variant == 9 || variant == 30 || variant == 31;
  ME_CHECK(ok, "this conv variant needs a -DME_DEBUG_VARIANTS build of libme_amd.so (tuning / ablation kernels)");
#endif
  g_conv_variant = variant;
  return 0;
}

int me_debug_conv_timing(uint64_t *out8, int32_t reset) {
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out8 != nullptr) {
    ME_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(d_conv_timing), sizeof(h)));
    for (int i = 0; i < 8; ++i) out8[i] = h[i];
  }
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ME_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_conv_timing), z, sizeof(z)));
  }
  return 0;
}


int me_transpose_kernel_f32(const float *w, int64_t volume, int32_t c_in, int32_t c_out, float *wt,
                            void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  const dim3 grid((unsigned)ceil_div(c_out, 32), (unsigned)ceil_div(c_in, 32), (unsigned)volume);
  hipLaunchKernelGGL(k_transpose_kernel, grid, dim3(256), 0, stream, w, volume, c_in, c_out, wt);
  ME_LAUNCH_CHECK();
  return 0;
}

int64_t me_conv_wgrad_workspace_bytes(const int64_t *k_offsets, int64_t volume, int32_t c_in, int32_t c_out) {
  if (volume < 1 || c_in <= 0 || c_out <= 0) return 256;
  // the larger need of the two kernels (k_wgrad_lds_f32 / k_wgrad_f32)
  const WgradGeom g = wgrad_geom(k_offsets[volume], volume, c_in, c_out);
  const WgradGeom h = wgrad_geom_staged(k_offsets[volume], c_in, c_out, 3);   // (k_wgrad_f32x3: three per CU)
  const int64_t a = (g.ranges + volume) * g.slot_floats, b = (h.ranges + volume) * h.slot_floats;
  return align_up((a > b ? a : b) * 4, 256);
}

void me_debug_set_wgrad_order(int mode) { g_wgrad_order = mode; }
void me_debug_set_wgrad_mb(int mb) { g_wgrad_mb = mb; }
void me_debug_set_wgrad_ws(int mode) { g_wgrad_ws = mode; }

void me_debug_set_wgrad_config(int depth, int wgs_per_cu) {
  g_wgrad_depth = depth;
  g_wgrad_wgs_per_cu = wgs_per_cu;
}

int me_conv_wgrad_f32(const float *x, int64_t n_in, int32_t c_in, const float *dy, int64_t n_out, int32_t c_out,
                      const int32_t *in_pairs,
                      const int32_t *out_pairs, const int64_t *k_offsets, const int64_t *k_offsets_dev,
                      int64_t volume, float *grad_w, void *workspace, int64_t workspace_bytes,
                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  // Default: the direct-from-global kernel.  The LDS-staged fp32 kernel (me_debug_set_wgrad_config(-2, 0)) measured
  // the same 82 TF on config 2 (profiles/r01_tune_wgrad_f32_staged.log): in fp32 the weight gradient is bound by
  // the ~4 TB/s at which 640 MB of random rows arrive from beyond the L2s, not by how the matrix pipe is fed
  // (it is 13 % faster on sparse maps and 40 % slower on 32 -> 32, so it is not the default).
  if (wgrad_use_split(c_in, c_out)) {
    ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
    ME_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0, "feature pointers must be 16-byte aligned");
    ME_CHECK(workspace_bytes >= me_conv_wgrad_workspace_bytes(k_offsets, volume, c_in, c_out), "workspace too small");
    const int64_t n_pairs = k_offsets[volume];
    const WgradGeom g = wgrad_geom_staged(n_pairs, c_in, c_out, 3);
    float *partial = reinterpret_cast<float *>(workspace);
    if (n_pairs > 0) {
      const WgRangeOrder order = wgrad_range_order(k_offsets, volume, n_pairs, g.ranges, (int64_t)g.n_cib * g.gz);
      const int rc = g.nb == 1 ? launch_wgrad_f32x3<1>(g, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev,
                                                        (int)volume, n_pairs, partial, order, stream)
                               : launch_wgrad_f32x3<2>(g, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev,
                                                        (int)volume, n_pairs, partial, order, stream);
      if (rc != 0) return rc;
    }
    const dim3 rgrid((unsigned)ceil_div(g.slot_floats, 256), (unsigned)volume);
    if (g.nb == 1)
      hipLaunchKernelGGL((k_wgrad_reduce<1, true>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev,
                         (int)volume, n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w);
    else
      hipLaunchKernelGGL((k_wgrad_reduce<2, true>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev,
                         (int)volume, n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w);
    ME_LAUNCH_CHECK();
    return 0;
  }
#ifndef ME_DEBUG_VARIANTS
  return wgrad_launch<float>(x, n_in, c_in, dy, n_out, c_out, in_pairs, out_pairs, k_offsets, k_offsets_dev, volume, grad_w,
                             workspace, workspace_bytes, stream);
#else
  if ((c_in % 4) != 0 || (c_out % 4) != 0 || g_wgrad_depth != -2)
    return wgrad_launch<float>(x, n_in, c_in, dy, n_out, c_out, in_pairs, out_pairs, k_offsets, k_offsets_dev, volume, grad_w,
                               workspace, workspace_bytes, stream);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(c_in > 0 && c_out > 0, "channel counts must be positive");
  ME_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0, "feature pointers must be 16-byte aligned");
  ME_CHECK(workspace_bytes >= me_conv_wgrad_workspace_bytes(k_offsets, volume, c_in, c_out), "workspace too small");
  const int64_t n_pairs = k_offsets[volume];
  const WgradGeom g = wgrad_geom_staged(n_pairs, c_in, c_out);
  float *partial = reinterpret_cast<float *>(workspace);
  if (n_pairs > 0) {
    int rc;
    if (g.nb == 1)
      rc = launch_wgrad_lds_f32<1, 2>(g, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs,
                                      partial, stream);
    else
      rc = launch_wgrad_lds_f32<2, 2>(g, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs,
                                      partial, stream);
    if (rc != 0) return rc;
  }
  const dim3 rgrid((unsigned)ceil_div(g.slot_floats, 256), (unsigned)volume);
  if (g.nb == 1)
    hipLaunchKernelGGL((k_wgrad_reduce<1, true>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev, (int)volume,
                       n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w);
  else
    hipLaunchKernelGGL((k_wgrad_reduce<2, true>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev, (int)volume,
                       n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w);
  ME_LAUNCH_CHECK();
  return 0;
#endif
}

}  // extern "C"

extern "C" {

int64_t me_conv_wgrad_workspace_bytes_bf16(const int64_t *k_offsets, int64_t volume, int32_t c_in, int32_t c_out) {
  if (volume < 1 || c_in <= 0 || c_out <= 0) return 256;
  // the larger of the kernels' needs (channel counts that are not multiples of 8 take k_wgrad_f32<__bf16>)
  const WgradGeom g4 = wgrad_geom_staged(k_offsets[volume], c_in, c_out, 2, 4);
  const WgradGeom g8 = wgrad_geom_staged(k_offsets[volume], c_in, c_out, 2, 8);
  const int64_t a4 = align_up((g4.ranges + volume) * g4.slot_floats * 4, 256);
  const int64_t a8 = align_up((g8.ranges + volume) * g8.slot_floats * 4, 256);
  const int64_t a = a4 > a8 ? a4 : a8;
  const int64_t b = me_conv_wgrad_workspace_bytes(k_offsets, volume, c_in, c_out);
  return a > b ? a : b;
}

int me_conv_wgrad_bf16(const uint16_t *x_, int64_t n_in, int32_t c_in, const uint16_t *dy_, int64_t n_out,
                       int32_t c_out, const int32_t *in_pairs,
                       const int32_t *out_pairs, const int64_t *k_offsets, const int64_t *k_offsets_dev,
                       int64_t volume, float *grad_w, void *workspace, int64_t workspace_bytes,
                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const __bf16 *x = reinterpret_cast<const __bf16 *>(x_);
  const __bf16 *dy = reinterpret_cast<const __bf16 *>(dy_);
  ME_CHECK(workspace_bytes >= me_conv_wgrad_workspace_bytes_bf16(k_offsets, volume, c_in, c_out),
           "workspace too small");
  // rows that are not whole 16-byte pieces: the fp32-MFMA kernel with bf16 loads
  if ((c_in % 8) != 0 || (c_out % 8) != 0 || g_wgrad_depth < 0)
    return wgrad_launch<__bf16>(x, n_in, c_in, dy, n_out, c_out, in_pairs, out_pairs, k_offsets, k_offsets_dev, volume, grad_w,
                                workspace, workspace_bytes, stream);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0, "feature pointers must be 16-byte aligned");
  const int64_t n_pairs = k_offsets[volume];
  const int mb = wgrad_bf16_mb(c_in, c_out);
  const WgradGeom g = wgrad_geom_staged(n_pairs, c_in, c_out, 2, mb);
  float *partial = reinterpret_cast<float *>(workspace);
  if (n_pairs > 0) {
    int rc;
#define ME_WG16(NBV, KSV, MBV)                                                                                      \
  launch_wgrad_bf16<NBV, KSV, MBV>(g, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs, \
                                   partial, stream)
    if (g.nb == 1) rc = ME_WG16(1, 2, 4);
    else if (mb == 8) rc = ME_WG16(2, 2, 8);
    else rc = ME_WG16(2, 2, 4);
#undef ME_WG16
    if (rc != 0) return rc;
  }
  const dim3 rgrid((unsigned)ceil_div(g.slot_floats, 256), (unsigned)volume);
#define ME_WGRAD_REDUCE_TR(NBV, MBV)                                                                               \
  hipLaunchKernelGGL((k_wgrad_reduce<NBV, true, MBV>), rgrid, dim3(64 * kWgReducePhases), 0, stream, partial, k_offsets_dev, (int)volume, \
                     n_pairs > 0 ? n_pairs : 1, (int)g.ranges, g.n_cib, g.n_cob, c_in, c_out, grad_w)
  if (g.nb == 1) ME_WGRAD_REDUCE_TR(1, 4);
  else if (mb == 8) ME_WGRAD_REDUCE_TR(2, 8);
  else ME_WGRAD_REDUCE_TR(2, 4);
#undef ME_WGRAD_REDUCE_TR
  ME_LAUNCH_CHECK();
  return 0;
}

int me_conv_forward_naive_f32(const float *in_feat, int32_t c_in, const float *w, int32_t c_out,
                              const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                              int64_t volume, int64_t n_pairs, float *out_feat, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_pairs == 0) return 0;
  const int64_t total = n_pairs * c_out;
  hipLaunchKernelGGL(k_naive_forward, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, in_feat,
                     c_in, w, c_out, in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs, out_feat);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_conv_backward_naive_f32(const float *in_feat, int32_t c_in, const float *grad_out, int32_t c_out,
                               const float *w, const int32_t *in_pairs, const int32_t *out_pairs,
                               const int64_t *k_offsets_dev, int64_t volume, int64_t n_pairs, float *grad_in,
                               float *grad_w, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_pairs > 0) {
    const int64_t total = n_pairs * c_in;
    hipLaunchKernelGGL(k_naive_dgrad, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, grad_out,
                       c_out, w, c_in, in_pairs, out_pairs, k_offsets_dev, (int)volume, n_pairs, grad_in);
    ME_LAUNCH_CHECK();
  }
  const int64_t tw = volume * c_in * c_out;
  hipLaunchKernelGGL(k_naive_wgrad, dim3((unsigned)ceil_div(tw, 256)), dim3(256), 0, stream, in_feat, c_in,
                     grad_out, c_out, in_pairs, out_pairs, k_offsets_dev, (int)volume, grad_w);
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_transpose_kernel));
}
