// bf16 sparse convolution (forward / dgrad), wave-specialised (round 4).
//
// k_conv_tile_bf16 (conv_bf16.hip) gives every wave of a workgroup every job of a batch, in lockstep: index window ->
// gather -> barrier -> stage write -> barrier -> operands -> MFMA -> accumulate.  Its phase counters say what that costs
// (docs/HISTORY.md 10.4): 3,200 - 4,600 cycles per batch for ~200 cycles of matrix work, waves waiting on s_waitcnt and
// barriers 42 % of their cycles, issuing 24 %; more workgroups per CU, deeper prefetch, a second stage buffer, split-K
// and an offset-synchronous schedule each moved it by a few per cent.  The fp32 kernel on the bf16 pipe
// (k_conv_tile_f32x3_ws) pushes SIX MFMAs per product and twice the gather bytes through the same plan in 1.6x the
// time — its structure, not its arithmetic, is the difference.  This is that structure with one operand plane:
//   * waves 4-7 ONLY produce: index window three batches ahead, rows two batches ahead (two register sets), stage write
//     into the free one of TWO stage buffers (swizzled, no padding: conv_ws.hpp StageLayout);
//   * waves 0-3 ONLY multiply, NC / 4 columns each: every LDS operation of a batch is inline asm in a fixed order with
//     counted waits (consume_batch_ws), weights of the next batch requested behind the first operand reads;
//   * ONE barrier per batch separates the two streams; wave w and w + 4 share a SIMD (matrix stream at priority 2).
// Same plan, same packed weights (k_pack_weights_bf16: the image is the MFMA A operand either way), same sums in the
// same order as k_conv_tile_bf16: bit-identical output (tests/test_gpu_bf16.py).  Batch-norm statistics of the tile in
// the epilogue as there.  Not covered (k_conv_tile_bf16 keeps them): multi-offset batch fusion for sparse maps, split-K,
// 256-channel chunks, 32-column slabs, source channels that are no multiple of the chunk, 64-bit gather offsets.
// Reference: src/convolution_kernel.cu:320-496 (one gather-GEMM-scatter launch per kernel offset).
#include "conv_common.hpp"
#include "conv_ws.hpp"

namespace me {

// accumulator tile + TWO stage buffers (one plane + the target indices of 64 rows each)
__host__ __device__ constexpr int conv_bf16_ws_lds(int nc, int kc, int tile_rows) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + 2 * ME_MAX_BATCH_GROUPS * 16 * (x3_stage_ld(kc) * 2 + 4);
}

// Phase counters of a -DME_WS_TIMING build (scripts/ws_phase_timing.py; s_memtime ticks, summed over wave 0 — a multiplier —
// and wave 4 — a producer — of every workgroup): [0] multiplier work between barriers, [1] multiplier barrier wait,
// [2] producer work, [3] producer barrier wait, [4] batches, [5] tile prologue (to the first barrier), [6] epilogue, [7] tiles
#ifdef ME_WS_TIMING
__device__ unsigned long long d_ws_timing[8];
#define ME_WS_SYNC(ROLE)                                                        \
  do {                                                                          \
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();                \
    tm_[2 * (ROLE)] += t0_ - t_prev_;                                           \
    __syncthreads();                                                            \
    t_prev_ = __builtin_amdgcn_s_memtime();                                     \
    tm_[2 * (ROLE) + 1] += t_prev_ - t0_;                                       \
  } while (0)
#else
#define ME_WS_SYNC(ROLE) __syncthreads()
#endif

// DEPTH: register sets of gathered rows in flight per producer thread — the rows of batch x + DEPTH are requested while
// batch x is staged.  A batch lasts 1,500 - 2,000 cycles here and a miss to HBM under load about as long: two sets
// (the fp32 kernel's pipeline, whose batches last twice as long) leave the producers waiting for rows.
//
// FUSE (sparse maps, the hosts' density rule): runs of single-group batches of consecutive offsets — what the plan of a
// sparse map consists of — are staged and multiplied together, up to MAXSUB offsets per barrier, each group with its own
// offset's weights (consume_super_ws); producers and multipliers walk the same super-batch sequence off the batch
// descriptors.  Same sums in the same order: bit-identical to the unfused launch.  MEASURED SLOWER than k_conv_tile_bf16's
// fused launch (96 -> 96 on 200k voxels, 8.7 pairs per item: 178 us against 92): one eight-wave workgroup per CU and
// super-batches of two offsets do not beat three four-wave workgroups there.  Instantiated in the tuning build only
// (-DME_DEBUG_VARIANTS, me_debug_set_bf16_ws_fuse(1)); sparse launches stay with k_conv_tile_bf16.
// NCW: multiplier waves — 4 (NC / 4 columns each) or, for 128-column slabs, 8 (16 columns each: two multipliers per SIMD
// next to the producer, one's MFMAs cover the other's LDS waits; the fp32 kernel's default there).
template <int NC, int KC, int DEPTH, bool FUSE = false, int NCW = 4>
__global__ __launch_bounds__((NCW + 4) * 64, 1) void k_conv_tile_bf16_ws(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, __bf16 *__restrict__ dst, int64_t n_tgt, int tile_rows,
    float *__restrict__ stat_mean, float *__restrict__ stat_m2) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef StageLayout<KC> SL;
  static_assert(NC == 64 || NC == 128, "multiplier waves of 16 or 32 columns");
  static_assert(NCW == 4 || (NCW == 8 && NC == 128 && !FUSE), "four multiplier waves, or eight on a 128-column slab");
  constexpr int CB = NC / (16 * NCW);  // 16-column blocks per multiplier wave
  constexpr int NTP = 256;             // producer threads (four waves)
  constexpr int NT = NCW * 64 + NTP;
  constexpr int WAVES = NT / 64;
  constexpr int LD = SL::kLd;
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int KS = KC / 32;
  constexpr int F8 = KC / 8;           // 16-byte pieces per gathered row
  constexpr int CAP = ME_MAX_BATCH_GROUPS * 16;
  constexpr int ITER = (CAP * F8 + NTP - 1) / NTP;
  constexpr int PLANE = CAP * LD;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);                              // [(tile_rows + 1) x ACC_LD]
  __bf16 *s_a = reinterpret_cast<__bf16 *>(s_acc + (tile_rows + 1) * ACC_LD);  // [2][64 x LD]
  int32_t *s_dst = reinterpret_cast<int32_t *>(s_a + 2 * PLANE);               // [2][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15;
  const int q = lane >> 4;
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];   // heaviest-first dispatch order (me_plan_build)
  const int col_base = blockIdx.y * NC;
  const int nchunks = c_src / KC;                           // (whole chunks: host-checked)
  const int ncb = (c_dst + 15) / 16;

  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  const int n_it = nb * nchunks;     // chunk-major, as k_conv_tile_bf16 walks them
#ifdef ME_WS_TIMING
  unsigned long long tm_[4] = {0, 0, 0, 0};
  const unsigned long long t_start_ = __builtin_amdgcn_s_memtime();
  unsigned long long t_prev_ = t_start_, t_first_ = 0, t_loop_end_ = 0;
#endif
  struct Desc {
    int chunk, g0, ng, k;
  };
  auto locate = [&](int it) {
    int r = min(it, n_it - 1);
    Desc d;
    d.chunk = 0;
    while (r >= nb) {
      r -= nb;
      ++d.chunk;
    }
    const i32x2 v = *reinterpret_cast<const i32x2 *>(batch_desc + 2 * (int64_t)(b0 + r));
    d.g0 = v.x;
    d.ng = v.y & 255;
    d.k = (int)((uint32_t)v.y >> 8);
    return d;
  };

  // super-batch walker (FUSE): the next super-batch of the tile, or nsub = 0 behind the last one; every wave walks the
  // same sequence.  MAXSUB offsets per super-batch while their weights fit the registers of a multiplier wave.
  constexpr int MAXSUB = !FUSE ? 1 : (KS <= 2 ? 4 : 2);
  struct Super {
    int chunk, g0, ng, nsub;
    int k[MAXSUB];
  };
  int cur_chunk = 0, cur_r = 0;
  auto next_super = [&]() {
    Super sb;
    const bool valid = cur_chunk < nchunks && nb > 0;
    const int r = valid ? cur_r : max(nb - 1, 0);
    sb.chunk = valid ? cur_chunk : max(nchunks - 1, 0);
    const int avail = valid ? nb - cur_r : 1;
    i32x2 dd[MAXSUB];
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j)   // (descriptors behind the tile's last batch are readable: me_plan_max_groups)
      dd[j] = *reinterpret_cast<const i32x2 *>(batch_desc + 2 * (int64_t)(b0 + r + j));
    sb.g0 = dd[0].x;
    sb.ng = 0;
    sb.nsub = 0;
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j) {
      const int g = dd[j].y & 255;
      // (only single-group batches are fused: group r <-> sub-batch r)
      const bool take = j == 0 || (sb.nsub == j && j < avail && sb.ng == j && g == 1);
      sb.k[j] = take ? (int)((uint32_t)dd[j].y >> 8) : sb.k[j > 0 ? j - 1 : 0];
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

  if (n_it > 0 && wave >= NCW) {
    // ------------------------------------------------ producer waves ------------------------------------------------
    const int ptid = tid - NCW * 64;
    bf16x8 stage[DEPTH][ITER];
    int32_t dstv[DEPTH];
    int32_t sidx[2][ITER];
    // the 64-entry index window of a batch is read to its end (the plan is followed by 64 readable entries); slots
    // behind the batch's own groups and padding slots (-1) gather row 0: nobody multiplies the former, the products of
    // the latter land in the dummy accumulator row
    auto load_sidx = [&](const auto &d, int32_t (&sx)[ITER]) {
      const char *pb = reinterpret_cast<const char *>(plan_src + (int64_t)d.g0 * 16);
#pragma unroll
      for (int j = 0; j < ITER; ++j)
        sx[j] = *reinterpret_cast<const int32_t *>(pb + (unsigned)(min((j * NTP + ptid) / F8, CAP - 1) * 4));
    };
    const char *srcb = reinterpret_cast<const char *>(src);
    const unsigned row_bytes = (unsigned)c_src * 2u;
    auto gather = [&](const auto &d, const int32_t (&sx)[ITER], bf16x8 (&st)[ITER], int32_t &dv) {
      const int c0 = d.chunk * KC;
      dv = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)d.g0 * 16) +
                                             (unsigned)(min(ptid, CAP - 1) * 4));
#pragma unroll
      for (int j = 0; j < ITER; ++j) {
        const int idx = j * NTP + ptid;
        const int ch = c0 + (idx % F8) * 8;
        const int sr = idx / F8 < d.ng * 16 ? max(sx[j], 0) : 0;
        st[j] = *reinterpret_cast<const bf16x8 *>(srcb + (__umul24((unsigned)sr, row_bytes) + (unsigned)ch * 2u));
      }
    };
    auto write_stage = [&](const bf16x8 (&st)[ITER], int32_t dv, int buf) {
      __bf16 *base = s_a + buf * PLANE;
#pragma unroll
      for (int j = 0; j < ITER; ++j) {
        const int idx = j * NTP + ptid;
        const int r = idx / F8;
        if (ITER * NTP == CAP * F8 || r < CAP) *reinterpret_cast<bf16x8 *>(base + SL::off(r, idx % F8)) = st[j];
      }
      if (ptid < CAP) s_dst[buf * CAP + ptid] = dv;
    };
    static_assert(DEPTH == 2 || DEPTH == 4, "register sets: the loop is unrolled by DEPTH, buffers alternate");
    if constexpr (!FUSE) {
      // produce(x): store batch x (register set x % DEPTH) into buffer x & 1, request the indices of batch x + DEPTH + 1 and,
      // with the indices requested one step ago, the rows of batch x + DEPTH into the set just stored
      Desc d0 = locate(DEPTH), d1 = locate(DEPTH + 1), d2 = locate(DEPTH + 2);   // batches x + DEPTH, + 1, + 2 of the next produce
      auto produce = [&](int x, auto set_) {
        constexpr int SET = decltype(set_)::value;     // = x % DEPTH
        write_stage(stage[SET], dstv[SET], SET & 1);
        load_sidx(d1, sidx[(SET + 1) & 1]);
        gather(d0, sidx[SET & 1], stage[SET], dstv[SET]);
        d0 = d1;
        d1 = d2;
        d2 = locate(x + DEPTH + 3);
      };
      {
        // prologue: the first DEPTH batches two at a time (two index sets), then the indices of batch DEPTH
#pragma unroll
        for (int j = 0; j < DEPTH; j += 2) {
          const Desc da = locate(j), db = locate(j + 1);
          load_sidx(da, sidx[0]);
          load_sidx(db, sidx[1]);
          gather(da, sidx[0], stage[j], dstv[j]);
          gather(db, sidx[1], stage[j + 1], dstv[j + 1]);
        }
        load_sidx(d0, sidx[0]);
        produce(0, std::integral_constant<int, 0>{});   // batch 0 -> buffer 0
      }
      __syncthreads();
#ifdef ME_WS_TIMING
      t_prev_ = t_first_ = __builtin_amdgcn_s_memtime();
#endif
      // iteration it: batch it + 1 is staged while batch it is multiplied
      int it = 0;
      for (; it + DEPTH <= n_it; it += DEPTH) {
        produce(it + 1, std::integral_constant<int, 1 % DEPTH>{});
        ME_WS_SYNC(1);
        produce(it + 2, std::integral_constant<int, 2 % DEPTH>{});
        ME_WS_SYNC(1);
        if constexpr (DEPTH == 4) {
          produce(it + 3, std::integral_constant<int, 3>{});
          ME_WS_SYNC(1);
          produce(it + 4, std::integral_constant<int, 0>{});
          ME_WS_SYNC(1);
        }
      }
      if (it < n_it) {
        produce(it + 1, std::integral_constant<int, 1 % DEPTH>{});
        ME_WS_SYNC(1);
        ++it;
      }
      if constexpr (DEPTH == 4) {
        if (it < n_it) {
          produce(it + 1, std::integral_constant<int, 2>{});
          ME_WS_SYNC(1);
          ++it;
        }
        if (it < n_it) {
          produce(it + 1, std::integral_constant<int, 3>{});
          ME_WS_SYNC(1);
          ++it;
        }
      }
    } else {
      // the same pipeline over super-batches: d0 / d1 / d2 are the super-batches x + DEPTH, + 1, + 2 of the next produce.
      // Bit j of `live` says whether super-batch it + j exists (it = the one the multipliers are at): the loop ends with
      // theirs.  produce() walks one super-batch further and records it at bit `at`.
      Super first[DEPTH];
      unsigned live = 0u;
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        first[j] = next_super();
        live |= (first[j].nsub > 0 ? 1u : 0u) << j;
      }
      Super d0 = next_super(), d1 = next_super(), d2 = next_super();
      live |= (d0.nsub > 0 ? 1u : 0u) << DEPTH;
      live |= (d1.nsub > 0 ? 1u : 0u) << (DEPTH + 1);
      live |= (d2.nsub > 0 ? 1u : 0u) << (DEPTH + 2);
      auto produce = [&](auto set_, int at) {
        constexpr int SET = decltype(set_)::value;
        write_stage(stage[SET], dstv[SET], SET & 1);
        load_sidx(d1, sidx[(SET + 1) & 1]);
        gather(d0, sidx[SET & 1], stage[SET], dstv[SET]);
        d0 = d1;
        d1 = d2;
        d2 = next_super();
        live |= (d2.nsub > 0 ? 1u : 0u) << at;
      };
#pragma unroll
      for (int j = 0; j < DEPTH; j += 2) {
        load_sidx(first[j], sidx[0]);
        load_sidx(first[j + 1], sidx[1]);
        gather(first[j], sidx[0], stage[j], dstv[j]);
        gather(first[j + 1], sidx[1], stage[j + 1], dstv[j + 1]);
      }
      load_sidx(d0, sidx[0]);
      produce(std::integral_constant<int, 0>{}, DEPTH + 3);   // super-batch 0 -> buffer 0; walks to super-batch DEPTH + 3
      __syncthreads();
      // iteration it (while super-batch it exists): super-batch it + 1 is staged while super-batch it is multiplied
#define ME_WS_STEP(SETV)                                              \
  if (!(live & 1u)) break;                                            \
  produce(std::integral_constant<int, SETV>{}, DEPTH + 4);            \
  __syncthreads();                                                    \
  live >>= 1;
      for (;;) {
        ME_WS_STEP(1 % DEPTH)
        ME_WS_STEP(2 % DEPTH)
        if constexpr (DEPTH == 4) {
          ME_WS_STEP(3)
          ME_WS_STEP(0)
        }
      }
#undef ME_WS_STEP
    }
  } else if (n_it > 0) {
    // ----------------------------------------------- multiplier waves -----------------------------------------------
    __builtin_amdgcn_s_setprio(2);   // (the matrix stream wins the issue arbitration against the producer wave of its SIMD)
    const int cbi0 = col_base / 16 + wave * CB;
    bf16x8 w[2][CB][1][KS];
    auto load_w = [&](const Desc &d, bf16x8 (&wd)[CB][1][KS]) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const bf16x8 *p = wp + ((((int64_t)d.k * nchunks + d.chunk) * ncb + min(cbi0 + c, ncb - 1)) * KS) * 64 + lane;
#pragma unroll
        for (int v = 0; v < KS; ++v) wd[c][0][v] = p[v * 64];
      }
    };
    int pofs[KS];
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) pofs[sx] = ((sx * 4 + q) ^ SL::swz(i16)) * 8;
    auto multiply = [&](const Desc &d, const bf16x8 (&wc)[CB][1][KS], int buf, auto &&next_w) {
      const __bf16 *rowp = s_a + buf * PLANE + i16 * LD;
      const int32_t *dstp = s_dst + buf * CAP + i16;
      float *accp = &s_acc[wave * CB * 16 + q * 4];
if (d.ng >= 4) consume_batch_ws<2, 2, CB, KC, 1>(rowp, pofs, wc, dstp, accp, ACC_LD, next_w);
      else if (d.ng == 3) consume_batch_ws<2, 1, CB, KC, 1>(rowp, pofs, wc, dstp, accp, ACC_LD, next_w);
      else if (d.ng == 2) consume_batch_ws<2, 0, CB, KC, 1>(rowp, pofs, wc, dstp, accp, ACC_LD, next_w);
      else consume_batch_ws<1, 0, CB, KC, 1>(rowp, pofs, wc, dstp, accp, ACC_LD, next_w);
    };
    if constexpr (!FUSE) {
      Desc dA = locate(0), dB = locate(1);
      load_w(dA, w[0]);
      __syncthreads();                      // batch 0 is staged
#ifdef ME_WS_TIMING
      t_prev_ = t_first_ = __builtin_amdgcn_s_memtime();
#endif
      auto iteration = [&](int it, int P, bf16x8 (&w_cu)[CB][1][KS], bf16x8 (&w_nx)[CB][1][KS]) {
        multiply(dA, w_cu, P, [&]() { load_w(dB, w_nx); });
        ME_WS_SYNC(0);
        dA = dB;
        dB = locate(it + 2);
      };
      int it = 0;
      for (; it + 1 < n_it; it += 2) {
        iteration(it, 0, w[0], w[1]);
        iteration(it + 1, 1, w[1], w[0]);
      }
      if (it < n_it) iteration(it, 0, w[0], w[1]);
    } else {
      // one weight slice per offset of a super-batch, two register sets: the next super-batch's are requested while this
      // one is multiplied
      bf16x8 wf[2][MAXSUB][CB][1][KS];
      auto load_wf = [&](const Super &sb, bf16x8 (&wd)[MAXSUB][CB][1][KS]) {
#pragma unroll
        for (int j = 0; j < MAXSUB; ++j) {
          if (j == 0 || j < sb.nsub) {     // wave-uniform: a dense batch loads one slice
#pragma unroll
            for (int c = 0; c < CB; ++c) {
              const bf16x8 *p = wp + ((((int64_t)sb.k[j] * nchunks + sb.chunk) * ncb + min(cbi0 + c, ncb - 1)) * KS) * 64 + lane;
#pragma unroll
              for (int v = 0; v < KS; ++v) wd[j][c][0][v] = p[v * 64];
            }
          }
        }
      };
      Super sA = next_super(), sB = next_super(), sC = next_super();
      load_wf(sA, wf[0]);
      __syncthreads();                      // super-batch 0 is staged
      auto iteration = [&](int P, const bf16x8 (&w_cu)[MAXSUB][CB][1][KS], bf16x8 (&w_nx)[MAXSUB][CB][1][KS]) {
        const __bf16 *rowp = s_a + P * PLANE + i16 * LD;
        const int32_t *dstp = s_dst + P * CAP + i16;
        float *accp = &s_acc[wave * CB * 16 + q * 4];
        consume_super_ws<MAXSUB, CB, KC, 1>(rowp, pofs, w_cu, sA.nsub, sA.ng, dstp, accp, ACC_LD, [&]() { load_wf(sB, w_nx); });
        __syncthreads();
        sA = sB;
        sB = sC;
        sC = next_super();
      };
      for (;;) {
        if (sA.nsub == 0) break;
        iteration(0, wf[0], wf[1]);
        if (sA.nsub == 0) break;
        iteration(1, wf[1], wf[0]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    __syncthreads();
  }

#ifdef ME_WS_TIMING
  t_loop_end_ = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: every target row of the tile is written exactly once, rounded to bf16 (RNE); the tile's batch-norm
  // statistics ride along (see k_conv_tile_bf16: same arithmetic, this kernel's thread count) ----
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const bool vec_out = (c_dst % 4) == 0;
  int32_t *s_ord = reinterpret_cast<int32_t *>(s_a);   // (the stage buffers are free now)
  if (order != nullptr) {
#pragma unroll
    for (int j = 0; j < ORD; ++j)
      if (j * NT + tid < tile_rows) s_ord[j * NT + tid] = my_ord[j];
    __syncthreads();
  }
  constexpr int G4 = NC / 4;                       // threads per tile row = four-column groups
  const bool do_stats = stat_mean != nullptr;      // uniform
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (do_stats) {
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&s_acc[(tid % G4) * 4]);   // row 0 of the tile: the shift
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[t] = (float)(__bf16)v0[t];
  }
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / G4;
    const int c4 = x % G4;
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
    float *s_st = s_acc + ACC_LD;                 // [WAVES][G4][8] behind row 0 (8 x 32 x 8 floats <= 16 rows; twelve
                                                  // waves: 24 rows — the launcher keeps shorter tiles on eight waves)
    if (lane < G4) {
      float *wv = s_st + (wave * G4 + lane) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wv[t] = st1[t];
        wv[4 + t] = st2[t];
      }
    }
    __syncthreads();
    if (tid < NC && col_base + tid < c_dst) {
      const int c4 = tid >> 2, t = tid & 3;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int wv = 0; wv < WAVES; ++wv) {
        a += s_st[(wv * G4 + c4) * 8 + t];
        b += s_st[(wv * G4 + c4) * 8 + 4 + t];
      }
      const float shift = (float)(__bf16)s_acc[tid];
      const float cnt = (float)rows_here, m = a / cnt;
      stat_mean[(int64_t)tile * c_dst + col_base + tid] = shift + m;
      stat_m2[(int64_t)tile * c_dst + col_base + tid] = fmaxf(b - a * m, 0.f);
    }
  }
#ifdef ME_WS_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  if (!FUSE && n_it > 0 && lane == 0 && (wave == 0 || wave == NCW)) {
    const unsigned long long t_end_ = __builtin_amdgcn_s_memtime();
    const int role = wave == 0 ? 0 : 1;
    atomicAdd(&d_ws_timing[2 * role], tm_[2 * role]);
    atomicAdd(&d_ws_timing[2 * role + 1], tm_[2 * role + 1]);
    if (wave == 0) {
      atomicAdd(&d_ws_timing[4], (unsigned long long)n_it);
      atomicAdd(&d_ws_timing[5], t_first_ - t_start_);
      atomicAdd(&d_ws_timing[6], t_end_ - t_loop_end_);
      atomicAdd(&d_ws_timing[7], 1ull);
    }
  }
#endif
}

// ---- launch (called by conv_bf16.hip's dispatcher) -------------------------------------------------------------------
bool conv_bf16_ws_shape(int nc, int kc) { return (nc == 64 || nc == 128) && (kc == 32 || kc == 64 || kc == 96 || kc == 128); }

int conv_bf16_ws_lds_bytes(int nc, int kc, int tile_rows) { return conv_bf16_ws_lds(nc, kc, tile_rows); }

int g_bf16_ws_depth = 4;   // me_debug_set_bf16_ws_depth: 2 | 4 register sets of rows in flight
int g_bf16_ws_ncw = 0;     // me_debug_set_bf16_ws_ncw: 0 policy | 4 | 8 multiplier waves on 128-column slabs

template <int NC, int KC>
static int launch_ws(const __bf16 *src, int c_src, const bf16x8 *wp, int c_dst, int slabs, const int32_t *plan_src,
                     const int32_t *plan_dst, const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order,
                     __bf16 *dst, int64_t n_tgt, int tile_rows, hipStream_t stream, float *stat_mean, float *stat_m2, bool fuse) {
  const int lds = conv_bf16_ws_lds(NC, KC, tile_rows);
  typedef void (*kernel_t)(const __bf16 *, int, const bf16x8 *, int, const int32_t *, const int32_t *, const int32_t *,
                           const int32_t *, const int32_t *, __bf16 *, int64_t, int, float *, float *);
  // (two register sets of rows in flight instead of four — me_debug_set_bf16_ws_depth(2) — measure the same on every
  // MinkUNet34C layer, profiles/r06_ws_depth.log: instantiated in the tuning build only)
#ifdef ME_DEBUG_VARIANTS
  const bool deep = g_bf16_ws_depth != 2;
  kernel_t fn = deep ? &k_conv_tile_bf16_ws<NC, KC, 4> : &k_conv_tile_bf16_ws<NC, KC, 2>;
#else
  const bool deep = true;
  kernel_t fn = &k_conv_tile_bf16_ws<NC, KC, 4>;
#endif
  int which = deep ? 1 : 0, threads = 512;
  if constexpr (NC == 128) {
    // Eight multiplier waves where the slab runs one workgroup per CU anyway (chunks of 96 / 128 channels: > 128 registers):
    // measured 4 - 11 % faster there (128 -> 128 @80k 72 -> 67 us, 384 -> 256 121 -> 108), 30 % SLOWER where four multipliers
    // leave room for two workgroups per CU (64 -> 128 @100k 57 -> 74 us): profiles/r04_ws_sweep_ncw.log.
    // (The statistics epilogue of twelve waves needs >= 24 tile rows.)
    const bool eight = g_bf16_ws_ncw == 8 || (g_bf16_ws_ncw == 0 && KC >= 96);
    if (eight && tile_rows >= 24 && !fuse) {
      fn = &k_conv_tile_bf16_ws<NC, KC, 4, false, 8>;
      which = 3;
      threads = 768;
    }
  }
#ifdef ME_DEBUG_VARIANTS   // measured 2x SLOWER than k_conv_tile_bf16's fused launch on MinkUNet34C's sparse levels
  if (fuse) {               // (profiles/r04_layers_minkunet34c_bf16_ws_fuse.log): tuning build only
    fn = &k_conv_tile_bf16_ws<NC, KC, 4, true>;
    which = 2;
    threads = 512;
  }
#else
  if (fuse) return -1;
#endif
  static bool attr_set[4] = {false, false, false, false};   // per instantiation
  if (lds > 48 * 1024 && !attr_set[which]) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_set[which] = true;
  }
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)slabs);
  hipLaunchKernelGGL(fn, grid, dim3((unsigned)threads), (size_t)lds, stream, src, c_src, wp, c_dst, plan_src, plan_dst, batch_desc,
                     tile_bptr, order, dst, n_tgt, tile_rows, stat_mean, stat_m2);
  ME_LAUNCH_CHECK();
  return 0;
}

// -1: the shape / tile is not this kernel's (the caller runs k_conv_tile_bf16)
int launch_conv_bf16_ws(int nc, int kc, const void *src, int c_src, const void *wp, int c_dst, int slabs,
                        const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                        const int32_t *tile_bptr, const int32_t *order, void *dst, int64_t n_tgt, int tile_rows,
                        hipStream_t stream, float *stat_mean, float *stat_m2, bool fuse) {
  if (!conv_bf16_ws_shape(nc, kc) || c_src % kc != 0 || conv_bf16_ws_lds(nc, kc, tile_rows) > kLdsBudget) return -1;
#define ME_WS(NCV, KCV)                                                                                                     \
  if (nc == NCV && kc == KCV)                                                                                               \
  return launch_ws<NCV, KCV>(reinterpret_cast<const __bf16 *>(src), c_src, reinterpret_cast<const bf16x8 *>(wp), c_dst,     \
                             slabs, plan_src, plan_dst, batch_desc, tile_bptr, order, reinterpret_cast<__bf16 *>(dst), n_tgt, \
                             tile_rows, stream, stat_mean, stat_m2, fuse)
  ME_WS(64, 32);
  ME_WS(64, 64);
  ME_WS(64, 96);
  ME_WS(64, 128);
  ME_WS(128, 32);
  ME_WS(128, 64);
  ME_WS(128, 96);
  ME_WS(128, 128);
#undef ME_WS
  return -1;
}

}  // namespace me

extern "C" void me_debug_set_bf16_ws_depth(int depth) { me::g_bf16_ws_depth = depth; }
extern "C" void me_debug_set_bf16_ws_ncw(int ncw) { me::g_bf16_ws_ncw = ncw; }

// phase counters of a -DME_WS_TIMING build (zeros otherwise); reset != 0 clears them
extern "C" int me_debug_ws_timing(uint64_t *out8, int32_t reset) {
#ifdef ME_WS_TIMING
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out8 != nullptr) {
    ME_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(me::d_ws_timing), sizeof(h)));
    for (int i = 0; i < 8; ++i) out8[i] = h[i];
  }
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ME_HIP(hipMemcpyToSymbol(HIP_SYMBOL(me::d_ws_timing), z, sizeof(z)));
  }
#else
  if (out8 != nullptr)
    for (int i = 0; i < 8; ++i) out8[i] = 0;
  (void)reset;
#endif
  return 0;
}

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv_bf16_ws(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_conv_tile_bf16_ws<64, 64, 4>));
}
