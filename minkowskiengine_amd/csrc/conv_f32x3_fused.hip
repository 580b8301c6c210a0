// fp32 sparse convolution (forward / dgrad) on SPARSE maps, on the bf16 matrix pipe (round 4).
//
// BASELINE config 5 (4-D, K = 81, 32 -> 64 channels, 7 pairs per (tile, offset) item) ran on k_conv_tile_f32 with
// multi-offset batches: the fp32 MFMA (v_mfma_f32_16x16x4_f32) blocks its SIMD for every other wave, and 19 % of its
// peak was the result.  The split kernels (conv_f32x3.hip) had no multi-offset batches, so the policy kept them away
// from sparse maps.  This kernel is k_conv_tile_bf16's lock-step loop with batch fusion (up to MAXSUB single-group
// batches of consecutive offsets per barrier pair, each group with its own offset's weights) on three operand planes:
//   * fp32 rows are gathered into registers, split exactly into three bf16 terms while they are written to the swizzled
//     LDS stage planes (StageLayout, split3_raw + the non-finite redo path of conv_common.hpp);
//   * the weights come from the fp32 register image that me_conv_pack_weights_f32 makes for k_conv_tile_f32 — the entry
//     point, the packed image and the plan are those of me_conv_target_f32_fused: no host change — re-addressed per lane
//     into the bf16 MFMA's operand layout and split in registers (weight side: exact);
//   * six v_mfma_f32_16x16x32_bf16 per product block, smallest terms first, fp32 accumulation (consume_super_ws<.., 3>);
//     the accumulator tile is updated in batch order.  Fixed order: bitwise reproducible; error against float64 that of
//     the other split kernels (tests/test_gpu_conv.py).
// Reference: src/convolution_kernel.cu:320-496 in fp32 (AT_DISPATCH_FLOATING_TYPES, src/convolution_gpu.cu:137-155).
//
// MEASURED (profiles/r04_conv4d_f32x3_fused.log): parity and fp32-grade error as designed, but 2x SLOWER than
// k_conv_tile_f32's fused launch on config 5 (forward 490 vs 248 us, dgrad 491 vs 256): the launch is bound by the
// lock-step chain of a super-batch (3 - 4 us each), not by the fp32 MFMA — and this kernel's 232 - 256 registers leave two
// workgroups per CU where the fp32 kernel has three.  Instantiated in the tuning build only (-DME_DEBUG_VARIANTS).
#include "conv_common.hpp"
#include "conv_ws.hpp"

namespace me {
#ifdef ME_DEBUG_VARIANTS

// accumulator tile + ONE stage buffer (three planes + the target indices of 64 rows)
__host__ __device__ constexpr int conv_x3f_lds(int nc, int kc, int tile_rows) {
  return (tile_rows + 1) * (nc + kAccPad) * 4 + ME_MAX_BATCH_GROUPS * 16 * (3 * x3_stage_ld(kc) * 2 + 4);
}

template <int NC, int KC>
__global__ __launch_bounds__(NC * 4, 2) void k_conv_tile_f32x3_fused(
    const float *__restrict__ src, int c_src, const f32x4 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ batch_desc, const int32_t *__restrict__ tile_bptr,
    const int32_t *__restrict__ order, float *__restrict__ dst, int64_t n_tgt, int tile_rows) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef StageLayout<KC> SL;
  static_assert(NC == 32 || NC == 64, "two or four waves of 16 columns");
  static_assert(KC == 32 || KC == 64, "one or two MFMA steps per chunk");
  constexpr int WAVES = NC / 16;
  constexpr int NT = WAVES * 64;
  constexpr int LD = SL::kLd;
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int KS = KC / 32;
  constexpr int KQ4 = KC / 16;         // 16-byte quads of the fp32 weight image per (offset, chunk, column block)
  constexpr int F8 = KC / 8;           // 8-channel pieces (32 bytes of fp32) per gathered row
  constexpr int CAP = ME_MAX_BATCH_GROUPS * 16;
  constexpr int ITER = (CAP * F8 + NT - 1) / NT;
  constexpr int PLANE = CAP * LD;
  constexpr int MAXSUB = KS <= 1 ? 4 : 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);                              // [(tile_rows + 1) x ACC_LD]
  __bf16 *s_a = reinterpret_cast<__bf16 *>(s_acc + (tile_rows + 1) * ACC_LD);  // [3][64 x LD]
  int32_t *s_dst = reinterpret_cast<int32_t *>(s_a + 3 * PLANE);               // [64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15;
  const int q = lane >> 4;
  const int tile = tile_bptr[gridDim.x + 1 + blockIdx.x];   // heaviest-first dispatch order (me_plan_build)
  const int col_base = blockIdx.y * NC;
  const int nchunks = c_src / KC;                           // (whole chunks: host-checked)
  const int ncb = (c_dst + 15) / 16;
  const int cb = min(col_base / 16 + wave, ncb - 1);

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

  // super-batch walker (as k_conv_tile_bf16 / k_conv_tile_bf16_ws): up to MAXSUB single-group batches of consecutive
  // offsets, or one batch of up to four groups
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

  f32x4 stage[ITER][2];
  int32_t dstv = tile_rows;
  int32_t sidx[ITER];
  f32x4 wraw[MAXSUB][KS][2];           // next super-batch's weights as they lie in the fp32 image
  bf16x8 wreg[MAXSUB][1][3][KS];       // this super-batch's, split into three bf16 planes

  auto load_sidx = [&](int g0) {
    const char *pb = reinterpret_cast<const char *>(plan_src + (int64_t)g0 * 16);
#pragma unroll
    for (int j = 0; j < ITER; ++j)
      sidx[j] = *reinterpret_cast<const int32_t *>(pb + (unsigned)(min((j * NT + tid) / F8, CAP - 1) * 4));
  };
  const char *srcb = reinterpret_cast<const char *>(src);
  const unsigned row_bytes = (unsigned)c_src * 4u;
  auto gather = [&](const Super &sb) {
    const int c0 = sb.chunk * KC;
    dstv = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan_dst + (int64_t)sb.g0 * 16) +
                                             (unsigned)(min(tid, CAP - 1) * 4));
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int idx = j * NT + tid;
      const int ch = c0 + (idx % F8) * 8;
      const int sr = idx / F8 < sb.ng * 16 ? max(sidx[j], 0) : 0;   // (slots behind the super-batch's groups: row 0)
      const f32x4 *p = reinterpret_cast<const f32x4 *>(srcb + (__umul24((unsigned)sr, row_bytes) + (unsigned)ch * 4u));
      stage[j][0] = p[0];
      stage[j][1] = p[1];
    }
  };
  auto write_stage = [&]() {
    uint32_t flag = 0u;      // non-finite rows: conv_common.hpp (split3_flag / split3_fix)
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
      const int idx = j * NT + tid;
      const int r = idx / F8;
      u32x4 p1, p2, p3;
      split3_raw(stage[j][0], stage[j][1], p1, p2, p3);
      flag = split3_flag(flag, p3);
      if (ITER * NT == CAP * F8 || r < CAP) {
        __bf16 *o = s_a + SL::off(r, idx % F8);
        *reinterpret_cast<u32x4 *>(o) = p1;
        *reinterpret_cast<u32x4 *>(o + PLANE) = p2;
        *reinterpret_cast<u32x4 *>(o + 2 * PLANE) = p3;
      }
    }
    if (__builtin_expect(__any(split3_suspect(flag)), 0)) {   // rare: split again, exactly, over the first attempt
#pragma unroll
      for (int j = 0; j < ITER; ++j) {
        const int idx = j * NT + tid;
        const int r = idx / F8;
        u32x4 p1, p2, p3;
        split3_fix(stage[j][0], stage[j][1], p1, p2, p3);
        if (ITER * NT == CAP * F8 || r < CAP) {
          __bf16 *o = s_a + SL::off(r, idx % F8);
          *reinterpret_cast<u32x4 *>(o) = p1;
          *reinterpret_cast<u32x4 *>(o + PLANE) = p2;
          *reinterpret_cast<u32x4 *>(o + 2 * PLANE) = p3;
        }
      }
    }
    if (tid < CAP) s_dst[tid] = dstv;
  };
  // The fp32 image holds, for quad v' and lane (q', i16), the four weights of channels (4 v' + q') * 4 .. + 3 of column
  // cb * 16 + i16 (k_pack_weights); the bf16 MFMA's A operand wants channels s * 32 + q * 8 .. + 7 in lane (q, i16):
  // pieces 8 s + 2 q and 8 s + 2 q + 1, i.e. quad 2 s + (q >> 1), lanes (2 (q & 1)) * 16 + i16 and the next 16.
  auto load_w = [&](const Super &sb) {
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j) {
      if (j == 0 || j < sb.nsub) {     // wave-uniform
        const f32x4 *p = wp + ((((int64_t)sb.k[j] * nchunks + sb.chunk) * ncb + cb) * KQ4) * 64 + (2 * (q & 1)) * 16 + i16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          wraw[j][s][0] = p[(2 * s + (q >> 1)) * 64];
          wraw[j][s][1] = p[(2 * s + (q >> 1)) * 64 + 16];
        }
      }
    }
  };
  auto split_w = [&](int nsub) {
#pragma unroll
    for (int j = 0; j < MAXSUB; ++j) {
      if (j > 0 && j >= nsub) continue;   // wave-uniform: registers that were not loaded are not split
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 p1, p2, p3;
        split3<true>(wraw[j][s][0], wraw[j][s][1], p1, p2, p3);
        wreg[j][0][0][s] = __builtin_bit_cast(bf16x8, p1);
        wreg[j][0][1][s] = __builtin_bit_cast(bf16x8, p2);
        wreg[j][0][2][s] = __builtin_bit_cast(bf16x8, p3);
      }
    }
  };
  int pofs[KS];
#pragma unroll
  for (int sx = 0; sx < KS; ++sx) pofs[sx] = ((sx * 4 + q) ^ SL::swz(i16)) * 8;

  Super sA = next_super();
  if (sA.nsub > 0) {
    Super sB = next_super();
    Super sC = next_super();
    load_w(sA);
    load_sidx(sA.g0);
    gather(sA);
    load_sidx(sB.g0);
    while (sA.nsub > 0) {
      __syncthreads();
      write_stage();
      split_w(sA.nsub);
      __syncthreads();
      load_w(sB);
      gather(sB);
      load_sidx(sC.g0);
      consume_super_ws<MAXSUB, 1, KC, 3>(s_a + i16 * LD, pofs, wreg, sA.nsub, sA.ng, s_dst + i16, &s_acc[wave * 16 + q * 4], ACC_LD,
                                         []() {});
      sA = sB;
      sB = sC;
      sC = next_super();
    }
  }
  __syncthreads();

  // every target row of the tile is written exactly once (rows without neighbours get zeros)
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const bool vec_out = (c_dst % 4) == 0;
  int32_t *s_ord = reinterpret_cast<int32_t *>(s_a);   // the stage buffer is free now
  if (order != nullptr) {
#pragma unroll
    for (int j = 0; j < ORD; ++j)
      if (j * NT + tid < tile_rows) s_ord[j * NT + tid] = my_ord[j];
    __syncthreads();
  }
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
}

#endif  // ME_DEBUG_VARIANTS

int g_f32_fused_split = 0;   // me_debug_set_f32_fused_split: 1 = on where instantiated (tuning build), 0 never (default)

#ifdef ME_DEBUG_VARIANTS

template <int NC, int KC>
static int launch_x3f(const float *src, int c_src, const float *wp, int c_dst, int slabs, const int32_t *plan_src,
                      const int32_t *plan_dst, const int32_t *batch_desc, const int32_t *tile_bptr, const int32_t *order,
                      float *dst, int64_t n_tgt, int tile_rows, hipStream_t stream) {
  const int lds = conv_x3f_lds(NC, KC, tile_rows);
  static bool attr_set = false;   // per instantiation
  if (lds > 48 * 1024 && !attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_tile_f32x3_fused<NC, KC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_set = true;
  }
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)slabs);
  hipLaunchKernelGGL((k_conv_tile_f32x3_fused<NC, KC>), grid, dim3(NC * 4), (size_t)lds, stream, src, c_src,
                     reinterpret_cast<const f32x4 *>(wp), c_dst, plan_src, plan_dst, batch_desc, tile_bptr, order, dst, n_tgt,
                     tile_rows);
  ME_LAUNCH_CHECK();
  return 0;
}

#endif  // ME_DEBUG_VARIANTS

// -1: not this kernel's launch (me_conv_target_f32_fused runs k_conv_tile_f32 then).  (nc, kc) = conv_variant's choice
// for the fp32 image: the kernel re-addresses that image, so it must agree with how it was packed.
int launch_conv_f32x3_fused(int nc, int kc, const float *src, int64_t n_src, int c_src, const float *wp, int c_dst, int slabs,
                            const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                            const int32_t *tile_bptr, const int32_t *order, float *dst, int64_t n_tgt, int tile_rows,
                            hipStream_t stream) {
#ifndef ME_DEBUG_VARIANTS
  return -1;
#else
  if (g_f32_fused_split == 0) return -1;
  if ((nc != 32 && nc != 64) || (kc != 32 && kc != 64) || c_src % kc != 0) return -1;
  if (!(n_src > 0 && n_src < (1ll << 24) && n_src * c_src * 4 < (1ll << 32))) return -1;   // 32-bit gather offsets
  if (conv_x3f_lds(nc, kc, tile_rows) > kLdsBudget) return -1;
#define ME_X3F(NCV, KCV)                                                                                                    \
  if (nc == NCV && kc == KCV)                                                                                               \
  return launch_x3f<NCV, KCV>(src, c_src, wp, c_dst, slabs, plan_src, plan_dst, batch_desc, tile_bptr, order, dst, n_tgt, \
                              tile_rows, stream)
  ME_X3F(64, 32);
  ME_X3F(64, 64);
  ME_X3F(32, 32);
  ME_X3F(32, 64);
#undef ME_X3F
  return -1;
#endif
}

}  // namespace me

extern "C" void me_debug_set_f32_fused_split(int mode) { me::g_f32_fused_split = mode; }
