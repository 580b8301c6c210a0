// bf16 sparse convolution for layers with AT MOST EIGHT source channels — a network's stem (MinkUNet: 3 -> 32 channels,
// 5^3 = 125 offsets) — with the kernel offsets stacked into the MFMA's reduction dimension (round 5).
//
// The tile-plan kernels treat every (tile, offset) item as its own batch: with 8 (padded) source channels a 32-channel
// MFMA step is three quarters padding, and a 125-offset stem on 200k voxels is 93 us at 5 TFLOP/s.  Here ONE
// v_mfma_f32_16x16x32_bf16 multiplies FOUR offsets at once: lane (i16, q) of the B operand holds the 8 channels of target
// row i16's neighbour at offset 4 j + q — one 16-byte load straight from the feature matrix, whose rows ARE 16 bytes —
// and the A operand holds W[4 j + q][0..7][column]: 32 steps cover 125 offsets.  Output-stationary: a wave owns 16 G
// target rows (G = 4: 64) x all output columns in registers; the weights (K x 8 x Cout, 64 KB as MFMA fragments for the
// stem) are built from the layer's own kernel tensor into LDS once per workgroup, which then walks tiles of 64 G rows; no
// plan, no packed image, no barrier inside a tile's walk.  The walk is a software pipeline of pinned requests (inline
// asm, counted s_waitcnt): neighbour indices six quads ahead, rows two quads ahead, every request unconditional
// (clamped) — a predicated load serialises on its own wait.  Same semantics as me_conv_target_bf16 (weights rounded to
// bf16, exact products, fp32 sums in a fixed order — ascending offset quads —, one rounding), batch-norm partials per tile.
// MI355X, 200k voxels, 3 -> 32 channels, 5^3 offsets: 94 us on the tile-plan kernel, 40 us here (profiles/r05_stem_*).
// Reference: src/convolution_kernel.cu:320-496 (one gather - GEMM - scatter launch per offset: 125 launches for the stem).
#include "conv_common.hpp"
#include <stdlib.h>

namespace me {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kStemWaves = 4;      // waves per workgroup; a wave owns 16 G rows of a tile of 64 G rows
constexpr int kStemIdxAhead = 6;   // neighbour indices are requested six offset quads ahead, rows two quads ahead

__host__ __device__ constexpr int conv_stem_lds(int nq4, int cb) {
  const int nc = cb * 16;
  return nq4 * cb * 64 * 16 + 2 * nc * 4 + 2 * kStemWaves * nc * 2 * 4;
}

// W fragments of the whole kernel into LDS: fragment (j, cb), lane (i16, q) = W[4 j + q][0..7][cb * 16 + i16], zeros for the
// offsets beyond `volume` (the walk runs over nq4 = a multiple of four quads and loads clamped neighbours there)
template <int CB, typename WT>
__device__ __forceinline__ void stem_stage_weights(bf16x8 *s_w, const WT *__restrict__ w, int transposed, int volume, int nq4,
                                                   int tid) {
  constexpr int NC = CB * 16;
  constexpr int NT = 64 * kStemWaves, U = 4;      // U entries per thread in flight (8 loads each)
  const int total = nq4 * CB * 64;
  for (int e0 = tid; e0 < total; e0 += NT * U) {
    float f[U][8];
    float live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = min(e0 + u * NT, total - 1);
      const int l = e & 63, cb = (e >> 6) % CB, j = (e >> 6) / CB;
      const int k = 4 * j + (l >> 4), col = cb * 16 + (l & 15);
      const int kc = min(k, volume - 1);
      live[u] = k < volume ? 1.f : 0.f;
      const WT *p = transposed ? w + ((int64_t)kc * NC + col) * 8 : w + (int64_t)kc * 8 * NC + col;
      const int step = transposed ? 1 : NC;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) f[u][ch] = (float)p[ch * step];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bf16x8 v;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) v[ch] = (__bf16)(f[u][ch] * live[u]);
      if (e0 + u * NT < total) s_w[e0 + u * NT] = v;
    }
  }
}

template <int CB, int G>
__global__ __launch_bounds__(64 * kStemWaves, 2) void k_conv_stem_bf16(
    const __bf16 *__restrict__ src, const void *__restrict__ w_, int w_is_f32, int transposed, int volume,
    const int32_t *__restrict__ tbl, const int32_t *__restrict__ col_order, const int32_t *__restrict__ out_order,
    __bf16 *__restrict__ dst, int64_t n_tgt, float *__restrict__ stat_mean, float *__restrict__ stat_m2) {
  constexpr int NC = CB * 16;
  constexpr int TILE = 16 * G * kStemWaves;
  constexpr int D = kStemIdxAhead;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nq4 = (volume + 15) / 16 * 4;                  // offset quads, rounded up to the unroll of the walk
  bf16x8 *s_w = reinterpret_cast<bf16x8 *>(smem);          // [nq4][CB][64 lanes]
  float *s_sh = reinterpret_cast<float *>(smem + (size_t)nq4 * CB * 64 * 16);   // [2][NC]   first row of the tile (the shift)
  float *s_st = s_sh + 2 * NC;                                                   // [2][waves][NC][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i16 = lane & 15, q = lane >> 4;

  if (w_is_f32) stem_stage_weights<CB>(s_w, reinterpret_cast<const float *>(w_), transposed, volume, nq4, tid);
  else stem_stage_weights<CB>(s_w, reinterpret_cast<const __bf16 *>(w_), transposed, volume, nq4, tid);
  __syncthreads();

  const int64_t tiles = (n_tgt + TILE - 1) / TILE;
  const uint32_t k_pitch = (uint32_t)n_tgt * 4u;                       // bytes between two offsets of the table
  const uint32_t k_last = (uint32_t)(volume - 1) * k_pitch;
  const __bf16 z = (__bf16)0.f;
  const bf16x8 zero = bf16x8{z, z, z, z, z, z, z, z};
  int parity = 0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x, parity ^= 1) {
    const int64_t row0 = tile * TILE + wave * (16 * G);
    uint32_t col4[G];      // byte offset of this lane's table column, per 16-row group
    int64_t prow[G];       // tile position of the lane's row
    bool ok[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      prow[g] = row0 + g * 16 + i16;
      ok[g] = prow[g] < n_tgt;
      const int64_t pc = ok[g] ? prow[g] : n_tgt - 1;
      col4[g] = (uint32_t)(col_order ? (int64_t)col_order[pc] : pc) * 4u;
    }
    f32x4 acc[G][CB];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- the walk over the offset quads.  Rings of four slots (slot = quad mod 4): neighbour indices ix, "no neighbour"
    // flags, operand rows x.  Step jj: rows of quad jj + 2 are requested (their indices arrived), indices of quad jj + D,
    // then quad jj is multiplied.  Every request is issued unconditionally (clamped), so the counted waits hold.
    // Every ring register is a TIED operand of its load ("+v"): one register per slot through the whole walk — a copy of a
    // register whose load is still in flight (what a rotated ring costs at the loop's back edge) would read stale data.
    int32_t ix[4][G];
    bf16x8 x[4][G];
    uint32_t keep[4][G];   // all ones / zero: the row has a neighbour at the quad's offset (made opaque: not recomputed from ix later)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        ix[d][g] = 0;
        x[d][g] = zero;
      }
    }
    uint32_t kq = (uint32_t)q * k_pitch;            // byte offset of offset 4 jj + q, clamped to the last offset below
    auto issue_idx = [&](int32_t (&dst_ix)[G]) {
      const uint32_t kk = min(kq, k_last);
      kq += 4u * k_pitch;
#pragma unroll
      for (int g = 0; g < G; ++g)
        asm volatile("global_load_dword %0, %1, %2" : "+v"(dst_ix[g]) : "v"(kk + col4[g]), "s"(tbl) : "memory");
    };
    auto issue_rows = [&](const int32_t (&src_ix)[G], bf16x8 (&dst_x)[G], uint32_t (&dst_keep)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        dst_keep[g] = ~(uint32_t)(src_ix[g] >> 31);
        asm volatile("" : "+v"(dst_keep[g]));
        const uint32_t off = (uint32_t)max(src_ix[g], 0) << 4;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst_x[g]) : "v"(off), "s"(src) : "memory");
      }
    };
    // prologue: indices of quads 0 .. D-1, rows of quads 0 and 1 (plain waits: once per tile)
#pragma unroll
    for (int d = 0; d < 4; ++d) issue_idx(ix[d]);
#pragma unroll
    for (int g = 0; g < G; ++g) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ix[0][g]), "+v"(ix[1][g]) : : "memory");
    // (in the order of the steady state — rows of quad s, indices of quad s + 4 —: the counted waits below count on it)
    issue_rows(ix[0], x[0], keep[0]);
    issue_idx(ix[0]);      // quad 4
    issue_rows(ix[1], x[1], keep[1]);
    issue_idx(ix[1]);      // quad 5
    static_assert(D == 6, "the prologue and the counted waits below are written for six quads of index look-ahead");
    for (int j = 0; j < nq4; j += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        constexpr int kYoungerIdx = 2 * G * (D - 3);     // requests issued after the indices of quad jj + 2
        constexpr int kYoungerRows = 5 * G;              // ... after the rows of quad jj, at the point they are used
        const int s2 = (u + 2) & 3;
        // A: the indices of quad jj + 2 have arrived -> request its rows
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ix[s2][g]) : "n"(kYoungerIdx) : "memory");
        issue_rows(ix[s2], x[s2], keep[s2]);
        // B: indices of quad jj + D (slot (jj + D) mod 4 = the slot just consumed)
        issue_idx(ix[s2]);
        // C: the rows of quad jj have arrived -> multiply
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[u][g]) : "n"(kYoungerRows) : "memory");
        bf16x8 wf[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) wf[c] = s_w[((j + u) * CB + c) * 64 + lane];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bf16x8 xv = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, x[u][g]) & keep[u][g]);
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xv, acc[g][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the surplus requests behind the last quad)

    // ---- epilogue of the tile: one rounding to bf16, 8-byte stores straight from the accumulators (lane (i16, q) holds
    // row i16, columns 4 q .. 4 q + 3 of every 16-column block); batch-norm partials of the STORED values: mean and M2
    // about the tile's first row, rows summed in a fixed order (16-row groups in a lane, then a butterfly over the rows
    // of a group, then the waves)
    bf16x4 ov[G][CB];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const f32x4 v = acc[g][c];
        ov[g][c] = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      }
      if (ok[g]) {
        const int64_t grow = out_order ? (int64_t)out_order[prow[g]] : prow[g];
#pragma unroll
        for (int c = 0; c < CB; ++c) *reinterpret_cast<bf16x4 *>(dst + grow * NC + c * 16 + q * 4) = ov[g][c];
      }
    }
    if (stat_mean != nullptr) {
      float *sh = s_sh + parity * NC;
      float *stp = s_st + parity * (kStemWaves * NC * 2);
      if (wave == 0 && i16 == 0) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
#pragma unroll
          for (int t = 0; t < 4; ++t) sh[c * 16 + q * 4 + t] = (float)ov[0][c][t];
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CB; ++c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float shift = sh[c * 16 + q * 4 + t];
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const float d = ok[g] ? (float)ov[g][c][t] - shift : 0.f;
            a += d;
            b = fmaf(d, d, b);
          }
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            a += __shfl_xor(a, m);
            b += __shfl_xor(b, m);
          }
          if (i16 == 0) {
            stp[(wave * NC + c * 16 + q * 4 + t) * 2] = a;
            stp[(wave * NC + c * 16 + q * 4 + t) * 2 + 1] = b;
          }
        }
      }
      __syncthreads();
      if (tid < NC) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int wv = 0; wv < kStemWaves; ++wv) {
          a += stp[(wv * NC + tid) * 2];
          b += stp[(wv * NC + tid) * 2 + 1];
        }
        const float cnt = (float)min((int64_t)TILE, n_tgt - tile * TILE), m = a / cnt;
        stat_mean[tile * NC + tid] = sh[tid] + m;
        stat_m2[tile * NC + tid] = fmaxf(b - a * m, 0.f);
      }
    }
  }
}

int g_stem_mode = -1;    // me_debug_set_stem: -1 policy (ME_AMD_STEM), 0 never, 1 wherever the shape is supported
int g_stem_g = 0;        // 16-row groups per wave: 0 policy (4: tiles of 256 rows), 1 | 2 | 4

// (MI355X, 200k voxels: 3 -> 32 over 5^3 offsets 58 / 45 / 40 us with 1 / 2 / 4 groups; profiles/r05_stem_sweep.txt)
static int stem_groups() { return g_stem_g ? g_stem_g : 4; }

static bool stem_shape(int64_t n_tgt, int64_t volume, int c_src, int c_dst) {
  return volume >= 1 && volume <= 512 && c_src == 8 && (c_dst == 16 || c_dst == 32 || c_dst == 64) &&
         conv_stem_lds((int)((volume + 15) / 16 * 4), c_dst / 16) <= kLdsBudget / 2 &&
         // 32-bit byte offsets into the neighbour table — of the clamped look-ahead as well: the walk counts up to offset
         // volume + 42 before it clamps to the last one
         (volume + 43) * std::max<int64_t>(n_tgt, 1) < (1ll << 30);
}

}  // namespace me


using namespace me;

extern "C" void me_debug_set_stem(int mode, int groups) {
  g_stem_mode = mode;
  g_stem_g = (groups == 1 || groups == 2 || groups == 4) ? groups : 0;
}

// 1: this launch runs on the stacked-offset kernel (feature rows of exactly 8 channels — pad with zeros —, 16 / 32 / 64
// output channels, bf16 features; ME_AMD_STEM=0 keeps the tile-plan kernels).  The policy both hosts follow.
extern "C" int32_t me_conv_stem_use_bf16(int64_t n_tgt, int64_t volume, int32_t c_src, int32_t c_dst) {
  if (n_tgt <= 0 || !stem_shape(n_tgt, volume, c_src, c_dst)) return 0;
  int mode = g_stem_mode;
  if (mode < 0) {
    static int env_mode = -2;
    if (env_mode == -2) {
      const char *e = getenv("ME_AMD_STEM");
      env_mode = (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : -1);
    }
    mode = env_mode;
  }
  if (mode >= 0) return mode ? 1 : 0;
  return volume >= 8 ? 1 : 0;     // (a 1 x 1 layer has nothing to stack)
}

extern "C" int32_t me_conv_stem_tile_rows(void) { return 16 * stem_groups() * kStemWaves; }

// dst[t, :] = sum over offsets k of src[tbl[k][col(t)], :] @ w[k]   for feature rows of exactly 8 bf16 channels (16 bytes:
// pad with zeros), w the layer's own kernel tensor [volume, 8, c_dst] (transposed != 0: [volume, c_dst, 8]) in fp32 or
// bf16 — no packed image.  tbl / col_order / out_order as me_conv_halo_bf16; part_* per tile of me_conv_stem_tile_rows()
// rows, or NULL.
extern "C" int me_conv_stem_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const void *w_dev, int32_t w_is_f32,
                                 int32_t transposed, int64_t volume, int32_t c_dst, const int32_t *tbl_dev,
                                 const int32_t *col_order_dev, const int32_t *out_order_dev, uint16_t *dst_feat_dev,
                                 int64_t n_tgt, float *part_mean_dev, float *part_m2_dev, void *stream) {
  ME_CHECK((part_mean_dev == nullptr) == (part_m2_dev == nullptr), "statistics: both partial arrays or none");
  if (n_tgt > 0 && n_src <= 0) {
    // a source side without rows: every neighbour is absent -> zeros (and statistics of zeros)
    ME_CHECK(dst_feat_dev != nullptr && c_dst > 0, "null argument");
    hipStream_t st0 = (hipStream_t)stream;
    ME_HIP(hipMemsetAsync(dst_feat_dev, 0, (size_t)n_tgt * c_dst * 2, st0));
    if (part_mean_dev != nullptr) {
      const size_t bytes = (size_t)ceil_div(n_tgt, me_conv_stem_tile_rows()) * c_dst * 4;
      ME_HIP(hipMemsetAsync(part_mean_dev, 0, bytes, st0));
      ME_HIP(hipMemsetAsync(part_m2_dev, 0, bytes, st0));
    }
    return 0;
  }
  ME_CHECK(src_feat_dev && w_dev && tbl_dev && dst_feat_dev, "null argument");
  ME_CHECK(c_src == 8, "stacked-offset kernel: feature rows must be padded to 8 channels (16 bytes)");
  ME_CHECK(stem_shape(n_tgt, volume, c_src, c_dst), "no stacked-offset kernel for this shape");
  ME_CHECK((uintptr_t)src_feat_dev % 16 == 0 && (uintptr_t)dst_feat_dev % 8 == 0, "stacked-offset kernel: feature rows must be 16-byte aligned");
  ME_CHECK(n_src < (1ll << 28), "stacked-offset kernel: source rows");
  if (n_tgt <= 0) return 0;
  const int nq4 = (int)((volume + 15) / 16 * 4), cb = c_dst / 16, g = stem_groups();
  const int lds = conv_stem_lds(nq4, cb);
  const int64_t tiles = ceil_div(n_tgt, 16 * g * kStemWaves);
  // persistent workgroups (the weights are staged once per workgroup), two per CU; the same number of tiles for all
  const int64_t slots = (int64_t)device_cu_count() * 2;
  const int64_t rounds = ceil_div(tiles, slots);
  const dim3 grid((unsigned)ceil_div(tiles, rounds));
  hipStream_t st = (hipStream_t)stream;
  const __bf16 *src = reinterpret_cast<const __bf16 *>(src_feat_dev);
  __bf16 *dst = reinterpret_cast<__bf16 *>(dst_feat_dev);
#define ME_STEM(CBV, GV)                                                                                                    \
  if (cb == CBV && g == GV) {                                                                                               \
    static bool attr_set = false;                                                                                           \
    if (lds > 48 * 1024 && !attr_set) {                                                                                     \
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_stem_bf16<CBV, GV>),                                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget / 2));                              \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((k_conv_stem_bf16<CBV, GV>), grid, dim3(64 * kStemWaves), (size_t)lds, st, src, w_dev, w_is_f32,     \
                       transposed, (int)volume, tbl_dev, col_order_dev, out_order_dev, dst, n_tgt, part_mean_dev,           \
                       part_m2_dev);                                                                                        \
    ME_LAUNCH_CHECK();                                                                                                      \
    return 0;                                                                                                               \
  }
  ME_STEM(1, 1) ME_STEM(1, 2) ME_STEM(1, 4)
  ME_STEM(2, 1) ME_STEM(2, 2) ME_STEM(2, 4)
  ME_STEM(4, 1) ME_STEM(4, 2) ME_STEM(4, 4)
#undef ME_STEM
  ME_FAIL("no stacked-offset kernel instantiation for this shape");
}

// code-object preload (me_preload, coords.hip)
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv_stem(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_conv_stem_bf16<2, 4>));
}
