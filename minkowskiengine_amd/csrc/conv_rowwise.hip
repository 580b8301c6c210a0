// Row-wise convolution launches (round 6): kernel-map sides on which every TARGET row has EXACTLY ONE pair.
//
// Three families of layers are such sides (MinkUNet34C: 24 of its 126 forward / input-gradient launches):
//   * kernel volume 1, stride 1 — the reference runs them as `input.F.mm(kernel)`
//     (MinkowskiEngine/MinkowskiConvolution.py:304-308): out[i] = x[i] . W, a dense row-contiguous product;
//   * the FINE side of a kernel_size == stride map (MinkUNet's 2^3 stride-2 layers): the forward launch of the
//     transposed (up-sampling) layer and the input gradient of the strided (down-sampling) one — every fine voxel lies in
//     exactly one window, so out[fine] = x[parent(fine)] . W[child index]: a permuted product, nothing is accumulated;
// The tile-plan kernels (conv_bf16.hip) serve those launches with index windows, padded groups, a staged gather and an
// fp32 accumulator tile in LDS that is read-modified-written per pair — machinery for SUMS over several pairs per row
// that such a side does not have (profiles/r05_layers_minkunet34c_bf16_final.log: 128 -> 96 on 200k rows forward
// 56 us against 11 us of compulsory HBM traffic; 32 -> 32 down-sampling input gradient at 11.5 TFLOP/s).
//
// k_conv_rowwise_bf16<NCB, G>: a workgroup takes an ITEM = 64 G consecutive pairs of ONE offset k of the pair list
// (pairs of an offset are contiguous: kernel_map.hpp:40-53 layout); wave w owns G 16-row groups of them and NCB 16-column
// blocks of the output IN REGISTERS.  The x rows go from global memory straight into the MFMA operand registers — lane
// (i16, q) loads channels 32 s + 8 q .. + 7 of row i16 with one 16-byte load, exactly the 16x16x32 B-operand layout —
// through a ring that runs two or four 32-channel steps ahead ACROSS items (a workgroup walks a run of consecutive items;
// the pair indices arrive two items ahead); the weights W[k] come from the layer's packed image (the tile kernels' own:
// pack.hip), staged into LDS once per OFFSET the workgroup meets (coalesced 16-byte copies; conflict-free ds_read_b128
// per fragment) and shared by the four waves.  No index window, no padding groups, no stage buffer, no accumulator in
// LDS, no barrier inside an item.  Every output row is written once (one rounding: fp32 sums over the channels in ascending
// 32-channel steps -> bf16, RNE: the semantics of the tile-plan kernels), as whole rows: the accumulators pass through a
// wave-private LDS tile that turns "4 columns of 16 rows per lane group" into 16-byte pieces along the rows.
//
// Bound: HBM streaming (x rows in, output rows out, 8 bytes of indices per pair); the weights are L2 -> LDS traffic of
// Cin x Cout x 2 bytes per 64 G pairs.
#include <algorithm>

#include "conv_common.hpp"

namespace me {

constexpr int kRwMaxLds = 64 * 1024;  // W[k] of one column slab (policy: beyond it one workgroup per CU is left and the tile-plan kernel wins)

// D: 32-channel steps of x rows in flight per row group (the ring).
// (No batch-norm statistics epilogue: built and measured — per-item butterflies, then running per-lane sums reduced once
// per workgroup; a workgroup walks only 2 - 4 items, so either costs the K = 1 forward launch 13 - 20 us on 200k rows,
// more than the pass over the output it saves: gpurun_out/r06_layers_rowwise_*.log.  The host keeps a K = 1 forward launch
// whose statistics are wanted on the tile-plan kernel unless the map is large: csrc_host/manager.cpp.)
template <int NCB, int G, int D>
__global__ __launch_bounds__(256) void k_conv_rowwise_bf16(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int ks /* 32-channel steps per chunk */,
    int nchunks, int ncb_total, const int32_t *__restrict__ src_rows, const int32_t *__restrict__ tgt_rows,
    const int64_t *__restrict__ koffs, int volume, __bf16 *__restrict__ dst, int c_dst, int items_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 64 * G;
  const int steps = nchunks * ks;
  const int steps_pad = (steps + D - 1) / D * D;
  bf16x8 *s_w = reinterpret_cast<bf16x8 *>(smem);     // [steps][NCB][64 lanes] fragments of W[k], this column slab
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int cb0 = blockIdx.y * NCB;

  // ---- items: offset k and pair range [e0, e1) of item b (every wave computes it: no LDS, no barrier) ----
  const int64_t lo = lane < volume ? koffs[lane] : 0, hi = lane < volume ? koffs[lane + 1] : 0;
  const uint32_t cnt = (uint32_t)((hi - lo + ROWS - 1) / ROWS);
  const uint32_t incl = wave_inclusive_scan(cnt);
  const uint32_t total = __shfl(incl, 63, 64);
  const uint32_t b0 = blockIdx.x * (uint32_t)items_per_wg;
  if (b0 >= total) return;                             // (the grid is sized by an upper bound of the item count)
  const uint32_t b1 = min(total, b0 + (uint32_t)items_per_wg);
  auto locate = [&](uint32_t b, int &k, int64_t &e0, int64_t &e1) {
    b = min(b, total - 1);
    const unsigned long long owner = __ballot(b >= incl - cnt && b < incl);
    k = __builtin_ctzll(owner);
    const uint32_t excl_k = __shfl(incl - cnt, k, 64);
    e0 = __shfl(lo, k, 64) + (int64_t)(b - excl_k) * ROWS;
    e1 = __shfl(hi, k, 64);
  };
  // pair indices of this wave's row groups of an item (lane (i16, q): row i16 of every group); pairs beyond the item are
  // clamped to its first pair (loaded, multiplied, never stored: target -1)
  auto load_idx = [&](int64_t e0, int64_t e1, int32_t (&sr)[G], int32_t (&tr)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t e = e0 + (wave * G + g) * 16 + i16;
      const bool ok = e < e1;
      sr[g] = src_rows[ok ? e : e0];
      const int32_t t = tgt_rows[ok ? e : e0];
      tr[g] = ok ? t : -1;
    }
  };

  int k_cur, k_nxt, k_nn;
  int64_t e0_cur, e1_cur, e0_nxt, e1_nxt, e0_nn, e1_nn;
  int32_t srow[G], trow[G], srow_nxt[G], trow_nxt[G], srow_nn[G], trow_nn[G];
  locate(b0, k_cur, e0_cur, e1_cur);
  load_idx(e0_cur, e1_cur, srow, trow);
  locate(b0 + 1, k_nxt, e0_nxt, e1_nxt);
  load_idx(e0_nxt, e1_nxt, srow_nxt, trow_nxt);

  // ---- x rows: a ring of D steps per group that runs ACROSS items (slot = step % D; steps_pad is a multiple of D) ----
  bf16x8 xr[D][G];
  const __bf16 *xbase[G], *xbase_nxt[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    xbase[g] = src + (int64_t)srow[g] * c_src + q * 8;
    xbase_nxt[g] = src + (int64_t)srow_nxt[g] * c_src + q * 8;
  }
  // step s' >= steps_pad: step s' - steps_pad of the NEXT item; a padding step / channels beyond the row: a valid
  // address (offset 0), the value is never used / masked at use
  auto request = [&](int sp, int j) {
    const bool next = sp >= steps_pad;
    const int s = next ? sp - steps_pad : sp;
    const int off = (s < steps && s * 32 + q * 8 < c_src) ? s * 32 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g) xr[j][g] = *reinterpret_cast<const bf16x8 *>((next ? xbase_nxt[g] : xbase[g]) + off);
  };
#pragma unroll
  for (int j = 0; j < D; ++j) request(j, j);

  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  int k_staged = -1;
  for (uint32_t b = b0; b < b1; ++b) {
    // ---- W[k], this slab: packed image -> LDS, when the offset changes (a workgroup's items are consecutive) ----
    if (k_cur != k_staged) {                            // (uniform over the workgroup)
      if (k_staged >= 0) __syncthreads();               // everybody is done with the previous offset's fragments
      const int n = steps * NCB * 64;
      constexpr int UNR = 4;
      for (int base = 0; base < n; base += 256 * UNR) {
        bf16x8 w[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = min(base + u * 256 + tid, n - 1);
          const int s = idx / (NCB * 64), r = idx - s * (NCB * 64);
          const int cb = min(r >> 6, ncb_total - 1 - cb0), l = r & 63;      // (a block beyond c_dst: any valid address)
          const int c = s / ks, v = s - c * ks;
          w[u] = wp[((((int64_t)k_cur * nchunks + c) * ncb_total + cb0 + cb) * ks + v) * 64 + l];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = base + u * 256 + tid;
          if (idx < n) s_w[idx] = w[u];    // (a block beyond c_dst holds a copy of the last real block: never stored)
        }
      }
      __syncthreads();
      k_staged = k_cur;
    }
    // indices of the item after the next: in flight during this item's steps
    locate(b + 2, k_nn, e0_nn, e1_nn);
    load_idx(e0_nn, e1_nn, srow_nn, trow_nn);

    f32x4 acc[G][NCB];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[g][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s0 = 0; s0 < steps_pad; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        bf16x8 x[G];
        const bool in_row = s * 32 + q * 8 < c_src;
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = in_row ? xr[j][g] : zero8;
        request(s + D, j);                              // the slot is free: D steps ahead (maybe of the next item)
        if (s < steps) {                                // (uniform; a padding step multiplies nothing)
          const bf16x8 *wv = s_w + (s * NCB) * 64 + lane;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            const bf16x8 a = wv[cb * 64];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x[g], acc[g][cb], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: one rounding to bf16; lane (i16, q) holds columns 16 cb + 4 q .. + 3 of row i16.  The rows leave
    // through a wave-private LDS tile (16 rows x NC columns, pitch + 16 bytes: 16-byte aligned pieces, the 8-byte writes of
    // 16 rows x 4 quads at most two-way conflicting) as 16-BYTE pieces, consecutive lanes = consecutive
    // pieces of a row: whole rows per store instead of 32-byte fragments of 16 different rows — straight 8-byte stores
    // from the accumulators were 8.8 of the launch's 25 us on 200k x 128 -> 96 (scripts/rowwise_bench.py ablations) ----
    constexpr int NC = NCB * 16, PITCH = NC * 2 + 16, PPR = NC / 8;    // bytes per tile row, 16-byte pieces per row
    char *s_y = smem + (size_t)steps * NCB * 1024 + wave * (16 * PITCH);
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const bf16x4 y = {(__bf16)acc[g][cb][0], (__bf16)acc[g][cb][1], (__bf16)acc[g][cb][2], (__bf16)acc[g][cb][3]};
        *reinterpret_cast<bf16x4 *>(s_y + i16 * PITCH + (cb * 16 + q * 4) * 2) = y;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NC / 32; ++it) {
        const int p = it * 64 + lane, row = p / PPR, c8 = p - row * PPR;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(s_y + row * PITCH + c8 * 16);
        const int32_t t = __shfl(trow[g], row, 64);       // (lane `row` = (i16 = row, q = 0) holds the row's target)
        if (t >= 0 && (cb0 * 16 + c8 * 8) < c_dst)         // (c_dst % 8 == 0: a piece is inside the row or outside)
          *reinterpret_cast<u32x4 *>(dst + (int64_t)t * c_dst + cb0 * 16 + c8 * 8) = v;
      }
      __builtin_amdgcn_wave_barrier();                    // (the tile is rewritten by the next group)
    }
    // ---- rotate: next item becomes current ----
    k_cur = k_nxt; e0_cur = e0_nxt; e1_cur = e1_nxt;
    k_nxt = k_nn; e0_nxt = e0_nn; e1_nxt = e1_nn;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      srow[g] = srow_nxt[g]; trow[g] = trow_nxt[g];
      srow_nxt[g] = srow_nn[g]; trow_nxt[g] = trow_nn[g];
      xbase[g] = xbase_nxt[g];
      xbase_nxt[g] = src + (int64_t)srow_nxt[g] * c_src + q * 8;
    }
  }
}

// columns per workgroup slab (16-column blocks): the narrowest instantiation that covers c_dst, 128 columns at most
static int rowwise_ncb(int c_dst) {
  const int ncb = (int)ceil_div(c_dst, 16);
  return ncb <= 2 ? 2 : (ncb <= 4 ? 4 : (ncb <= 6 ? 6 : 8));
}

int g_rw_groups = 0;   // me_debug_set_rowwise_groups: 0 policy, 1 / 2 row groups per wave (tuning)

// launch geometry of a row-wise launch: a pure function of the shape and the pair bound (the host sizes the statistics
// partials by it before the launch)
struct RwGeom {
  int ks, nchunks, ncb_total, steps, ncb, slabs, lds, g, d;
  int64_t items, ipw, grid_x;
};

static RwGeom rowwise_geom(int64_t volume, int c_src, int c_dst, int64_t n_pairs_bound) {
  RwGeom r;
  const int kc = me_conv_pack_chunk_bf16(c_src, c_dst);
  r.ks = kc / 32;
  r.nchunks = (int)ceil_div(c_src, kc);
  r.ncb_total = (int)ceil_div(c_dst, 16);
  r.steps = r.nchunks * r.ks;
  r.ncb = rowwise_ncb(c_dst);
  r.slabs = (int)ceil_div(r.ncb_total, r.ncb);
  r.lds = r.steps * r.ncb * 1024 + 4 * 16 * (r.ncb * 32 + 16);   // W[k] fragments + the four waves' output tiles
  // row groups per wave: 128-pair items where they still fill the chip a few times over, 64-pair items on small maps
  r.g = g_rw_groups ? g_rw_groups : (n_pairs_bound >= (int64_t)device_cu_count() * 4 * 128 ? 2 : 1);
  r.d = r.steps <= 2 ? 2 : 4;
  r.items = ceil_div(n_pairs_bound, 64 * r.g) + volume;     // (every offset may end with a partial item)
  // A workgroup walks `ipw` consecutive items (the weights are staged once per offset it meets); the grid is sized to the
  // workgroups the chip holds at once: by registers four / three / two per CU (64-pair items / 128-pair items on up to
  // 64 columns / wider — `hipcc -Rpass-analysis=kernel-resource-usage`), and by the LDS image of W[k]
  int occ = r.g == 1 ? 4 : (r.ncb <= 4 ? 3 : 2);
  occ = (int)std::max<int64_t>(1, std::min<int64_t>(occ, kLdsBudget / std::max(r.lds, 1)));
  const int64_t slots = (int64_t)device_cu_count() * occ;
  r.ipw = std::max<int64_t>(1, ceil_div(r.items * r.slabs, slots));
  r.grid_x = ceil_div(r.items, r.ipw);
  return r;
}

}  // namespace me

using namespace me;

extern "C" {

void me_debug_set_rowwise_groups(int g) { g_rw_groups = (g == 1 || g == 2) ? g : 0; }

// 1: the row-wise kernel takes a (c_src, c_dst, volume) launch — whole 16-byte pieces of a source and of an output row, the offsets within one wave's scan, W[k] of one column slab within 64 KB of LDS
int32_t me_conv_rowwise_supported_bf16(int64_t volume, int32_t c_src, int32_t c_dst) {
  if (volume < 1 || volume > 64 || c_src < 8 || (c_src % 8) != 0 || c_dst < 8 || (c_dst % 8) != 0) return 0;
  const int kc = me_conv_pack_chunk_bf16(c_src, c_dst);
  if (kc < 32 || (kc % 32) != 0) return 0;
  const int64_t steps = ceil_div(c_src, kc) * (kc / 32);
  return steps * rowwise_ncb(c_dst) * 1024 <= kRwMaxLds ? 1 : 0;
}

int me_conv_rowwise_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const uint16_t *packed_w_dev,
                         int64_t volume, int32_t c_dst, const int32_t *src_rows_dev, const int32_t *tgt_rows_dev,
                         const int64_t *k_offsets_dev, int64_t n_pairs_bound, uint16_t *dst_feat_dev, int64_t n_tgt,
                         void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(me_conv_rowwise_supported_bf16(volume, c_src, c_dst), "shape not supported (me_conv_rowwise_supported_bf16)");
  ME_CHECK(n_pairs_bound >= 0 && n_tgt >= 0 && n_src >= 0, "negative size");
  if (n_pairs_bound == 0 || n_tgt == 0) return 0;
  ME_CHECK(src_feat_dev && packed_w_dev && src_rows_dev && tgt_rows_dev && k_offsets_dev && dst_feat_dev,
           "device pointers must not be null");
  ME_CHECK((uintptr_t)src_feat_dev % 16 == 0 && (uintptr_t)dst_feat_dev % 16 == 0 && (uintptr_t)packed_w_dev % 16 == 0,
           "feature / weight pointers must be 16-byte aligned");
  const RwGeom r = rowwise_geom(volume, c_src, c_dst, n_pairs_bound);
  const int ks = r.ks, nchunks = r.nchunks, ncb_total = r.ncb_total, ncb = r.ncb, lds = r.lds, g = r.g;
  ME_CHECK(r.items < (1ll << 31), "too many pairs for one launch");
  const int64_t ipw = r.ipw;
  const dim3 grid((unsigned)r.grid_x, (unsigned)r.slabs), block(256);
  const __bf16 *src = reinterpret_cast<const __bf16 *>(src_feat_dev);
  const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(packed_w_dev);
  __bf16 *dst = reinterpret_cast<__bf16 *>(dst_feat_dev);
#define ME_RW(NCBV, GV, DV)                                                                                           \
  do {                                                                                                                \
    auto fn = &k_conv_rowwise_bf16<NCBV, GV, DV>;                                                                     \
    static bool attr_set = false;                                                                                     \
    if (lds > 48 * 1024 && !attr_set) {                                                                               \
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                 kLdsBudget));                                                                        \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL(fn, grid, block, (size_t)lds, stream, src, c_src, wp, ks, nchunks, ncb_total, src_rows_dev,    \
                       tgt_rows_dev, k_offsets_dev, (int)volume, dst, c_dst, (int)ipw);                               \
  } while (0)
#define ME_RW_D(NCBV, GV)                \
  do {                                   \
    if (r.d == 2) ME_RW(NCBV, GV, 2);  \
    else ME_RW(NCBV, GV, 4);           \
  } while (0)
#define ME_RW_G(NCBV)             \
  do {                            \
    if (g == 2) ME_RW_D(NCBV, 2); \
    else ME_RW_D(NCBV, 1);        \
  } while (0)
  switch (ncb) {
    case 2: ME_RW_G(2); break;
    case 4: ME_RW_G(4); break;
    case 6: ME_RW_G(6); break;
    default: ME_RW_G(8); break;
  }
#undef ME_RW_G
#undef ME_RW_D
#undef ME_RW
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip)
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv_rowwise(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_conv_rowwise_bf16<2, 1, 2>));
}
