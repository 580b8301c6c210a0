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
#include "common.hpp"

namespace me {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGB = 4;            // groups (of 16 gathered rows) per LDS batch
constexpr int kRows = kGB * 16;   // gathered rows per batch
constexpr int kLdsBudget = 160 * 1024;

constexpr int kAccPad = 4;  // accumulator row stride NC + 4 floats: spreads the row-scattered adds over banks

// LDS bytes of one workgroup of k_conv_target_f32<NC, KC> with `tile_rows` target rows
// (accumulator tile + one dummy row for padding slots, gathered-row tile, plan slice)
__host__ __device__ constexpr int conv_lds_bytes(int NC, int KC, int tile_rows) {
  return (tile_rows + 1) * (NC + kAccPad) * 4 + kRows * (KC + 4) * 4 + kRows * 4 + kGB * 4 + 32;
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
template <int R, int KQ, int A_LD, int ACC_LD>
__device__ __forceinline__ void mma_groups(const float *__restrict__ a0p, const float (&wreg)[KQ],
                                           const int32_t *__restrict__ dstp, float *__restrict__ accp) {
  int d[R];
#pragma unroll
  for (int r = 0; r < R; ++r) d[r] = dstp[r * 16];
  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s4 = 0; s4 < KQ / 4; ++s4) {
    f32x4 a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = *reinterpret_cast<const f32x4 *>(a0p + r * 16 * A_LD + s4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s4 * 4 + j], a[r][j], acc[r], 0, 0, 0);
    }
  }
  f32x4 old[R];
#pragma unroll
  for (int r = 0; r < R; ++r) old[r] = *reinterpret_cast<const f32x4 *>(accp + d[r] * ACC_LD);
#pragma unroll
  for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4 *>(accp + d[r] * ACC_LD) = old[r] + acc[r];
}

// Packed weights: the exact register image of the kernel.  For offset k, source-channel chunk c,
// 16-column block cb and k-step quad v, lane (q = lane >> 4, i16 = lane & 15) finds its four weights
//   W[k][c*KC + q*KQ + v*4 + j][cb*16 + i16],  j = 0..3
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
    const int ch = c * KC + q * KQ + v * 4 + j;
    float val = 0.f;
    if (ch < c_src && col < c_dst) {
      // plain: w is [K, c_src, c_dst]; transposed (dgrad): w is the forward kernel [K, c_dst, c_src]
      val = transposed ? w[(k * c_dst + col) * c_src + ch] : w[(k * c_src + ch) * c_dst + col];
    }
    out[j] = val;
  }
  wp[e] = out;
}

// VAR bits (tuning / ablation; 0 is the shipped configuration):
//   2: runs of up to 4 groups instead of 2 (more accumulators in flight, more registers)
//   4: plan indices fetched one batch ahead only (dependent index -> row load chain per batch)
// EXACT: c_src is a multiple of KC (and of 4): the gather needs no channel guards.
template <int NC, int KC, int VAR, bool EXACT>
__global__ __launch_bounds__(NC * 4, 3) void k_conv_target_f32(
    const float *__restrict__ src, int c_src, const f32x4 *__restrict__ wp, int c_dst,
    const int32_t *__restrict__ plan_src, const int32_t *__restrict__ plan_dst,
    const int32_t *__restrict__ group_k, const int32_t *__restrict__ tile_gptr,
    float *__restrict__ dst, int64_t n_tgt, int tile_rows) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int WAVES = NC / 16;
  constexpr int NT = WAVES * 64;
  constexpr int A_LD = KC + 4;         // floats; +16 B per row spreads ds_read_b128 over the banks
  constexpr int ACC_LD = NC + kAccPad;
  constexpr int KQ = KC / 4;           // MFMA k-steps per chunk (= weight registers per lane)
  constexpr int F4_PER_ROW = KC / 4;   // 16-byte pieces per gathered row
  constexpr int ITER = kRows * F4_PER_ROW / NT;
  constexpr int MAXRUN = (VAR & 2) ? 4 : 2;
  static_assert(kRows * F4_PER_ROW % NT == 0, "gather work must divide evenly");
  static_assert(KC % 16 == 0, "KC must be a multiple of 16");
  static_assert(kGB == 4, "the batch metadata is read as one int4");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);              // [(tile_rows + 1) x ACC_LD]
  float *s_a = s_acc + (tile_rows + 1) * ACC_LD;               // [kRows x A_LD]
  int32_t *s_dst = reinterpret_cast<int32_t *>(s_a + kRows * A_LD);  // [kRows]
  int32_t *s_k = s_dst + kRows;                                // [kGB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i16 = lane & 15;  // gathered row (MFMA B column) / weight column (MFMA A row) of this lane
  const int q = lane >> 4;    // MFMA k index of this lane; after the MFMA: output columns q*4 .. q*4+3
  const int tile = blockIdx.x;
  const int col_base = blockIdx.y * NC;
  const int g_begin = tile_gptr[tile];
  const int g_end = tile_gptr[tile + 1];
  const bool vec_ok = (c_src % 4) == 0;
  const int nchunks = (c_src + KC - 1) / KC;
  const int ncb = (c_dst + 15) / 16;
  const int cb = col_base / 16 + wave;   // this wave's 16-column block (may lie beyond c_dst: zeros)

  for (int x = tid; x < (tile_rows + 1) * ACC_LD / 4; x += NT)
    reinterpret_cast<f32x4 *>(s_acc)[x] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0, chunk = 0; c0 < c_src; c0 += KC, ++chunk) {
    float wreg[KQ];
    int cur_k = -1;
    // software pipeline registers: gathered rows of the NEXT batch, its plan slice, and the source
    // row indices of the batch after that (so no load in the loop waits for another load)
    f32x4 stage[ITER];
    int32_t sidx[ITER];
    int32_t dst_r = tile_rows, k_r = -1;

    auto load_w = [&](int k) {
      if (cb < ncb) {
        const f32x4 *p = wp + ((((int64_t)k * nchunks + chunk) * ncb + cb) * (KQ / 4)) * 64 + lane;
#pragma unroll
        for (int v = 0; v < KQ / 4; ++v) {
          const f32x4 t = p[v * 64];
          wreg[v * 4 + 0] = t.x;
          wreg[v * 4 + 1] = t.y;
          wreg[v * 4 + 2] = t.z;
          wreg[v * 4 + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < KQ; ++s) wreg[s] = 0.f;
      }
    };
    auto load_idx = [&](int gb) {
      const int nrows = (gb < g_end) ? min(kGB, g_end - gb) * 16 : 0;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int r = (it * NT + tid) / F4_PER_ROW;
        sidx[it] = (r < nrows) ? plan_src[(int64_t)gb * 16 + r] : -1;
      }
    };
    auto load_meta = [&](int gb) {
      const int ng = min(kGB, g_end - gb);
      if (tid < kRows) dst_r = (tid < ng * 16) ? plan_dst[(int64_t)gb * 16 + tid] : tile_rows;
      if (tid < kGB) k_r = (tid < ng) ? group_k[gb + tid] : -1;
    };
    // issue the gather loads of the batch whose indices sit in sidx (global -> registers)
    auto gather_issue = [&]() {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int ch = c0 + ((it * NT + tid) % F4_PER_ROW) * 4;
        const int s = sidx[it];
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (EXACT) {
          // unconditional 16-byte load (padding slots read row 0 and are zeroed by a select)
          const f32x4 ld = *reinterpret_cast<const f32x4 *>(src + (int64_t)max(s, 0) * c_src + ch);
          t = (s >= 0) ? ld : t;
        } else if (s >= 0 && ch < c_src) {
          const float *rowp = src + (int64_t)s * c_src + ch;
          if (vec_ok) {
            t = *reinterpret_cast<const f32x4 *>(rowp);
          } else {
            t.x = rowp[0];
            if (ch + 1 < c_src) t.y = rowp[1];
            if (ch + 2 < c_src) t.z = rowp[2];
            if (ch + 3 < c_src) t.w = rowp[3];
          }
        }
        stage[it] = t;
      }
    };

    if (g_begin < g_end) {
      load_idx(g_begin);
      load_meta(g_begin);
      gather_issue();
      if (!(VAR & 4)) load_idx(g_begin + kGB);
    }

    for (int gb = g_begin; gb < g_end; gb += kGB) {
      const int ng = min(kGB, g_end - gb);
      __syncthreads();  // consumers of the previous batch are done with s_a / s_dst / s_k
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = it * NT + tid;
        *reinterpret_cast<f32x4 *>(&s_a[(idx / F4_PER_ROW) * A_LD + (idx % F4_PER_ROW) * 4]) = stage[it];
      }
      if (tid < kRows) s_dst[tid] = dst_r;
      if (tid < kGB) s_k[tid] = k_r;
      __syncthreads();
      // the next batch's rows fly while this batch is multiplied; its indices were fetched a batch ago
      if (gb + kGB < g_end) {
        if (VAR & 4) load_idx(gb + kGB);
        load_meta(gb + kGB);
        gather_issue();
        if (!(VAR & 4)) load_idx(gb + 2 * kGB);
      }

      const i32x4 kv = *reinterpret_cast<const i32x4 *>(s_k);
      const int kk0 = __builtin_amdgcn_readfirstlane(kv.x), kk1 = __builtin_amdgcn_readfirstlane(kv.y);
      const int kk2 = __builtin_amdgcn_readfirstlane(kv.z), kk3 = __builtin_amdgcn_readfirstlane(kv.w);
      auto ksel = [&](int g) { return g == 0 ? kk0 : (g == 1 ? kk1 : (g == 2 ? kk2 : kk3)); };

      int g = 0;
      while (g < ng) {
        const int k0 = ksel(g);
        int run = 1;
        while (run < MAXRUN && g + run < ng && ksel(g + run) == k0) ++run;
        if (k0 != cur_k) {
          cur_k = k0;
          load_w(k0);
        }
        const float *a0p = &s_a[(g * 16 + i16) * A_LD + q * KQ];
        const int32_t *dstp = &s_dst[g * 16 + i16];
        // after the MFMA this lane holds columns wave*16 + q*4 .. +3 of target row s_dst[g*16 + i16];
        // columns are private to this wave -> plain LDS read-add-write in a fixed order (bitwise
        // reproducible); padding slots land in the dummy row `tile_rows`.
        float *accp = &s_acc[wave * 16 + q * 4];
        if (MAXRUN > 2 && run == 4) mma_groups<4, KQ, A_LD, ACC_LD>(a0p, wreg, dstp, accp);
        else if (MAXRUN > 2 && run == 3) mma_groups<3, KQ, A_LD, ACC_LD>(a0p, wreg, dstp, accp);
        else if (run == 2) mma_groups<2, KQ, A_LD, ACC_LD>(a0p, wreg, dstp, accp);
        else mma_groups<1, KQ, A_LD, ACC_LD>(a0p, wreg, dstp, accp);
        g += run;
      }
    }
    __syncthreads();  // all reads of s_a done before the next chunk restages it
  }

  __syncthreads();
  const int64_t row0 = (int64_t)tile * tile_rows;
  const int rows_here = (int)min((int64_t)tile_rows, n_tgt - row0);
  const bool vec_out = (c_dst % 4) == 0;
  for (int x = tid; x < tile_rows * NC / 4; x += NT) {
    const int row = x / (NC / 4);
    const int c4 = x % (NC / 4);
    const int cc = col_base + c4 * 4;
    if (row < rows_here && cc < c_dst) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_acc[row * ACC_LD + c4 * 4]);
      float *o = dst + (row0 + row) * c_dst + cc;
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
constexpr int kWgPB = 32;       // pairs per LDS batch
constexpr int kWgLD = 64 + 16;  // floats per staged row (stride = 16 mod 32 banks)

// chunk -> (k, first pair, last pair): chunks are ME_WGRAD_CHUNK pairs of one offset
__device__ __forceinline__ void wgrad_locate_chunk(const int64_t *__restrict__ koffs, int volume,
                                                   int chunk, int &k_out, int64_t &e0, int64_t &e1) {
  int c = 0;
  k_out = -1;
  e0 = e1 = 0;
  for (int k = 0; k < volume; ++k) {
    const int64_t b = koffs[k], e = koffs[k + 1];
    const int nck = (int)((e - b + ME_WGRAD_CHUNK - 1) / ME_WGRAD_CHUNK);
    if (chunk < c + nck) {
      k_out = k;
      e0 = b + (int64_t)(chunk - c) * ME_WGRAD_CHUNK;
      e1 = min(e, e0 + ME_WGRAD_CHUNK);
      return;
    }
    c += nck;
  }
}

__global__ __launch_bounds__(256) void k_wgrad_f32(const float *__restrict__ x, int c_in,
                                                  const float *__restrict__ dy, int c_out,
                                                  const int32_t *__restrict__ in_pairs,
                                                  const int32_t *__restrict__ out_pairs,
                                                  const int64_t *__restrict__ koffs, int volume,
                                                  float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float s_x[kWgPB * kWgLD];
  __shared__ __attribute__((aligned(16))) float s_y[kWgPB * kWgLD];
  __shared__ int s_k;
  __shared__ int64_t s_e[2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x;
  const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  if (tid == 0) {
    int k;
    int64_t e0, e1;
    wgrad_locate_chunk(koffs, volume, chunk, k, e0, e1);
    s_k = k;
    s_e[0] = e0;
    s_e[1] = e1;
  }
  __syncthreads();
  const int64_t e0 = s_e[0], e1 = s_e[1];
  const int wm = wave >> 1, wn = wave & 1;
  const bool vx = (c_in % 4) == 0, vy = (c_out % 4) == 0;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // 32 rows x 16 float4 per operand = 512 float4; 256 threads -> 2 per thread per operand
  f32x4 sx[2], sy[2];
  auto issue = [&](int64_t eb) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx >> 4;
      const int p = idx & 15;
      const int64_t e = eb + r;
      f32x4 tx = {0.f, 0.f, 0.f, 0.f}, ty = {0.f, 0.f, 0.f, 0.f};
      if (e < e1) {
        const int ci = ci0 + p * 4, co = co0 + p * 4;
        if (ci < c_in) {
          const float *xp = x + (int64_t)in_pairs[e] * c_in + ci;
          if (vx) {
            tx = *reinterpret_cast<const f32x4 *>(xp);
          } else {
            tx.x = xp[0];
            if (ci + 1 < c_in) tx.y = xp[1];
            if (ci + 2 < c_in) tx.z = xp[2];
            if (ci + 3 < c_in) tx.w = xp[3];
          }
        }
        if (co < c_out) {
          const float *yp = dy + (int64_t)out_pairs[e] * c_out + co;
          if (vy) {
            ty = *reinterpret_cast<const f32x4 *>(yp);
          } else {
            ty.x = yp[0];
            if (co + 1 < c_out) ty.y = yp[1];
            if (co + 2 < c_out) ty.z = yp[2];
            if (co + 3 < c_out) ty.w = yp[3];
          }
        }
      }
      sx[it] = tx;
      sy[it] = ty;
    }
  };

  if (e0 < e1) issue(e0);
  for (int64_t eb = e0; eb < e1; eb += kWgPB) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx >> 4, p = idx & 15;
      *reinterpret_cast<f32x4 *>(&s_x[r * kWgLD + p * 4]) = sx[it];
      *reinterpret_cast<f32x4 *>(&s_y[r * kWgLD + p * 4]) = sy[it];
    }
    __syncthreads();
    if (eb + kWgPB < e1) issue(eb + kWgPB);
    const float *ap = &s_x[(lane >> 5) * kWgLD + wm * 32 + (lane & 31)];
    const float *bp = &s_y[(lane >> 5) * kWgLD + wn * 32 + (lane & 31)];
#pragma unroll
    for (int s = 0; s < kWgPB / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * 2 * kWgLD], bp[s * 2 * kWgLD], acc, 0, 0, 0);
  }

  float *pp = partial + (int64_t)chunk * c_in * c_out;
  const int colo = co0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < c_in && colo < c_out) pp[(int64_t)row * c_out + colo] = acc[r];
  }
}

// grad_w[k][idx] = sum of the partial tiles of offset k's chunks, in chunk order (deterministic)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial,
                                                     const int64_t *__restrict__ koffs, int volume,
                                                     int64_t cc, float *__restrict__ grad_w) {
  __shared__ int s_c[2];
  const int k = blockIdx.y;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int kk = 0; kk < k; ++kk)
      c += (int)((koffs[kk + 1] - koffs[kk] + ME_WGRAD_CHUNK - 1) / ME_WGRAD_CHUNK);
    s_c[0] = c;
    s_c[1] = c + (int)((koffs[k + 1] - koffs[k] + ME_WGRAD_CHUNK - 1) / ME_WGRAD_CHUNK);
  }
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cc) return;
  float s = 0.f;
  for (int c = s_c[0]; c < s_c[1]; ++c) s += partial[(int64_t)c * cc + idx];
  grad_w[(int64_t)k * cc + idx] = s;
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
  int nc;  // output columns per workgroup (64 or 32); waves per workgroup = nc / 16
  int kc;  // source-channel chunk (64, 32 or 16)
};

static ConvVariant conv_variant(int c_src, int c_dst) {
  ConvVariant v;
  // columns per workgroup: 64 unless the last 64-column slab would be at most half full
  const int rem = c_dst % 64;
  v.nc = (rem != 0 && rem <= 32) ? 32 : 64;
  // source-channel chunk: the largest of {64, 32, 16} that adds at most 16 channels of padding
  if (c_src % 64 == 0) v.kc = 64;
  else if (c_src % 32 == 0) v.kc = 32;
  else if (c_src <= 16) v.kc = 16;
  else if (c_src <= 32) v.kc = 32;
  else v.kc = (align_up(c_src, 64) - c_src <= 16) ? 64 : ((align_up(c_src, 32) - c_src <= 16) ? 32 : 16);
  return v;
}

static int device_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;  // MI355X
  }
  return cus;
}

int g_conv_variant = 0;  // me_debug_set_conv_variant

template <int NC, int KC, int VAR>
static int launch_conv_target(const float *src, int c_src, const float *wp, int c_dst,
                              const int32_t *plan_src, const int32_t *plan_dst, const int32_t *group_k,
                              const int32_t *tile_gptr, float *dst, int64_t n_tgt, int tile_rows,
                              hipStream_t stream) {
  const int lds = conv_lds_bytes(NC, KC, tile_rows);
  ME_CHECK(lds <= kLdsBudget, "tile_rows too large for the LDS of one workgroup");
  const bool exact = (c_src % KC) == 0;
  static int max_lds_set[2] = {0, 0};  // per instantiation
  if (lds > 64 * 1024 && lds > max_lds_set[exact]) {
    const void *fn = exact ? reinterpret_cast<const void *>(&k_conv_target_f32<NC, KC, VAR, true>)
                           : reinterpret_cast<const void *>(&k_conv_target_f32<NC, KC, VAR, false>);
    ME_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    max_lds_set[exact] = kLdsBudget;
  }
  const dim3 grid((unsigned)ceil_div(n_tgt, tile_rows), (unsigned)ceil_div(c_dst, NC));
  const f32x4 *wp4 = reinterpret_cast<const f32x4 *>(wp);
  if (exact)
    hipLaunchKernelGGL((k_conv_target_f32<NC, KC, VAR, true>), grid, dim3(NC * 4), (size_t)lds, stream, src,
                       c_src, wp4, c_dst, plan_src, plan_dst, group_k, tile_gptr, dst, n_tgt, tile_rows);
  else
    hipLaunchKernelGGL((k_conv_target_f32<NC, KC, VAR, false>), grid, dim3(NC * 4), (size_t)lds, stream, src,
                       c_src, wp4, c_dst, plan_src, plan_dst, group_k, tile_gptr, dst, n_tgt, tile_rows);
  ME_LAUNCH_CHECK();
  return 0;
}

static int64_t wgrad_num_chunks(const int64_t *k_offsets, int64_t volume) {
  int64_t c = 0;
  for (int64_t k = 0; k < volume; ++k) c += ceil_div(k_offsets[k + 1] - k_offsets[k], ME_WGRAD_CHUNK);
  return c;
}

}  // namespace me

using namespace me;

extern "C" {

int32_t me_conv_choose_tile_rows(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src,
                                 int32_t c_dst) {
  if (n_tgt <= 0 || volume <= 0 || c_src <= 0 || c_dst <= 0) return 128;
  const ConvVariant v = conv_variant(c_src, c_dst);
  const int waves = v.nc / 16;
  const int64_t slabs = ceil_div(c_dst, v.nc);
  const int chunks = (int)ceil_div(c_src, v.kc);
  // resident workgroups per CU: 4 waves per SIMD by registers (3 for the <32,64> variant), then
  // whatever the LDS allows
  const int occ_regs = ((v.nc == 32 && v.kc == 64) ? 12 : 16) / waves;
  const int cus = device_cu_count();
  // occupancy p of a neighbour offset (centre excluded) -> expected 16-row groups of a (tile, k)
  const double p = volume > 1 ? (double)(n_pairs > n_tgt ? n_pairs - n_tgt : 0) / ((double)(volume - 1) * n_tgt)
                              : 0.0;
  double best_cost = 1e300;
  int best_t = 0;
  for (int occ = occ_regs; occ >= 1; --occ) {
    const int64_t slots = (int64_t)cus * occ;
    for (int rounds = 1; rounds <= 64; ++rounds) {
      int64_t t = ceil_div(n_tgt * slabs, slots * rounds);
      if (t < ME_GROUP_ROWS) t = ME_GROUP_ROWS;
      if (t > ME_MAX_TILE_ROWS) continue;
      if ((int64_t)conv_lds_bytes(v.nc, v.kc, (int)t) * occ > kLdsBudget) continue;
      const double m = (double)t * p;
      const double g_side = m < 6.0 ? (m <= 0 ? 0.0 : (1.0 - exp(-m)) * (1.0 + m / 16.0)) : m / 16.0 + 0.5;
      const double groups = (double)ceil_div(t, 16) + (double)(volume - 1) * g_side;
      const int64_t items = ceil_div(n_tgt, t) * slabs;
      const double real_rounds = (double)ceil_div(items, slots);
      // time ~ rounds x groups of one item x waves sharing a SIMD (+ a per-batch overhead term)
      const double cost = real_rounds * (groups * chunks) * (double)(occ * waves) / 4.0 *
                          (1.0 + 0.15 * 12.0 / (occ * waves));
      if (cost < best_cost) {
        best_cost = cost;
        best_t = (int)t;
      }
      if (t == ME_GROUP_ROWS) break;
    }
  }
  return best_t > 0 ? best_t : 128;
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
  if (v.kc == 64)
    hipLaunchKernelGGL(k_pack_weights<64>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  else if (v.kc == 32)
    hipLaunchKernelGGL(k_pack_weights<32>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  else
    hipLaunchKernelGGL(k_pack_weights<16>, grid, block, 0, stream, w, c_src, c_dst, transposed, nchunks, ncb, wp4, total);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_conv_target_f32(const float *src, int64_t n_src, int32_t c_src, const float *wp, int64_t volume,
                       int32_t c_dst, const int32_t *plan_src, const int32_t *plan_dst,
                       const int32_t *group_k, const int32_t *tile_gptr, float *dst, int64_t n_tgt,
                       int32_t tile_rows, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)n_src;
  (void)volume;
  ME_CHECK(c_src > 0 && c_dst > 0, "channel counts must be positive");
  ME_CHECK(tile_rows >= ME_GROUP_ROWS && tile_rows <= ME_MAX_TILE_ROWS, "tile_rows out of range");
  ME_CHECK((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)wp % 16 == 0,
           "feature and weight pointers must be 16-byte aligned");
  if (n_tgt == 0) return 0;
  const ConvVariant v = conv_variant(c_src, c_dst);
#define ME_CONV_ARGS src, c_src, wp, c_dst, plan_src, plan_dst, group_k, tile_gptr, dst, n_tgt, tile_rows, stream
  if (g_conv_variant != 0 && v.nc == 64 && v.kc == 64) {  // ablation builds exist for the headline shape only
    switch (g_conv_variant) {
      case 2: return launch_conv_target<64, 64, 2>(ME_CONV_ARGS);
      case 4: return launch_conv_target<64, 64, 4>(ME_CONV_ARGS);
      case 6: return launch_conv_target<64, 64, 6>(ME_CONV_ARGS);
      default: break;
    }
  }
#define ME_CONV_CASE(NCV, KCV) return launch_conv_target<NCV, KCV, 0>(ME_CONV_ARGS)
  if (v.nc == 32) {
    if (v.kc == 64) ME_CONV_CASE(32, 64);
    if (v.kc == 32) ME_CONV_CASE(32, 32);
    ME_CONV_CASE(32, 16);
  } else {
    if (v.kc == 64) ME_CONV_CASE(64, 64);
    if (v.kc == 32) ME_CONV_CASE(64, 32);
    ME_CONV_CASE(64, 16);
  }
#undef ME_CONV_CASE
#undef ME_CONV_ARGS
}

void me_debug_set_conv_variant(int variant) { g_conv_variant = variant; }

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
  int64_t chunks = wgrad_num_chunks(k_offsets, volume);
  if (chunks < 1) chunks = 1;
  return align_up(chunks * (int64_t)c_in * c_out * 4, 256);
}

int me_conv_wgrad_f32(const float *x, int32_t c_in, const float *dy, int32_t c_out, const int32_t *in_pairs,
                      const int32_t *out_pairs, const int64_t *k_offsets, const int64_t *k_offsets_dev,
                      int64_t volume, float *grad_w, void *workspace, int64_t workspace_bytes,
                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(c_in > 0 && c_out > 0, "channel counts must be positive");
  ME_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0, "feature pointers must be 16-byte aligned");
  ME_CHECK(workspace_bytes >= me_conv_wgrad_workspace_bytes(k_offsets, volume, c_in, c_out),
           "workspace too small");
  const int64_t chunks = wgrad_num_chunks(k_offsets, volume);
  float *partial = reinterpret_cast<float *>(workspace);
  if (chunks > 0) {
    const dim3 grid((unsigned)chunks, (unsigned)ceil_div(c_in, 64), (unsigned)ceil_div(c_out, 64));
    hipLaunchKernelGGL(k_wgrad_f32, grid, dim3(256), 0, stream, x, c_in, dy, c_out, in_pairs, out_pairs,
                       k_offsets_dev, (int)volume, partial);
    ME_LAUNCH_CHECK();
  }
  const int64_t cc = (int64_t)c_in * c_out;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)ceil_div(cc, 256), (unsigned)volume), dim3(256), 0,
                     stream, partial, k_offsets_dev, (int)volume, cc, grad_w);
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
