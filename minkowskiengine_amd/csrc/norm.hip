// Batch normalisation over the rows of a sparse tensor's feature matrix [n, c] for gfx950 (MI355X).
//
// The reference's MinkowskiBatchNorm is torch.nn.BatchNorm1d applied to the feature matrix
// (MinkowskiEngine/MinkowskiNormalization.py:35-82); MinkUNet34C runs 62 of them per forward pass (SURVEY 8f
// rank 1).  torch's channels-last statistics kernels take ~70 us per call on a 200k x 32..256 matrix (29 %
// of a bf16 MinkUNet34C step, profiles/r01_rocprof_kernel_stats_minkunet34c_bf16_first.csv), far from the
// HBM time of a 13..100 MB read, so the layer gets its own kernels here.  All of it is HBM-bound streaming:
//
//   statistics  k_bn_partial   one workgroup per chunk of rows; a thread owns a 16-byte channel piece and
//                              strides over the rows of the chunk (coalesced rows), shifted sums
//                              (shift = first row of the chunk) -> (mean, M2) of the chunk and channel
//               k_bn_final     chunks combined per channel with Chan's formula in a FIXED order (bitwise
//                              reproducible), mean / rstd written, running statistics updated
//   forward     k_bn_apply     y = x * a[c] + b[c]   (a = gamma * rstd, b = beta - mean * a)
//   backward    k_bn_bwd_partial / k_bn_bwd_final   sum dy, sum dy * xhat per channel (two levels, fixed order)
//               k_bn_bwd_apply dx = a * (dy - sum_dy / n - xhat * sum_dy_xhat / n)
// T = float or __bf16 rows; statistics and parameters are fp32.
#include "conv_common.hpp"

namespace me {

constexpr int kBnMaxChunks = 512;
constexpr int kBnRowsPerThread = 8;   // fully unrolled: 8 rows in flight per thread (2 or 4 with more workgroups
                                      // measured 2x slower: the loads in flight per thread matter, not the grid size)

template <typename T, int V>
struct Row {
  float v[V];
};
template <typename T, int V>
__device__ __forceinline__ Row<T, V> load_row(const T *p) {
  Row<T, V> r;
  if constexpr (V == 1) {
    r.v[0] = (float)p[0];
  } else {
    typedef T tvec __attribute__((ext_vector_type(V)));
    const tvec t = *reinterpret_cast<const tvec *>(p);
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = (float)t[j];
  }
  return r;
}
template <typename T, int V>
__device__ __forceinline__ void store_row(T *p, const Row<T, V> &r) {
  if constexpr (V == 1) {
    p[0] = (T)r.v[0];
  } else {
    typedef T tvec __attribute__((ext_vector_type(V)));
    tvec t;
#pragma unroll
    for (int j = 0; j < V; ++j) t[j] = (T)r.v[j];
    *reinterpret_cast<tvec *>(p) = t;
  }
}

// rows of chunk g: [g * n / G, (g + 1) * n / G)
__device__ __forceinline__ int64_t chunk_begin(int64_t g, int64_t n, int64_t G) { return g * n / G; }

// LDS of the partial kernels: s_red[R][2c] (one row of 2c sums per row lane) | s_out[2c] | s_tmp[256] | s_shift[c]
__host__ __device__ constexpr size_t bn_partial_lds_bytes(int c, int row_lanes) {
  return ((size_t)row_lanes * 2 * c + 2 * c + 256 + c) * sizeof(float);
}

// s_out[q] = sum over the row lanes l (ascending) of s_red[l * 2c + q], by the whole workgroup: with fewer than 256
// values (c < 128) G = 256 / 2c threads share a value — contiguous lane ranges, their G partial sums added in range
// order — instead of c / V threads walking all R lanes (R = 32 - 64 on the narrow layers: 500 - 1000 serial LDS reads
// on a handful of threads were 5 - 10 us of every partial kernel).  A fixed order: bitwise reproducible.
__device__ __forceinline__ void bn_reduce_lanes(const float *__restrict__ s_red, float *__restrict__ s_out,
                                                float *__restrict__ s_tmp, int c, int R) {
  const int NV = 2 * c, tid = (int)threadIdx.x, NT = (int)blockDim.x;
  __syncthreads();
  if (NV * 2 > NT) {
    for (int q = tid; q < NV; q += NT) {
      float a = 0.f;
      for (int l = 0; l < R; ++l) a += s_red[l * NV + q];
      s_out[q] = a;
    }
  } else {
    const int G = NT / NV, g = tid / NV, q = tid % NV;
    if (g < G) {
      float a = 0.f;
      for (int l = g * R / G; l < (g + 1) * R / G; ++l) a += s_red[l * NV + q];
      s_tmp[g * NV + q] = a;
    }
    __syncthreads();
    if (tid < NV) {
      float a = 0.f;
      for (int gg = 0; gg < G; ++gg) a += s_tmp[gg * NV + tid];
      s_out[tid] = a;
    }
  }
  __syncthreads();
}

// Per chunk and channel: mean and M2 = sum (x - mean)^2, from sums shifted by the chunk's first row.
// Block layout: P = c / V pieces per row, R = blockDim / P row lanes; thread (lane rl, piece p) takes rows
// r0 + rl, r0 + rl + R, ...; the R lanes are combined through LDS in lane order.
// Round 3: every load of a batch of kBnRowsPerThread rows is UNCONDITIONAL (row index clamped to the chunk, the
// contribution masked) and issued before the first use.  With `if (r < r1) load` hipcc emitted a branch, a load and
// an s_waitcnt vmcnt(0) per row — eight dependent memory round trips per thread, 8 - 16 us for ANY matrix size
// (profiles/r03_bn_kernels_by_shape_before.log).
template <typename T, int V>
__global__ __launch_bounds__(256) void k_bn_partial(const T *__restrict__ x, int64_t n, int c, int chunks,
                                                   float *__restrict__ part_mean, float *__restrict__ part_m2) {
  extern __shared__ float s_red[];  // bn_partial_lds_bytes
  const int P = c / V;
  const int R = max(1, (int)blockDim.x / P);
  float *s_out = s_red + (size_t)R * 2 * c, *s_tmp = s_out + 2 * c, *s_shift = s_tmp + 256;
  const int64_t r0 = chunk_begin(blockIdx.x, n, chunks), r1 = chunk_begin(blockIdx.x + 1, n, chunks);
  for (int p0 = 0; p0 < P; p0 += blockDim.x) {  // one pass unless c / V > blockDim
    const int p = p0 + (int)threadIdx.x % min(P, (int)blockDim.x);
    const int rl = (int)threadIdx.x / min(P, (int)blockDim.x);
    const bool active = rl < R && p < P;
    float s1[V], s2[V], shift[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = shift[j] = 0.f;
    if (active && r0 < r1) {
      const T *xp = x + p * V;
      int64_t rb = r0 + rl;
      Row<T, V> t[kBnRowsPerThread];
      auto load_batch = [&]() {
#pragma unroll
        for (int i = 0; i < kBnRowsPerThread; ++i)
          t[i] = load_row<T, V>(xp + min(rb + (int64_t)i * R, r1 - 1) * c);
      };
      auto add_batch = [&]() {
#pragma unroll
        for (int i = 0; i < kBnRowsPerThread; ++i) {
          const float w = rb + (int64_t)i * R < r1 ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = (t[i].v[j] - shift[j]) * w;
            s1[j] += d;
            s2[j] += d * d;
          }
        }
      };
      if (rb < r1) {   // first batch and the shift row in one round trip
        load_batch();
        const Row<T, V> k = load_row<T, V>(xp + r0 * c);
#pragma unroll
        for (int j = 0; j < V; ++j) shift[j] = k.v[j];
        add_batch();
        for (rb += (int64_t)kBnRowsPerThread * R; rb < r1; rb += (int64_t)kBnRowsPerThread * R) {
          load_batch();
          add_batch();
        }
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s_red[(rl * 2 + 0) * c + p * V + j] = s1[j];
        s_red[(rl * 2 + 1) * c + p * V + j] = s2[j];
        if (rl == 0) s_shift[p * V + j] = shift[j];
      }
    }
  }
  bn_reduce_lanes(s_red, s_out, s_tmp, c, R);
  const float cnt = (float)(r1 - r0);
  for (int ch = (int)threadIdx.x; ch < c; ch += (int)blockDim.x) {
    const float a = s_out[ch], b = s_out[c + ch];
    const float m = cnt > 0.f ? a / cnt : 0.f;
    part_mean[(int64_t)blockIdx.x * c + ch] = s_shift[ch] + m;
    part_m2[(int64_t)blockIdx.x * c + ch] = cnt > 0.f ? fmaxf(b - a * m, 0.f) : 0.f;
  }
}

// Combine the chunks of each channel: ONE WAVE per channel (4 channels per block); lane l takes chunks l, l + 64, ...
// in order, then the 64 lane results are added in a fixed shuffle tree (the same every run: bitwise reproducible).
// Writes mean, rstd (biased variance + eps) and updates the running statistics (unbiased variance, torch's convention).

// tile_rows > 0: the partials are those of a convolution's tiles (k_conv_tile_bf16's statistics epilogue: tile g
// holds rows [g * tile_rows, min((g + 1) * tile_rows, n)), any number of tiles); 0: k_bn_partial's equal chunks.
__global__ __launch_bounds__(256) void k_bn_final(const float *__restrict__ part_mean,
                                                 const float *__restrict__ part_m2, int64_t n, int c, int chunks,
                                                 float eps, float momentum, float *__restrict__ mean_out,
                                                 float *__restrict__ rstd_out, float *__restrict__ running_mean,
                                                 float *__restrict__ running_var,
                                                 int64_t *__restrict__ num_batches_tracked, int tile_rows) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  if (ch >= c) return;  // whole wave
  float rm_old = 0.f, rv_old = 0.f;   // (requested with the partials, not behind the merge)
  if (running_mean != nullptr && lane == 0) {
    rm_old = running_mean[ch];
    rv_old = running_var[ch];
  }
  // Merge of the chunks (round 3, second form): every chunk's mean is taken relative to ONE shift — chunk 0's mean, the
  // same for all lanes — so the merge is three weighted sums, A = sum n_g d_g, B = sum (M2_g + n_g d_g^2), N = sum n_g with
  // d_g = mean_g - shift, added lane by lane and then across the lanes in a fixed shuffle tree; mean = shift + A / N,
  // M2 = B - A^2 / N.  No division inside the loop or the tree (Chan's pairwise update costs two per merge: 16 merges
  // per lane + 6 tree levels were ~3 us of this kernel's 7).  Fixed order: bitwise reproducible.
  constexpr int L = kBnMaxChunks / 64;   // chunks per lane and pass: all partials requested before they are used
  // rows of chunk g = chunk_begin(g + 1) - chunk_begin(g) = q + ((g + 1) * rem) / G - (g * rem) / G with n = q * G + rem:
  // one 64-bit division per thread instead of two per chunk
  const int64_t cq_rows = tile_rows > 0 ? 0 : n / chunks;
  const uint32_t rem = (uint32_t)(n - cq_rows * chunks), G = (uint32_t)chunks;
  const float shift = part_mean[ch];     // chunk 0 (every lane reads the same word)
  float sa = 0.f, sb = 0.f, sn = 0.f;
  for (int g0 = 0; g0 < chunks; g0 += 64 * L) {   // one pass unless a convolution had more than 512 tiles
    float pm[L], pq[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int g = g0 + lane + i * 64;
      const int64_t gc = min(g, chunks - 1);   // unconditional loads (a conditional one costs a branch and a full wait)
      pm[i] = part_mean[gc * c + ch];
      pq[i] = part_m2[gc * c + ch];
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int g = g0 + lane + i * 64;
      float bn;
      if (tile_rows > 0) {
        bn = (float)min((int64_t)tile_rows, n - (int64_t)g * tile_rows);
      } else {
        const uint32_t extra = ((uint32_t)(g + 1) * rem) / G - ((uint32_t)g * rem) / G;
        bn = (float)(cq_rows + extra);
      }
      if (g >= chunks) bn = 0.f;
      const float d = pm[i] - shift;
      sa = fmaf(bn, d, sa);
      sb += g < chunks ? fmaf(bn * d, d, pq[i]) : 0.f;
      sn += bn;
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {  // lane l absorbs lane l + off: a fixed tree
    const float ta = __shfl_down(sa, off, 64), tb = __shfl_down(sb, off, 64), tn = __shfl_down(sn, off, 64);
    if ((lane & (2 * off - 1)) == 0) {
      sa += ta;
      sb += tb;
      sn += tn;
    }
  }
  const float cn = sn;
  const float am = cn > 0.f ? sa / cn : 0.f;
  const float cm = shift + am;
  const float cq = fmaxf(sb - sa * am, 0.f);
  if (lane != 0) return;
  const float var = cn > 0.f ? cq / cn : 0.f;
  mean_out[ch] = cm;
  rstd_out[ch] = rsqrtf(var + eps);
  if (running_mean != nullptr) {
    const float unbiased = cn > 1.f ? cq / (cn - 1.f) : var;
    running_mean[ch] = (1.f - momentum) * rm_old + momentum * cm;
    running_var[ch] = (1.f - momentum) * rv_old + momentum * unbiased;
  }
}

// y = x * a[c] + b[c] with a = gamma * rstd, b = beta - mean * a (gamma / beta may be NULL: 1 / 0).
// Same thread layout as k_bn_partial — a thread owns one channel piece, keeps its a / b in registers and walks
// kBnRowsPerThread rows of the block's row range (no per-element index arithmetic, coalesced rows).

// per-channel parameters of V consecutive channels (gamma / beta may be NULL: 1 / 0 — one uniform branch, not one
// per element)
template <int V>
__device__ __forceinline__ void load_affine(const float *__restrict__ gamma, const float *__restrict__ beta, int ch0,
                                            float (&ga)[V], float (&be)[V]) {
  if (gamma != nullptr) {
#pragma unroll
    for (int j = 0; j < V; ++j) ga[j] = gamma[ch0 + j];
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) ga[j] = 1.f;
  }
  if (beta != nullptr) {
#pragma unroll
    for (int j = 0; j < V; ++j) be[j] = beta[ch0 + j];
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) be[j] = 0.f;
  }
}

// `skip` (SKIP): the residual branch of a ResNet block — y = relu(x * a + b + skip) in ONE pass instead of a
// batch-norm apply, an addition and a ReLU (seven passes over the matrix -> three; MinkUNet34C has 23 such blocks).
// The normalised value is rounded to T before the addition and the sum is rounded again, exactly as the separate
// kernels do (the fused result is bit-identical to them).
// All row loads of a thread (kBnRowsPerThread rows, clamped to the matrix) are issued first, unconditionally, then
// the parameters, then the arithmetic; only the stores are predicated (see k_bn_partial).
template <typename T, int V, bool SKIP>
__global__ __launch_bounds__(256) void k_bn_apply(const T *__restrict__ x, int64_t n, int c,
                                                 const float *__restrict__ mean, const float *__restrict__ rstd,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 T *__restrict__ y, int relu, const T *__restrict__ skip) {
  const int P = c / V;
  const int W = min(P, (int)blockDim.x);
  const int R = max(1, (int)blockDim.x / P);
  const int rl = (int)threadIdx.x / W;
  const int64_t r0 = (int64_t)blockIdx.x * R * kBnRowsPerThread;
  if (rl >= R) return;
  for (int p = (int)threadIdx.x % W; p < P; p += W) {
    Row<T, V> t[kBnRowsPerThread], sk[SKIP ? kBnRowsPerThread : 1];
#pragma unroll
    for (int i = 0; i < kBnRowsPerThread; ++i) {
      const int64_t off = min(r0 + rl + (int64_t)i * R, n - 1) * c + p * V;
      t[i] = load_row<T, V>(x + off);
      if constexpr (SKIP) sk[i] = load_row<T, V>(skip + off);
    }
    float a[V], b[V];
    load_affine<V>(gamma, beta, p * V, a, b);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      a[j] *= rstd[p * V + j];
      b[j] = fmaf(-mean[p * V + j], a[j], b[j]);
    }
#pragma unroll
    for (int i = 0; i < kBnRowsPerThread; ++i) {
      const int64_t r = r0 + rl + (int64_t)i * R;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if constexpr (SKIP) {
          const float z = (float)(T)fmaf(t[i].v[j], a[j], b[j]);   // what the unfused apply kernel would have stored
          t[i].v[j] = (float)(T)(z + sk[i].v[j]);                  // ... and the unfused addition
        } else {
          t[i].v[j] = fmaf(t[i].v[j], a[j], b[j]);
        }
        if (relu) t[i].v[j] = fmaxf(t[i].v[j], 0.f);
      }
      if (r < n) store_row<T, V>(y + r * c + p * V, t[i]);
    }
  }
}

// per chunk and channel: sum dy and sum dy * xhat (xhat = (x - mean) * rstd); same layout as k_bn_partial.
// YOUT: the residual form — the ReLU mask comes from the stored forward output (it also holds the skip branch).
template <typename T, int V, bool YOUT>
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const T *__restrict__ x, const T *__restrict__ dy,
                                                       int64_t n, int c, int chunks,
                                                       const float *__restrict__ mean,
                                                       const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int relu,
                                                       float *__restrict__ part_dy,
                                                       float *__restrict__ part_dyx, const T *__restrict__ yout) {
  extern __shared__ float s_red[];  // bn_partial_lds_bytes
  const int P = c / V;
  const int R = max(1, (int)blockDim.x / P);
  float *s_out = s_red + (size_t)R * 2 * c, *s_tmp = s_out + 2 * c;
  const int64_t r0 = chunk_begin(blockIdx.x, n, chunks), r1 = chunk_begin(blockIdx.x + 1, n, chunks);
  constexpr int RB = kBnRowsPerThread / 2;   // rows in flight per thread (x, dy and the output: 8 - 12 loads)
  for (int p0 = 0; p0 < P; p0 += blockDim.x) {
    const int p = p0 + (int)threadIdx.x % min(P, (int)blockDim.x);
    const int rl = (int)threadIdx.x / min(P, (int)blockDim.x);
    const bool active = rl < R && p < P;
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
      float m[V], rs[V], ga[V], be[V];
      bool first = true;
      for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)RB * R) {
        Row<T, V> tx[RB], tg[RB], ty[YOUT ? RB : 1];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int64_t off = min(rb + (int64_t)i * R, r1 - 1) * c + p * V;
          tx[i] = load_row<T, V>(x + off);
          tg[i] = load_row<T, V>(dy + off);
          if constexpr (YOUT) ty[i] = load_row<T, V>(yout + off);
        }
        if (first) {   // (behind the first batch of row loads: one round trip, not two)
          load_affine<V>(gamma, beta, p * V, ga, be);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            m[j] = mean[p * V + j];
            rs[j] = rstd[p * V + j];
          }
          first = false;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const bool valid = rb + (int64_t)i * R < r1;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float xh = (tx[i].v[j] - m[j]) * rs[j];
            // fused ReLU: the gradient passes where the forward output was positive — xh * gamma + beta recomputed,
            // or (residual form) the stored output itself
            bool pass;
            if constexpr (YOUT) pass = ty[i].v[j] > 0.f;
            else pass = fmaf(xh, ga[j], be[j]) > 0.f;
            const float g = (!valid || (relu && !pass)) ? 0.f : tg[i].v[j];
            s1[j] += g;
            s2[j] = fmaf(g, xh, s2[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s_red[(rl * 2 + 0) * c + p * V + j] = s1[j];
        s_red[(rl * 2 + 1) * c + p * V + j] = s2[j];
      }
    }
  }
  bn_reduce_lanes(s_red, s_out, s_tmp, c, R);
  for (int ch = (int)threadIdx.x; ch < c; ch += (int)blockDim.x) {
    part_dy[(int64_t)blockIdx.x * c + ch] = s_out[ch];
    part_dyx[(int64_t)blockIdx.x * c + ch] = s_out[c + ch];
  }
}

// sums of the chunks per channel in a fixed order: grad_beta = sum dy, grad_gamma = sum dy * xhat
// (one wave per channel and a fixed shuffle tree, as k_bn_final)
__global__ __launch_bounds__(256) void k_bn_bwd_final(const float *__restrict__ part_dy,
                                                     const float *__restrict__ part_dyx, int c, int chunks,
                                                     float *__restrict__ sum_dy, float *__restrict__ sum_dyx) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;  // whole wave
  float a = 0.f, b = 0.f;
  constexpr int L = kBnMaxChunks / 64;
  float pa[L], pb[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int g = lane + i * 64;
    const int64_t gc = min(g, chunks - 1);
    pa[i] = part_dy[gc * c + ch];
    pb[i] = part_dyx[gc * c + ch];
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    if (lane + i * 64 < chunks) {
      a += pa[i];
      b += pb[i];
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ta = __shfl_down(a, off, 64), tb = __shfl_down(b, off, 64);
    if ((lane & (2 * off - 1)) == 0) {
      a += ta;
      b += tb;
    }
  }
  if (lane == 0) {
    sum_dy[ch] = a;
    sum_dyx[ch] = b;
  }
}

// dx = gamma * rstd * (dy - sum_dy / n - xhat * sum_dyx / n) = dy * ca + x * cb + cc per channel
// (thread layout and load-first structure of k_bn_apply).  YOUT: residual form (mask from the stored output);
// dskip (optional, uniform): the masked gradient, i.e. the gradient of the residual branch.
template <typename T, int V, bool YOUT>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T *__restrict__ x, const T *__restrict__ dy,
                                                     int64_t n, int c, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd,
                                                     const float *__restrict__ gamma,
                                                     const float *__restrict__ sum_dy,
                                                     const float *__restrict__ sum_dyx, T *__restrict__ dx,
                                                     const float *__restrict__ beta, int relu,
                                                     const T *__restrict__ yout, T *__restrict__ dskip) {
  const int P = c / V;
  const int W = min(P, (int)blockDim.x);
  const int R = max(1, (int)blockDim.x / P);
  const int rl = (int)threadIdx.x / W;
  constexpr int RB = kBnRowsPerThread;
  const int64_t r0 = (int64_t)blockIdx.x * R * RB;
  if (rl >= R) return;
  const float inv_n = 1.f / (float)n;
  for (int p = (int)threadIdx.x % W; p < P; p += W) {
    Row<T, V> tx[RB], tg[RB], ty[YOUT ? RB : 1];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int64_t off = min(r0 + rl + (int64_t)i * R, n - 1) * c + p * V;
      tx[i] = load_row<T, V>(x + off);
      tg[i] = load_row<T, V>(dy + off);
      if constexpr (YOUT) ty[i] = load_row<T, V>(yout + off);
    }
    float ca[V], cb[V], cc[V], za[V], zb[V];
    load_affine<V>(gamma, beta, p * V, za, zb);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int ch = p * V + j;
      const float rs = rstd[ch], mu = mean[ch];
      const float a = za[j] * rs;
      const float k = sum_dyx[ch] * inv_n * rs;   // xhat * sum_dyx / n = (x - mean) * k
      ca[j] = a;
      cb[j] = -a * k;
      cc[j] = a * (mu * k - sum_dy[ch] * inv_n);
      za[j] = a;                                    // forward output z = x * za + zb (fused-ReLU mask)
      zb[j] = fmaf(-mu, a, zb[j]);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int64_t r = r0 + rl + (int64_t)i * R;
      Row<T, V> out, gs;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        bool pass;
        if constexpr (YOUT) pass = ty[i].v[j] > 0.f;
        else pass = fmaf(tx[i].v[j], za[j], zb[j]) > 0.f;
        const float g = (relu && !pass) ? 0.f : tg[i].v[j];
        gs.v[j] = g;
        // (explicit FMAs: the residual and the plain instantiation must contract alike — their results are compared
        // bit for bit with the three separate operators)
        out.v[j] = fmaf(g, ca[j], fmaf(tx[i].v[j], cb[j], cc[j]));
      }
      if (r < n) {
        store_row<T, V>(dx + r * c + p * V, out);
        if (dskip != nullptr) store_row<T, V>(dskip + r * c + p * V, gs);   // gradient of the residual branch
      }
    }
  }
}

// chunks of a reduction over n rows by workgroups of `row_lanes` row lanes: one batch of rows in flight per thread
// where that fills the chip (rows_per_thread * row_lanes rows per chunk), more per thread beyond kBnMaxChunks chunks
static int bn_chunks(int64_t n, int row_lanes, int rows_per_thread) {
  int64_t g = ceil_div(n, (int64_t)row_lanes * rows_per_thread);
  if (g > kBnMaxChunks) g = kBnMaxChunks;
  if (g < 1) g = 1;
  return (int)g;
}
static int bn_chunks_max(int64_t n) { return bn_chunks(n, 1, kBnRowsPerThread / 2); }   // workspace bound

template <typename T>
static int bn_stats(const T *x, int64_t n, int c, float eps, float momentum, float *mean, float *rstd,
                    float *running_mean, float *running_var, int64_t *num_batches_tracked, float *ws,
                    hipStream_t stream) {
  constexpr int W = 16 / (int)sizeof(T);  // channels per 16-byte access
  const bool aligned = (uintptr_t)x % 16 == 0;
  const int v = (aligned && c % W == 0) ? W : ((aligned && c % 4 == 0) ? 4 : 1);
  const int P = c / v;
  const int R = P >= 256 ? 1 : 256 / P;
  const int chunks = bn_chunks(n, R, kBnRowsPerThread);
  float *pm = ws, *pq = ws + (int64_t)chunks * c;
  const size_t lds = bn_partial_lds_bytes(c, R);
  ME_CHECK(lds <= 64 * 1024, "channel count too large for the batch-norm kernels");
  if (v == W) hipLaunchKernelGGL((k_bn_partial<T, W>), dim3(chunks), dim3(256), lds, stream, x, n, c, chunks, pm, pq);
  else if (v == 4) hipLaunchKernelGGL((k_bn_partial<T, 4>), dim3(chunks), dim3(256), lds, stream, x, n, c, chunks, pm, pq);
  else hipLaunchKernelGGL((k_bn_partial<T, 1>), dim3(chunks), dim3(256), lds, stream, x, n, c, chunks, pm, pq);
  hipLaunchKernelGGL(k_bn_final, dim3((unsigned)ceil_div(c, 4)), dim3(256), 0, stream, pm, pq, n, c, chunks, eps,
                     momentum, mean, rstd, running_mean, running_var, num_batches_tracked, 0);
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int bn_apply(const T *x, int64_t n, int c, const float *mean, const float *rstd, const float *gamma,
                    const float *beta, T *y, int relu, hipStream_t stream, const T *skip = nullptr) {
  constexpr int W = 16 / (int)sizeof(T);  // channels per 16-byte access
  const bool aligned = (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)skip % 16 == 0;
  const int v = (aligned && c % W == 0) ? W : ((aligned && c % 4 == 0) ? 4 : 1);
  const int pieces = c / v;
  const dim3 grid((unsigned)ceil_div(n, (int64_t)(pieces >= 256 ? 1 : 256 / pieces) * kBnRowsPerThread));
#define ME_BN_APPLY(VV)                                                                                            \
  do {                                                                                                             \
    if (skip != nullptr)                                                                                           \
      hipLaunchKernelGGL((k_bn_apply<T, VV, true>), grid, dim3(256), 0, stream, x, n, c, mean, rstd, gamma, beta, y, \
                         relu, skip);                                                                              \
    else                                                                                                           \
      hipLaunchKernelGGL((k_bn_apply<T, VV, false>), grid, dim3(256), 0, stream, x, n, c, mean, rstd, gamma, beta, y, \
                         relu, skip);                                                                              \
  } while (0)
  if (v == W) ME_BN_APPLY(W);
  else if (v == 4) ME_BN_APPLY(4);
  else ME_BN_APPLY(1);
#undef ME_BN_APPLY
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int bn_backward(const T *x, const T *dy, int64_t n, int c, const float *mean, const float *rstd,
                       const float *gamma, const float *beta, int relu, T *dx, float *grad_gamma, float *grad_beta,
                       float *ws, hipStream_t stream, const T *yout = nullptr, T *dskip = nullptr) {
  constexpr int W = 16 / (int)sizeof(T);
  const bool vec = (c % 4) == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 &&
                   (uintptr_t)yout % 16 == 0 && (uintptr_t)dskip % 16 == 0;
  const int v = (vec && c % W == 0) ? W : (vec ? 4 : 1);
  const int P = c / v;
  const int R = P >= 256 ? 1 : 256 / P;
  const int chunks = bn_chunks(n, R, kBnRowsPerThread / 2);
  float *pa = ws, *pb = ws + (int64_t)chunks * c;
  const size_t lds = bn_partial_lds_bytes(c, R);
  ME_CHECK(lds <= 64 * 1024, "channel count too large for the batch-norm kernels");
#define ME_BN_BWD_PARTIAL(VV)                                                                                      \
  do {                                                                                                             \
    if (yout != nullptr)                                                                                           \
      hipLaunchKernelGGL((k_bn_bwd_partial<T, VV, true>), dim3(chunks), dim3(256), lds, stream, x, dy, n, c, chunks, \
                         mean, rstd, gamma, beta, relu, pa, pb, yout);                                             \
    else                                                                                                           \
      hipLaunchKernelGGL((k_bn_bwd_partial<T, VV, false>), dim3(chunks), dim3(256), lds, stream, x, dy, n, c, chunks, \
                         mean, rstd, gamma, beta, relu, pa, pb, yout);                                             \
  } while (0)
  if (v == W) ME_BN_BWD_PARTIAL(W);
  else if (v == 4) ME_BN_BWD_PARTIAL(4);
  else ME_BN_BWD_PARTIAL(1);
#undef ME_BN_BWD_PARTIAL
  hipLaunchKernelGGL(k_bn_bwd_final, dim3((unsigned)ceil_div(c, 4)), dim3(256), 0, stream, pa, pb, c, chunks,
                     grad_beta, grad_gamma);
  const int pieces = c / v;
  const dim3 grid((unsigned)ceil_div(n, (int64_t)(pieces >= 256 ? 1 : 256 / pieces) * kBnRowsPerThread));
#define ME_BN_BWD_APPLY(VV)                                                                                        \
  do {                                                                                                             \
    if (yout != nullptr)                                                                                           \
      hipLaunchKernelGGL((k_bn_bwd_apply<T, VV, true>), grid, dim3(256), 0, stream, x, dy, n, c, mean, rstd, gamma, \
                         grad_beta, grad_gamma, dx, beta, relu, yout, dskip);                                      \
    else                                                                                                           \
      hipLaunchKernelGGL((k_bn_bwd_apply<T, VV, false>), grid, dim3(256), 0, stream, x, dy, n, c, mean, rstd, gamma, \
                         grad_beta, grad_gamma, dx, beta, relu, yout, dskip);                                      \
  } while (0)
  if (v == W) ME_BN_BWD_APPLY(W);
  else if (v == 4) ME_BN_BWD_APPLY(4);
  else ME_BN_BWD_APPLY(1);
#undef ME_BN_BWD_APPLY
  ME_LAUNCH_CHECK();
  return 0;
}

}  // namespace me

using namespace me;

extern "C" {

int64_t me_bn_workspace_bytes(int64_t n, int32_t c) {
  return align_up((int64_t)bn_chunks_max(n) * c * 2 * 4, 256);
}

int me_bn_stats(const void *x, int32_t is_bf16, int64_t n, int32_t c, float eps, float momentum, float *mean,
                float *rstd, float *running_mean, float *running_var, int64_t *num_batches_tracked, void *workspace,
                int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n > 0 && c > 0, "batch norm needs at least one row and one channel");
  ME_CHECK(workspace_bytes >= me_bn_workspace_bytes(n, c), "workspace too small");
  float *ws = reinterpret_cast<float *>(workspace);
  if (is_bf16)
    return bn_stats<__bf16>(reinterpret_cast<const __bf16 *>(x), n, c, eps, momentum, mean, rstd, running_mean,
                            running_var, num_batches_tracked, ws, stream);
  return bn_stats<float>(reinterpret_cast<const float *>(x), n, c, eps, momentum, mean, rstd, running_mean,
                         running_var, num_batches_tracked, ws, stream);
}

int me_bn_stats_from_tiles(const float *part_mean, const float *part_m2, int64_t n, int32_t c, int32_t tile_rows,
                           float eps, float momentum, float *mean, float *rstd, float *running_mean,
                           float *running_var, int64_t *num_batches_tracked, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n > 0 && c > 0 && tile_rows > 0, "batch norm needs at least one row, one channel and a tile height");
  ME_CHECK(part_mean != nullptr && part_m2 != nullptr, "the tile partials must be given");
  const int64_t tiles = ceil_div(n, (int64_t)tile_rows);
  ME_CHECK(tiles < (1ll << 30), "too many tiles");
  hipLaunchKernelGGL(k_bn_final, dim3((unsigned)ceil_div(c, 4)), dim3(256), 0, stream, part_mean, part_m2, n, c,
                     (int)tiles, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked,
                     (int)tile_rows);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_bn_apply(const void *x, int32_t is_bf16, int64_t n, int32_t c, const float *mean, const float *rstd,
                const float *gamma, const float *beta, int32_t relu, void *y, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(c > 0, "invalid channel count");
  if (n == 0) return 0;
  if (is_bf16)
    return bn_apply<__bf16>(reinterpret_cast<const __bf16 *>(x), n, c, mean, rstd, gamma, beta,
                            reinterpret_cast<__bf16 *>(y), relu, stream);
  return bn_apply<float>(reinterpret_cast<const float *>(x), n, c, mean, rstd, gamma, beta,
                         reinterpret_cast<float *>(y), relu, stream);
}

int me_bn_apply_residual(const void *x, const void *skip, int32_t is_bf16, int64_t n, int32_t c, const float *mean,
                         const float *rstd, const float *gamma, const float *beta, int32_t relu, void *y,
                         void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(c > 0 && skip != nullptr, "invalid channel count / missing residual branch");
  if (n == 0) return 0;
  if (is_bf16)
    return bn_apply<__bf16>(reinterpret_cast<const __bf16 *>(x), n, c, mean, rstd, gamma, beta,
                            reinterpret_cast<__bf16 *>(y), relu, stream, reinterpret_cast<const __bf16 *>(skip));
  return bn_apply<float>(reinterpret_cast<const float *>(x), n, c, mean, rstd, gamma, beta,
                         reinterpret_cast<float *>(y), relu, stream, reinterpret_cast<const float *>(skip));
}

int me_bn_backward_residual(const void *x, const void *dy, const void *yout, int32_t is_bf16, int64_t n, int32_t c,
                            const float *mean, const float *rstd, const float *gamma, const float *beta, int32_t relu,
                            void *dx, void *dskip, float *grad_gamma, float *grad_beta, void *workspace,
                            int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n > 0 && c > 0, "batch norm needs at least one row and one channel");
  ME_CHECK(workspace_bytes >= me_bn_workspace_bytes(n, c), "workspace too small");
  ME_CHECK(!relu || yout != nullptr, "the fused ReLU of the residual form needs the forward output");
  float *ws = reinterpret_cast<float *>(workspace);
  if (is_bf16)
    return bn_backward<__bf16>(reinterpret_cast<const __bf16 *>(x), reinterpret_cast<const __bf16 *>(dy), n, c, mean,
                               rstd, gamma, beta, relu, reinterpret_cast<__bf16 *>(dx), grad_gamma, grad_beta, ws,
                               stream, relu ? reinterpret_cast<const __bf16 *>(yout) : nullptr,
                               reinterpret_cast<__bf16 *>(dskip));
  return bn_backward<float>(reinterpret_cast<const float *>(x), reinterpret_cast<const float *>(dy), n, c, mean, rstd,
                            gamma, beta, relu, reinterpret_cast<float *>(dx), grad_gamma, grad_beta, ws, stream,
                            relu ? reinterpret_cast<const float *>(yout) : nullptr, reinterpret_cast<float *>(dskip));
}

int me_bn_backward(const void *x, const void *dy, int32_t is_bf16, int64_t n, int32_t c, const float *mean,
                   const float *rstd, const float *gamma, const float *beta, int32_t relu, void *dx,
                   float *grad_gamma, float *grad_beta, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n > 0 && c > 0, "batch norm needs at least one row and one channel");
  ME_CHECK(workspace_bytes >= me_bn_workspace_bytes(n, c), "workspace too small");
  float *ws = reinterpret_cast<float *>(workspace);
  if (is_bf16)
    return bn_backward<__bf16>(reinterpret_cast<const __bf16 *>(x), reinterpret_cast<const __bf16 *>(dy), n, c, mean,
                               rstd, gamma, beta, relu, reinterpret_cast<__bf16 *>(dx), grad_gamma, grad_beta, ws,
                               stream);
  return bn_backward<float>(reinterpret_cast<const float *>(x), reinterpret_cast<const float *>(dy), n, c, mean, rstd,
                            gamma, beta, relu, reinterpret_cast<float *>(dx), grad_gamma, grad_beta, ws, stream);
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_norm(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_bn_final));
}
