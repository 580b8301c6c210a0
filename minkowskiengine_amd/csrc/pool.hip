// Pooling and broadcast kernels for gfx950 (MI355X): local sum / average / max pooling, global pooling
// over the rows of each batch index, and broadcast of one feature row per batch index.
//
// Replace NonzeroAvgPooling{Forward,Backward}KernelGPU, MaxPooling{Forward,Backward}KernelGPU and the
// broadcast kernels (src/pooling_avg_kernel.cu, src/pooling_max_kernel.cu, src/broadcast_kernel.cu; CPU
// twins src/pooling_avg_kernel.hpp:41-150, src/pooling_max_kernel.hpp:36-117,
// src/broadcast_kernel.hpp:35-160).  The reference scatters over the pair lists (cuSPARSE SpMM for the
// sums, one atomic per element otherwise); here every kernel is TARGET-stationary on the dense
// neighbour tables the convolution already uses (nbr[k][target row] -> source row or -1): a thread owns
// a (target row, 16-byte channel piece), walks the kernel offsets in ascending k and writes its result
// once — no atomics, no zero-fill pass, and the summation order is the reference CPU order (k ascending),
// so sums and averages are bitwise reproducible.  All of them are HBM-bound gathers:
// bytes = 4*C*(pairs + targets) + 4*K*targets of table.
#include "common.hpp"

#include <float.h>
#include <initializer_list>

namespace me {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// V consecutive channels of one row, held in fp32 whatever the storage type T (float or __bf16: bf16 features are
// summed / compared in fp32 and rounded once at the store, like the convolution kernels).  One access of
// V * sizeof(T) bytes: 16 bytes for (float, 4) and (__bf16, 8), 8 bytes for (__bf16, 4), one element for V = 1.
template <int V>
struct Piece {
  float v[V];
};
template <typename T, int V>
__device__ __forceinline__ Piece<V> load_piece(const T *p) {
  Piece<V> r;
  if constexpr (V == 1) {
    r.v[0] = (float)*p;
  } else {
    typedef T tvec __attribute__((ext_vector_type(V)));
    const tvec t = *reinterpret_cast<const tvec *>(p);
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = (float)t[j];
  }
  return r;
}
template <typename T, int V>
__device__ __forceinline__ void store_piece(T *p, const Piece<V> &r) {
  if constexpr (V == 1) {
    *p = (T)r.v[0];
  } else {
    typedef T tvec __attribute__((ext_vector_type(V)));
    tvec t;
#pragma unroll
    for (int j = 0; j < V; ++j) t[j] = (T)r.v[j];
    *reinterpret_cast<tvec *>(p) = t;
  }
}

// dst[t] = sum over k of src[tbl[k][t]]            (src_count == nullptr)
//        = sum over k of src[s] / src_count[s]      (src_count != nullptr: average-pooling backward,
//                                                    src/pooling_avg_kernel.hpp:118-127)
// then divided by the number of summed rows when `average` (forward, :96-108); that number is written to
// dst_count when given.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_pool_sum(const T *__restrict__ src, int c,
                                                 const int32_t *__restrict__ tbl, int64_t n_tgt, int volume,
                                                 const float *__restrict__ src_count, int average,
                                                 T *__restrict__ dst, float *__restrict__ dst_count) {
  const int pieces = c / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tgt * pieces) return;
  const int64_t t = idx / pieces;
  const int ch = (int)(idx % pieces) * V;
  Piece<V> acc;
#pragma unroll
  for (int j = 0; j < V; ++j) acc.v[j] = 0.f;
  float cnt = 0.f;
  for (int k = 0; k < volume; ++k) {
    const int32_t s = tbl[(int64_t)k * n_tgt + t];
    if (s < 0) continue;
    const Piece<V> x = load_piece<T, V>(src + (int64_t)s * c + ch);
    if (src_count) {
      const float d = src_count[s];
      if (d > 0.f) {
#pragma unroll
        for (int j = 0; j < V; ++j) acc.v[j] += x.v[j] / d;
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) acc.v[j] += x.v[j];
    }
    cnt += 1.f;
  }
  if (average && cnt > 0.f) {
#pragma unroll
    for (int j = 0; j < V; ++j) acc.v[j] /= cnt;
  }
  store_piece<T, V>(dst + t * c + ch, acc);
  if (dst_count && ch == 0) dst_count[t] = cnt;
}

// dst[t][c] = max over k of src[tbl[k][t]][c], mask[t][c] = flat index (source row * C + c) of the
// first maximum in k order, -FLT_MAX / -1 for rows without neighbours (src/pooling_max_kernel.hpp:36-96)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_pool_max(const T *__restrict__ src, int c,
                                                 const int32_t *__restrict__ tbl, int64_t n_tgt, int volume,
                                                 T *__restrict__ dst, int32_t *__restrict__ mask) {
  const int pieces = c / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tgt * pieces) return;
  const int64_t t = idx / pieces;
  const int ch = (int)(idx % pieces) * V;
  Piece<V> best;
  int32_t arg[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    best.v[j] = -FLT_MAX;
    arg[j] = -1;
  }
  for (int k = 0; k < volume; ++k) {
    const int32_t s = tbl[(int64_t)k * n_tgt + t];
    if (s < 0) continue;
    const Piece<V> x = load_piece<T, V>(src + (int64_t)s * c + ch);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (best.v[j] < x.v[j]) {
        best.v[j] = x.v[j];
        arg[j] = s * c + ch + j;
      }
    }
  }
  store_piece<T, V>(dst + t * c + ch, best);
#pragma unroll
  for (int j = 0; j < V; ++j) mask[t * c + ch + j] = arg[j];
}

// grad_in[i][c] = sum over k of grad_out[o][c] for the output rows o = tblT[k][i] whose maximum came
// from (i, c)  (src/pooling_max_kernel.hpp:98-117, without its scatter)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_pool_max_backward(const T *__restrict__ grad_out, int c,
                                                          const int32_t *__restrict__ tbl_in, int64_t n_in,
                                                          int volume, const int32_t *__restrict__ mask,
                                                          T *__restrict__ grad_in) {
  const int pieces = c / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_in * pieces) return;
  const int64_t i = idx / pieces;
  const int ch = (int)(idx % pieces) * V;
  Piece<V> acc;
#pragma unroll
  for (int j = 0; j < V; ++j) acc.v[j] = 0.f;
  for (int k = 0; k < volume; ++k) {
    const int32_t o = tbl_in[(int64_t)k * n_in + i];
    if (o < 0) continue;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (mask[(int64_t)o * c + ch + j] == (int32_t)(i * c + ch + j)) acc.v[j] += (float)grad_out[(int64_t)o * c + ch + j];
    }
  }
  store_piece<T, V>(grad_in + i * c + ch, acc);
}

// ---- global pooling: reduce the rows of each batch index (origin map row) -------------------------
// Stage 1: a workgroup reduces one chunk of consecutive rows into partial[chunk][b][c].  Rows of a scene are
// normally consecutive, so nearly every chunk holds ONE batch index: then a thread owns a 16-byte channel piece
// and every (256 / pieces)-th row, and the row lanes are combined through LDS in lane order.  A chunk that
// mixes batch indices takes the general path: one thread per channel walks the rows and adds every run of
// equal batch indices.  Either way the summation order is fixed (deterministic); the argmax is the first row
// that attains the maximum.  Stage 2: one wave per (batch index, channel) merges the chunks in a fixed
// shuffle tree.
static inline int64_t global_chunk_rows(int64_t n) {
  const int64_t l = ceil_div(n < 1 ? 1 : n, 512);
  return l < 256 ? 256 : l;
}

template <typename T, bool MAX, int V>
__global__ __launch_bounds__(256) void k_global_partial(const T *__restrict__ src,
                                                       const T *__restrict__ src2, int c,
                                                       const int32_t *__restrict__ batch_row, int64_t n,
                                                       int64_t chunk_rows, int n_batch,
                                                       float *__restrict__ partial,
                                                       int32_t *__restrict__ partial_arg,
                                                       float *__restrict__ partial_cnt) {
  extern __shared__ float s_red[];  // [R][c] values (+ [R][c] argmax as int)
  const int64_t chunk = blockIdx.x;
  const int64_t r0 = chunk * chunk_rows;
  const int64_t r1 = min(n, r0 + chunk_rows);
  float *pp = partial + chunk * n_batch * c;
  int32_t *pa = MAX ? partial_arg + chunk * n_batch * c : nullptr;
  const int b0 = batch_row[r0];
  int mixed = 0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) mixed |= (batch_row[r] != b0);
  if (!__syncthreads_or(mixed)) {
    const int P = c / V;                      // the host launches V = 4 only when P <= 256
    const int R = max(1, (int)blockDim.x / P);
    const int p = (int)threadIdx.x % P, rl = (int)threadIdx.x / P;
    const bool active = rl < R;
    float acc[V];
    int32_t arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      acc[j] = MAX ? -FLT_MAX : 0.f;
      arg[j] = -1;
    }
    if (active) {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += R) {
        Piece<V> x = load_piece<T, V>(src + r * c + p * V);
        if (src2) {
          const Piece<V> y = load_piece<T, V>(src2 + r * c + p * V);
#pragma unroll
          for (int j = 0; j < V; ++j) x.v[j] *= y.v[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (MAX) {
            if (acc[j] < x.v[j]) {
              acc[j] = x.v[j];
              arg[j] = (int32_t)(r * c + p * V + j);
            }
          } else {
            acc[j] += x.v[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s_red[rl * c + p * V + j] = acc[j];
        if (MAX) reinterpret_cast<int32_t *>(s_red)[(R + rl) * c + p * V + j] = arg[j];
      }
    }
    __syncthreads();
    if (active && rl == 0) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float a = MAX ? -FLT_MAX : 0.f;
        int32_t g = -1;
        for (int l = 0; l < R; ++l) {  // fixed order; ties keep the smaller flat index = the earlier row
          const float v = s_red[l * c + p * V + j];
          if (MAX) {
            const int32_t vg = reinterpret_cast<const int32_t *>(s_red)[(R + l) * c + p * V + j];
            if (a < v || (a == v && vg >= 0 && (g < 0 || vg < g))) {
              a = v;
              g = vg;
            }
          } else {
            a += v;
          }
        }
        pp[(int64_t)b0 * c + p * V + j] = a;
        if (MAX) pa[(int64_t)b0 * c + p * V + j] = g;
      }
      if (!MAX && p == 0 && partial_cnt) partial_cnt[chunk * n_batch + b0] = (float)(r1 - r0);
    }
    return;
  }
  // general path: runs of equal batch indices, one thread per channel
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    int cur = -1;
    float acc = 0.f;
    int32_t arg = -1;
    float cnt = 0.f;
    auto flush = [&]() {
      if (cur < 0) return;
      if (MAX) {
        if (pp[(int64_t)cur * c + ch] < acc) {
          pp[(int64_t)cur * c + ch] = acc;
          pa[(int64_t)cur * c + ch] = arg;
        }
      } else {
        pp[(int64_t)cur * c + ch] += acc;
        if (ch == 0 && partial_cnt) partial_cnt[chunk * n_batch + cur] += cnt;
      }
    };
    for (int64_t r = r0; r < r1; ++r) {
      const int b = batch_row[r];
      if (b != cur) {
        flush();
        cur = b;
        acc = MAX ? -FLT_MAX : 0.f;
        arg = -1;
        cnt = 0.f;
      }
      float x = (float)src[r * c + ch];
      if (src2) x *= (float)src2[r * c + ch];
      if (MAX) {
        if (acc < x) {
          acc = x;
          arg = (int32_t)(r * c + ch);
        }
      } else {
        acc += x;
        cnt += 1.f;
      }
    }
    flush();
  }
}

// one wave per (batch index, channel): lane l merges chunks l, l + 64, ... in order, then a fixed shuffle tree
template <bool MAX>
__global__ __launch_bounds__(256) void k_global_final(const float *__restrict__ partial,
                                                     const int32_t *__restrict__ partial_arg,
                                                     const float *__restrict__ partial_cnt, int64_t chunks,
                                                     int n_batch, int c, int average,
                                                     float *__restrict__ dst, int32_t *__restrict__ dst_arg,
                                                     float *__restrict__ dst_cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, channel)
  if (idx >= (int64_t)n_batch * c) return;                              // whole wave
  const int b = (int)(idx / c);
  float a = MAX ? -FLT_MAX : 0.f, cnt = 0.f;
  int32_t g = -1;
  auto merge = [&](float v, int32_t vg) {
    if (MAX) {
      if (a < v || (a == v && vg >= 0 && (g < 0 || vg < g))) {
        a = v;
        g = vg;
      }
    } else {
      a += v;
    }
  };
  for (int64_t q = lane; q < chunks; q += 64) {
    merge(partial[q * n_batch * c + idx], MAX ? partial_arg[q * n_batch * c + idx] : -1);
    if (!MAX && partial_cnt) cnt += partial_cnt[q * n_batch + b];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float v = __shfl_down(a, off, 64), vc = __shfl_down(cnt, off, 64);
    const int32_t vg = __shfl_down(g, off, 64);
    if ((lane & (2 * off - 1)) == 0) {
      merge(v, vg);
      cnt += vc;
    }
  }
  if (lane != 0) return;
  if (MAX) {
    dst[idx] = a;
    dst_arg[idx] = g;
  } else {
    if (average && cnt > 0.f) a /= cnt;
    dst[idx] = a;
    if (dst_cnt && idx % c == 0) dst_cnt[b] = cnt;
  }
}

// out[i][c] = in[i][c] (+ | *) glob[batch_row[i]][c]; with in == nullptr: out[i][c] = glob[...][c]
template <typename T, int V>
__global__ __launch_bounds__(256) void k_broadcast(const T *__restrict__ in, const T *__restrict__ glob,
                                                  const int32_t *__restrict__ batch_row, int64_t n, int c,
                                                  int multiply, T *__restrict__ out) {
  const int pieces = c / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * pieces) return;
  const int64_t i = idx / pieces;
  const int ch = (int)(idx % pieces) * V;
  const Piece<V> g = load_piece<T, V>(glob + (int64_t)batch_row[i] * c + ch);
  Piece<V> r = g;
  if (in) {
    const Piece<V> x = load_piece<T, V>(in + i * c + ch);
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = multiply ? x.v[j] * g.v[j] : x.v[j] + g.v[j];
  }
  store_piece<T, V>(out + i * c + ch, r);
}

// Segmented sum / mean of feature rows (voxelisation of duplicate coordinates, the reference's
// UNWEIGHTED_SUM / UNWEIGHTED_AVERAGE quantisation modes: MinkowskiSparseTensor.py:317-341 via
// MinkowskiSPMMFunction -> cuSPARSE coo_spmm).  Segment s owns the rows perm[seg[s] .. seg[s+1]) — the
// input rows of one voxel in input order — so the summation order is fixed (no atomics, bitwise
// reproducible).  A thread owns one 16-byte channel piece of one segment.
template <int V>
__global__ __launch_bounds__(256) void k_segment_sum(const float *__restrict__ src, int c,
                                                    const int64_t *__restrict__ perm,
                                                    const int64_t *__restrict__ seg, int64_t n_seg, int average,
                                                    float *__restrict__ dst) {
  const int pieces = c / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_seg * pieces) return;
  const int64_t s = idx / pieces;
  const int ch = (int)(idx % pieces) * V;
  const int64_t b = seg[s], e = seg[s + 1];
  Piece<V> acc;
#pragma unroll
  for (int j = 0; j < V; ++j) acc.v[j] = 0.f;
  for (int64_t i = b; i < e; ++i) {
    const Piece<V> x = load_piece<float, V>(src + perm[i] * c + ch);
#pragma unroll
    for (int j = 0; j < V; ++j) acc.v[j] += x.v[j];
  }
  if (average && e > b) {
    const float inv = (float)(e - b);
#pragma unroll
    for (int j = 0; j < V; ++j) acc.v[j] /= inv;
  }
  store_piece<float, V>(dst + s * c + ch, acc);
}

}  // namespace me

using namespace me;

namespace {

// widest piece (elements) for rows of c channels of T at the given addresses: 16 bytes when everything is
// 16-byte aligned, 8 bytes for bf16 rows of a multiple of 4 channels, else one element
template <typename T>
int piece_width(int c, std::initializer_list<const void *> ptrs) {
  constexpr int W = 16 / (int)sizeof(T);
  bool a16 = true, a8 = true;
  for (const void *p : ptrs) {
    if (p == nullptr) continue;
    a16 = a16 && ((uintptr_t)p % 16 == 0);
    a8 = a8 && ((uintptr_t)p % 8 == 0);
  }
  if (c % W == 0 && a16) return W;
  if (sizeof(T) == 2 && c % 4 == 0 && a8) return 4;
  return 1;
}

#define ME_POOL_DISPATCH_V(T, v, ...)                                                            \
  do {                                                                                           \
    if ((v) == 8) { constexpr int V = (sizeof(T) == 2 ? 8 : 4); __VA_ARGS__; }                   \
    else if ((v) == 4) { constexpr int V = 4; __VA_ARGS__; }                                     \
    else { constexpr int V = 1; __VA_ARGS__; }                                                   \
  } while (0)

template <typename T>
int pool_sum(const T *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume, const float *src_count,
             int32_t average, T *dst, float *dst_count, hipStream_t stream) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_tgt == 0) return 0;
  const int v = piece_width<T>(c, {src, dst});
  const int64_t total = n_tgt * (c / v);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  ME_POOL_DISPATCH_V(T, v, hipLaunchKernelGGL((k_pool_sum<T, V>), grid, block, 0, stream, src, c, tbl, n_tgt,
                                              (int)volume, src_count, average, dst, dst_count));
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int pool_max(const T *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume, T *dst, int32_t *mask,
             hipStream_t stream) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_tgt == 0) return 0;
  const int v = piece_width<T>(c, {src, dst});
  const int64_t total = n_tgt * (c / v);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  ME_POOL_DISPATCH_V(T, v, hipLaunchKernelGGL((k_pool_max<T, V>), grid, block, 0, stream, src, c, tbl, n_tgt,
                                              (int)volume, dst, mask));
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int pool_max_backward(const T *grad_out, int32_t c, const int32_t *tbl_in, int64_t n_in, int64_t volume,
                      const int32_t *mask, T *grad_in, hipStream_t stream) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_in == 0) return 0;
  const int v = piece_width<T>(c, {grad_in});
  const int64_t total = n_in * (c / v);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  ME_POOL_DISPATCH_V(T, v, hipLaunchKernelGGL((k_pool_max_backward<T, V>), grid, block, 0, stream, grad_out, c,
                                              tbl_in, n_in, (int)volume, mask, grad_in));
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int broadcast(const T *in, const T *glob, const int32_t *batch_row, int64_t n, int32_t c, int32_t multiply, T *out,
              hipStream_t stream) {
  ME_CHECK(c > 0, "invalid channel count");
  if (n == 0) return 0;
  const int v = piece_width<T>(c, {glob, out, in});
  const int64_t total = n * (c / v);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  ME_POOL_DISPATCH_V(T, v, hipLaunchKernelGGL((k_broadcast<T, V>), grid, block, 0, stream, in, glob, batch_row, n, c,
                                              multiply, out));
  ME_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int global_pool(const T *src, const T *src2, int32_t c, const int32_t *batch_row, int64_t n, int32_t n_batch,
                int32_t mode, float *dst, int32_t *dst_arg, float *dst_count, void *workspace,
                int64_t workspace_bytes, hipStream_t stream) {
  ME_CHECK(c > 0 && n_batch > 0, "invalid channel count or batch size");
  ME_CHECK(mode >= 0 && mode <= 2, "mode must be 0 (sum), 1 (average) or 2 (max)");
  ME_CHECK(mode != 2 || dst_arg != nullptr, "max pooling needs the argmax output");
  ME_CHECK(workspace_bytes >= me_global_pool_workspace_bytes(n, n_batch, c), "workspace too small");
  const int64_t chunk_rows = global_chunk_rows(n);
  const int64_t chunks = ceil_div(n < 1 ? 1 : n, chunk_rows);
  const int64_t vsz = align_up(chunks * n_batch * c * 4, 256);
  float *partial = reinterpret_cast<float *>(workspace);
  int32_t *partial_arg = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(workspace) + vsz);
  float *partial_cnt = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + 2 * vsz);
  const int64_t total = (int64_t)n_batch * c;
  // a thread owns 4 channels (one 16-byte piece of a float row, 8 bytes of a bf16 row) or one channel
  const bool vec = (c % 4) == 0 && c / 4 <= 256 && (uintptr_t)src % 16 == 0 &&
                   (src2 == nullptr || (uintptr_t)src2 % 16 == 0);
  const int P = vec ? c / 4 : (c < 256 ? c : 256);
  const int R = 256 / P < 1 ? 1 : 256 / P;
  const size_t lds = (size_t)R * c * 4 * (mode == 2 ? 2 : 1);
  ME_CHECK(vec || c <= 256, "global pooling of more than 256 channels needs a channel count that is a multiple of 4");
  ME_CHECK(lds <= 64 * 1024, "channel count too large for global pooling");
  if (mode == 2) {
    // -FLT_MAX / -1 start values (0xff7fffff = -FLT_MAX as a 32-bit pattern)
    ME_HIP(hipMemsetAsync(partial_arg, 0xff, (size_t)chunks * total * 4, stream));
    ME_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(partial), (int)0xff7fffff, (size_t)chunks * total, stream));
    if (n > 0) {
      if (vec)
        hipLaunchKernelGGL((k_global_partial<T, true, 4>), dim3((unsigned)chunks), dim3(256), lds, stream, src, src2, c,
                           batch_row, n, chunk_rows, n_batch, partial, partial_arg, partial_cnt);
      else
        hipLaunchKernelGGL((k_global_partial<T, true, 1>), dim3((unsigned)chunks), dim3(256), lds, stream, src, src2, c,
                           batch_row, n, chunk_rows, n_batch, partial, partial_arg, partial_cnt);
      ME_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_global_final<true>, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, stream, partial,
                       partial_arg, partial_cnt, chunks, n_batch, c, 0, dst, dst_arg, dst_count);
  } else {
    ME_HIP(hipMemsetAsync(partial, 0, (size_t)chunks * total * 4, stream));
    ME_HIP(hipMemsetAsync(partial_cnt, 0, (size_t)chunks * n_batch * 4, stream));
    if (n > 0) {
      if (vec)
        hipLaunchKernelGGL((k_global_partial<T, false, 4>), dim3((unsigned)chunks), dim3(256), lds, stream, src, src2, c,
                           batch_row, n, chunk_rows, n_batch, partial, partial_arg, partial_cnt);
      else
        hipLaunchKernelGGL((k_global_partial<T, false, 1>), dim3((unsigned)chunks), dim3(256), lds, stream, src, src2, c,
                           batch_row, n, chunk_rows, n_batch, partial, partial_arg, partial_cnt);
      ME_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_global_final<false>, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, stream, partial,
                       partial_arg, partial_cnt, chunks, n_batch, c, mode == 1, dst, dst_arg, dst_count);
  }
  ME_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int64_t me_global_pool_workspace_bytes(int64_t n, int32_t n_batch, int32_t c) {
  const int64_t chunks = ceil_div(n < 1 ? 1 : n, global_chunk_rows(n));
  // partial values | partial argmax | partial counts
  return align_up(chunks * n_batch * c * 4, 256) * 2 + align_up(chunks * n_batch * 4, 256);
}

int me_pool_sum_f32(const float *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume,
                    const float *src_count, int32_t average, float *dst, float *dst_count, void *stream) {
  return pool_sum<float>(src, c, tbl, n_tgt, volume, src_count, average, dst, dst_count, (hipStream_t)stream);
}
int me_pool_sum_bf16(const uint16_t *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume,
                     const float *src_count, int32_t average, uint16_t *dst, float *dst_count, void *stream) {
  return pool_sum<__bf16>((const __bf16 *)src, c, tbl, n_tgt, volume, src_count, average, (__bf16 *)dst, dst_count,
                          (hipStream_t)stream);
}

int me_pool_max_f32(const float *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume, float *dst,
                    int32_t *mask, void *stream) {
  return pool_max<float>(src, c, tbl, n_tgt, volume, dst, mask, (hipStream_t)stream);
}
int me_pool_max_bf16(const uint16_t *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume, uint16_t *dst,
                     int32_t *mask, void *stream) {
  return pool_max<__bf16>((const __bf16 *)src, c, tbl, n_tgt, volume, (__bf16 *)dst, mask, (hipStream_t)stream);
}

int me_pool_max_backward_f32(const float *grad_out, int32_t c, const int32_t *tbl_in, int64_t n_in,
                             int64_t volume, const int32_t *mask, float *grad_in, void *stream) {
  return pool_max_backward<float>(grad_out, c, tbl_in, n_in, volume, mask, grad_in, (hipStream_t)stream);
}
int me_pool_max_backward_bf16(const uint16_t *grad_out, int32_t c, const int32_t *tbl_in, int64_t n_in,
                              int64_t volume, const int32_t *mask, uint16_t *grad_in, void *stream) {
  return pool_max_backward<__bf16>((const __bf16 *)grad_out, c, tbl_in, n_in, volume, mask, (__bf16 *)grad_in,
                                   (hipStream_t)stream);
}

int me_global_pool_f32(const float *src, const float *src2, int32_t c, const int32_t *batch_row, int64_t n,
                       int32_t n_batch, int32_t mode, float *dst, int32_t *dst_arg, float *dst_count,
                       void *workspace, int64_t workspace_bytes, void *stream) {
  return global_pool<float>(src, src2, c, batch_row, n, n_batch, mode, dst, dst_arg, dst_count, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
int me_global_pool_bf16(const uint16_t *src, const uint16_t *src2, int32_t c, const int32_t *batch_row, int64_t n,
                        int32_t n_batch, int32_t mode, float *dst, int32_t *dst_arg, float *dst_count,
                        void *workspace, int64_t workspace_bytes, void *stream) {
  return global_pool<__bf16>((const __bf16 *)src, (const __bf16 *)src2, c, batch_row, n, n_batch, mode, dst, dst_arg,
                             dst_count, workspace, workspace_bytes, (hipStream_t)stream);
}

int me_broadcast_f32(const float *in, const float *glob, const int32_t *batch_row, int64_t n, int32_t c,
                     int32_t multiply, float *out, void *stream) {
  return broadcast<float>(in, glob, batch_row, n, c, multiply, out, (hipStream_t)stream);
}
int me_broadcast_bf16(const uint16_t *in, const uint16_t *glob, const int32_t *batch_row, int64_t n, int32_t c,
                      int32_t multiply, uint16_t *out, void *stream) {
  return broadcast<__bf16>((const __bf16 *)in, (const __bf16 *)glob, batch_row, n, c, multiply, (__bf16 *)out,
                           (hipStream_t)stream);
}

int me_segment_sum_f32(const float *src, int32_t c, const int64_t *perm, const int64_t *seg_offsets, int64_t n_seg,
                       int32_t average, float *dst, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(c > 0, "invalid channel count");
  if (n_seg == 0) return 0;
  const bool vec = (c % 4) == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
  const int64_t total = n_seg * (vec ? c / 4 : c);
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (vec) hipLaunchKernelGGL(k_segment_sum<4>, grid, block, 0, stream, src, c, perm, seg_offsets, n_seg, average, dst);
  else hipLaunchKernelGGL(k_segment_sum<1>, grid, block, 0, stream, src, c, perm, seg_offsets, n_seg, average, dst);
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_pool(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_pool_sum<float, 4>));
}
