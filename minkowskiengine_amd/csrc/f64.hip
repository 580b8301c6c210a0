// float64 features on the same coordinate / kernel maps (round 4).
//
// The reference instantiates every feature operator for float AND double (AT_DISPATCH_FLOATING_TYPES,
// src/convolution_gpu.cu:137-155, the same in the pooling / broadcast translation units), and its own tests are float64
// `gradcheck`s (MinkowskiEngine/utils/gradcheck.py:34-57, tests/python/convolution.py, pooling.py, broadcast.py).  The
// float64 path exists for exactly that — checking gradients and arbitrating fp32 results — so these kernels are
// correctness-first: plain `fma` in double, one thread per output element, TARGET-stationary on the dense neighbour
// tables like pool.hip (no atomics: every sum runs over the kernel offsets in ascending k, then the channels in
// ascending order — deterministic, and the order of the reference's CPU loops, src/convolution_kernel.hpp:62-70).
// No tile plan, no packed weights: the fp32 / bf16 kernels are the product's hot path, this is its yardstick.
#include "common.hpp"

#include <float.h>

namespace me {

// dst[t][j] = sum_k sum_i src[tbl[k][t]][i] * W_k[i][j];  W_k[i][j] = w[k][i][j] (forward: w is [K, c_src, c_dst])
// or w[k][j][i] (TRANSPOSED, the input gradient: w is the forward kernel [K, c_dst, c_src]).
// A thread owns one (target row, column); the 16 columns of a quarter wave read neighbouring weights.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void k_conv_target_f64(const double *__restrict__ src, int c_src,
                                                        const double *__restrict__ w, int c_dst,
                                                        const int32_t *__restrict__ tbl, int64_t n_tgt, int volume,
                                                        double *__restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tgt * c_dst) return;
  const int64_t t = idx / c_dst;
  const int j = (int)(idx % c_dst);
  double acc = 0.0;
  for (int k = 0; k < volume; ++k) {
    const int32_t s = tbl[(int64_t)k * n_tgt + t];
    if (s < 0) continue;
    const double *x = src + (int64_t)s * c_src;
    const double *wk = w + (int64_t)k * c_src * c_dst;
    if (TRANSPOSED) {
      const double *wr = wk + (int64_t)j * c_src;          // w[k][j][i], i contiguous
      for (int i = 0; i < c_src; ++i) acc = fma(x[i], wr[i], acc);
    } else {
      for (int i = 0; i < c_src; ++i) acc = fma(x[i], wk[(int64_t)i * c_dst + j], acc);
    }
  }
  dst[idx] = acc;
}

// grad_w[k][i][j] = sum over the pairs e of offset k (ascending) of x[in[e]][i] * dy[out[e]][j]
// (src/convolution_kernel.hpp:128-142).  One workgroup per (k, i); thread = column j, strided when c_out > 256.
__global__ __launch_bounds__(256) void k_conv_wgrad_f64(const double *__restrict__ x, int c_in,
                                                       const double *__restrict__ dy, int c_out,
                                                       const int32_t *__restrict__ in_pairs,
                                                       const int32_t *__restrict__ out_pairs,
                                                       const int64_t *__restrict__ koffs, double *__restrict__ grad_w) {
  const int k = blockIdx.x / c_in, i = blockIdx.x % c_in;
  const int64_t e0 = koffs[k], e1 = koffs[k + 1];
  for (int j = threadIdx.x; j < c_out; j += blockDim.x) {
    double acc = 0.0;
    for (int64_t e = e0; e < e1; ++e)
      acc = fma(x[(int64_t)in_pairs[e] * c_in + i], dy[(int64_t)out_pairs[e] * c_out + j], acc);
    grad_w[((int64_t)k * c_in + i) * c_out + j] = acc;
  }
}

// ---- pooling / broadcast: the kernels of pool.hip in double, one element per thread -------------------------------------
__global__ __launch_bounds__(256) void k_pool_sum_f64(const double *__restrict__ src, int c,
                                                     const int32_t *__restrict__ tbl, int64_t n_tgt, int volume,
                                                     const float *__restrict__ src_count, int average,
                                                     double *__restrict__ dst, float *__restrict__ dst_count) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tgt * c) return;
  const int64_t t = idx / c;
  const int ch = (int)(idx % c);
  double acc = 0.0;
  float cnt = 0.f;
  for (int k = 0; k < volume; ++k) {
    const int32_t s = tbl[(int64_t)k * n_tgt + t];
    if (s < 0) continue;
    const double v = src[(int64_t)s * c + ch];
    if (src_count) {
      const float d = src_count[s];
      if (d > 0.f) acc += v / (double)d;
    } else {
      acc += v;
    }
    cnt += 1.f;
  }
  if (average && cnt > 0.f) acc /= (double)cnt;
  dst[idx] = acc;
  if (dst_count && ch == 0) dst_count[t] = cnt;
}

__global__ __launch_bounds__(256) void k_pool_max_f64(const double *__restrict__ src, int c,
                                                     const int32_t *__restrict__ tbl, int64_t n_tgt, int volume,
                                                     double *__restrict__ dst, int32_t *__restrict__ mask) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tgt * c) return;
  const int64_t t = idx / c;
  const int ch = (int)(idx % c);
  double best = -DBL_MAX;
  int32_t arg = -1;
  for (int k = 0; k < volume; ++k) {
    const int32_t s = tbl[(int64_t)k * n_tgt + t];
    if (s < 0) continue;
    const double v = src[(int64_t)s * c + ch];
    if (best < v) {
      best = v;
      arg = s * c + ch;
    }
  }
  dst[idx] = best;
  mask[idx] = arg;
}

__global__ __launch_bounds__(256) void k_pool_max_backward_f64(const double *__restrict__ grad_out, int c,
                                                              const int32_t *__restrict__ tbl_in, int64_t n_in,
                                                              int volume, const int32_t *__restrict__ mask,
                                                              double *__restrict__ grad_in) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_in * c) return;
  const int64_t i = idx / c;
  const int ch = (int)(idx % c);
  double acc = 0.0;
  for (int k = 0; k < volume; ++k) {
    const int32_t o = tbl_in[(int64_t)k * n_in + i];
    if (o < 0) continue;
    if (mask[(int64_t)o * c + ch] == (int32_t)idx) acc += grad_out[(int64_t)o * c + ch];
  }
  grad_in[idx] = acc;
}

// dst[b][ch] = sum | mean | max over the rows r with batch_row[r] == b of src[r][ch] (* src2[r][ch] when given), rows in
// ascending order; one thread per (b, ch) walks all rows (a yardstick, not a hot path)
__global__ __launch_bounds__(256) void k_global_pool_f64(const double *__restrict__ src, const double *__restrict__ src2,
                                                        int c, const int32_t *__restrict__ batch_row, int64_t n,
                                                        int n_batch, int mode, double *__restrict__ dst,
                                                        int32_t *__restrict__ dst_arg, float *__restrict__ dst_cnt) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_batch * c) return;
  const int b = (int)(idx / c), ch = (int)(idx % c);
  double a = mode == 2 ? -DBL_MAX : 0.0;
  int32_t arg = -1;
  float cnt = 0.f;
  for (int64_t r = 0; r < n; ++r) {
    if (batch_row[r] != b) continue;
    double v = src[r * c + ch];
    if (src2) v *= src2[r * c + ch];
    if (mode == 2) {
      if (a < v) {
        a = v;
        arg = (int32_t)(r * c + ch);
      }
    } else {
      a += v;
    }
    cnt += 1.f;
  }
  if (mode == 1 && cnt > 0.f) a /= (double)cnt;
  dst[idx] = a;
  if (dst_arg) dst_arg[idx] = arg;
  if (dst_cnt && ch == 0) dst_cnt[b] = cnt;
}

__global__ __launch_bounds__(256) void k_broadcast_f64(const double *__restrict__ in, const double *__restrict__ glob,
                                                      const int32_t *__restrict__ batch_row, int64_t n, int c,
                                                      int multiply, double *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * c) return;
  const int64_t i = idx / c;
  const double g = glob[(int64_t)batch_row[i] * c + idx % c];
  out[idx] = in ? (multiply ? in[idx] * g : in[idx] + g) : g;
}

}  // namespace me

using namespace me;

extern "C" {

int me_conv_target_f64(const double *src, int64_t n_src, int32_t c_src, const double *w, int32_t transposed,
                       int64_t volume, int32_t c_dst, const int32_t *tbl, double *dst, int64_t n_tgt, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)n_src;
  ME_CHECK(c_src > 0 && c_dst > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_tgt == 0) return 0;
  const dim3 grid((unsigned)ceil_div(n_tgt * c_dst, 256)), block(256);
  if (transposed)
    hipLaunchKernelGGL(k_conv_target_f64<true>, grid, block, 0, stream, src, c_src, w, c_dst, tbl, n_tgt, (int)volume, dst);
  else
    hipLaunchKernelGGL(k_conv_target_f64<false>, grid, block, 0, stream, src, c_src, w, c_dst, tbl, n_tgt, (int)volume, dst);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_conv_wgrad_f64(const double *x, int32_t c_in, const double *dy, int32_t c_out, const int32_t *in_pairs,
                      const int32_t *out_pairs, const int64_t *k_offsets_dev, int64_t volume, double *grad_w,
                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(c_in > 0 && c_out > 0 && volume >= 1 && volume * c_in < (1ll << 31), "invalid weight shape");
  hipLaunchKernelGGL(k_conv_wgrad_f64, dim3((unsigned)(volume * c_in)), dim3(c_out >= 256 ? 256 : (c_out + 63) / 64 * 64),
                     0, stream, x, c_in, dy, c_out, in_pairs, out_pairs, k_offsets_dev, grad_w);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_pool_sum_f64(const double *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume,
                    const float *src_count, int32_t average, double *dst, float *dst_count, void *stream_) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_tgt == 0) return 0;
  hipLaunchKernelGGL(k_pool_sum_f64, dim3((unsigned)ceil_div(n_tgt * c, 256)), dim3(256), 0, (hipStream_t)stream_, src, c,
                     tbl, n_tgt, (int)volume, src_count, average, dst, dst_count);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_pool_max_f64(const double *src, int32_t c, const int32_t *tbl, int64_t n_tgt, int64_t volume, double *dst,
                    int32_t *mask, void *stream_) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_tgt == 0) return 0;
  hipLaunchKernelGGL(k_pool_max_f64, dim3((unsigned)ceil_div(n_tgt * c, 256)), dim3(256), 0, (hipStream_t)stream_, src, c,
                     tbl, n_tgt, (int)volume, dst, mask);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_pool_max_backward_f64(const double *grad_out, int32_t c, const int32_t *tbl_in, int64_t n_in, int64_t volume,
                             const int32_t *mask, double *grad_in, void *stream_) {
  ME_CHECK(c > 0 && volume >= 1 && volume <= 65535, "invalid channel count or kernel volume");
  if (n_in == 0) return 0;
  hipLaunchKernelGGL(k_pool_max_backward_f64, dim3((unsigned)ceil_div(n_in * c, 256)), dim3(256), 0, (hipStream_t)stream_,
                     grad_out, c, tbl_in, n_in, (int)volume, mask, grad_in);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_global_pool_f64(const double *src, const double *src2, int32_t c, const int32_t *batch_row, int64_t n,
                       int32_t n_batch, int32_t mode, double *dst, int32_t *dst_arg, float *dst_count, void *stream_) {
  ME_CHECK(c > 0 && n_batch >= 0 && mode >= 0 && mode <= 2, "invalid global pooling arguments");
  if (n_batch == 0) return 0;
  hipLaunchKernelGGL(k_global_pool_f64, dim3((unsigned)ceil_div((int64_t)n_batch * c, 256)), dim3(256), 0,
                     (hipStream_t)stream_, src, src2, c, batch_row, n, n_batch, mode, dst, dst_arg, dst_count);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_broadcast_f64(const double *in, const double *glob, const int32_t *batch_row, int64_t n, int32_t c,
                     int32_t multiply, double *out, void *stream_) {
  ME_CHECK(c > 0, "invalid channel count");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_broadcast_f64, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, (hipStream_t)stream_, in, glob,
                     batch_row, n, c, multiply, out);
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_f64(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_conv_target_f64<true>));
}
