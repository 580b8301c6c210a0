// Weight packing for MANY convolution layers in one launch (round 3).
//
// The tile kernels read their weights as the register image of the MFMA A operand (k_pack_weights_bf16 in
// conv_bf16.hip, k_pack_weights_f32x3 in conv_f32x3.hip), packed from the reference layout kernel[K, Cin, Cout]
// (src/convolution_kernel.hpp:62-70) — once for the forward launch and once, transposed, for the input gradient.  A
// MinkUNet34C training step therefore issued 126 pack launches of 3 - 6 us (5 % of a bf16 step,
// profiles/r02_rocprof_kernel_stats_minkunet34c_bf16_final.csv), all of them independent of any activation: the
// weights only change at the optimizer step.  me_conv_pack_weights_multi packs any number of (layer, direction) jobs
// with ONE launch: a thread finds its job by binary search in the prefix of the jobs' element counts and then does
// exactly what the single-layer pack kernels do (same layouts, same rounding / exact split: bit-identical images).
#include "conv_common.hpp"

namespace me {

constexpr int kPackMaxJobs = 1024;

__global__ __launch_bounds__(256) void k_pack_weights_multi(const me_pack_job *__restrict__ jobs, int n_jobs,
                                                           const int64_t *__restrict__ prefix) {
  __shared__ int64_t s_prefix[kPackMaxJobs + 1];
  for (int i = threadIdx.x; i <= n_jobs; i += blockDim.x) s_prefix[i] = prefix[i];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= s_prefix[n_jobs]) return;
  int lo = 0, hi = n_jobs;            // largest j with prefix[j] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (s_prefix[mid] <= t) lo = mid; else hi = mid;
  }
  const me_pack_job jb = jobs[lo];
  const int64_t e = t - s_prefix[lo];
  const int KS = jb.kc / 32;
  const int lane = (int)(e % 64);
  int64_t r = e / 64;
  const int v = (int)(r % KS);
  r /= KS;
  const int cb = (int)(r % jb.ncb);
  r /= jb.ncb;
  const int c = (int)(r % jb.nchunks);
  const int64_t k = r / jb.nchunks;
  const int q = lane >> 4, i16 = lane & 15;
  const int col = cb * 16 + i16;
  float val[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = c * jb.kc + v * 32 + q * 8 + j;
    val[j] = 0.f;
    if (ch < jb.c_src && col < jb.c_dst) {
      // plain: w is [K, c_src, c_dst]; transposed (dgrad): w is the forward kernel [K, c_dst, c_src]
      const int64_t idx = jb.transposed ? (k * jb.c_dst + col) * jb.c_src + ch : (k * jb.c_src + ch) * jb.c_dst + col;
      val[j] = jb.w_is_f32 ? reinterpret_cast<const float *>(jb.w)[idx]
                           : (float) reinterpret_cast<const __bf16 *>(jb.w)[idx];
    }
  }
  if (jb.mode == ME_PACK_BF16) {
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (__bf16)val[j];   // RNE, as k_pack_weights_bf16
    reinterpret_cast<bf16x8 *>(jb.wp)[e] = out;
  } else {                                                  // ME_PACK_F32X3: three bf16 planes of the exact split
    u32x4 p1, p2, p3;
    split3<true>(f32x4{val[0], val[1], val[2], val[3]}, f32x4{val[4], val[5], val[6], val[7]}, p1, p2, p3);
    u32x4 *wp = reinterpret_cast<u32x4 *>(jb.wp);
    const int64_t base = ((((k * jb.nchunks + c) * jb.ncb + cb) * 3) * KS + v) * 64 + lane;
    wp[base] = p1;
    wp[base + (int64_t)KS * 64] = p2;
    wp[base + (int64_t)2 * KS * 64] = p3;
  }
}

}  // namespace me

using namespace me;

extern "C" {

// geometry of one job (host): fills kc / nchunks / ncb / threads from the tile kernel's own variant selection
int me_conv_pack_job_init(me_pack_job *job) {
  ME_CHECK(job != nullptr && job->volume >= 1 && job->c_src > 0 && job->c_dst > 0, "invalid pack job");
  int32_t kc = 0;
  if (job->mode == ME_PACK_BF16) kc = me_conv_pack_chunk_bf16(job->c_src, job->c_dst);
  else if (job->mode == ME_PACK_F32X3) kc = me_conv_pack_chunk_f32x3(job->c_src, job->c_dst);
  ME_CHECK(kc >= 32 && kc % 32 == 0, "unknown pack mode");
  ME_CHECK(job->mode != ME_PACK_F32X3 || job->w_is_f32, "the split kernels take fp32 weights");
  job->kc = kc;
  job->nchunks = (int32_t)ceil_div(job->c_src, kc);
  job->ncb = (int32_t)ceil_div(job->c_dst, 16);
  job->threads = job->volume * job->nchunks * job->ncb * (kc / 32) * 64;
  return 0;
}

int me_conv_pack_weights_multi(const me_pack_job *jobs_dev, int32_t n_jobs, const int64_t *thread_prefix_dev,
                               int64_t total_threads, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n_jobs >= 0 && n_jobs <= kPackMaxJobs, "at most 1024 pack jobs per launch");
  if (n_jobs == 0 || total_threads <= 0) return 0;
  ME_CHECK(jobs_dev != nullptr && thread_prefix_dev != nullptr, "job table / prefix must not be null");
  hipLaunchKernelGGL(k_pack_weights_multi, dim3((unsigned)ceil_div(total_threads, 256)), dim3(256), 0, stream, jobs_dev,
                     (int)n_jobs, thread_prefix_dev);
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_pack(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_pack_weights_multi));
}
