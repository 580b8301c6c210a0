// Would wave specialisation help the fp32 tile kernel?  Two kernels with the SAME per-batch work as
// k_conv_tile_f32<64,64> on config 2 (4 groups of 16 gathered rows x 64 channels per batch, W_k slice of
// 16 registers per consumer wave, 64 MFMAs per wave per batch, LDS accumulate at scattered rows):
//   unified      every wave gathers, stages, multiplies (4 waves, 2 barriers per batch)  = the shipped structure
//   specialised  4 producer waves gather + stage into a double buffer, 4 consumer waves multiply
//                (8 waves, 1 barrier per batch)
// Build: hipcc --offload-arch=gfx950 -O3 -o ws_probe ws_probe.hip ; ./ws_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));        \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int kRows = 100000, kC = 64, kTile = 160, kAccLd = 68;   // accumulator rows padded like the kernel
constexpr int kStageWords = 64 * 64;                               // 4 groups x 16 rows x 64 channels

// one batch of the consumer side: operands from `stage`, weights in w[16], accumulate at rows drow0/1
__device__ __forceinline__ void multiply(const float *stage, float *acc, const float (&w)[16], int lane, int wave,
                                         int it) {
  const f32x4 *st4 = reinterpret_cast<const f32x4 *>(stage);
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    f32x4 s[8];
    // group g row (lane & 15), 16-byte piece (lane >> 4) + 4 * i : 4 pieces per group, two groups
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = st4[((2 * pair) * 16 + (lane & 15)) * 16 + (((lane >> 4) + 4 * i) ^ (lane & 15))];
      s[4 + i] = st4[((2 * pair + 1) * 16 + (lane & 15)) * 16 + (((lane >> 4) + 4 * i) ^ (lane & 15))];
    }
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * i + j], s[i][j], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * i + j], s[4 + i][j], a1, 0, 0, 0);
      }
    // scattered accumulate: row depends on (it, pair, lane & 15), 16 columns of this wave
    const int r0 = ((it * 37 + pair * 11 + (lane & 15) * 7) * 13) % kTile;
    const int r1 = ((it * 37 + pair * 11 + 5 + (lane & 15) * 7) * 13 + 3) % kTile;
    f32x4 *p0 = reinterpret_cast<f32x4 *>(acc + r0 * kAccLd + wave * 16 + (lane >> 4) * 4);
    f32x4 *p1 = reinterpret_cast<f32x4 *>(acc + r1 * kAccLd + wave * 16 + (lane >> 4) * 4);
    f32x4 c0 = *p0, c1 = *p1;
    *p0 = c0 + a0;
    *p1 = c1 + a1;
  }
}

__device__ __forceinline__ void load_w(float (&w)[16], const float *wpack, int k, int wave, int lane) {
  const f32x4 *p = reinterpret_cast<const f32x4 *>(wpack) + ((size_t)(k % 27) * 4 + wave) * 256 + lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 v = p[i * 64];
    w[4 * i] = v[0], w[4 * i + 1] = v[1], w[4 * i + 2] = v[2], w[4 * i + 3] = v[3];
  }
}

// ---- unified: the shipped structure ---------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void k_unified(const float *feat, const float *wpack, const unsigned *plan,
                                                     float *out, int iters) {
  extern __shared__ float lds[];
  float *acc = lds, *stage = lds + kTile * kAccLd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kTile * kAccLd + kStageWords; i += 256) lds[i] = 0.f;
  const unsigned *myplan = plan + (size_t)blockIdx.x * iters * 64;
  // thread t stages piece (t & 15) of rows (t >> 4) + 16 g, g = 0..3
  unsigned idx[4], idx_n[4];
  f32x4 g[4];
  float w[16], wn[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) idx[i] = myplan[(tid >> 4) + 16 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const f32x4 *>(feat + (size_t)idx[i] * kC)[tid & 15];
#pragma unroll
  for (int i = 0; i < 4; ++i) idx_n[i] = myplan[64 + (tid >> 4) + 16 * i];
  load_w(w, wpack, 0, wave, lane);
  for (int it = 0; it < iters; ++it) {
    __syncthreads();                                   // barrier A: everyone done reading the stage
#pragma unroll
    for (int i = 0; i < 4; ++i)
      reinterpret_cast<f32x4 *>(stage)[((tid >> 4) + 16 * i) * 16 + ((tid & 15) ^ ((tid >> 4) & 15))] = g[i];
    __syncthreads();                                   // barrier B: stage visible
    // next batch: weights, gather, and the indices of the batch after it
    load_w(wn, wpack, it + 1, wave, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const f32x4 *>(feat + (size_t)idx_n[i] * kC)[tid & 15];
    const int nn = it + 2 < iters ? it + 2 : it;
#pragma unroll
    for (int i = 0; i < 4; ++i) idx_n[i] = myplan[(size_t)nn * 64 + (tid >> 4) + 16 * i];
    __builtin_amdgcn_sched_barrier(0);                 // keep the loads ahead of the multiply
    multiply(stage, acc, w, lane, wave, it);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wn[i];
  }
  __syncthreads();
  float s = 0;
  for (int i = tid; i < kTile * kAccLd; i += 256) s += acc[i];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

// ---- specialised: producers (waves 4..7) gather and stage, consumers (waves 0..3) multiply ----------
__global__ __launch_bounds__(512, 2) void k_special(const float *feat, const float *wpack, const unsigned *plan,
                                                     float *out, int iters) {
  extern __shared__ float lds[];
  float *acc = lds, *stage = lds + kTile * kAccLd;     // two stage buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kTile * kAccLd + 2 * kStageWords; i += 512) lds[i] = 0.f;
  const unsigned *myplan = plan + (size_t)blockIdx.x * iters * 64;
  __syncthreads();
  if (wave >= 4) {
    const int t = tid - 256;
    unsigned idx[4];
    f32x4 g[4], gn[4];
    // prologue: batch 0 staged, batch 1 in flight, indices of batch 2 loaded
#pragma unroll
    for (int i = 0; i < 4; ++i) idx[i] = myplan[(t >> 4) + 16 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const f32x4 *>(feat + (size_t)idx[i] * kC)[t & 15];
#pragma unroll
    for (int i = 0; i < 4; ++i) idx[i] = myplan[64 + (t >> 4) + 16 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      reinterpret_cast<f32x4 *>(stage)[((t >> 4) + 16 * i) * 16 + ((t & 15) ^ ((t >> 4) & 15))] = g[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const f32x4 *>(feat + (size_t)idx[i] * kC)[t & 15];
#pragma unroll
    for (int i = 0; i < 4; ++i) idx[i] = myplan[128 + (t >> 4) + 16 * i];
    __syncthreads();                                   // batch 0 visible
    // two batches per trip so that the gathered rows never move between registers (a copy would wait for the
    // gather that was only just issued)
    // indices of batch it + 3 go out BEFORE the gather of batch it + 2, so that waiting for them at the top of the
    // next step does not wait for that gather (loads retire in order)
    unsigned idx2[4];
    auto step = [&](int it, f32x4 (&fly)[4], f32x4 (&landed)[4], unsigned (&iuse)[4], unsigned (&iload)[4]) {
      __builtin_amdgcn_sched_barrier(0);               // address arithmetic of this batch stays below the barrier
      const int nn = it + 3 < iters ? it + 3 : it;
#pragma unroll
      for (int i = 0; i < 4; ++i) iload[i] = myplan[(size_t)nn * 64 + (t >> 4) + 16 * i];
#pragma unroll
      for (int i = 0; i < 4; ++i) fly[i] = reinterpret_cast<const f32x4 *>(feat + (size_t)iuse[i] * kC)[t & 15];
      __builtin_amdgcn_sched_barrier(0);
      float *buf = stage + ((it + 1) & 1) * kStageWords;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        reinterpret_cast<f32x4 *>(buf)[((t >> 4) + 16 * i) * 16 + ((t & 15) ^ ((t >> 4) & 15))] = landed[i];
      __syncthreads();
    };
    for (int it = 0; it < iters; it += 2) {
      step(it, gn, g, idx, idx2);
      step(it + 1, g, gn, idx2, idx);
    }
  } else {
    float w[16], wn[16];
    load_w(w, wpack, 0, wave, lane);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      load_w(wn, wpack, it + 1, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      multiply(stage + (it & 1) * kStageWords, acc, w, lane, wave, it);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = wn[i];
      __syncthreads();
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = tid; i < kTile * kAccLd; i += 512) s += acc[i];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;     // batches per workgroup (the kernel: 32 per tile)
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;     // full rounds of resident workgroups
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *feat, *wpack, *out;
  unsigned *plan;
  const int max_grid = cus * 3 * rounds;
  CHECK(hipMalloc(&feat, (size_t)kRows * kC * 4));
  CHECK(hipMalloc(&wpack, (size_t)27 * 4 * 256 * 16));
  CHECK(hipMalloc(&out, (size_t)max_grid * 512 * 4));
  CHECK(hipMalloc(&plan, (size_t)max_grid * iters * 64 * 4));
  std::vector<float> hf((size_t)kRows * kC);
  for (size_t i = 0; i < hf.size(); ++i) hf[i] = 1e-3f * (float)(i % 97);
  CHECK(hipMemcpy(feat, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(wpack, hf.data(), (size_t)27 * 4 * 256 * 16, hipMemcpyHostToDevice));
  std::vector<unsigned> hp((size_t)max_grid * iters * 64);
  unsigned s = 12345;
  // a tile's rows come from a neighbourhood: rows within +-2000 of a per-workgroup base (Z-order locality is absent
  // in the real map too: unsorted rows), drawn uniformly
  for (size_t i = 0; i < hp.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    hp[i] = (s >> 8) % kRows;
  }
  CHECK(hipMemcpy(plan, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto report = [&](const char *name, int grid, float ms) {
    const double mfma = (double)grid * 4 * iters * 64;
    printf("%-34s grid %5d  %8.3f ms  %6.1f TFLOP/s  %5.1f us per 32 batches per round\n", name, grid, ms,
           mfma * 2048 / (ms * 1e-3) / 1e12, ms * 1e3 / rounds * 32 / iters);
  };
  const int lds_u = (kTile * kAccLd + kStageWords) * 4, lds_s = (kTile * kAccLd + 2 * kStageWords) * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unified), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_special), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("LDS per workgroup: unified %d B, specialised %d B\n", lds_u, lds_s);
  for (int wpc : {2, 1}) {
    const int lds = wpc == 2 ? lds_u : 100 * 1024;
    const int grid = cus * wpc * rounds;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_unified, dim3(grid), dim3(256), lds, 0, feat, wpack, plan, out, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      char name[64];
      snprintf(name, sizeof name, "unified, %d workgroups/CU", wpc);
      if (rep) report(name, grid, ms);
    }
  }
  for (int wpc : {2, 1}) {
    const int lds = wpc == 2 ? lds_s : 100 * 1024;
    const int grid = cus * wpc * rounds;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_special, dim3(grid), dim3(512), lds, 0, feat, wpack, plan, out, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      char name[64];
      snprintf(name, sizeof name, "specialised, %d workgroups/CU", wpc);
      if (rep) report(name, grid, ms);
    }
  }
  CHECK(hipGetLastError());
  return 0;
}
