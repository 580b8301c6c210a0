// Sustained v_mfma_f32_16x16x4_f32 rate on gfx950 under different co-running loads.  The roofline peak
// (157.3 TFLOP/s) assumes every SIMD issues one MFMA per 32 cycles at 2.4 GHz; this measures what a loop
// of register-operand MFMAs actually sustains (power/clock management included) with 1..3 waves per SIMD,
// and the same loop with the LDS operand reads / accumulator read-add-write / VALU work of the conv tile
// kernel mixed in.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
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

// MODE 0: MFMAs only (4 accumulator chains, operands in registers)
// MODE 1: + 8 ds_read_b128 operand reads per 32 MFMAs (the conv kernel's ratio), data fed to the MFMAs
// MODE 2: MODE 1 + accumulator read-add-write in LDS per 32 MFMAs
// MODE 3: MODE 2 + 4 global 16-byte loads per 64 MFMAs (gather stand-in, L2-resident)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, const float *src, int iters, int lds_words) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 1e-3f * (float)(i & 15);
  __syncthreads();
  f32x4 acc[4];
  float w[16];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) w[i] = 1e-3f * (float)(lane + i);
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = 1e-3f * (float)(lane - i);
  const f32x4 *stage = reinterpret_cast<const f32x4 *>(lds) + lane;       // 64 lanes x 16 B, conflict-free
  f32x4 *accp = reinterpret_cast<f32x4 *>(lds) + 2048 + wave * 256 + lane;
  const f32x4 *gsrc = reinterpret_cast<const f32x4 *>(src) + (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 g[4] = {};
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = __builtin_nontemporal_load(gsrc + ((it * 4 + i) & 63) * 65536);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (MODE >= 1) {
        f32x4 s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = stage[(i + half * 8 + (it & 1) * 16) * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          b[2 * i] = s[i][0] + s[i][2];
          b[2 * i + 1] = s[i][1] + s[i][3];
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[2 * half] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], b[q], acc[2 * half], 0, 0, 0);
        acc[2 * half + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], b[15 - q], acc[2 * half + 1], 0, 0, 0);
      }
      if (MODE >= 2) {
        f32x4 a0 = accp[0], a1 = accp[64];
        accp[0] = a0 + acc[2 * half];
        accp[64] = a1 + acc[2 * half + 1];
        acc[2 * half] = f32x4{0, 0, 0, 0};
        acc[2 * half + 1] = f32x4{0, 0, 0, 0};
      }
    }
    if (MODE >= 3) {
      f32x4 *st = reinterpret_cast<f32x4 *>(lds) + 1024 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) st[i * 256] = g[i];
    }
  }
  f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
  if (MODE >= 2) r += accp[0];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

template <int MODE>
static void run(int wg_per_cu, int iters, float *out, const float *src, int cus) {
  // LDS padding sets the occupancy: 160 KB per CU
  const int lds_bytes = wg_per_cu == 1 ? 120 * 1024 : wg_per_cu == 2 ? 72 * 1024 : 48 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rate<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes));
  const int grid = cus * wg_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int lds_words = 12 * 1024;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), lds_bytes, 0, out, src, iters, lds_words);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)grid * 4 * iters * 64;
    const double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    if (rep == 1)
      printf("mode %d  %d wave(s)/SIMD  %8.3f ms  %6.1f TFLOP/s  = %4.1f %% of 157.3  (issue-bound clock %.2f GHz)\n", MODE,
             wg_per_cu, ms, tf, 100 * tf / 157.3, mfma * 32 / (cus * 4) / (ms * 1e-3) / 1e9);
  }
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, clock %d MHz; %d iterations of 64 MFMAs per wave\n", p.name, cus, p.clockRate / 1000, iters);
  float *out, *src;
  CHECK(hipMalloc(&out, (size_t)cus * 3 * 256 * 4));
  CHECK(hipMalloc(&src, (size_t)64 * 65536 * 16 + (size_t)cus * 3 * 256 * 16));
  CHECK(hipMemset(src, 0, (size_t)64 * 65536 * 16 + (size_t)cus * 3 * 256 * 16));
  for (int w = 1; w <= 3; ++w) run<0>(w, iters, out, src, cus);
  for (int w = 1; w <= 3; ++w) run<1>(w, iters, out, src, cus);
  for (int w = 1; w <= 3; ++w) run<2>(w, iters, out, src, cus);
  for (int w = 1; w <= 3; ++w) run<3>(w, iters, out, src, cus);
  // short launches (the conv kernel runs ~0.2 ms): does a cold start clock differently?
  for (int it : {50, 100, 200}) {
    printf("short launch, %d iterations: ", it);
    run<0>(3, it, out, src, cus);
  }
  return 0;
}
