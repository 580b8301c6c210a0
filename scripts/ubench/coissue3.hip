// Third co-issue probe on gfx950 (see coissue.hip, coissue2.hip):
//   T1  roles swapped: the v_add waves are the OLDER waves (0-3), the MFMA waves the younger (4-7)
//   T2  same wave: v_mfma_f32_16x16x4_f32 followed by 6 independent v_add_u32 -> cycles per MFMA (32 = co-execution)
//   T3  same wave: 4 MFMAs then one global_load_dwordx4
//   T4  bf16 matrix pipe: dense v_mfma_f32_16x16x32_bf16 on waves 0-3 | v_add_u32 on waves 4-7
//   T5  dense v_mfma_f32_32x32x2_f32 | v_add_u32
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue3 coissue3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// KIND: 0 f32 16x16x4, 1 bf16 16x16x32, 2 f32 32x32x2, 3 f32 16x16x4 + 6 v_add in the same wave,
//       4 f32 16x16x4 x4 + 1 global_load in the same wave
template <int KIND>
__device__ __forceinline__ float mfma_loop(int iters, int lane, const f32x4 *p) {
  float res = 0;
  if (KIND == 1) {
    f32x4 acc[4] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(1e-2f * lane), b[i] = (__bf16)(1e-2f * i);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 64; ++j)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n s_nop 15");
    res = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
  } else if (KIND == 2) {
    f32x16 acc[2] = {};
    const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 32; ++j)       // 32 x (64 cycles) = the same pipe time as 64 x 16x16x4
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
    res = acc[0][0] + acc[1][0];
  } else {
    f32x4 acc[4] = {};
    const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
    unsigned r[6];
    for (int i = 0; i < 6; ++i) r[i] = lane + i;
    f32x4 v[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
        if (KIND == 3) {
#pragma unroll
          for (int k = 0; k < 6; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(lane));
        }
        if (KIND == 4 && (j & 3) == 3)
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v[(j >> 2) & 3]) : "v"(p), "i"(((j >> 2) & 3) * 1024));
      }
      if (KIND == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15");
    res = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + (float)(r[0] + r[1] + r[2] + r[3] + r[4] + r[5]) + v[0][0] + v[1][0] + v[2][0] + v[3][0];
  }
  return res;
}

template <int KIND, bool SWAP>
__global__ __launch_bounds__(512) void k_co(float *out, long long *ticks, const f32x4 *gsrc, int iters_m, int iters_x, int mask) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int role = SWAP ? 1 - (wave >> 2) : (wave >> 2);
  float res = 0;
  __syncthreads();
  long long t0 = wall_clock64(), t1 = t0;
  const f32x4 *p = gsrc + (size_t)blockIdx.x * 4096 + (wave & 3) * 1024 + lane;
  if (role == 0) {
    if (mask & 1) { res = mfma_loop<KIND>(iters_m, lane, p); t1 = wall_clock64(); }
  } else if (mask & 2) {
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = lane + i;
    for (int it = 0; it < iters_x; ++it)
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[j & 7]) : "v"(lane));
    for (int i = 0; i < 8; ++i) res += (float)r[i];
    t1 = wall_clock64();
  }
  if (lane == 0) ticks[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + tid] = res;
}

template <int KIND, bool SWAP>
static void run(const char *name, float *out, long long *ticks, const f32x4 *gsrc, int cus, int im, int ix, int n_mfma) {
  std::vector<long long> h((size_t)cus * 8);
  double r[4][2] = {};
  for (int mask = 1; mask <= 3; ++mask) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k_co<KIND, SWAP>), dim3(cus), dim3(512), 0, 0, out, ticks, gsrc, im, ix, mask);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double s[2] = {0, 0};
    for (int b = 0; b < cus; ++b)
      for (int w = 0; w < 8; ++w) s[SWAP ? 1 - (w >> 2) : (w >> 2)] += (double)h[(size_t)b * 8 + w];
    r[mask][0] = s[0] / (cus * 4) * 10e-3;
    r[mask][1] = s[1] / (cus * 4) * 10e-3;
  }
  printf("%-58s MFMA alone %6.1f us (%5.1f cycles each @2.4GHz) | v_add alone %6.1f | together: MFMA %6.1f, v_add %6.1f\n", name,
         r[1][0], r[1][0] * 2400.0 / ((double)im * n_mfma), r[2][1], r[3][0], r[3][1]);
}

int main(int argc, char **argv) {
  const int im = argc > 1 ? atoi(argv[1]) : 200, ix = argc > 2 ? atoi(argv[2]) : 400;
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *out; long long *ticks; f32x4 *gsrc;
  CHECK(hipMalloc(&out, (size_t)cus * 512 * 4));
  CHECK(hipMalloc(&ticks, (size_t)cus * 8 * 8));
  CHECK(hipMalloc(&gsrc, (size_t)cus * 4096 * 16 + 65536));
  CHECK(hipMemset(gsrc, 0, (size_t)cus * 4096 * 16 + 65536));
  run<0, false>("f32 16x16x4 on waves 0-3 | v_add on waves 4-7", out, ticks, gsrc, cus, im, ix, 64);
  run<0, true>("T1 f32 16x16x4 on waves 4-7 | v_add on waves 0-3 (older)", out, ticks, gsrc, cus, im, ix, 64);
  run<3, false>("T2 same wave: MFMA + 6 v_add | v_add on waves 4-7", out, ticks, gsrc, cus, im, ix, 64);
  run<4, false>("T3 same wave: 4 MFMA + 1 global_load | v_add on waves 4-7", out, ticks, gsrc, cus, im, ix, 64);
  run<1, false>("T4 bf16 16x16x32 on waves 0-3 | v_add on waves 4-7", out, ticks, gsrc, cus, im, ix, 64);
  run<2, false>("T5 f32 32x32x2 on waves 0-3 | v_add on waves 4-7", out, ticks, gsrc, cus, im, ix, 32);
  return 0;
}
