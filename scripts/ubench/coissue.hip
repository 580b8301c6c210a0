// Which instruction classes can a SIMD of gfx950 issue for one wave while another wave on the SAME SIMD keeps the
// matrix pipe busy with v_mfma_f32_16x16x4_f32?  One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run a
// pure MFMA loop, waves 4-7 (their SIMD neighbours) run a loop of class X.  Each wave times itself with the
// 100 MHz wall clock; every class is run alone and together with the MFMA waves.
//   alone == together  -> X co-issues with MFMA for free
//   together == sum    -> X and MFMA serialise on the SIMD
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
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

enum { X_VALU32, X_VALU64, X_VMOV, X_DSREAD, X_DSWRITE, X_GLOAD, X_SALU, X_MFMA, X_COUNT };
static const char *kNames[] = {"v_add_u32 x64",        "64-bit mul-add x64 (address math)", "v_mov_b32 x64",
                               "ds_read_b128 x64",     "ds_write_b128 x64",                 "global_load_dwordx4 x64 (L2)",
                               "s_add/s_mul x64",      "v_mfma_f32_16x16x4_f32 x64"};

__device__ __forceinline__ void mfma64(f32x4 (&acc)[4], float a, float b) {
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
}

template <int X>
__global__ __launch_bounds__(512) void k_co(float *out, long long *ticks, const f32x4 *gsrc, int iters_m, int iters_x,
                                            int mask) {
  __shared__ f32x4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, role = wave >> 2;
  for (int i = tid; i < 2048; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float res = 0;
  long long t0 = wall_clock64(), t1 = t0;
  if (role == 0) {
    if (mask & 1) {
      f32x4 acc[4] = {};
      const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
      for (int it = 0; it < iters_m; ++it) mfma64(acc, a, b);
      res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
      t1 = wall_clock64();
    }
  } else if (mask & 2) {
    if (X == X_MFMA) {
      f32x4 acc[4] = {};
      const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
      for (int it = 0; it < iters_x; ++it) mfma64(acc, a, b);
      res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (X == X_VALU32) {
      unsigned r[8];
      for (int i = 0; i < 8; ++i) r[i] = lane + i;
      for (int it = 0; it < iters_x; ++it)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[j & 7]) : "v"(lane));
      for (int i = 0; i < 8; ++i) res += (float)r[i];
    } else if (X == X_VALU64) {
      unsigned long long r[8];
      const unsigned long long m = 0x9E3779B97F4A7C15ull + lane;
      for (int i = 0; i < 8; ++i) r[i] = lane + i;
      for (int it = 0; it < iters_x; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) r[j & 7] = r[j & 7] * (unsigned)(m >> (j & 7)) + m;
        asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
      }
      for (int i = 0; i < 8; ++i) res += (float)(r[i] >> 40);
    } else if (X == X_VMOV) {
      unsigned r[8];
      for (int i = 0; i < 8; ++i) r[i] = lane + i;
      for (int it = 0; it < iters_x; ++it)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(r[j & 7]) : "v"(r[(j + 1) & 7]));
      for (int i = 0; i < 8; ++i) res += (float)r[i];
    } else if (X == X_DSREAD) {
      f32x4 v[8];
      const unsigned addr = (unsigned)(size_t)(&lds[0]) + lane * 16;
      for (int it = 0; it < iters_x; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j & 7]) : "v"(addr), "i"((j & 15) * 1024));
          if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      for (int i = 0; i < 8; ++i) res += v[i][0];
    } else if (X == X_DSWRITE) {
      const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
      const unsigned addr = (unsigned)(size_t)(&lds[0]) + lane * 16 + (wave - 4) * 1024;
      for (int it = 0; it < iters_x; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"((j & 3) * 4096));
          if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    } else if (X == X_GLOAD) {
      f32x4 v[8];
      const f32x4 *p = gsrc + (size_t)blockIdx.x * 4096 + (wave - 4) * 1024 + lane;
      for (int it = 0; it < iters_x; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v[j & 7]) : "v"(p), "i"((j & 3) * 1024));
          if ((j & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      for (int i = 0; i < 8; ++i) res += v[i][0];
    } else if (X == X_SALU) {
      unsigned s0 = blockIdx.x + 1, s1 = __builtin_amdgcn_readfirstlane(wave + 3);
      for (int it = 0; it < iters_x; ++it)
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
          asm volatile("s_add_i32 %0, %0, %1" : "+s"(s1) : "s"(s0));
        }
      res = (float)(s0 + s1);
    }
    t1 = wall_clock64();
  }
  if (lane == 0) ticks[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + tid] = res;
}

template <int X>
static void run(float *out, long long *ticks, const f32x4 *gsrc, int cus, int iters_m, int iters_x) {
  std::vector<long long> h((size_t)cus * 8);
  double r[4][2] = {};
  for (int mask = 1; mask <= 3; ++mask) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_co<X>, dim3(cus), dim3(512), 0, 0, out, ticks, gsrc, iters_m, iters_x, mask);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double s[2] = {0, 0};
    for (int b = 0; b < cus; ++b)
      for (int w = 0; w < 8; ++w) s[w >> 2] += (double)h[(size_t)b * 8 + w];
    r[mask][0] = s[0] / (cus * 4) * 10e-3;   // 100 MHz ticks -> us
    r[mask][1] = s[1] / (cus * 4) * 10e-3;
  }
  printf("%-36s  MFMA alone %7.1f us | X alone %7.1f us | together: MFMA %7.1f us, X %7.1f us  (sum of alone %7.1f)\n",
         kNames[X], r[1][0], r[2][1], r[3][0], r[3][1], r[1][0] + r[2][1]);
}

int main(int argc, char **argv) {
  const int iters_m = argc > 1 ? atoi(argv[1]) : 200, iters_x = argc > 2 ? atoi(argv[2]) : 400;
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *out;
  long long *ticks;
  f32x4 *gsrc;
  CHECK(hipMalloc(&out, (size_t)cus * 512 * 4));
  CHECK(hipMalloc(&ticks, (size_t)cus * 8 * 8));
  CHECK(hipMalloc(&gsrc, (size_t)cus * 4096 * 16 + 65536));
  CHECK(hipMemset(gsrc, 0, (size_t)cus * 4096 * 16 + 65536));
  printf("%d CUs; MFMA loop %d x 64, X loop %d x 64 (per wave); waves 0-3 MFMA, waves 4-7 X\n", cus, iters_m, iters_x);
  run<X_MFMA>(out, ticks, gsrc, cus, iters_m, iters_m);
  run<X_VALU32>(out, ticks, gsrc, cus, iters_m, iters_x);
  run<X_VALU64>(out, ticks, gsrc, cus, iters_m, iters_x / 4);
  run<X_VMOV>(out, ticks, gsrc, cus, iters_m, iters_x);
  run<X_DSREAD>(out, ticks, gsrc, cus, iters_m, iters_x / 2);
  run<X_DSWRITE>(out, ticks, gsrc, cus, iters_m, iters_x / 2);
  run<X_GLOAD>(out, ticks, gsrc, cus, iters_m, iters_x / 8);
  run<X_SALU>(out, ticks, gsrc, cus, iters_m, iters_x);
  return 0;
}
