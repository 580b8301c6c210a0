// Fourth co-issue probe (gfx950): do SALU and LDS instructions of a neighbour wave issue while fp32 MFMAs run?
// The neighbour's loop and its end timestamp contain no VALU / VMEM instruction (those are known to starve).
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue4 coissue4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// X: 0 SALU (s_mul_i32 / s_add_i32), 1 ds_read_b128, 2 ds_write_b128, 3 s_load_dwordx4 (scalar cache)
template <int X>
__global__ __launch_bounds__(512) void k_co(float *out, long long *ticks, const unsigned *ssrc, int iters_m, int iters_x, int mask) {
  __shared__ f32x4 lds[4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, role = wave >> 2;
  for (int i = tid; i < 4096; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  const unsigned addr = (unsigned)(size_t)(&lds[0]) + lane * 16;
  const f32x4 wv = {1.f * lane, 2.f, 3.f, 4.f};
  __syncthreads();
  float res = 0;
  long long t0 = wall_clock64(), t1 = t0;
  if (role == 0) {
    if (mask & 1) {
      f32x4 acc[4] = {};
      const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
      asm volatile("s_nop 15\n s_nop 15");
      res = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
      t1 = wall_clock64();
    }
  } else if (mask & 2) {
    unsigned s0 = blockIdx.x + 1, s1 = blockIdx.x * 7 + 3;
    f32x4 v[8];
    for (int it = 0; it < iters_x; ++it) {
      if (X == 0) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
          asm volatile("s_add_i32 %0, %0, %1" : "+s"(s1) : "s"(s0));
        }
      } else if (X == 1) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j & 7]) : "v"(addr), "i"((j & 15) * 1024));
          if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      } else if (X == 2) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(wv), "i"(32768 + (j & 7) * 1024));
          if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      } else {
        const unsigned *p = ssrc + (blockIdx.x & 7) * 64;
        unsigned a4[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(*reinterpret_cast<__uint128_t *>(a4)) : "s"(p), "i"((j & 7) * 16));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(a4[0]));
        }
      }
    }
    t1 = wall_clock64();                     // s_memrealtime + s_waitcnt: no VALU before this point
    if (X == 1) for (int i = 0; i < 8; ++i) res += v[i][0];
    res += (float)(s0 + s1);
  }
  if (lane == 0) ticks[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + tid] = res;
}

template <int X>
static void run(const char *name, float *out, long long *ticks, const unsigned *ssrc, int cus, int im, int ix) {
  std::vector<long long> h((size_t)cus * 8);
  double r[4][2] = {};
  for (int mask = 1; mask <= 3; ++mask) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k_co<X>), dim3(cus), dim3(512), 0, 0, out, ticks, ssrc, im, ix, mask);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double s[2] = {0, 0};
    for (int b = 0; b < cus; ++b)
      for (int w = 0; w < 8; ++w) s[w >> 2] += (double)h[(size_t)b * 8 + w];
    r[mask][0] = s[0] / (cus * 4) * 10e-3;
    r[mask][1] = s[1] / (cus * 4) * 10e-3;
  }
  fflush(stdout);
  printf("%-28s MFMA alone %6.1f us | X alone %6.1f | together: MFMA %6.1f, X %6.1f us\n", name, r[1][0], r[2][1], r[3][0], r[3][1]);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int im = argc > 1 ? atoi(argv[1]) : 200, ix = argc > 2 ? atoi(argv[2]) : 400;
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *out; long long *ticks; unsigned *ssrc;
  CHECK(hipMalloc(&out, (size_t)cus * 512 * 4));
  CHECK(hipMalloc(&ticks, (size_t)cus * 8 * 8));
  CHECK(hipMalloc(&ssrc, 65536));
  CHECK(hipMemset(ssrc, 0, 65536));
  run<0>("s_mul_i32 + s_add_i32 x32", out, ticks, ssrc, cus, im, ix * 4);
  run<1>("ds_read_b128 x64", out, ticks, ssrc, cus, im, ix / 2);
  run<2>("ds_write_b128 x64", out, ticks, ssrc, cus, im, ix / 4);
  return 0;
}
