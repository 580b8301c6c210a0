// Follow-up of coissue.hip: a wave that issues v_mfma_f32_16x16x4_f32 back to back starves the VALU / VMEM
// instructions of the other waves of its SIMD.  Which remedy lets the neighbour issue without slowing the MFMA wave?
//   MV 0  dense MFMAs (baseline)            MV 1  s_nop 7 after every MFMA        MV 2  3 x s_nop 7 after every MFMA
//   MV 3  s_nop 15 + s_nop 11 (28 cycles)   MV 4  one dependent chain             MV 5  s_sleep 0? -> not used
//   PV 0  default priorities                PV 1  neighbour raises its priority (s_setprio 3)
//   PV 2  MFMA wave lowers nothing, neighbour prio 3 and MFMA wave prio 0 explicitly
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue2 coissue2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <int MV>
__device__ __forceinline__ void mfma64(f32x4 (&acc)[4], float a, float b) {
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 &d = acc[MV == 4 ? 0 : c];
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
      if (MV == 1) asm volatile("s_nop 7");
      if (MV == 2) asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
      if (MV == 3) asm volatile("s_nop 15\n s_nop 11");
    }
}

// X: 0 v_add_u32, 1 global_load_dwordx4, 2 mixed (the conv kernel's issue block: 12 loads + ~60 VALU per 64 MFMAs)
template <int MV, int X, int PV>
__global__ __launch_bounds__(512) void k_co(float *out, long long *ticks, const f32x4 *gsrc, int iters_m, int iters_x,
                                            int mask) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, role = wave >> 2;
  float res = 0;
  __syncthreads();
  long long t0 = wall_clock64(), t1 = t0;
  if (role == 0) {
    if (mask & 1) {
      if (PV == 2) __builtin_amdgcn_s_setprio(0);
      f32x4 acc[4] = {};
      const float a = 1e-3f * lane, b = 1e-3f * (63 - lane);
      for (int it = 0; it < iters_m; ++it) mfma64<MV>(acc, a, b);
      asm volatile("s_nop 15\n s_nop 15");
      res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
      t1 = wall_clock64();
    }
  } else if (mask & 2) {
    if (PV >= 1) __builtin_amdgcn_s_setprio(3);
    if (X == 0) {
      unsigned r[8];
      for (int i = 0; i < 8; ++i) r[i] = lane + i;
      for (int it = 0; it < iters_x; ++it)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[j & 7]) : "v"(lane));
      for (int i = 0; i < 8; ++i) res += (float)r[i];
    } else {
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
    }
    t1 = wall_clock64();
  }
  if (lane == 0) ticks[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + tid] = res;
}

template <int MV, int X, int PV>
static void run(const char *name, float *out, long long *ticks, const f32x4 *gsrc, int cus, int iters_m, int iters_x) {
  std::vector<long long> h((size_t)cus * 8);
  double r[4][2] = {};
  for (int mask = 1; mask <= 3; ++mask) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k_co<MV, X, PV>), dim3(cus), dim3(512), 0, 0, out, ticks, gsrc, iters_m, iters_x, mask);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double s[2] = {0, 0};
    for (int b = 0; b < cus; ++b)
      for (int w = 0; w < 8; ++w) s[w >> 2] += (double)h[(size_t)b * 8 + w];
    r[mask][0] = s[0] / (cus * 4) * 10e-3;
    r[mask][1] = s[1] / (cus * 4) * 10e-3;
  }
  printf("%-52s MFMA alone %6.1f | X alone %6.1f | together: MFMA %6.1f, X %6.1f us\n", name, r[1][0], r[2][1], r[3][0],
         r[3][1]);
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
  printf("times in us; MFMA loop %d x 64 per wave; X loop %d x 64 v_add_u32 or %d x 64 global_load_dwordx4\n", im, ix, ix / 8);
#define RUN(MV, X, PV, name) run<MV, X, PV>(name, out, ticks, gsrc, cus, im, X == 0 ? ix : ix / 8)
  RUN(0, 0, 0, "dense MFMA | v_add");
  RUN(0, 0, 1, "dense MFMA | v_add, neighbour s_setprio 3");
  RUN(0, 0, 2, "dense MFMA prio 0 | v_add prio 3");
  RUN(1, 0, 0, "MFMA + s_nop 7 | v_add");
  RUN(2, 0, 0, "MFMA + 3 x s_nop 7 | v_add");
  RUN(3, 0, 0, "MFMA + s_nop 15 + s_nop 11 | v_add");
  RUN(3, 0, 1, "MFMA + s_nop 15 + s_nop 11 | v_add prio 3");
  RUN(4, 0, 0, "one dependent MFMA chain | v_add");
  RUN(4, 0, 1, "one dependent MFMA chain | v_add prio 3");
  RUN(0, 1, 0, "dense MFMA | global_load");
  RUN(0, 1, 1, "dense MFMA | global_load prio 3");
  RUN(2, 1, 0, "MFMA + 3 x s_nop 7 | global_load");
  RUN(4, 1, 0, "one dependent MFMA chain | global_load");
  return 0;
}
