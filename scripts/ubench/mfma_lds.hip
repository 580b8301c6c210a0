// Micro-benchmark of the convolution kernel's inner loop on gfx950: v_mfma_f32_16x16x4_f32 fed from LDS with
// an LDS read-add-write per 16 MFMAs, at 1-3 waves per SIMD, with and without barriers.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds scripts/ubench/mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bits: 1 = operands from LDS (8 x ds_read_b128 per unit), 2 = RMW into LDS accumulator at pseudo-random
// rows, 4 = two barriers per 4 units (like one batch), 8 = single chain (dependent MFMAs)
template <int MODE>
__global__ __launch_bounds__(256) void k_unit(const float *__restrict__ in, float *__restrict__ out, int iters,
                                              int acc_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_a = smem;                 // 64 rows x 68 floats
  float *s_acc = smem + 64 * 68;     // acc_rows x 68 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  for (int x = tid; x < 64 * 68 + acc_rows * 68; x += 256) smem[x] = in[x % 4096];
  __syncthreads();
  float w[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) w[s] = in[lane + s];
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  uint32_t rng = tid * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 4) __syncthreads();
    if (MODE & 4) __syncthreads();
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {   // 2 units of 2 groups = 64 MFMAs per wave per iteration
      const float *a0p = &s_a[((u * 2) * 16 + i16) * 68 + q * 16];
      const float *a1p = &s_a[((u * 2 + 1) * 16 + i16) * 68 + q * 16];
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 a0, a1;
        if (MODE & 1) {
          a0 = *reinterpret_cast<const f32x4 *>(a0p + s4 * 4);
          a1 = *reinterpret_cast<const f32x4 *>(a1p + s4 * 4);
        } else {
          a0 = f32x4{w[s4], w[s4 + 1], w[s4 + 2], w[s4 + 3]};
          a1 = f32x4{w[s4 + 4], w[s4 + 5], w[s4 + 6], w[s4 + 7]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s4 * 4 + j], a0[j], c0, 0, 0, 0);
          if (MODE & 8) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s4 * 4 + j], a1[j], c0, 0, 0, 0);
          else c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s4 * 4 + j], a1[j], c1, 0, 0, 0);
        }
      }
      if (MODE & 2) {
        rng = rng * 1664525u + 1013904223u;
        const int d0 = (rng >> 8) % acc_rows, d1 = (rng >> 20) % acc_rows;
        float *p0 = &s_acc[d0 * 68 + wave * 16 + q * 4], *p1 = &s_acc[d1 * 68 + wave * 16 + q * 4];
        const f32x4 o0 = *reinterpret_cast<f32x4 *>(p0), o1 = *reinterpret_cast<f32x4 *>(p1);
        *reinterpret_cast<f32x4 *>(p0) = o0 + c0;
        *reinterpret_cast<f32x4 *>(p1) = o1 + c1;
      } else {
        total += c0 + c1;
      }
    }
  }
  if (MODE & 2) total += *reinterpret_cast<f32x4 *>(&s_acc[(tid % acc_rows) * 68]);
  out[(size_t)blockIdx.x * 256 + tid] = total.x + total.y + total.z + total.w;
}

template <int MODE>
void run(const char *name, const float *in, float *out, int wgs_per_cu, int acc_rows) {
  const int iters = 2000;
  const int lds = (64 * 68 + acc_rows * 68) * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_unit<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k_unit<MODE>, dim3(grid), dim3(256), lds, 0, in, out, 10, acc_rows);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(k_unit<MODE>, dim3(grid), dim3(256), lds, 0, in, out, iters, acc_rows);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  const double mfma = (double)grid * 4 * iters * 64;  // MFMAs
  const double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
  printf("%-34s wgs/CU %d  lds %6d B  %7.3f ms  %6.1f TF  %5.1f cycles/MFMA/SIMD @2.4GHz\n", name, wgs_per_cu, lds, ms,
         tf, ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * 64));
}

int main() {
  float *in, *out;
  hipMalloc(&in, 1 << 20);
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMemset(in, 0, 1 << 20);
  for (int occ = 1; occ <= 3; ++occ) {
    run<0>("registers only, 2 chains", in, out, occ, 16);
    run<8>("registers only, 1 chain", in, out, occ, 16);
    run<1>("operands from LDS", in, out, occ, 16);
    run<3>("operands from LDS + RMW", in, out, occ, 131);
    run<7>("LDS + RMW + 2 barriers / 4 units", in, out, occ, 131);
    run<5>("LDS + 2 barriers / 4 units", in, out, occ, 16);
  }
  return 0;
}
