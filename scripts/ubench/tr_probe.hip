// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds element index as u16; lane l passes the byte address
// addr[l]; prints what every lane receives.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k(const int *addr, s16x4 *out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  lds_v4 *p = (lds_v4 *)((__attribute__((address_space(3))) char *)lds + addr[l]);
  out[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}

int main() {
  int h_addr[64];
  // case A: row-major image, row stride 64 elements (128 B): lane i of 16-group g points at row (4g + i/4), col 4*(i%4)
  for (int mode = 0; mode < 2; ++mode) {
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      if (mode == 0) h_addr[l] = ((4 * g + i / 4) * 64 + 4 * (i % 4)) * 2;
      else h_addr[l] = ((4 * g + (i % 4)) * 64 + 4 * (i / 4)) * 2;  // alternative lane -> (row, col) assignment
    }
    int *d_addr; s16x4 *d_out; s16x4 h_out[64];
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("mode %d (elements shown as row:col of a 64-wide image)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr r%d:c%-2d ->", l, h_addr[l] / 2 / 64, (h_addr[l] / 2) % 64);
      for (int j = 0; j < 4; ++j) printf(" %d:%-2d", (uint16_t)h_out[l][j] / 64, (uint16_t)h_out[l][j] % 64);
      printf("\n");
    }
  }
  return 0;
}
