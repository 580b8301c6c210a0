// How fast can every CU stream the SAME weight tensor out of L2?  (round 4)
//
// The coarse MinkUNet levels (256 -> 256 channels on 5k / 21k voxels) run one workgroup per CU, and every workgroup
// walks the whole packed weight tensor (27 offsets x 64 KB per 128-column slab) once per tile: 453 MB of L2 -> CU
// traffic for 3.5 MB of weights.  The tile kernel sees ~30 B / clk / CU there (docs/HISTORY.md 10.4).  Is that the chip's
// limit for this pattern, or the kernel's pipeline?  Variants (each wave loads 8 x 16 B per lane and batch, exactly
// the kernel's weight slice):
//   depth D      batches of loads in flight per wave (1, 2, 4), consumed by an xor chain
//   waves        8 waves x 1 workgroup per CU, 4 waves x 2, 16 waves x 1
//   rotate       workgroup i starts at offset (i * 7) % 27 instead of 0: no lockstep on one 64 KB slice
//   dma          global_load_lds_dwordx4 into an LDS ring (no VGPR round trip), consumed by ds_read
// Prints us per launch and B / clk / CU at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip ; ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));        \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int kOffsets = 27;
constexpr int kSliceBytes = 64 * 1024;                       // one offset, one 128-column slab, 256 channels (bf16)
constexpr int kSlabs = 2;
constexpr int kWBytes = kOffsets * kSlabs * kSliceBytes;     // 3.5 MB

// WAVES waves per workgroup; every wave owns 1 / 8 of a slice (8 KB = 8 loads of 16 B per lane) when WAVES == 8,
// 2 / 8 when WAVES == 4 (16 loads), 1 / 16 when WAVES == 16 (4 loads): the workgroup always moves 64 KB per batch.
template <int WAVES, int DEPTH, bool ROTATE>
__global__ __launch_bounds__(WAVES * 64) void k_stream(const u32x4 *__restrict__ w, u32x4 *__restrict__ out, int rounds) {
  constexpr int LOADS = 64 * 1024 / 16 / (WAVES * 64);       // 16-byte loads per lane and batch
  const int tid = threadIdx.x;
  const int slab = blockIdx.x & 1;
  const int start = ROTATE ? (int)((blockIdx.x >> 1) * 7u % kOffsets) : 0;
  u32x4 buf[DEPTH][LOADS];
  u32x4 acc = {0, 0, 0, 0};
  const int total = kOffsets * rounds;
  auto issue = [&](int b, u32x4 (&dst)[LOADS]) {
    int k = (start + b) % kOffsets;
    const u32x4 *p = w + ((size_t)(k * kSlabs + slab) * (kSliceBytes / 16)) + tid;
#pragma unroll
    for (int j = 0; j < LOADS; ++j) dst[j] = p[j * WAVES * 64];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, buf[d]);
  for (int b = 0; b < total; b += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < LOADS; ++j) acc ^= buf[d][j];
      issue(min(b + d + DEPTH, total - 1), buf[d]);
    }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * blockDim.x + tid] = acc;
}

// LDS-DMA: every wave fills its share of a ring slot with global_load_lds_dwordx4 (16 B per lane, lane-linear LDS
// image at M0), SLOTS slots of 64 KB... the ring is [SLOTS][64 KB]; consumption = one ds_read_b128 per 16 B.
template <int WAVES, int SLOTS>
__global__ __launch_bounds__(WAVES * 64) void k_stream_dma(const u32x4 *__restrict__ w, u32x4 *__restrict__ out, int rounds) {
  constexpr int LOADS = 64 * 1024 / 16 / (WAVES * 64);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = blockIdx.x & 1;
  u32x4 acc = {0, 0, 0, 0};
  const int total = kOffsets * rounds;
  auto issue = [&](int b, int slot) {
    const int k = b % kOffsets;
    const u32x4 *p = w + ((size_t)(k * kSlabs + slab) * (kSliceBytes / 16));
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      const int piece = j * WAVES + wave;                       // 1 KB pieces of the slice
      const unsigned lds_base = (unsigned)(slot * kSliceBytes + piece * 1024);
      // (the LDS address of a DMA load is M0 + instruction offset + lane * 16; the base is wave-uniform)
      __builtin_amdgcn_global_load_lds(p + piece * 64 + lane,
                                       (__attribute__((address_space(3))) void *)(smem + lds_base), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < SLOTS - 1; ++s) issue(s, s);
  for (int b = 0; b < total; ++b) {
    const int slot = b % SLOTS;
    // wait for slot b (the oldest LOADS * (number of younger batches) may stay in flight)
    if (SLOTS == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);       // vmcnt(0)
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SLOTS - 2) * LOADS));
    __syncthreads();
    issue(min(b + SLOTS - 1, total - 1), (b + SLOTS - 1) % SLOTS);
    const u32x4 *s4 = reinterpret_cast<const u32x4 *>(smem + slot * kSliceBytes);
#pragma unroll
    for (int j = 0; j < LOADS; ++j) acc ^= s4[j * WAVES * 64 + tid];
    __syncthreads();
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * blockDim.x + tid] = acc;
}

template <typename F>
static void run(const char *name, F launch, int wgs, int rounds) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  const int reps = 20;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes_per_wg = (double)kOffsets * rounds * kSliceBytes;
  const double us = best * 1e3;
  const int cus = 256;
  const double wg_per_cu = (double)wgs / cus;
  printf("%-44s wgs %4d rounds %d  best %8.2f us  avg %8.2f us  %6.1f B/clk/CU  %6.2f TB/s aggregate\n", name, wgs,
         rounds, us, sum / reps * 1e3, bytes_per_wg * wg_per_cu / (us * 1e-6 * 2.4e9), bytes_per_wg * wgs / (us * 1e-6) / 1e12);
}

int main() {
  u32x4 *w, *out;
  CHECK(hipMalloc(&w, kWBytes));
  CHECK(hipMalloc(&out, 1 << 24));
  std::vector<uint32_t> h(kWBytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
  CHECK(hipMemcpy(w, h.data(), kWBytes, hipMemcpyHostToDevice));
  const int rounds = 2;   // the weight tensor is walked twice per launch (54 batches): less launch overhead in the figure
#define RUN(NAME, KERNEL, WGS, THREADS, LDS) \
  run(NAME, [&] { hipLaunchKernelGGL(KERNEL, dim3(WGS), dim3(THREADS), LDS, 0, w, out, rounds); }, WGS, rounds)
  RUN("8 waves, depth 1", (k_stream<8, 1, false>), 256, 512, 0);
  RUN("8 waves, depth 2 (the tile kernel)", (k_stream<8, 2, false>), 256, 512, 0);
  RUN("8 waves, depth 4", (k_stream<8, 4, false>), 256, 512, 0);
  RUN("8 waves, depth 2, rotated start", (k_stream<8, 2, true>), 256, 512, 0);
  RUN("8 waves, depth 4, rotated start", (k_stream<8, 4, true>), 256, 512, 0);
  RUN("4 waves x 2 workgroups per CU, depth 2", (k_stream<4, 2, false>), 512, 256, 0);
  RUN("4 waves x 2 workgroups per CU, depth 2, rot", (k_stream<4, 2, true>), 512, 256, 0);
  RUN("16 waves, depth 2", (k_stream<16, 2, false>), 256, 1024, 0);
  RUN("16 waves, depth 4", (k_stream<16, 4, false>), 256, 1024, 0);
  RUN("8 waves x 2 workgroups per CU, depth 2", (k_stream<8, 2, false>), 512, 512, 0);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream_dma<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream_dma<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  RUN("LDS-DMA ring, 8 waves, 2 slots", (k_stream_dma<8, 2>), 256, 512, 2 * kSliceBytes);
  RUN("LDS-DMA ring, 4 waves, 2 slots", (k_stream_dma<4, 2>), 256, 256, 2 * kSliceBytes);
  // half the CUs only (one workgroup on every other CU cannot be forced; 128 workgroups land on 128 CUs)
  RUN("8 waves, depth 2, 128 workgroups", (k_stream<8, 2, false>), 128, 512, 0);
  RUN("8 waves, depth 2, 64 workgroups", (k_stream<8, 2, false>), 64, 512, 0);
  return 0;
}
