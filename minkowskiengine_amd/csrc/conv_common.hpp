// Shared by the convolution translation units (conv.hip: fp32, conv_bf16.hip: bf16 features).
#pragma once
#include "common.hpp"

namespace me {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLdsBudget = 160 * 1024;
constexpr int kAccPad = 4;  // accumulator row stride NC + 4 floats: spreads the row-scattered adds over banks

static inline int device_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;  // MI355X
  }
  return cus;
}

// eight fp32 values -> the three bf16 planes of the exact split (conv_f32x3.hip: a = a1 + a2 + a3 with
// a1 = the upper 16 bits of a's encoding, a2 = the upper 16 bits of a - a1, a3 = a - a1 - a2; all exact)
__device__ __forceinline__ void split3(const f32x4 &lo, const f32x4 &hi, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
  float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t b1[8], b2[8], b3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    b1[e] = __float_as_uint(a[e]);
    const float r = a[e] - __uint_as_float(b1[e] & 0xffff0000u);
    b2[e] = __float_as_uint(r);
    b3[e] = __float_as_uint(r - __uint_as_float(b2[e] & 0xffff0000u));
  }
  // upper halves of two encodings -> one dword (element 2j in the lower half)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p1[j] = __builtin_amdgcn_perm(b1[2 * j + 1], b1[2 * j], 0x07060302u);
    p2[j] = __builtin_amdgcn_perm(b2[2 * j + 1], b2[2 * j], 0x07060302u);
    p3[j] = __builtin_amdgcn_perm(b3[2 * j + 1], b3[2 * j], 0x07060302u);
  }
}

// tile-plan geometry (conv.hip)
struct PlanShape {
  int nc, slabs, chunks;   // columns per workgroup, column slabs, source-channel chunks
  double group_cycles;     // matrix-pipe (or LDS) cycles of one 16-row group and chunk in one wave
  int stage_row_bytes;     // LDS bytes per staged row (+ its target index)
  int max_occ = 0;         // resident workgroups per CU the kernel's registers allow (0: three waves per SIMD)
};
int plan_tile_rows(const PlanShape &s, int64_t n_tgt, int64_t volume, int64_t n_pairs);

}  // namespace me
