// Shared by the convolution translation units (conv.hip: fp32, conv_bf16.hip: bf16 features).
#pragma once
#include "common.hpp"

namespace me {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

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

// tile-plan geometry (conv.hip)
struct PlanShape {
  int nc, slabs, chunks;   // columns per workgroup, column slabs, source-channel chunks
  double group_cycles;     // matrix-pipe (or LDS) cycles of one 16-row group and chunk in one wave
  int stage_row_bytes;     // LDS bytes per staged row (+ its target index)
  int max_occ = 0;         // resident workgroups per CU the kernel's registers allow (0: three waves per SIMD)
};
int plan_tile_rows(const PlanShape &s, int64_t n_tgt, int64_t volume, int64_t n_pairs);

}  // namespace me
