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
// a1 = the upper 16 bits of a's encoding, a2 = the upper 16 bits of a - a1, a3 = a - a1 - a2; all exact).
//
// Non-finite inputs.  a = +-inf or NaN makes a - a1 a NaN, and the six products would turn every infinity into NaN
// where fp32 arithmetic (the reference's sgemm, src/math_functions_cpu.cpp:29-46) keeps it.  The kept products are
// (row plane, weight plane) = (1,1) (1,2) (2,1) (1,3) (2,2) (3,1): a value in plane 3 meets ONLY plane 1 of the other
// operand, which is zero only for |x| < 2^-126.  So a non-finite ROW value (WEIGHT_SIDE = false: gathered features,
// upstream gradients, the x rows of the weight gradient) is stored as planes (0, 0, a): its one product a * w1 is
// +-inf with the sign of a * w (NaN for NaN, or for w = 0 — exactly fp32 semantics).  A non-finite value on the
// WEIGHT side (packed weights, the dy rows of the weight gradient) becomes NaN in all three planes: every output it
// reaches is NaN (fp32 would keep the sign of an infinite weight — the one divergence left; inf * inf must not fall
// into the dropped (3,3) product and vanish).
//
// Cost.  The hot kernels split with split3_raw (no handling) and OR the third plane's packed words into a flag
// (split3_flag: two v_or3 per eight values): the second remainder of a non-finite input is a NaN, so its bf16 exponent
// field is all ones and survives the OR.  Once per staged batch split3_suspect tests the flag; if it fires the batch is
// split again with split3_fix (exact, per element) and stored over the first attempt — rare, out of line.  A FALSE
// alarm needs an OR-ed exponent field of all ones, i.e. some remainder |a3| >= 2, i.e. an input of magnitude >= 2^16:
// such batches take the slow path too, with the same results.
template <bool WEIGHT_SIDE = false>
__device__ __forceinline__ void split3_fix(const f32x4 &lo, const f32x4 &hi, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
  float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t b1[8], b2[8], b3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t bits = __float_as_uint(a[e]);
    b1[e] = bits;
    const float r = a[e] - __uint_as_float(b1[e] & 0xffff0000u);
    b2[e] = __float_as_uint(r);
    b3[e] = __float_as_uint(r - __uint_as_float(b2[e] & 0xffff0000u));
    const bool nonfinite = (bits & 0x7f800000u) == 0x7f800000u;
    // the bf16 that stands for it: +-inf, or a quiet NaN (a payload in the low 16 bits must not truncate to inf)
    const uint32_t t = (bits & 0xffff0000u) | ((bits & 0x007fffffu) ? 0x00400000u : 0u);
    if (WEIGHT_SIDE) {
      b1[e] = nonfinite ? 0x7fc00000u : b1[e];
      b2[e] = nonfinite ? 0x7fc00000u : b2[e];
      b3[e] = nonfinite ? 0x7fc00000u : b3[e];
    } else {
      b1[e] = nonfinite ? 0u : b1[e];
      b2[e] = nonfinite ? 0u : b2[e];
      b3[e] = nonfinite ? t : b3[e];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p1[j] = __builtin_amdgcn_perm(b1[2 * j + 1], b1[2 * j], 0x07060302u);
    p2[j] = __builtin_amdgcn_perm(b2[2 * j + 1], b2[2 * j], 0x07060302u);
    p3[j] = __builtin_amdgcn_perm(b3[2 * j + 1], b3[2 * j], 0x07060302u);
  }
}

__device__ __forceinline__ void split3_raw(const f32x4 &lo, const f32x4 &hi, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
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
__device__ __forceinline__ uint32_t split3_flag(uint32_t flag, const u32x4 &p3) {
#ifdef ME_SPLIT_NO_NONFINITE
  return flag;
#else
  return (flag | p3[0] | p3[1]) | (p3[2] | p3[3]);
#endif
}
// lane predicate: some value whose third plane went into `flag` may have been non-finite
__device__ __forceinline__ bool split3_suspect(uint32_t flag) {
  const uint32_t t = flag | (flag << 16);
  return (t & 0x7f800000u) == 0x7f800000u;
}

// one piece with the handling inline (cold callers: the weight pack kernels)
template <bool WEIGHT_SIDE = false>
__device__ __forceinline__ void split3(const f32x4 &lo, const f32x4 &hi, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
  split3_raw(lo, hi, p1, p2, p3);
  if (__builtin_expect(__any(split3_suspect(split3_flag(0u, p3))), 0)) split3_fix<WEIGHT_SIDE>(lo, hi, p1, p2, p3);
}

// tile-plan geometry (conv.hip)
struct PlanShape {
  int nc, slabs, chunks;   // columns per workgroup, column slabs, source-channel chunks
  double group_cycles;     // matrix-pipe (or LDS) cycles of one 16-row group and chunk in one wave
  int stage_row_bytes;     // LDS bytes per staged row (+ its target index)
  int max_occ = 0;         // resident workgroups per CU the kernel's registers allow (0: three waves per SIMD)
  int wave_slots = 0;      // waves per CU the kernel's register budget admits (0: 12 = three per SIMD)
};
int plan_tile_rows(const PlanShape &s, int64_t n_tgt, int64_t volume, int64_t n_pairs);

// fp32 rows on the bf16 pipe with multi-offset batches (conv_f32x3_fused.hip), dispatched by me_conv_target_f32_fused
int launch_conv_f32x3_fused(int nc, int kc, const float *src, int64_t n_src, int c_src, const float *wp, int c_dst, int slabs,
                            const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                            const int32_t *tile_bptr, const int32_t *order, float *dst, int64_t n_tgt, int tile_rows,
                            hipStream_t stream);

// wave-specialised bf16 tile kernel (conv_bf16_ws.hip), dispatched by conv_bf16.hip
bool conv_bf16_ws_shape(int nc, int kc);
int conv_bf16_ws_lds_bytes(int nc, int kc, int tile_rows);
int launch_conv_bf16_ws(int nc, int kc, const void *src, int c_src, const void *wp, int c_dst, int slabs,
                        const int32_t *plan_src, const int32_t *plan_dst, const int32_t *batch_desc,
                        const int32_t *tile_bptr, const int32_t *order, void *dst, int64_t n_tgt, int tile_rows,
                        hipStream_t stream, float *stat_mean, float *stat_m2, bool fuse);

}  // namespace me
