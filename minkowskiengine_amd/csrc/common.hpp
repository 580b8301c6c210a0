// Shared device/host helpers for the gfx950 hot path.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/me_amd.h"
#include "me_amd_debug.h"

namespace me {

constexpr int kWave = 64;
constexpr uint64_t kEmptySlot = ~0ull;

// ---- error plumbing ---------------------------------------------------------------------------
extern thread_local char g_last_error[512];
inline int fail(const char *file, int line, const char *msg) {
  snprintf(g_last_error, sizeof(g_last_error), "%s:%d, %s", file, line, msg);
  return 1;
}
#define ME_FAIL(msg) return ::me::fail(__FILE__, __LINE__, msg)
#define ME_CHECK(cond, msg)                                                                      \
  do {                                                                                           \
    if (!(cond)) return ::me::fail(__FILE__, __LINE__, "assertion (" #cond ") failed. " msg);    \
  } while (0)
#define ME_HIP(expr)                                                                             \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) return ::me::fail(__FILE__, __LINE__, hipGetErrorString(e__));        \
  } while (0)
#define ME_LAUNCH_CHECK() ME_HIP(hipGetLastError())

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- wave primitives --------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of a 64-bit ballot mask below this lane
__device__ __forceinline__ uint32_t mask_prefix(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// ---- coordinate hashing -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// MurmurHash3_x86_32-style mix over NCOL int32 words (seed 0x9747b28c).  Any good mix works: row
// order never depends on it (compaction is by input row), only probe lengths do.
template <int NCOL>
__device__ __forceinline__ uint32_t hash_coords(const int32_t (&c)[NCOL]) {
  uint32_t h = 0x9747b28cu;
#pragma unroll
  for (int i = 0; i < NCOL; ++i) {
    uint32_t k = (uint32_t)c[i];
    k *= 0xcc9e2d51u;
    k = rotl32(k, 15);
    k *= 0x1b873593u;
    h ^= k;
    h = rotl32(h, 13);
    h = h * 5u + 0xe6546b64u;
  }
  h ^= (uint32_t)(NCOL * 4);
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

template <int NCOL>
__device__ __forceinline__ void load_coords(const int32_t *__restrict__ base, int64_t row,
                                            int32_t (&c)[NCOL]) {
  if constexpr (NCOL == 4) {
    const int4 v = *reinterpret_cast<const int4 *>(base + row * 4);
    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
  } else if constexpr (NCOL == 2) {
    const int2 v = *reinterpret_cast<const int2 *>(base + row * 2);
    c[0] = v.x; c[1] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < NCOL; ++i) c[i] = base[row * NCOL + i];
  }
}

template <int NCOL>
__device__ __forceinline__ void store_coords(int32_t *__restrict__ base, int64_t row,
                                             const int32_t (&c)[NCOL]) {
  if constexpr (NCOL == 4) {
    *reinterpret_cast<int4 *>(base + row * 4) = make_int4(c[0], c[1], c[2], c[3]);
  } else if constexpr (NCOL == 2) {
    *reinterpret_cast<int2 *>(base + row * 2) = make_int2(c[0], c[1]);
  } else {
#pragma unroll
    for (int i = 0; i < NCOL; ++i) base[row * NCOL + i] = c[i];
  }
}

template <int NCOL>
__device__ __forceinline__ bool coords_equal(const int32_t (&a)[NCOL], const int32_t (&b)[NCOL]) {
  bool eq = true;
#pragma unroll
  for (int i = 0; i < NCOL; ++i) eq = eq && (a[i] == b[i]);
  return eq;
}

// Look a key up in an open-addressing table of {tag:32 | row:32} slots (linear probing).
// Returns the row or -1.  The probe count is bounded by the capacity, so a corrupt table cannot
// hang the GPU.
template <int NCOL>
__device__ __forceinline__ int32_t table_find(const uint64_t *__restrict__ table, uint32_t mask,
                                              const int32_t *__restrict__ map_coords,
                                              const int32_t (&key)[NCOL]) {
  const uint32_t h = hash_coords<NCOL>(key);
  uint32_t pos = h & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint64_t cur = table[pos];
    if (cur == kEmptySlot) return -1;
    if ((uint32_t)(cur >> 32) == h) {
      const uint32_t r = (uint32_t)cur;
      int32_t other[NCOL];
      load_coords<NCOL>(map_coords, r, other);
      if (coords_equal<NCOL>(other, key)) return (int32_t)r;
    }
    pos = (pos + 1) & mask;
  }
  return -1;
}

// ---- device-wide exclusive scan of uint32 (three small launches; deterministic) -----------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanBlock = kScanThreads * kScanItems;  // 1024 items per block
int64_t scan_workspace_bytes(int64_t n);
// out[i] = sum_{j<i} in[j];  *total_dev (uint32, device) = sum of all.  in == out allowed.
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total_dev, void *ws,
                       int64_t ws_bytes, hipStream_t stream);

// dispatch a functor templated on NCOL
#define ME_DISPATCH_NCOL(ncol, ...)                                                              \
  switch (ncol) {                                                                                \
    case 2: { constexpr int NCOL = 2; __VA_ARGS__; break; }                                      \
    case 3: { constexpr int NCOL = 3; __VA_ARGS__; break; }                                      \
    case 4: { constexpr int NCOL = 4; __VA_ARGS__; break; }                                      \
    case 5: { constexpr int NCOL = 5; __VA_ARGS__; break; }                                      \
    case 6: { constexpr int NCOL = 6; __VA_ARGS__; break; }                                      \
    case 7: { constexpr int NCOL = 7; __VA_ARGS__; break; }                                      \
    case 8: { constexpr int NCOL = 8; __VA_ARGS__; break; }                                      \
    default: ME_FAIL("coordinate size (D+1) must be in [2, 8]");                                 \
  }

}  // namespace me
