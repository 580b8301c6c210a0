// Coordinate hash map, kernel-map construction and tile plans for gfx950 (MI355X).
//
// Replaces the reference's CoordinateMapGPU (src/coordinate_map_gpu.cu) and gpu_kernel_map
// (src/kernel_map.cuh).  Design differences (MI355X-first, not a translation):
//   * 8-byte slots {hash tag : 32 | row : 32} in one flat open-addressing table, keys are NOT
//     pointers: a miss costs one coalescable 8-byte read, a hit one more 16-byte read of the row's
//     coordinates (the reference dereferences a pointer key on every compare,
//     src/coordinate.hpp:223-241).
//   * duplicates resolve with a 64-bit atomicMin on the slot, so the FIRST input row wins
//     deterministically (the reference GPU path is racy; its CPU path is first-wins).
//   * the kernel map is produced as a dense neighbour table nbr[k][out_row] (coalesced writes,
//     one thread per (out row, offset)) and compacted with wavefront ballot + mbcnt prefix sums
//     into per-offset pair lists sorted by output row; no sort_by_key, no remove_if.
#include "common.hpp"

namespace me {

thread_local char g_last_error[512] = "";

// =================================================================================================
// scan
// =================================================================================================
// T = uint32_t, or uint64_t for two 32-bit counters scanned at once (high and low word: me_plan_build)
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan_t(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *s_wave /*[kScanThreads / 64]*/, T &block_total) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const T incl = wave_inclusive_scan_t(v);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  T wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const T t = s_wave[w];
    if (w < wave) wave_off += t;
    total += t;
  }
  __syncthreads();  // s_wave may be reused by the caller
  block_total = total;
  return wave_off + incl - v;
}

template <typename T>
__global__ __launch_bounds__(kScanThreads) void k_scan_block_sums(const T *__restrict__ in, int64_t n,
                                                                 T *__restrict__ bsums) {
  __shared__ T s_wave[8];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanItems;
  T s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) s += in[base + j];
  T total;
  (void)block_exclusive_scan(s, s_wave, total);
  if (threadIdx.x == 0) bsums[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place, grand total to *total_dev
template <typename T>
__global__ __launch_bounds__(kScanThreads) void k_scan_of_sums(T *__restrict__ bsums, int64_t nb,
                                                              T *__restrict__ total_dev) {
  __shared__ T s_wave[8];
  T carry = 0;
  for (int64_t base = 0; base < nb; base += kScanThreads) {
    const int64_t i = base + threadIdx.x;
    const T v = (i < nb) ? bsums[i] : (T)0;
    T total;
    const T ex = block_exclusive_scan(v, s_wave, total);
    if (i < nb) bsums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && total_dev) *total_dev = carry;
}

template <typename T>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const T *__restrict__ in, T *__restrict__ out,
                                                            int64_t n, const T *__restrict__ bsums) {
  __shared__ T s_wave[8];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanItems;
  T v[kScanItems];
  T s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = (base + j < n) ? in[base + j] : (T)0;
    s += v[j];
  }
  T total;
  T ex = block_exclusive_scan(s, s_wave, total) + bsums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// short inputs: ONE workgroup of 1024 threads walks the array 8192 items at a time with a running carry (one launch
// instead of three; the map builders of a network issue ~100 scans per step, most of them a few thousand to a few
// ten thousand items long)
constexpr int kScanSingleThreads = 1024, kScanSingleItems = 8;
template <typename T>
__global__ __launch_bounds__(kScanSingleThreads) void k_scan_single(const T *__restrict__ in, T *__restrict__ out,
                                                                   int64_t n, T *__restrict__ total_dev) {
  constexpr int kWaves = kScanSingleThreads / 64;
  __shared__ T s_wave[kWaves];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  T carry = 0;
  for (int64_t blk = 0; blk < n; blk += kScanSingleThreads * kScanSingleItems) {
    const int64_t base = blk + (int64_t)threadIdx.x * kScanSingleItems;
    T v[kScanSingleItems];
    T s = 0;
#pragma unroll
    for (int j = 0; j < kScanSingleItems; ++j) {
      v[j] = (base + j < n) ? in[base + j] : (T)0;
      s += v[j];
    }
    const T incl = wave_inclusive_scan_t(s);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    T wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const T t = s_wave[w];
      if (w < wave) wave_off += t;
      total += t;
    }
    __syncthreads();
    T ex = carry + wave_off + incl - s;
#pragma unroll
    for (int j = 0; j < kScanSingleItems; ++j) {
      if (base + j < n) out[base + j] = ex;
      ex += v[j];
    }
    carry += total;
  }
  if (threadIdx.x == 0 && total_dev) *total_dev = carry;
}

constexpr int64_t kScanSingleMax = 4 * kScanSingleThreads * kScanSingleItems;  // 32768 items

int64_t scan_workspace_bytes(int64_t n) { return align_up((ceil_div(n, kScanBlock) + 1) * 8, 256); }

template <typename T>
static int exclusive_scan(const T *in, T *out, int64_t n, T *total_dev, void *ws, int64_t ws_bytes,
                          hipStream_t stream) {
  if (n <= 0) {
    if (total_dev) ME_HIP(hipMemsetAsync(total_dev, 0, sizeof(T), stream));
    return 0;
  }
  if (n <= kScanSingleMax) {
    hipLaunchKernelGGL(k_scan_single<T>, dim3(1), dim3(kScanSingleThreads), 0, stream, in, out, n, total_dev);
    ME_LAUNCH_CHECK();
    return 0;
  }
  ME_CHECK(ws_bytes >= scan_workspace_bytes(n), "scan workspace too small");
  T *bsums = reinterpret_cast<T *>(ws);
  const int64_t nb = ceil_div(n, kScanBlock);
  hipLaunchKernelGGL(k_scan_block_sums<T>, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n, bsums);
  ME_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_of_sums<T>, dim3(1), dim3(kScanThreads), 0, stream, bsums, nb, total_dev);
  ME_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_apply<T>, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, out, n, bsums);
  ME_LAUNCH_CHECK();
  return 0;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total_dev, void *ws,
                       int64_t ws_bytes, hipStream_t stream) {
  return exclusive_scan<uint32_t>(in, out, n, total_dev, ws, ws_bytes, stream);
}

// =================================================================================================
// insert_and_map
// =================================================================================================
template <int NCOL>
__global__ __launch_bounds__(256) void k_insert(const int32_t *__restrict__ coords, int64_t n,
                                               uint64_t *table, uint32_t mask,
                                               uint32_t *__restrict__ slot_of_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t key[NCOL];
  load_coords<NCOL>(coords, i, key);
  const uint32_t h = hash_coords<NCOL>(key);
  const uint64_t mine = ((uint64_t)h << 32) | (uint32_t)i;
  uint32_t pos = h & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    // claim first, look afterwards (round 6): at <= 50 % load most first probes hit an empty slot, and a plain load before
    // the CAS was a second dependent round trip to the L2 for every one of them
    const uint64_t cur = atomicCAS(reinterpret_cast<unsigned long long *>(&table[pos]), (unsigned long long)kEmptySlot,
                                   (unsigned long long)mine);
    if (cur == kEmptySlot) {
      slot_of_row[i] = pos;
      return;
    }
    if ((uint32_t)(cur >> 32) == h) {
      const uint32_t r = (uint32_t)cur;
      int32_t other[NCOL];
      load_coords<NCOL>(coords, r, other);
      if (coords_equal<NCOL>(other, key)) {
        // same coordinate: the smallest input row owns the slot (tag bits are equal, so a
        // 64-bit min is a min over the row field).
        atomicMin(reinterpret_cast<unsigned long long *>(&table[pos]), (unsigned long long)mine);
        slot_of_row[i] = pos;
        return;
      }
    }
    pos = (pos + 1) & mask;
  }
  slot_of_row[i] = 0xffffffffu;  // table full: cannot happen with capacity >= 2n
}

// flag[i] = 1 iff input row i is the first occurrence of its coordinate; wrow[i] = winning row
__global__ __launch_bounds__(256) void k_insert_resolve(const uint64_t *__restrict__ table,
                                                       const uint32_t *__restrict__ slot_of_row,
                                                       int64_t n, uint32_t *__restrict__ wrow,
                                                       uint32_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = slot_of_row[i];
  const uint32_t w = (s == 0xffffffffu) ? (uint32_t)i : (uint32_t)table[s];
  wrow[i] = w;
  flag[i] = (w == (uint32_t)i) ? 1u : 0u;
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_insert_finalize(
    const int32_t *__restrict__ coords, int64_t n, uint64_t *table,
    const uint32_t *__restrict__ slot_of_row, const uint32_t *__restrict__ wrow,
    const uint32_t *__restrict__ newid, int32_t *__restrict__ coords_unique,
    int64_t *__restrict__ unique_map, int64_t *__restrict__ inverse_map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t w = wrow[i];
  const uint32_t id = newid[w];
  inverse_map[i] = (int64_t)id;
  if (w == (uint32_t)i) {
    unique_map[id] = i;
    int32_t key[NCOL];
    load_coords<NCOL>(coords, i, key);
    store_coords<NCOL>(coords_unique, id, key);
    const uint32_t s = slot_of_row[i];
    if (s != 0xffffffffu) table[s] = (table[s] & 0xffffffff00000000ull) | (uint64_t)id;
  }
}

// ---- fused resolve / rank / emit (round 6) -----------------------------------------------------------------------------
// The pipeline above needed resolve -> scan (ONE 1024-thread block for up to 2^20 rows: 25 us at 100k rows, the longest
// kernel of an insert) -> finalize, plus a bounding-box kernel with its own initialising copy.  Here the rank of a first
// occurrence is never materialised: k_insert_flags leaves, per 64 consecutive rows, the ballot of the "first occurrence"
// flags and the count of flags before them inside their 4096-row block, per block its flag count and bounding box;
// k_insert_emit rebuilds any row's new index as  base[block] + prefix[64-row group] + popcount(ballot below the row)
// — three small loads — after every block has scanned the block counts (n / 1024 of them) in LDS.  No scan kernel, no
// bounding-box kernel, no initialising copies: insert = fill + 3 launches.  Same results (first occurrence wins, unique rows
// in input order): bit-identical maps.
constexpr int kInsBlockRows = 1024;               // rows per block of the two kernels below (256 threads x 4: 98 blocks at 100k rows)
constexpr int kInsBlockShift = 10;
constexpr int kInsJ = kInsBlockRows / 256;        // rows per thread; the block holds 4 * kInsJ groups of 64 rows
constexpr int kInsMaxBlocks = 8192;               // block counts scanned in LDS by every block: n <= 2^23 rows (beyond: the scan pipeline)

template <int NCOL>
__global__ __launch_bounds__(256) void k_insert_flags(const int32_t *__restrict__ coords, const uint64_t *__restrict__ table,
                                                     const uint32_t *__restrict__ slot_of_row, int64_t n,
                                                     uint32_t *__restrict__ wrow, uint64_t *__restrict__ flagbits,
                                                     uint32_t *__restrict__ group_prefix, uint32_t *__restrict__ blk_count,
                                                     int32_t *__restrict__ blk_bbox) {
  __shared__ uint32_t s_cnt[64];
  __shared__ int32_t s_lo[4][NCOL], s_hi[4][NCOL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kInsBlockRows;
  int32_t lo[NCOL], hi[NCOL];
#pragma unroll
  for (int d = 0; d < NCOL; ++d) {
    lo[d] = INT32_MAX;
    hi[d] = INT32_MIN;
  }
  if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kInsJ; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;      // 64-row group j * 4 + wave of the block
    bool first = false;
    if (i < n) {
      const uint32_t s = slot_of_row[i];
      const uint32_t w = (s == 0xffffffffu) ? (uint32_t)i : (uint32_t)table[s];
      wrow[i] = w;
      first = w == (uint32_t)i;
      int32_t c[NCOL];
      load_coords<NCOL>(coords, i, c);
#pragma unroll
      for (int d = 0; d < NCOL; ++d) {
        lo[d] = min(lo[d], c[d]);
        hi[d] = max(hi[d], c[d]);
      }
    }
    const unsigned long long bits = __ballot(first);
    if (lane == 0) {
      const int64_t grp = (base >> 6) + j * 4 + wave;
      if (grp * 64 < n) flagbits[grp] = bits;
      s_cnt[j * 4 + wave] = (uint32_t)__popcll(bits);
    }
  }
#pragma unroll
  for (int d = 0; d < NCOL; ++d) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = min(lo[d], __shfl_xor(lo[d], off, 64));
      hi[d] = max(hi[d], __shfl_xor(hi[d], off, 64));
    }
    if (lane == 0) {
      s_lo[wave][d] = lo[d];
      s_hi[wave][d] = hi[d];
    }
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t c = s_cnt[lane];
    const uint32_t incl = wave_inclusive_scan(c);
    const int64_t grp = (base >> 6) + lane;
    if (lane < 4 * kInsJ && grp * 64 < n) group_prefix[grp] = incl - c;
    if (lane == 63) blk_count[blockIdx.x] = incl;
    if (lane < NCOL) {
      blk_bbox[(int64_t)blockIdx.x * 2 * NCOL + lane] = min(min(s_lo[0][lane], s_lo[1][lane]), min(s_lo[2][lane], s_lo[3][lane]));
      blk_bbox[(int64_t)blockIdx.x * 2 * NCOL + NCOL + lane] =
          max(max(s_hi[0][lane], s_hi[1][lane]), max(s_hi[2][lane], s_hi[3][lane]));
    }
  }
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_insert_emit(const int32_t *__restrict__ coords, int64_t n, uint64_t *table,
                                                    const uint32_t *__restrict__ slot_of_row,
                                                    const uint32_t *__restrict__ wrow, const uint64_t *__restrict__ flagbits,
                                                    const uint32_t *__restrict__ group_prefix,
                                                    const uint32_t *__restrict__ blk_count, const int32_t *__restrict__ blk_bbox,
                                                    int nb, int32_t *__restrict__ coords_unique,
                                                    int64_t *__restrict__ unique_map, int64_t *__restrict__ inverse_map,
                                                    uint32_t *__restrict__ total_and_bbox) {
  __shared__ uint32_t s_base[kInsMaxBlocks];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (this thread's row: requested before the scan below, used after it)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const uint32_t w = valid ? wrow[i] : 0u;
  const uint32_t grp = w >> 6;
  const uint32_t gp = valid ? group_prefix[grp] : 0u;
  const uint64_t fb = valid ? flagbits[grp] : 0ull;
  // exclusive scan of the block counts, the same in every block: wave 0 walks them 64 at a time
  __shared__ uint32_t s_total;
  if (wave == 0) {
    uint32_t run = 0;
    for (int b0 = 0; b0 < nb; b0 += 64) {
      const int b = b0 + lane;
      const uint32_t c = b < nb ? blk_count[b] : 0u;
      const uint32_t incl = wave_inclusive_scan(c);
      if (b < nb) s_base[b] = run + incl - c;
      run += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_total = run;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    // the number of unique rows and the bounding box of the INPUT rows (the one read-back of an insert); the per-block
    // boxes are folded by all threads of this block — 2 NCOL columns x 256 / (2 NCOL) strided walkers, LDS atomics —:
    // eight threads walking nb dependent loads each made this block the whole kernel's duration (15 of 17 us at 100k rows)
    __shared__ int32_t s_bb[2 * (ME_MAX_DIM + 1)];
    if (threadIdx.x < 2 * NCOL) s_bb[threadIdx.x] = threadIdx.x < NCOL ? INT32_MAX : INT32_MIN;
    __syncthreads();
    constexpr int kWalkers = 256 / (2 * NCOL);
    const int d = threadIdx.x % (2 * NCOL), j0 = threadIdx.x / (2 * NCOL);
    if (j0 < kWalkers) {
      int32_t r = d < NCOL ? INT32_MAX : INT32_MIN;
      for (int b = j0; b < nb; b += kWalkers) {
        const int32_t x = blk_bbox[(int64_t)b * 2 * NCOL + d];
        r = d < NCOL ? min(r, x) : max(r, x);
      }
      if (d < NCOL) atomicMin(&s_bb[d], r);
      else atomicMax(&s_bb[d], r);
    }
    __syncthreads();
    if (threadIdx.x == 0) total_and_bbox[0] = s_total;
    if (threadIdx.x < 2 * NCOL) reinterpret_cast<int32_t *>(total_and_bbox)[1 + threadIdx.x] = s_bb[threadIdx.x];
  }
  // (one row per thread, as many blocks as that takes: the ranks only need the scanned counts of the 1024-row blocks)
  if (valid) {
    const uint32_t id = s_base[w >> kInsBlockShift] + gp + (uint32_t)__popcll(fb & ((1ull << (w & 63u)) - 1ull));
    inverse_map[i] = (int64_t)id;
    if (w == (uint32_t)i) {
      unique_map[id] = i;
      int32_t key[NCOL];
      load_coords<NCOL>(coords, i, key);
      store_coords<NCOL>(coords_unique, id, key);
      const uint32_t s = slot_of_row[i];
      if (s != 0xffffffffu) table[s] = (table[s] & 0xffffffff00000000ull) | (uint64_t)id;
    }
  }
}

// =================================================================================================
// stride / find
// =================================================================================================
struct StrideArg {
  int32_t ts[ME_MAX_DIM];
};

__device__ __forceinline__ int32_t floor_to_multiple(int32_t c, int32_t ts) {
  int32_t q = c / ts;
  if ((c % ts != 0) && ((c < 0) != (ts < 0))) --q;
  return q * ts;
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_stride(const int32_t *__restrict__ coords, int64_t n,
                                               StrideArg arg, int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c[NCOL];
  load_coords<NCOL>(coords, i, c);
#pragma unroll
  for (int d = 1; d < NCOL; ++d) c[d] = floor_to_multiple(c[d], arg.ts[d - 1]);
  store_coords<NCOL>(out, i, c);
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_find(const uint64_t *__restrict__ table, uint32_t mask,
                                             const int32_t *__restrict__ map_coords,
                                             const int32_t *__restrict__ queries, int64_t nq,
                                             int32_t *__restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  int32_t key[NCOL];
  load_coords<NCOL>(queries, i, key);
  rows[i] = table_find<NCOL>(table, mask, map_coords, key);
}

// =================================================================================================
// spatial keys
// =================================================================================================
// Z-order (Morton) key of a coordinate row: batch index in the top bits, then the bit-interleaved
// spatial coordinates in units of the tensor stride (biased to be non-negative, truncated to `bits`
// bits per axis — the key only has to give locality, not identity).  Rows sorted by this key form the
// tiles of the convolution plan: a tile's 27 offsets then gather (almost) the same ~3x tile-size source
// rows, which stay in the XCD's L2 instead of being fetched 27 times from HBM / Infinity Cache.
template <int NCOL>
__global__ __launch_bounds__(256) void k_spatial_keys(const int32_t *__restrict__ coords, int64_t n,
                                                     StrideArg ts, int bits, int64_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c[NCOL];
  load_coords<NCOL>(coords, i, c);
  uint64_t key = 0;
  const uint32_t bias = 1u << (bits - 1), mask = (1u << bits) - 1u;
  uint32_t v[NCOL - 1];
#pragma unroll
  for (int d = 0; d < NCOL - 1; ++d) v[d] = ((uint32_t)(floor_to_multiple(c[d + 1], ts.ts[d]) / ts.ts[d]) + bias) & mask;
  for (int b = 0; b < bits; ++b) {
#pragma unroll
    for (int d = 0; d < NCOL - 1; ++d) key |= (uint64_t)((v[d] >> b) & 1u) << (b * (NCOL - 1) + d);
  }
  key |= (uint64_t)((uint32_t)c[0] & 0x7fu) << 56;
  keys[i] = (int64_t)key;
}

// =================================================================================================
// kernel map
// =================================================================================================
// neighbour coordinate of offset k: src/kernel_region.hpp:198-247
template <int NCOL>
__device__ __forceinline__ void region_coordinate_at(const me_region &rg, int32_t k,
                                                     const int32_t (&src)[NCOL],
                                                     int32_t (&dst)[NCOL]) {
  dst[0] = src[0];
  if (rg.region_type == ME_REGION_HYPER_CUBE) {
    int32_t rem = k;
#pragma unroll
    for (int d = 0; d < NCOL - 1; ++d) {
      const int32_t ks = rg.kernel_size[d];
      const int32_t idx = rem % ks;
      rem /= ks;
      const int32_t step = rg.dilation[d] * rg.tensor_stride[d];
      dst[d + 1] = src[d + 1] + ((ks % 2 == 0) ? idx : (idx - ks / 2)) * step;
    }
  } else {  // HYPER_CROSS: centre first, then per axis the (ks-1) off-centre taps
#pragma unroll
    for (int d = 1; d < NCOL; ++d) dst[d] = src[d];
    if (k == 0) return;
    int32_t ind = k - 1;
    int axis = 0;
    while (axis < NCOL - 1) {
      if (ind < rg.kernel_size[axis] - 1) break;
      ind -= rg.kernel_size[axis] - 1;
      ++axis;
    }
    if (axis >= NCOL - 1) return;
    const int32_t r = (rg.kernel_size[axis] - 1) / 2;
    const int32_t off = (ind < r) ? (ind + 1) : (ind - 2 * r);
    // dst[axis + 1] with a runtime axis: unrolled select keeps the array in registers
#pragma unroll
    for (int d = 0; d < NCOL - 1; ++d)
      if (d == axis) dst[d + 1] += off * rg.dilation[d] * rg.tensor_stride[d];
  }
}

// Coordinates of a kernel region around every input coordinate (generative / transposed convolution with
// expand_coordinates: CoordinateMapCPU::stride_region, src/coordinate_map_cpu.hpp:446-487).  Candidate
// (row, k) is written at row * volume + k, so a first-occurrence dedup orders the new map by input row, then
// kernel offset.  `aligned` (may be NULL): 1 where every spatial coordinate is a multiple of the region's
// tensor stride times `align` (the non-transposed expand_coordinates case keeps only those).
template <int NCOL>
__global__ __launch_bounds__(256) void k_expand_region(const int32_t *__restrict__ coords, int64_t n,
                                                      me_region rg, int32_t volume, StrideArg align,
                                                      int32_t *__restrict__ out, uint8_t *__restrict__ aligned) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * volume) return;
  const int64_t row = idx / volume;
  const int32_t k = (int32_t)(idx % volume);
  int32_t src[NCOL], dst[NCOL];
  load_coords<NCOL>(coords, row, src);
  region_coordinate_at<NCOL>(rg, k, src, dst);
  store_coords<NCOL>(out, idx, dst);
  if (aligned != nullptr) {
    bool ok = true;
#pragma unroll
    for (int d = 0; d < NCOL - 1; ++d) ok = ok && (dst[d + 1] % align.ts[d] == 0);
    aligned[idx] = ok ? 1 : 0;
  }
}

// one thread per (output row u, offset k = blockIdx.y); lanes = consecutive u
template <int NCOL>
__global__ __launch_bounds__(256) void k_kmap_probe(const uint64_t *__restrict__ in_table,
                                                   uint32_t mask,
                                                   const int32_t *__restrict__ in_coords,
                                                   const int32_t *__restrict__ out_coords,
                                                   int64_t n_out, me_region rg,
                                                   int32_t *__restrict__ nbr,
                                                   uint32_t *__restrict__ wcount, int64_t nw) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = blockIdx.y;
  int32_t r = -1;
  if (u < n_out) {
    int32_t src[NCOL], key[NCOL];
    load_coords<NCOL>(out_coords, u, src);
    region_coordinate_at<NCOL>(rg, k, src, key);
    r = table_find<NCOL>(in_table, mask, in_coords, key);
    nbr[(int64_t)k * n_out + u] = r;
  }
  const unsigned long long m = __ballot(r >= 0);
  const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (lane_id() == 0 && wave_global < nw) wcount[(int64_t)k * nw + wave_global] = (uint32_t)__popcll(m);
}

__global__ void k_kmap_koffsets(const uint32_t *__restrict__ woffs, const uint32_t *__restrict__ total,
                                int64_t nw, int64_t volume, int64_t *__restrict__ koffs) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < volume) koffs[k] = (int64_t)woffs[k * nw];
  if (k == volume) koffs[k] = (int64_t)(*total);
}

__global__ __launch_bounds__(256) void k_kmap_compact(const int32_t *__restrict__ nbr, int64_t n_out,
                                                     const uint32_t *__restrict__ woffs, int64_t nw,
                                                     int32_t *__restrict__ in_pairs,
                                                     int32_t *__restrict__ out_pairs) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = blockIdx.y;
  const int32_t r = (u < n_out) ? nbr[(int64_t)k * n_out + u] : -1;
  const unsigned long long m = __ballot(r >= 0);
  const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 0) {
    const uint32_t dst = woffs[(int64_t)k * nw + wave_global] + mask_prefix(m);
    in_pairs[dst] = r;
    out_pairs[dst] = (int32_t)u;
  }
}

__global__ __launch_bounds__(256) void k_kmap_transpose(const int32_t *__restrict__ in_pairs,
                                                       const int32_t *__restrict__ out_pairs,
                                                       const int64_t *__restrict__ koffs,
                                                       int64_t n_in, int32_t *__restrict__ nbrT) {
  const int32_t k = blockIdx.y;
  const int64_t e0 = koffs[k], e1 = koffs[k + 1];
  for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1;
       e += (int64_t)gridDim.x * blockDim.x)
    nbrT[(int64_t)k * n_in + in_pairs[e]] = out_pairs[e];
}

// =================================================================================================
// tile plan
// =================================================================================================
// one wavefront per (tile, offset) item; tiles hold `tile_rows` consecutive target rows.  An item's
// valid entries are padded to groups of 16 (one MFMA N-tile) and its groups are cut into BATCHES of at
// most `batch_groups` groups: a batch is what the convolution kernel stages in LDS at once, and it
// never mixes offsets (one weight slice per batch).
__device__ __forceinline__ uint64_t plan_count_item(const int32_t *__restrict__ tbl, const int32_t *__restrict__ order,
                                                    int64_t n_tgt, int64_t volume, int64_t item, int tile_rows,
                                                    int batch_groups) {
  const int64_t t = item / volume, k = item % volume;
  const int64_t row0 = t * tile_rows;
  uint32_t count = 0;
  for (int c = 0; c < tile_rows; c += 64) {
    const int local = c + lane_id();
    const int64_t pos = row0 + local;
    int32_t r = -1;
    if (local < tile_rows && pos < n_tgt) r = tbl[k * n_tgt + (order ? (int64_t)order[pos] : pos)];
    count += (uint32_t)__popcll(__ballot(r >= 0));
  }
  const uint32_t groups = (count + ME_GROUP_ROWS - 1) / ME_GROUP_ROWS;
  // groups in the low word, batches in the high word: ONE scan yields both offsets
  return ((uint64_t)((groups + batch_groups - 1) / batch_groups) << 32) | groups;
}

__global__ __launch_bounds__(256) void k_plan_count(const int32_t *__restrict__ tbl,
                                                   const int32_t *__restrict__ order, int64_t n_tgt,
                                                   int64_t volume, int64_t n_items, int tile_rows,
                                                   int batch_groups, uint64_t *__restrict__ count_gb) {
  const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= n_items) return;  // wave-uniform
  const uint64_t c = plan_count_item(tbl, order, n_tgt, volume, item, tile_rows, batch_groups);
  if (lane_id() == 0) count_gb[item] = c;
}

// The valid entries of an item are not stored in row order: they are DEALT to the item's groups and
// slots by the residue of their target row.  The convolution kernel adds the 16x16 result of a group
// into its LDS accumulator with one 16-byte access per lane at the lane's target row; the LDS serves a
// 16-byte store in passes of 8 consecutive lanes, and two rows whose indices are congruent mod 8 fall
// into the same banks (the accumulator row stride is an odd number of 16-byte units).  Entries are
// therefore ordered by (row mod 8) — position p —, entry p goes to group p % groups, step s = p / groups,
// and even steps fill slots 0-7, odd steps slots 8-15: each half of a group then holds (close to) one
// row of every residue, instead of ~3 rows in the fullest residue for 8 random rows.
// (one wave per item; the plan's totals — tile_bptr[n_tiles], item_gptr[n_items] — are the caller's to store)
__device__ __forceinline__ void plan_fill_item(const int32_t *__restrict__ tbl, const int32_t *__restrict__ order,
                                               int64_t n_tgt, int64_t volume, int64_t n_items, int64_t item, int tile_rows,
                                               int batch_groups, uint64_t off_gb, int32_t *__restrict__ plan_src,
                                               int32_t *__restrict__ plan_dst, int32_t *__restrict__ batch_desc,
                                               int32_t *__restrict__ tile_bptr, int32_t *__restrict__ item_gptr) {
  const int lane = lane_id();
  const int64_t t = item / volume, k = item % volume;
  const int64_t row0 = t * tile_rows;
  // off_gb: first group (low word) and first batch (high word) of the item
  const uint32_t g0 = (uint32_t)off_gb;
  const int64_t slot0 = (int64_t)g0 * ME_GROUP_ROWS;
  constexpr int kChunks = ME_MAX_TILE_ROWS / 64;
  // local row = c * 64 + lane, so its residue mod 8 is lane & 7 in every chunk
  const unsigned long long class_mask = 0x0101010101010101ull << (lane & 7);
  int32_t r[kChunks];
  uint32_t my_class_count = 0, total = 0;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int local = c * 64 + lane;
    const int64_t pos = row0 + local;
    r[c] = -1;
    if (local < tile_rows && pos < n_tgt) r[c] = tbl[k * n_tgt + (order ? (int64_t)order[pos] : pos)];
    const unsigned long long m = __ballot(r[c] >= 0);
    my_class_count += (uint32_t)__popcll(m & class_mask);
    total += (uint32_t)__popcll(m);
  }
  const uint32_t groups = (total + ME_GROUP_ROWS - 1) / ME_GROUP_ROWS;
  const uint32_t padded = groups * ME_GROUP_ROWS;
  // entries of smaller residues come first (lanes 0-7 hold the counts of residues 0-7)
  uint32_t run = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t cj = __shfl(my_class_count, j, 64);
    if (j < (lane & 7)) run += cj;
  }
  auto place = [&](uint32_t p) {  // position in residue order -> slot of the plan
    const uint32_t gi = p % groups, s = p / groups;
    return slot0 + (int64_t)gi * ME_GROUP_ROWS + (((s & 1u) << 3) | (s >> 1));
  };
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const unsigned long long mine = __ballot(r[c] >= 0) & class_mask;
    if (r[c] >= 0) {
      const int64_t s = place(run + mask_prefix(mine));
      plan_src[s] = r[c];
      plan_dst[s] = c * 64 + lane;
    }
    run += (uint32_t)__popcll(mine);
  }
  // padding positions (fewer than 16): no source row, dummy accumulator row
  if (total + lane < padded) {
    const int64_t s = place(total + lane);
    plan_src[s] = -1;
    plan_dst[s] = tile_rows;
  }
  const uint32_t b0 = (uint32_t)(off_gb >> 32);
  const uint32_t nb = (groups + batch_groups - 1) / batch_groups;
  for (uint32_t j = lane; j < nb; j += 64) {
    // the groups are dealt evenly over the nb batches (5 groups -> 3 + 2, not 4 + 1): no one-group stragglers
    const uint32_t base = groups / nb, rem = groups % nb;
    const uint32_t first = j * base + min(j, rem);
    const uint32_t ng = base + (j < rem ? 1u : 0u);
    batch_desc[2 * (int64_t)(b0 + j)] = (int32_t)(g0 + first);
    batch_desc[2 * (int64_t)(b0 + j) + 1] = (int32_t)(((uint32_t)k << 8) | ng);
  }
  // The convolution kernels read the index window of a batch (ME_MAX_BATCH_GROUPS * 16 entries from its first
  // group) unconditionally: behind the last group of the plan follow 64 entries that gather row 0 into the dummy
  // row (me_plan_max_groups reserves them).
  if (item == n_items - 1) {
    plan_src[slot0 + (int64_t)padded + lane] = 0;
    plan_dst[slot0 + (int64_t)padded + lane] = tile_rows;
  }
  if (lane == 0) {
    item_gptr[item] = (int32_t)g0;
    if (k == 0) tile_bptr[t] = (int32_t)b0;
  }
}

__global__ __launch_bounds__(256) void k_plan_fill(const int32_t *__restrict__ tbl,
                                                  const int32_t *__restrict__ order, int64_t n_tgt,
                                                  int64_t volume, int64_t n_items, int tile_rows,
                                                  int batch_groups, const uint64_t *__restrict__ offs_gb,
                                                  const uint64_t *__restrict__ total_gb,
                                                  int32_t *__restrict__ plan_src,
                                                  int32_t *__restrict__ plan_dst,
                                                  int32_t *__restrict__ batch_desc,
                                                  int32_t *__restrict__ tile_bptr,
                                                  int32_t *__restrict__ item_gptr, int64_t n_tiles) {
  const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= n_items) return;  // wave-uniform
  plan_fill_item(tbl, order, n_tgt, volume, n_items, item, tile_rows, batch_groups, offs_gb[item], plan_src, plan_dst,
                 batch_desc, tile_bptr, item_gptr);
  if (item == 0 && lane_id() == 0) {
    tile_bptr[n_tiles] = (int32_t)(*total_gb >> 32);
    item_gptr[n_items] = (int32_t)(uint32_t)(*total_gb);
  }
}

// Dispatch order of the tiles: heaviest first (longest-processing-time-first list scheduling).  The hardware hands
// workgroups to CUs in blockIdx order as slots free up; a plan has about two rounds of tiles per slot and their work
// (16-row groups) varies by +-10 % — +-20 % for spatially compact tiles, whose density follows the scene — so in
// tile order the last round ends ragged (simulated makespan 1.04x / 1.11x the ideal on config 2; 1.01x sorted).
// One workgroup: min / max of the work, a 256-bin counting sort by descending work (the order inside a bin is
// arbitrary: it only permutes the dispatch, never a result).  perm = tile_bptr + n_tiles + 1.
// XCD > 0 (round 4): the hardware deals workgroups to the eight XCDs round-robin by their linear index, so slot l of the
// dispatch order runs on XCD l % 8 — and each XCD has its own 4 MB L2.  Tiles that are neighbours in the plan's tile
// order (spatially compact tiles: neighbours in space, whose 3^D halos overlap) are therefore handed to ONE XCD: the
// tile sequence is cut into XCD contiguous chunks (chunk x gets as many tiles as XCD x gets slots), heaviest first
// inside a chunk, and the j-th tile of chunk x goes to slot 8 j + x.  The order only permutes the dispatch.
// Measured (profiles/r04_tile_dispatch_sweep.log, r04_unet_tile_policy.log): 1 - 5 % on the large row-tiled bf16 layers
// taken alone, nothing on spatial tiles, nothing on the MinkUNet34C step (11.99 vs 12.01 ms) — opt-in
// (ME_AMD_TILE_DISPATCH=1 / me_debug_set_tile_dispatch), not the default.
int g_tile_dispatch = 0;   // me_debug_set_tile_dispatch: 0 = heaviest first over the whole launch, 1 = XCD chunks

template <int XCD>
__device__ __forceinline__ void plan_tile_order(const int32_t *__restrict__ item_gptr, int64_t volume, int64_t n_tiles,
                                                int32_t *__restrict__ perm) {   // one workgroup of 1024 threads
  constexpr int CH = XCD > 0 ? XCD : 1;          // chunks
  constexpr int NB = 256 * CH;                   // bins: chunk-major, heavy tiles in a chunk's low bins
  constexpr int PER = (NB + 1023) / 1024;        // bins per thread in the scan
  __shared__ int32_t s_min, s_max;
  __shared__ uint32_t s_bin[NB];
  __shared__ uint32_t s_wsum[16];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_min = INT32_MAX;
    s_max = 0;
  }
  for (int b = tid; b < NB; b += 1024) s_bin[b] = 0;
  __syncthreads();
  auto work_of = [&](int64_t t) { return item_gptr[(t + 1) * volume] - item_gptr[t * volume]; };
  // the first tile of every thread stays in a register (plans rarely have more tiles than the block has threads)
  const int32_t w_first = tid < n_tiles ? work_of(tid) : 0;
  auto work = [&](int64_t t) { return t == tid ? w_first : work_of(t); };
  int32_t lo = INT32_MAX, hi = 0;
  for (int64_t t = tid; t < n_tiles; t += blockDim.x) {
    const int32_t w = work(t);
    lo = min(lo, w);
    hi = max(hi, w);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {   // one LDS atomic per wave
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((tid & 63) == 0) {
    atomicMin(&s_min, lo);
    atomicMax(&s_max, hi);
  }
  __syncthreads();
  const int32_t wmin = s_min, span = max(s_max - s_min, 1);
  // chunk x holds tiles [first(x), first(x + 1)): first(x) = slots of XCDs 0 .. x - 1 = sum of ceil((n - x') / CH)
  auto first = [&](int x) {
    const int64_t q = n_tiles / CH, r = n_tiles % CH;
    return q * x + min((int64_t)x, r);
  };
  auto chunk_of = [&](int64_t t) {
    if (CH == 1) return 0;
    const int64_t q = n_tiles / CH, r = n_tiles % CH;
    return (int)(t < (q + 1) * r ? t / (q + 1) : r + (t - (q + 1) * r) / max(q, (int64_t)1));
  };
  auto bin = [&](int64_t t, int32_t w) { return chunk_of(t) * 256 + 255 - (int)(((int64_t)(w - wmin) * 255) / span); };
  for (int64_t t = tid; t < n_tiles; t += blockDim.x) atomicAdd(&s_bin[bin(t, work(t))], 1u);
  __syncthreads();
  // exclusive scan of the bins: PER consecutive bins per thread, a shuffle scan inside each wave + the wave totals (a
  // serial scan by one thread — 256 dependent LDS round trips — was most of this kernel's 10 us)
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = tid * PER + i < NB ? s_bin[tid * PER + i] : 0u;
    sum += v[i];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if ((tid & 63) >= off) incl += up;
  }
  if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
  for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (tid * PER + i < NB) s_bin[tid * PER + i] = base;
    base += v[i];
  }
  __syncthreads();
  for (int64_t t = tid; t < n_tiles; t += blockDim.x) {
    const uint32_t m = atomicAdd(&s_bin[bin(t, work(t))], 1u);   // rank in (chunk, weight) order
    if (CH == 1) {
      perm[m] = (int32_t)t;
    } else {
      const int x = chunk_of(t);
      perm[(int64_t)(m - (uint32_t)first(x)) * CH + x] = (int32_t)t;
    }
  }
}

template <int XCD>
__global__ __launch_bounds__(1024) void k_plan_tile_order(const int32_t *__restrict__ item_gptr, int64_t volume,
                                                         int64_t n_tiles, int32_t *__restrict__ perm) {
  plan_tile_order<XCD>(item_gptr, volume, n_tiles, perm);
}

// ---- all plans of a scene in four launches (round 4) ------------------------------------------------------------
// A network on a new scene needs ~45 plans (MinkUNet34C: 10 kernel maps x forward / dgrad x tile geometries); one
// me_plan_build each is 4 launches + 2 memsets of 4 - 8 us on mostly idle hardware, and ~55 us of host time: 2.5 ms of
// host and 1.2 ms of GPU time per scene.  me_plan_build_multi walks a table of jobs instead: the items of all jobs are
// numbered through (item_base), one wave per item finds its job by bisection; the scan and the dispatch order take one
// workgroup per job.  The arrays it writes are those of me_plan_build, bit for bit.
__device__ __forceinline__ int plan_job_of(const me_plan_job *__restrict__ jobs, int n_jobs, int64_t gi) {
  int lo = 0, hi = n_jobs - 1;   // last job with item_base <= gi (jobs without items share their successor's base)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].item_base <= gi) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_plan_count_multi(const me_plan_job *__restrict__ jobs, int n_jobs,
                                                         int64_t total_items, uint64_t *__restrict__ count_gb) {
  const int64_t gi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (gi >= total_items) return;  // wave-uniform
  const me_plan_job &j = jobs[plan_job_of(jobs, n_jobs, gi)];
  const uint64_t c = plan_count_item(j.tbl, j.order, j.n_tgt, j.volume, gi - j.item_base, j.tile_rows, j.batch_groups);
  if (lane_id() == 0) count_gb[gi] = c;
}

// exclusive scan of a job's item counts (one workgroup per job, 8192 items per pass with a running carry) + the
// plan's totals
__global__ __launch_bounds__(kScanSingleThreads) void k_plan_scan_multi(const me_plan_job *__restrict__ jobs,
                                                                       const uint64_t *__restrict__ count_gb,
                                                                       uint64_t *__restrict__ offs_gb) {
  constexpr int kWaves = kScanSingleThreads / 64;
  __shared__ uint64_t s_wave[kWaves];
  const me_plan_job &j = jobs[blockIdx.x];
  const uint64_t *in = count_gb + j.item_base;
  uint64_t *out = offs_gb + j.item_base;
  const int64_t n = j.n_items;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  uint64_t carry = 0;
  for (int64_t blk = 0; blk < n; blk += kScanSingleThreads * kScanSingleItems) {
    const int64_t base = blk + (int64_t)threadIdx.x * kScanSingleItems;
    uint64_t v[kScanSingleItems];
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanSingleItems; ++i) {
      v[i] = (base + i < n) ? in[base + i] : 0ull;
      sum += v[i];
    }
    const uint64_t incl = wave_inclusive_scan_t(sum);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint64_t t = s_wave[w];
      if (w < wave) wave_off += t;
      total += t;
    }
    __syncthreads();
    uint64_t ex = carry + wave_off + incl - sum;
#pragma unroll
    for (int i = 0; i < kScanSingleItems; ++i) {
      if (base + i < n) out[base + i] = ex;
      ex += v[i];
    }
    carry += total;
  }
  if (threadIdx.x == 0) {
    j.tile_bptr[j.n_tiles] = (int32_t)(carry >> 32);
    j.item_gptr[n] = (int32_t)(uint32_t)carry;
  }
}

__global__ __launch_bounds__(256) void k_plan_fill_multi(const me_plan_job *__restrict__ jobs, int n_jobs,
                                                        int64_t total_items, const uint64_t *__restrict__ offs_gb) {
  const int64_t gi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (gi >= total_items) return;  // wave-uniform
  const me_plan_job &j = jobs[plan_job_of(jobs, n_jobs, gi)];
  plan_fill_item(j.tbl, j.order, j.n_tgt, j.volume, j.n_items, gi - j.item_base, j.tile_rows, j.batch_groups, offs_gb[gi],
                 j.plan_src, j.plan_dst, j.batch_desc, j.tile_bptr, j.item_gptr);
}

template <int XCD>
__global__ __launch_bounds__(1024) void k_plan_tile_order_multi(const me_plan_job *__restrict__ jobs) {
  const me_plan_job &j = jobs[blockIdx.x];
  plan_tile_order<XCD>(j.item_gptr, j.volume, j.n_tiles, j.tile_bptr + j.n_tiles + 1);
}

// Voxel labels (the reference's quantize_label, src/quantization.cpp:140-196): a voxel keeps the label of
// its first point unless another point of the voxel disagrees -> ignore_label.  Every writer of a voxel
// writes the same value and the comparison reads the INPUT labels, so the result does not depend on the
// schedule.
__global__ __launch_bounds__(256) void k_quantize_labels_init(const int64_t *__restrict__ unique_map, int64_t n_unique,
                                                             const int32_t *__restrict__ labels,
                                                             int32_t *__restrict__ colabels) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_unique) colabels[u] = labels[unique_map[u]];
}

__global__ __launch_bounds__(256) void k_quantize_labels_mark(const int64_t *__restrict__ unique_map,
                                                             const int64_t *__restrict__ inverse_map,
                                                             const int32_t *__restrict__ labels, int64_t n,
                                                             int32_t ignore_label, int32_t *__restrict__ colabels) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int64_t u = inverse_map[row];
  if (labels[row] != labels[unique_map[u]]) colabels[u] = ignore_label;
}


// =================================================================================================
// spatial index + LDS-bucketed kernel map (round 2)
// =================================================================================================
// BASELINE's north_star asks for "LDS-bucketed open-address hashing with coalesced HBM reads".  The flat table
// above answers point queries; a KERNEL MAP asks, for every row, for its K spatial neighbours — N * K probes whose
// answers sit within a few cells of the query.  The bucket that serves them is therefore SPATIAL: the rows of a
// map are ordered by the supercell (16^3 cells for D = 3, 8^4 for D = 4, <= 4096 cells) that contains them
// (me_spatial_index_build: one key pass, a stable LSD radix sort, a dense supercell directory), and
// k_kmap_probe_lds gives every supercell of the query map to one workgroup, which
//   * reads the rows of the 3^D neighbouring supercells of the LOOKUP map as contiguous ranges of its sorted
//     coordinate array (coalesced 16-byte rows — the only HBM reads of the build) into a dense halo grid in LDS
//     (cell -> row id; (16 + 2 h)^3 int32 = 23 KiB for k = 3),
//   * answers all rows x K probes of the supercell with ONE ds_read each (own cell + a per-offset constant), and
//   * writes the neighbour table in POSITION space (position = rank in supercell order), where its supercell is a
//     contiguous range: coalesced stores.  Row-space consumers go through `order` (position -> row).
// No atomics, no hashing in the probe; the flat table remains the dedup structure of insert_and_map and the
// fallback for kernels whose halo does not fit the LDS or whose maps differ in tensor stride (strided maps: K = 8).

struct SpatialGrid {           // device copy of me_spatial_grid
  int32_t shift[ME_MAX_DIM];   // log2 of the supercell side per spatial axis (cells)
  int32_t sc_min[ME_MAX_DIM + 1];  // [0]: smallest batch index; [1 + d]: smallest supercell coordinate of axis d
  int32_t sc_dim[ME_MAX_DIM + 1];  // extents of the dense supercell directory (batch indices, supercells per axis)
  int32_t ts[ME_MAX_DIM];      // tensor stride = cell size
};

__device__ __forceinline__ int32_t floor_div(int32_t c, int32_t ts) {
  int32_t q = c / ts;
  if ((c % ts != 0) && ((c < 0) != (ts < 0))) --q;
  return q;
}

// linear supercell index of a coordinate row, or -1 outside the directory
template <int NCOL>
__device__ __forceinline__ int64_t supercell_of(const SpatialGrid &g, const int32_t (&c)[NCOL]) {
  int64_t lin = (int64_t)c[0] - g.sc_min[0];
  if (lin < 0 || lin >= g.sc_dim[0]) return -1;
#pragma unroll
  for (int d = 0; d < NCOL - 1; ++d) {
    const int32_t sc = (floor_div(c[d + 1], g.ts[d]) >> g.shift[d]) - g.sc_min[d + 1];
    if (sc < 0 || sc >= g.sc_dim[d + 1]) return -1;
    lin = lin * g.sc_dim[d + 1] + sc;
  }
  return lin;
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_sp_keys(const int32_t *__restrict__ coords, int64_t n, SpatialGrid g,
                                                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c[NCOL];
  load_coords<NCOL>(coords, i, c);
  const int64_t sc = supercell_of<NCOL>(g, c);   // inside by construction (the grid is the map's bounding box)
  keys[i] = sc < 0 ? 0u : (uint32_t)sc;
  vals[i] = (uint32_t)i;
}

// directory of the sorted keys: dir_start[sc] = first position whose key is >= sc (lower bound), dir_start[m] = n.
// (a histogram with one atomicAdd per row took 100 us on 100k rows in 125 supercells: contention; this is m
// independent binary searches over an L2-resident array)
__global__ __launch_bounds__(256) void k_sp_directory(const uint32_t *__restrict__ sorted_keys, int64_t n, int64_t m,
                                                     uint32_t *__restrict__ dir_start) {
  const int64_t sc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sc > m) return;
  int64_t lo = 0, hi = n;                         // first position with key >= sc
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)sorted_keys[mid] < sc) lo = mid + 1; else hi = mid;
  }
  dir_start[sc] = (uint32_t)lo;
}

// ---- stable LSD radix sort of (key, value) pairs, 8-bit digits ------------------------------------------------
constexpr int kRsTile = 1024;   // elements per block in both passes
template <typename K = uint32_t>
__global__ __launch_bounds__(256) void k_rs_hist(const K *__restrict__ keys, int64_t n, int shift,
                                                int64_t nblocks, uint32_t *__restrict__ hist) {
  __shared__ uint32_t s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll
  for (int j = 0; j < kRsTile / 256; ++j) {
    const int64_t e = base + j * 256 + threadIdx.x;
    if (e < n) atomicAdd(&s_h[(uint32_t)(keys[e] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];   // digit-major: one scan gives all offsets
}

// one wave per block walks its tile 64 elements at a time; inside a step the rank of an element among the
// equal digits of lower lanes comes from eight ballots (match-any), across steps from a running count per digit
// (vals_in == nullptr: the values are the positions themselves — the first pass of an argsort)
template <typename K = uint32_t>
__global__ __launch_bounds__(64) void k_rs_scatter(const K *__restrict__ keys_in,
                                                  const uint32_t *__restrict__ vals_in, int64_t n, int shift,
                                                  int64_t nblocks, const uint32_t *__restrict__ offs,
                                                  K *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
  __shared__ uint32_t s_cnt[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) s_cnt[d] = offs[(int64_t)d * nblocks + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kRsTile / 64; ++r) {
    const int64_t e = base + r * 64 + lane;
    const bool valid = e < n;
    const K key = valid ? keys_in[e] : (K)0;
    const uint32_t val = valid ? (vals_in != nullptr ? vals_in[e] : (uint32_t)e) : 0u;
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & lt);
    const uint32_t cnt = (uint32_t)__popcll(peers);
    uint32_t pos = 0;
    if (valid) pos = s_cnt[d] + rank;
    __syncthreads();                             // (one wave: orders the reads above before the updates below)
    if (valid && rank == 0) s_cnt[d] += cnt;
    __syncthreads();
    if (valid) {
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
  }
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_sp_finish(const int32_t *__restrict__ coords, int64_t n,
                                                  const uint32_t *__restrict__ sorted_rows,
                                                  int32_t *__restrict__ order, int32_t *__restrict__ pos_of_row,
                                                  int32_t *__restrict__ coords_sorted) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t row = sorted_rows[p];
  order[p] = (int32_t)row;
  pos_of_row[row] = (int32_t)p;
  int32_t c[NCOL];
  load_coords<NCOL>(coords, row, c);
  store_coords<NCOL>(coords_sorted, p, c);
}

struct ProbeLds {
  int32_t halo[ME_MAX_DIM];    // cells of halo on either side of a supercell
  int32_t gdim[ME_MAX_DIM];    // halo grid extents (supercell side + 2 * halo)
  int32_t grid_cells;          // product of gdim
  int32_t sc_cells;            // cells of a supercell (upper bound of its rows)
};

// One workgroup per supercell of the query map (grid-stride).  LDS: halo grid [grid_cells] | own cells
// [sc_cells] | offset deltas [volume] | neighbour ranges [2 * 3^D].
// wcount (may be NULL): per (offset, 64-position chunk) pair counts for the compaction, fused here (a wave re-reads
// its chunk's probes from LDS and ballots; chunks shared with a neighbouring supercell are added atomically —
// integer sums, order-independent); it must be zeroed before the launch.
template <int NCOL>
__global__ __launch_bounds__(256) void k_kmap_probe_lds(SpatialGrid gq, const int32_t *__restrict__ q_coords,
                                                       const uint32_t *__restrict__ q_dir, int64_t n_q,
                                                       int64_t m_q, SpatialGrid gl,
                                                       const int32_t *__restrict__ l_coords,
                                                       const int32_t *__restrict__ l_order,
                                                       const uint32_t *__restrict__ l_dir, me_region rg,
                                                       ProbeLds pl, int32_t volume, int32_t *__restrict__ nbr,
                                                       uint32_t *__restrict__ wcount, int64_t nw) {
  constexpr int D = NCOL - 1;
  extern __shared__ __attribute__((aligned(16))) int32_t s_mem[];
  int32_t *s_grid = s_mem;
  int32_t *s_qlin = s_grid + pl.grid_cells;
  int32_t *s_delta = s_qlin + pl.sc_cells;
  uint32_t *s_rng = reinterpret_cast<uint32_t *>(s_delta + volume);
  const int tid = threadIdx.x;
  // linear halo-grid step of every kernel offset (offsets of the region in cells of the lookup map)
  for (int k = tid; k < volume; k += blockDim.x) {
    int32_t zero[NCOL], off[NCOL];
#pragma unroll
    for (int d = 0; d < NCOL; ++d) zero[d] = 0;
    region_coordinate_at<NCOL>(rg, k, zero, off);
    int32_t lin = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) lin = lin * pl.gdim[d] + off[d + 1] / gl.ts[d];
    s_delta[k] = lin;
  }
  int n_nb = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) n_nb *= 3;
  for (int64_t sc = blockIdx.x; sc < m_q; sc += gridDim.x) {
    const uint32_t start = q_dir[sc], end = q_dir[sc + 1];
    const int rows = (int)(end - start);
    if (rows == 0) continue;                                   // workgroup-uniform
    // absolute supercell coordinates of this supercell and the origin cell of its halo grid
    int32_t A[D], org[D];
    int64_t rem = sc;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      A[d] = (int32_t)(rem % gq.sc_dim[d + 1]) + gq.sc_min[d + 1];
      rem /= gq.sc_dim[d + 1];
      org[d] = (A[d] << gq.shift[d]) - pl.halo[d];
    }
    const int32_t batch = (int32_t)rem + gq.sc_min[0];
    for (int x = tid; x < pl.grid_cells; x += blockDim.x) s_grid[x] = -1;
    // position ranges of the 3^D neighbouring supercells of the lookup map: fetched by 3^D threads at once (one
    // directory round trip per supercell instead of 3^D dependent ones)
    const int32_t bl = batch - gl.sc_min[0];
    for (int nb = tid; nb < n_nb; nb += blockDim.x) {
      int64_t lin_sc = bl;
      bool inside = bl >= 0 && bl < gl.sc_dim[0];
      int32_t dl[D];
      int t = nb;
#pragma unroll
      for (int d = D - 1; d >= 0; --d) {
        dl[d] = t % 3 - 1;
        t /= 3;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int32_t idx = A[d] + dl[d] - gl.sc_min[d + 1];
        inside = inside && idx >= 0 && idx < gl.sc_dim[d + 1];
        lin_sc = lin_sc * gl.sc_dim[d + 1] + idx;
      }
      s_rng[2 * nb] = inside ? l_dir[lin_sc] : 0u;
      s_rng[2 * nb + 1] = inside ? l_dir[lin_sc + 1] : 0u;
    }
    __syncthreads();
    // stage the lookup map's rows of those supercells (contiguous ranges of its sorted coordinate array).  The
    // ranges are walked as ONE concatenated list (exclusive prefix of their lengths in s_rng[2 nb + 1]): a loop over
    // the 3^D neighbours with a handful of rows each paid one memory latency per neighbour (81 in 4-D)
    if (tid == 0) {
      uint32_t run = 0;
      for (int nb = 0; nb < n_nb; ++nb) {
        const uint32_t len = s_rng[2 * nb + 1] - s_rng[2 * nb];
        s_rng[2 * nb + 1] = run;            // exclusive prefix
        run += len;
      }
      s_rng[2 * n_nb] = run;                // total rows to stage
    }
    __syncthreads();
    const uint32_t n_stage = s_rng[2 * n_nb];
    for (uint32_t j = tid; j < n_stage; j += blockDim.x) {
      int lo = 0, hi = n_nb - 1;            // last neighbour whose prefix is <= j (empty ranges share a prefix: the
      while (lo < hi) {                     // LAST of them is the non-empty one that owns j)
        const int mid = (lo + hi + 1) >> 1;
        if (s_rng[2 * mid + 1] <= j) lo = mid; else hi = mid - 1;
      }
      const uint32_t p = s_rng[2 * lo] + (j - s_rng[2 * lo + 1]);
      int32_t c[NCOL];
      load_coords<NCOL>(l_coords, p, c);
      int32_t lin = 0;
      bool ok = true;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int32_t loc = floor_div(c[d + 1], gl.ts[d]) - org[d];
        ok = ok && loc >= 0 && loc < pl.gdim[d];
        lin = lin * pl.gdim[d] + loc;
      }
      if (ok) s_grid[lin] = l_order[p];
    }
    // the halo-grid cell of every row of this supercell
    for (int i = tid; i < rows; i += blockDim.x) {
      int32_t c[NCOL];
      load_coords<NCOL>(q_coords, start + i, c);
      int32_t lin = 0;
#pragma unroll
      for (int d = 0; d < D; ++d) lin = lin * pl.gdim[d] + (floor_div(c[d + 1], gq.ts[d]) - org[d]);
      s_qlin[i] = lin;
    }
    __syncthreads();
    // (offset, 64-position chunk) items, one per wave and step: consecutive lanes = consecutive positions ->
    // coalesced stores; the ballot of the hits is the chunk's pair count
    const uint32_t c0 = start >> 6, c1 = (end - 1) >> 6;      // global 64-position chunks this supercell touches
    const int n_chunks = (int)(c1 - c0 + 1);
    const int lane = tid & 63, wave = tid >> 6;
    const int items = n_chunks * volume;
    for (int item = wave; item < items; item += (int)(blockDim.x >> 6)) {
      const int k = item / n_chunks;
      const uint32_t chunk = c0 + (uint32_t)(item - k * n_chunks);
      const uint32_t p = (chunk << 6) + (uint32_t)lane;
      int32_t r = -1;
      if (p >= start && p < end) {
        r = s_grid[s_qlin[p - start] + s_delta[k]];
        nbr[(int64_t)k * n_q + p] = r;
      }
      if (wcount != nullptr) {
        const uint32_t cnt = (uint32_t)__popcll(__ballot(r >= 0));
        if (lane == 0 && cnt != 0) {
          uint32_t *w = &wcount[(int64_t)k * nw + chunk];
          if ((chunk << 6) >= start && (chunk << 6) + 64 <= end) *w = cnt;   // chunk owned by this supercell alone
          else atomicAdd(w, cnt);
        }
      }
    }
    __syncthreads();
  }
}

// per-offset pair counts of an existing neighbour table (the counting half of k_kmap_probe)
__global__ __launch_bounds__(256) void k_kmap_count(const int32_t *__restrict__ nbr, int64_t n_out,
                                                   uint32_t *__restrict__ wcount, int64_t nw) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = blockIdx.y;
  const int32_t r = (u < n_out) ? nbr[(int64_t)k * n_out + u] : -1;
  const unsigned long long m = __ballot(r >= 0);
  const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (lane_id() == 0 && wave_global < nw) wcount[(int64_t)k * nw + wave_global] = (uint32_t)__popcll(m);
}

// compaction of a POSITION-space table: the pair of (offset k, position p) names the target row order[p]
__global__ __launch_bounds__(256) void k_kmap_compact_ordered(const int32_t *__restrict__ nbr,
                                                             const int32_t *__restrict__ order, int64_t n_out,
                                                             const uint32_t *__restrict__ woffs, int64_t nw,
                                                             int32_t *__restrict__ in_pairs,
                                                             int32_t *__restrict__ out_pairs) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = blockIdx.y;
  const int32_t r = (u < n_out) ? nbr[(int64_t)k * n_out + u] : -1;
  const unsigned long long m = __ballot(r >= 0);
  const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 0) {
    const uint32_t dst = woffs[(int64_t)k * nw + wave_global] + mask_prefix(m);
    in_pairs[dst] = r;
    out_pairs[dst] = order[u];
  }
}

__global__ __launch_bounds__(256) void k_kmap_transpose_ordered(const int32_t *__restrict__ in_pairs,
                                                               const int32_t *__restrict__ out_pairs,
                                                               const int64_t *__restrict__ koffs,
                                                               const int32_t *__restrict__ pos_in, int64_t n_in,
                                                               int32_t *__restrict__ nbrT) {
  const int32_t k = blockIdx.y;
  const int64_t e0 = koffs[k], e1 = koffs[k + 1];
  for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1;
       e += (int64_t)gridDim.x * blockDim.x)
    nbrT[(int64_t)k * n_in + pos_in[in_pairs[e]]] = out_pairs[e];
}

template <int NCOL>
__global__ __launch_bounds__(256) void k_bbox(const int32_t *__restrict__ coords, int64_t n,
                                             int32_t *__restrict__ bbox /* [2 * NCOL]: mins, then maxs */) {
  __shared__ int32_t s_lo[4][NCOL], s_hi[4][NCOL];
  int32_t lo[NCOL], hi[NCOL];
#pragma unroll
  for (int d = 0; d < NCOL; ++d) {
    lo[d] = INT32_MAX;
    hi[d] = INT32_MIN;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t c[NCOL];
    load_coords<NCOL>(coords, i, c);
#pragma unroll
    for (int d = 0; d < NCOL; ++d) {
      lo[d] = min(lo[d], c[d]);
      hi[d] = max(hi[d], c[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < NCOL; ++d) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = min(lo[d], __shfl_xor(lo[d], off, 64));
      hi[d] = max(hi[d], __shfl_xor(hi[d], off, 64));
    }
    if (lane_id() == 0) {
      s_lo[threadIdx.x >> 6][d] = lo[d];
      s_hi[threadIdx.x >> 6][d] = hi[d];
    }
  }
  __syncthreads();
  // one atomic per column and block (a few dozen blocks: the atomics of every wave serialised on 2 * NCOL
  // addresses and took 22 us for 100k rows)
  if (threadIdx.x < NCOL) {
    const int d = threadIdx.x;
    atomicMin(&bbox[d], min(min(s_lo[0][d], s_lo[1][d]), min(s_lo[2][d], s_lo[3][d])));
    atomicMax(&bbox[NCOL + d], max(max(s_hi[0][d], s_hi[1][d]), max(s_hi[2][d], s_hi[3][d])));
  }
}

}  // namespace me

// =================================================================================================
// C ABI
// =================================================================================================
using namespace me;

extern "C" {

int me_version(void) { return 160; }   // 100 * major + 10 * minor: see the changelog in include/me_amd.h
const char *me_last_error(void) { return g_last_error; }

int64_t me_region_volume(const me_region *rg) {
  if (!rg || rg->ncol < 2 || rg->ncol > ME_MAX_DIM + 1) return -1;
  int64_t v = 1;
  if (rg->region_type == ME_REGION_HYPER_CUBE) {
    for (int d = 0; d < rg->ncol - 1; ++d) v *= rg->kernel_size[d];
  } else if (rg->region_type == ME_REGION_HYPER_CROSS) {
    for (int d = 0; d < rg->ncol - 1; ++d) v += rg->kernel_size[d] - 1;
  } else {
    return -1;
  }
  return v;
}

int64_t me_hash_capacity(int64_t n) {
  int64_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static int64_t rs_blocks_fwd(int64_t n) { return ceil_div(n < 1 ? 1 : n, kRsTile); }   // blocks of the radix-sort passes

// workspace layout of insert: slot_of_row[n] | wrow[n] | flag/newid[n] | total | scan ws
static int64_t insert_ws_arrays(int64_t n) { return align_up(n * 4, 256) + 1024; }
int g_insert_fused = 1;   // me_debug_set_insert_fused: 0 = the resolve / scan / finalize / bbox pipeline of rounds 1 - 5
int64_t me_insert_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return 3 * insert_ws_arrays(n) + 256 + scan_workspace_bytes(n);
}

int me_coords_insert_and_map_bbox(const int32_t *coords, int64_t n, int32_t ncol, uint64_t *table,
                                  int64_t capacity, int32_t *coords_unique, int64_t *unique_map,
                                  int64_t *inverse_map, int64_t *n_unique, int32_t *bbox, void *workspace,
                                  int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(n >= 0 && n < (1ll << 31), "number of coordinates must fit in int32");
  ME_CHECK(capacity >= 2 * n && capacity >= 64 && (capacity & (capacity - 1)) == 0 &&
               capacity <= (1ll << 32),
           "capacity must be a power of two >= 2n");
  ME_CHECK(n_unique != nullptr, "n_unique must not be null");
  ME_CHECK(workspace_bytes >= me_insert_workspace_bytes(n), "workspace too small");
  ME_CHECK(ncol != 4 || ((uintptr_t)coords % 16 == 0 && (uintptr_t)coords_unique % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  ME_CHECK(ncol != 2 || ((uintptr_t)coords % 8 == 0 && (uintptr_t)coords_unique % 8 == 0),
           "coordinates with 2 columns must be 8-byte aligned");
  ME_HIP(hipMemsetAsync(table, 0xff, (size_t)capacity * 8, stream));
  if (n == 0) {
    *n_unique = 0;
    if (bbox)
      for (int d = 0; d < 2 * ncol; ++d) bbox[d] = 0;
    return 0;
  }
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t asz = insert_ws_arrays(n);
  uint32_t *slot_of_row = reinterpret_cast<uint32_t *>(ws);
  uint32_t *wrow = reinterpret_cast<uint32_t *>(ws + asz);
  uint32_t *flag = reinterpret_cast<uint32_t *>(ws + 2 * asz);
  uint32_t *total = reinterpret_cast<uint32_t *>(ws + 3 * asz);
  void *scan_ws = ws + 3 * asz + 256;
  const uint32_t mask = (uint32_t)(capacity - 1);
  const dim3 grid((unsigned)ceil_div(n, 256)), block(256);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_insert<NCOL>, grid, block, 0, stream, coords, n, table,
                                            mask, slot_of_row));
  ME_LAUNCH_CHECK();
  uint32_t host_buf[1 + 2 * (ME_MAX_DIM + 1)];
  const int64_t nb = ceil_div(n, (int64_t)kInsBlockRows);
  if (nb <= kInsMaxBlocks && g_insert_fused) {
    // fused resolve / rank / emit (see k_insert_flags): the `flag` array holds the ballots, the group prefixes and the
    // per-block counts / bounding boxes instead
    const int64_t groups = ceil_div(n, (int64_t)64);
    char *fb = reinterpret_cast<char *>(flag);
    uint64_t *flagbits = reinterpret_cast<uint64_t *>(fb);
    uint32_t *group_prefix = reinterpret_cast<uint32_t *>(fb + groups * 8);
    uint32_t *blk_count = group_prefix + groups;
    int32_t *blk_bbox = reinterpret_cast<int32_t *>(blk_count + nb);
    const dim3 bgrid((unsigned)nb);
    ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_insert_flags<NCOL>, bgrid, block, 0, stream, coords, table, slot_of_row, n,
                                              wrow, flagbits, group_prefix, blk_count, blk_bbox));
    ME_LAUNCH_CHECK();
    ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_insert_emit<NCOL>, grid, block, 0, stream, coords, n, table, slot_of_row,
                                              wrow, flagbits, group_prefix, blk_count, blk_bbox, (int)nb, coords_unique,
                                              unique_map, inverse_map, total));
    ME_LAUNCH_CHECK();
  } else {
  hipLaunchKernelGGL(k_insert_resolve, grid, block, 0, stream, table, slot_of_row, n, wrow, flag);
  ME_LAUNCH_CHECK();
  if (int rc = exclusive_scan_u32(flag, flag, n, total, scan_ws, scan_workspace_bytes(n), stream)) return rc;
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_insert_finalize<NCOL>, grid, block, 0, stream, coords, n,
                                            table, slot_of_row, wrow, flag, coords_unique,
                                            unique_map, inverse_map));
  ME_LAUNCH_CHECK();
  // the bounding box of the coordinates rides on the one read-back this function needs anyway: it sizes the dense
  // supercell directory of the map's spatial index (me_spatial_index_build) without a synchronisation of its own
  int32_t *bbox_dev = reinterpret_cast<int32_t *>(total + 1);   // (inside the 256-byte `total` slot)
  int32_t init[2 * (ME_MAX_DIM + 1)];   // source of an asynchronous copy: must live until the synchronisation below
  if (bbox) {
    for (int d = 0; d < ncol; ++d) {
      init[d] = INT32_MAX;
      init[ncol + d] = INT32_MIN;
    }
    ME_HIP(hipMemcpyAsync(bbox_dev, init, (size_t)2 * ncol * 4, hipMemcpyHostToDevice, stream));
    int64_t gb = ceil_div(n, 256 * 16);
    if (gb > 48) gb = 48;
    ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_bbox<NCOL>, dim3((unsigned)gb), dim3(256), 0, stream, coords, n,
                                              bbox_dev));
    ME_LAUNCH_CHECK();
  }
  ME_HIP(hipMemcpyAsync(host_buf, total, (size_t)(1 + (bbox ? 2 * ncol : 0)) * 4, hipMemcpyDeviceToHost, stream));
  ME_HIP(hipStreamSynchronize(stream));
  *n_unique = (int64_t)host_buf[0];
  if (bbox)
    for (int d = 0; d < 2 * ncol; ++d) bbox[d] = (int32_t)host_buf[1 + d];
  return 0;
  }
  {
    // the read-back lands in PINNED host memory (one small buffer per host thread, allocated once): a copy into pageable
    // stack memory goes through the runtime's staging path and made the synchronisation below ~10 us longer
    static thread_local uint32_t *pinned = nullptr;
    if (pinned == nullptr && hipHostMalloc(reinterpret_cast<void **>(&pinned), 256, hipHostMallocDefault) != hipSuccess) {
      pinned = nullptr;
      (void)hipGetLastError();
    }
    uint32_t *dst = pinned != nullptr ? pinned : host_buf;
    ME_HIP(hipMemcpyAsync(dst, total, (size_t)(1 + (bbox ? 2 * ncol : 0)) * 4, hipMemcpyDeviceToHost, stream));
    ME_HIP(hipStreamSynchronize(stream));
    *n_unique = (int64_t)dst[0];
    if (bbox)
      for (int d = 0; d < 2 * ncol; ++d) bbox[d] = (int32_t)dst[1 + d];
  }
  return 0;
}

void me_debug_set_insert_fused(int on) { g_insert_fused = on ? 1 : 0; }

int me_coords_insert_and_map(const int32_t *coords, int64_t n, int32_t ncol, uint64_t *table,
                             int64_t capacity, int32_t *coords_unique, int64_t *unique_map,
                             int64_t *inverse_map, int64_t *n_unique, void *workspace,
                             int64_t workspace_bytes, void *stream_) {
  return me_coords_insert_and_map_bbox(coords, n, ncol, table, capacity, coords_unique, unique_map, inverse_map,
                                       n_unique, nullptr, workspace, workspace_bytes, stream_);
}


int me_coords_stride(const int32_t *coords, int64_t n, int32_t ncol, const int32_t *out_ts,
                     int32_t *out_coords, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(ncol >= 2 && ncol <= ME_MAX_DIM + 1, "invalid coordinate size");
  StrideArg arg;
  for (int d = 0; d < ME_MAX_DIM; ++d) arg.ts[d] = 1;
  for (int d = 0; d < ncol - 1; ++d) {
    ME_CHECK(out_ts[d] > 0, "tensor stride must be positive");
    arg.ts[d] = out_ts[d];
  }
  if (n == 0) return 0;
  ME_CHECK(ncol != 4 || ((uintptr_t)coords % 16 == 0 && (uintptr_t)out_coords % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  const dim3 grid((unsigned)ceil_div(n, 256)), block(256);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_stride<NCOL>, grid, block, 0, stream, coords, n, arg, out_coords));
  ME_LAUNCH_CHECK();
  return 0;
}

int me_coords_expand_region(const int32_t *coords, int64_t n, int32_t ncol, const me_region *region,
                            const int32_t *align_stride, int32_t *out_coords, uint8_t *aligned, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(region != nullptr && region->ncol == ncol, "region / coordinate size mismatch");
  ME_CHECK(ncol >= 2 && ncol <= ME_MAX_DIM + 1, "invalid coordinate size");
  const int64_t volume = me_region_volume(region);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  if (n == 0) return 0;
  ME_CHECK(ncol != 4 || ((uintptr_t)coords % 16 == 0 && (uintptr_t)out_coords % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  StrideArg arg;
  for (int d = 0; d < ME_MAX_DIM; ++d) arg.ts[d] = 1;
  if (align_stride != nullptr)
    for (int d = 0; d < ncol - 1; ++d) {
      ME_CHECK(align_stride[d] > 0, "alignment stride must be positive");
      arg.ts[d] = align_stride[d];
    }
  const me_region rg = *region;
  const dim3 grid((unsigned)ceil_div(n * volume, 256)), block(256);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_expand_region<NCOL>, grid, block, 0, stream, coords, n, rg,
                                            (int32_t)volume, arg, out_coords,
                                            align_stride != nullptr ? aligned : nullptr));
  ME_LAUNCH_CHECK();
  return 0;
}

int me_coords_quantize_labels(const int64_t *unique_map, int64_t n_unique, const int64_t *inverse_map,
                              const int32_t *labels, int64_t n, int32_t ignore_label, int32_t *colabels,
                              void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_unique == 0 || n == 0) return 0;
  hipLaunchKernelGGL(k_quantize_labels_init, dim3((unsigned)ceil_div(n_unique, 256)), dim3(256), 0, stream,
                     unique_map, n_unique, labels, colabels);
  hipLaunchKernelGGL(k_quantize_labels_mark, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, unique_map,
                     inverse_map, labels, n, ignore_label, colabels);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_coords_spatial_keys(const int32_t *coords, int64_t n, int32_t ncol, const int32_t *tensor_stride,
                           int64_t *keys, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(ncol >= 2 && ncol <= ME_MAX_DIM + 1, "invalid coordinate size");
  StrideArg arg;
  for (int d = 0; d < ME_MAX_DIM; ++d) arg.ts[d] = 1;
  for (int d = 0; d < ncol - 1; ++d) {
    ME_CHECK(tensor_stride[d] > 0, "tensor stride must be positive");
    arg.ts[d] = tensor_stride[d];
  }
  if (n == 0) return 0;
  ME_CHECK(ncol != 4 || (uintptr_t)coords % 16 == 0, "coordinates with 4 columns must be 16-byte aligned");
  int bits = 56 / (ncol - 1);
  if (bits > 21) bits = 21;
  const dim3 grid((unsigned)ceil_div(n, 256)), block(256);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_spatial_keys<NCOL>, grid, block, 0, stream, coords, n, arg, bits, keys));
  ME_LAUNCH_CHECK();
  return 0;
}

// Rows of a coordinate map in Z-order: the STABLE argsort of me_coords_spatial_keys' keys, by the library's own LSD radix
// sort (round 6: the hosts called at::argsort — rocPRIM inside torch — here, on the plan build of the headline layer
// since its tiles are Morton-ordered; the reference uses thrust on its map path, src/coordinate_map_gpu.cu:766-772).
// Only the key bytes that can differ are sorted: with the bounding box of the rows (host ints, column minima then maxima,
// as me_coords_insert_and_map_bbox returns them; NULL: unknown) the varying bits of every axis are known — an axis
// whose biased values lo .. hi differ first at bit h varies in bits [0, h] — so a 70^3 scene takes three 8-bit passes
// over the interleaved coordinate bits (+ one over the batch byte when there are several scenes) instead of eight.
int64_t me_coords_zorder_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return 2 * align_up(n * 8, 256) + 2 * align_up(n * 4, 256) + align_up(256 * rs_blocks_fwd(n) * 4, 256) + 256 +
         scan_workspace_bytes(256 * rs_blocks_fwd(n));
}

int me_coords_zorder(const int32_t *coords, int64_t n, int32_t ncol, const int32_t *tensor_stride, const int32_t *bbox,
                     int32_t *order, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(ncol >= 2 && ncol <= ME_MAX_DIM + 1, "invalid coordinate size");
  ME_CHECK(n >= 0 && n < (1ll << 31), "number of rows must fit in int32");
  if (n == 0) return 0;
  ME_CHECK(order != nullptr && workspace != nullptr && workspace_bytes >= me_coords_zorder_workspace_bytes(n),
           "order / workspace");
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t ksz = align_up(n * 8, 256), vsz = align_up(n * 4, 256);
  uint64_t *keys[2] = {reinterpret_cast<uint64_t *>(ws), reinterpret_cast<uint64_t *>(ws + ksz)};
  uint32_t *vbuf = reinterpret_cast<uint32_t *>(ws + 2 * ksz);            // the one value buffer besides `order`
  const int64_t nblocks = rs_blocks_fwd(n);
  uint32_t *hist = reinterpret_cast<uint32_t *>(ws + 2 * ksz + 2 * vsz);
  uint32_t *hist_total = reinterpret_cast<uint32_t *>(ws + 2 * ksz + 2 * vsz + align_up(256 * nblocks * 4, 256));
  void *scan_ws = reinterpret_cast<char *>(hist_total) + 256;
  if (int rc = me_coords_spatial_keys(coords, n, ncol, tensor_stride, reinterpret_cast<int64_t *>(keys[0]), stream_)) return rc;
  // key bytes to sort (me_coords_spatial_keys: bit b of axis d at position b * D + d, the batch index from bit 56)
  const int D = ncol - 1;
  int kbits = 56 / D;
  if (kbits > 21) kbits = 21;
  int span = kbits * D;                       // interleaved coordinate bits that may vary
  bool batch_varies = true;
  if (bbox != nullptr) {
    const uint32_t bias = 1u << (kbits - 1), mask = (1u << kbits) - 1u;
    int top = 0;
    for (int d = 0; d < D; ++d) {
      const int32_t ts = tensor_stride[d];
      const int64_t lo = (int64_t)floor((double)bbox[1 + d] / ts), hi = (int64_t)floor((double)bbox[ncol + 1 + d] / ts);
      if (hi - lo >= (int64_t)mask) { top = kbits; continue; }          // wraps around the key's range: every bit
      const uint32_t a = ((uint32_t)lo + bias) & mask, b = ((uint32_t)hi + bias) & mask;
      // values lo .. hi (consecutive integers, biased, no wrap when a <= b) agree above the highest bit where a and b differ
      uint32_t x = a <= b ? (a ^ b) : mask;
      int h = 0;
      while (x) { ++h; x >>= 1; }
      if (h > top) top = h;
    }
    span = top * D;
    batch_varies = bbox[0] != bbox[ncol];
  }
  int shifts[9], np = 0;
  for (int sft = 0; sft < span; sft += 8) shifts[np++] = sft;
  if (batch_varies) shifts[np++] = 56;
  if (np == 0) shifts[np++] = 0;              // (one row / all keys equal: one pass produces the identity order)
  // the values ping-pong between `order` and vbuf so that the LAST pass lands in `order`
  uint32_t *ord = reinterpret_cast<uint32_t *>(order);
  int cur = 0;
  for (int p = 0; p < np; ++p) {
    uint32_t *v_out = ((np - 1 - p) % 2 == 0) ? ord : vbuf;
    const uint32_t *v_in = p == 0 ? nullptr : (v_out == ord ? vbuf : ord);
    hipLaunchKernelGGL(k_rs_hist<uint64_t>, dim3((unsigned)nblocks), dim3(256), 0, stream, keys[cur], n, shifts[p], nblocks, hist);
    ME_LAUNCH_CHECK();
    if (int rc = exclusive_scan_u32(hist, hist, 256 * nblocks, hist_total, scan_ws, scan_workspace_bytes(256 * nblocks), stream))
      return rc;
    hipLaunchKernelGGL(k_rs_scatter<uint64_t>, dim3((unsigned)nblocks), dim3(64), 0, stream, keys[cur], v_in, n, shifts[p],
                       nblocks, hist, keys[cur ^ 1], v_out);
    ME_LAUNCH_CHECK();
    cur ^= 1;
  }
  return 0;
}

int me_coords_find(const uint64_t *table, int64_t capacity, const int32_t *map_coords, int32_t ncol,
                   const int32_t *queries, int64_t nq, int32_t *rows, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be a power of two");
  if (nq == 0) return 0;
  ME_CHECK(ncol != 4 || ((uintptr_t)map_coords % 16 == 0 && (uintptr_t)queries % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  const dim3 grid((unsigned)ceil_div(nq, 256)), block(256);
  const uint32_t mask = (uint32_t)(capacity - 1);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_find<NCOL>, grid, block, 0, stream, table, mask, map_coords,
                                            queries, nq, rows));
  ME_LAUNCH_CHECK();
  return 0;
}

// workspace layout of the kernel map: wcount/woffs [volume * nw] | total | koffs[volume+1] | scan ws
static int64_t kmap_nw(int64_t n_out) { return ceil_div(n_out, 256) * 4; }
static int64_t kmap_counts_bytes(int64_t n_out, int64_t volume) {
  return align_up(volume * kmap_nw(n_out) * 4, 256);
}
int64_t me_kernel_map_workspace_bytes(int64_t n_out, int64_t volume) {
  if (n_out < 1) n_out = 1;
  return kmap_counts_bytes(n_out, volume) + 256 + align_up((volume + 1) * 8, 256) +
         scan_workspace_bytes(volume * kmap_nw(n_out));
}

int me_kernel_map_probe(const uint64_t *in_table, int64_t in_capacity, const int32_t *in_coords,
                        const int32_t *out_coords, int64_t n_out, const me_region *region,
                        int32_t *nbr, int64_t *k_offsets, int64_t *k_offsets_dev, void *workspace,
                        int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(region != nullptr, "region must not be null");
  const int64_t volume = me_region_volume(region);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(in_capacity >= 64 && (in_capacity & (in_capacity - 1)) == 0, "capacity must be a power of two");
  ME_CHECK(n_out >= 0 && n_out * volume < (1ll << 32), "n_out * volume must fit in 32 bits");
  for (int d = 0; d < region->ncol - 1; ++d)
    ME_CHECK(region->kernel_size[d] > 0 && region->dilation[d] > 0 && region->tensor_stride[d] > 0,
             "kernel size, dilation and tensor stride must be positive");
  if (region->region_type == ME_REGION_HYPER_CROSS)
    for (int d = 0; d < region->ncol - 1; ++d)
      ME_CHECK(region->kernel_size[d] % 2 == 1, "HYPER_CROSS needs odd kernel sizes");
  ME_CHECK(workspace_bytes >= me_kernel_map_workspace_bytes(n_out, volume), "workspace too small");
  ME_CHECK(k_offsets != nullptr || k_offsets_dev != nullptr, "k_offsets (host) or k_offsets_dev must be given");
  if (n_out == 0) {
    if (k_offsets)
      for (int64_t k = 0; k <= volume; ++k) k_offsets[k] = 0;
    if (k_offsets_dev) ME_HIP(hipMemsetAsync(k_offsets_dev, 0, (size_t)(volume + 1) * 8, stream));
    return 0;
  }
  const int ncol = region->ncol;
  ME_CHECK(ncol != 4 || ((uintptr_t)in_coords % 16 == 0 && (uintptr_t)out_coords % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t nw = kmap_nw(n_out);
  uint32_t *wcount = reinterpret_cast<uint32_t *>(ws);
  uint32_t *total = reinterpret_cast<uint32_t *>(ws + kmap_counts_bytes(n_out, volume));
  int64_t *koffs = k_offsets_dev ? k_offsets_dev
                                 : reinterpret_cast<int64_t *>(ws + kmap_counts_bytes(n_out, volume) + 256);
  void *scan_ws = ws + kmap_counts_bytes(n_out, volume) + 256 + align_up((volume + 1) * 8, 256);
  const dim3 grid((unsigned)ceil_div(n_out, 256), (unsigned)volume), block(256);
  const uint32_t mask = (uint32_t)(in_capacity - 1);
  const me_region rg = *region;
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_kmap_probe<NCOL>, grid, block, 0, stream, in_table, mask,
                                            in_coords, out_coords, n_out, rg, nbr, wcount, nw));
  ME_LAUNCH_CHECK();
  if (int rc = exclusive_scan_u32(wcount, wcount, volume * nw, total, scan_ws,
                                  scan_workspace_bytes(volume * nw), stream))
    return rc;
  hipLaunchKernelGGL(k_kmap_koffsets, dim3((unsigned)ceil_div(volume + 1, 256)), dim3(256), 0, stream,
                     wcount, total, nw, volume, koffs);
  ME_LAUNCH_CHECK();
  if (k_offsets != nullptr) {   // (NULL: no read-back, no synchronisation — the caller sizes the pair lists at their bound)
    ME_HIP(hipMemcpyAsync(k_offsets, koffs, (size_t)(volume + 1) * 8, hipMemcpyDeviceToHost, stream));
    ME_HIP(hipStreamSynchronize(stream));
  }
  return 0;
}

int me_kernel_map_compact(const int32_t *nbr, int64_t n_out, int64_t volume, int32_t *in_pairs,
                          int32_t *out_pairs, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(workspace_bytes >= me_kernel_map_workspace_bytes(n_out, volume), "workspace too small");
  if (n_out == 0) return 0;
  const int64_t nw = kmap_nw(n_out);
  const uint32_t *woffs = reinterpret_cast<const uint32_t *>(workspace);
  const dim3 grid((unsigned)ceil_div(n_out, 256), (unsigned)volume), block(256);
  hipLaunchKernelGGL(k_kmap_compact, grid, block, 0, stream, nbr, n_out, woffs, nw, in_pairs, out_pairs);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_kernel_map_transpose(const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                            int64_t volume, int64_t n_pairs, int64_t n_in, int32_t *nbrT, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  if (n_in == 0) return 0;
  ME_HIP(hipMemsetAsync(nbrT, 0xff, (size_t)volume * n_in * 4, stream));
  if (n_pairs == 0) return 0;
  int64_t bx = ceil_div(ceil_div(n_pairs, volume), 256);
  if (bx < 1) bx = 1;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(k_kmap_transpose, dim3((unsigned)bx, (unsigned)volume), dim3(256), 0, stream, in_pairs,
                     out_pairs, k_offsets_dev, n_in, nbrT);
  ME_LAUNCH_CHECK();
  return 0;
}

void me_debug_set_tile_dispatch(int mode) { g_tile_dispatch = mode; }

int64_t me_plan_num_tiles(int64_t n_tgt, int32_t tile_rows) {
  return tile_rows > 0 ? ceil_div(n_tgt, tile_rows) : -1;
}
int64_t me_plan_tile_bptr_elems(int64_t n_tgt, int32_t tile_rows) {
  const int64_t t = me_plan_num_tiles(n_tgt, tile_rows);
  return t < 0 ? -1 : 2 * t + 1;   // batch ranges [num_tiles + 1] + dispatch order [num_tiles]
}
int64_t me_plan_max_groups(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t tile_rows) {
  const int64_t items = me_plan_num_tiles(n_tgt, tile_rows) * volume;
  const int64_t nonempty = items < n_pairs ? items : n_pairs;
  // (+ 4 groups: the zero-filled over-read window behind the last batch, see k_plan_fill)
  return n_pairs / ME_GROUP_ROWS + nonempty + 1 + ME_MAX_BATCH_GROUPS;
}
// plan workspace: gcount | goffs | bcount | boffs [items each] | totals | scan ws
int64_t me_plan_workspace_bytes(int64_t n_tgt, int64_t volume, int32_t tile_rows) {
  const int64_t items = me_plan_num_tiles(n_tgt < 1 ? 1 : n_tgt, tile_rows) * volume;
  return 2 * align_up(items * 8, 256) + 256 + scan_workspace_bytes(items);
}

static void plan_dispatch_env() {
  static const bool env_read = [] {   // ME_AMD_TILE_DISPATCH=1: XCD chunks from the first plan on (tuning runs)
    const char *e = getenv("ME_AMD_TILE_DISPATCH");
    if (e != nullptr && *e != 0) g_tile_dispatch = atoi(e);
    return true;
  }();
  (void)env_read;
}

int me_plan_build(const int32_t *tbl, const int32_t *order, int64_t n_tgt, int64_t volume, int32_t tile_rows,
                  int32_t batch_groups, int32_t *plan_src, int32_t *plan_dst, int32_t *batch_desc, int32_t *tile_bptr,
                  int32_t *item_gptr, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(tile_rows >= ME_GROUP_ROWS && tile_rows <= ME_MAX_TILE_ROWS, "tile_rows out of range");
  ME_CHECK(batch_groups >= 1 && batch_groups <= ME_MAX_BATCH_GROUPS, "batch_groups out of range");
  ME_CHECK(workspace_bytes >= me_plan_workspace_bytes(n_tgt, volume, tile_rows), "workspace too small");
  const int64_t n_tiles = me_plan_num_tiles(n_tgt, tile_rows);
  if (n_tiles == 0) {
    ME_HIP(hipMemsetAsync(tile_bptr, 0, 4, stream));
    ME_HIP(hipMemsetAsync(item_gptr, 0, 4, stream));
    return 0;
  }
  const int64_t items = n_tiles * volume;
  ME_CHECK(items < (1ll << 31), "too many (tile, offset) items");
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t asz = align_up(items * 8, 256);
  uint64_t *count_gb = reinterpret_cast<uint64_t *>(ws);
  uint64_t *offs_gb = reinterpret_cast<uint64_t *>(ws + asz);
  uint64_t *total_gb = reinterpret_cast<uint64_t *>(ws + 2 * asz);
  void *scan_ws = ws + 2 * asz + 256;
  const dim3 grid((unsigned)ceil_div(items, 4)), block(256);
  hipLaunchKernelGGL(k_plan_count, grid, block, 0, stream, tbl, order, n_tgt, volume, items, (int)tile_rows,
                     (int)batch_groups, count_gb);
  ME_LAUNCH_CHECK();
  if (int rc = exclusive_scan<uint64_t>(count_gb, offs_gb, items, total_gb, scan_ws, scan_workspace_bytes(items),
                                        stream))
    return rc;
  hipLaunchKernelGGL(k_plan_fill, grid, block, 0, stream, tbl, order, n_tgt, volume, items, (int)tile_rows,
                     (int)batch_groups, offs_gb, total_gb, plan_src, plan_dst, batch_desc, tile_bptr, item_gptr,
                     n_tiles);
  ME_LAUNCH_CHECK();
  plan_dispatch_env();
  if (g_tile_dispatch == 1)
    hipLaunchKernelGGL(k_plan_tile_order<8>, dim3(1), dim3(1024), 0, stream, item_gptr, volume, n_tiles,
                       tile_bptr + n_tiles + 1);
  else
    hipLaunchKernelGGL(k_plan_tile_order<0>, dim3(1), dim3(1024), 0, stream, item_gptr, volume, n_tiles,
                       tile_bptr + n_tiles + 1);
  ME_LAUNCH_CHECK();
  return 0;
}

int64_t me_plan_jobs_init(me_plan_job *jobs, int32_t n_jobs) {
  if (jobs == nullptr || n_jobs < 0) return -1;
  int64_t base = 0;
  for (int32_t i = 0; i < n_jobs; ++i) {
    me_plan_job &j = jobs[i];
    if (j.volume < 1 || j.volume > 65535 || j.tile_rows < ME_GROUP_ROWS || j.tile_rows > ME_MAX_TILE_ROWS ||
        j.batch_groups < 1 || j.batch_groups > ME_MAX_BATCH_GROUPS || j.n_tgt < 0)
      return -1;
    j.n_tiles = me_plan_num_tiles(j.n_tgt, j.tile_rows);
    j.n_items = j.n_tiles * j.volume;
    j.item_base = base;
    base += j.n_items;
  }
  return base;
}

int64_t me_plan_multi_workspace_bytes(int64_t total_items) { return 2 * align_up((total_items < 1 ? 1 : total_items) * 8, 256); }

int me_plan_build_multi(const me_plan_job *jobs_host, const me_plan_job *jobs_dev, int32_t n_jobs, void *workspace,
                        int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_jobs == 0) return 0;
  ME_CHECK(jobs_host != nullptr && jobs_dev != nullptr && n_jobs > 0, "job tables must not be null");
  const int64_t total = jobs_host[n_jobs - 1].item_base + jobs_host[n_jobs - 1].n_items;
  for (int32_t i = 0; i < n_jobs; ++i) {
    const me_plan_job &j = jobs_host[i];
    ME_CHECK(j.n_tiles == me_plan_num_tiles(j.n_tgt, j.tile_rows) && j.n_items == j.n_tiles * j.volume &&
                 j.item_base == (i ? jobs_host[i - 1].item_base + jobs_host[i - 1].n_items : 0),
             "job table not initialised (me_plan_jobs_init)");
    ME_CHECK(j.tile_bptr != nullptr && j.item_gptr != nullptr, "plan arrays must not be null");
    ME_CHECK(j.n_items == 0 || (j.tbl != nullptr && j.plan_src != nullptr && j.plan_dst != nullptr && j.batch_desc != nullptr),
             "plan arrays must not be null");
  }
  ME_CHECK(total < (1ll << 31), "too many (tile, offset) items");
  ME_CHECK(workspace_bytes >= me_plan_multi_workspace_bytes(total), "workspace too small");
  uint64_t *count_gb = reinterpret_cast<uint64_t *>(workspace);
  uint64_t *offs_gb = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(workspace) + align_up((total < 1 ? 1 : total) * 8, 256));
  if (total > 0) {
    hipLaunchKernelGGL(k_plan_count_multi, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, stream, jobs_dev, (int)n_jobs,
                       total, count_gb);
    ME_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_plan_scan_multi, dim3((unsigned)n_jobs), dim3(kScanSingleThreads), 0, stream, jobs_dev, count_gb,
                     offs_gb);
  ME_LAUNCH_CHECK();
  if (total > 0) {
    hipLaunchKernelGGL(k_plan_fill_multi, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, stream, jobs_dev, (int)n_jobs,
                       total, offs_gb);
    ME_LAUNCH_CHECK();
    plan_dispatch_env();
    if (g_tile_dispatch == 1) hipLaunchKernelGGL(k_plan_tile_order_multi<8>, dim3((unsigned)n_jobs), dim3(1024), 0, stream, jobs_dev);
    else hipLaunchKernelGGL(k_plan_tile_order_multi<0>, dim3((unsigned)n_jobs), dim3(1024), 0, stream, jobs_dev);
    ME_LAUNCH_CHECK();
  }
  return 0;
}


// ---- spatial index + LDS-bucketed kernel map -----------------------------------------------------------------
static SpatialGrid to_device_grid(const me_spatial_grid *g) {
  SpatialGrid d;
  for (int i = 0; i < ME_MAX_DIM; ++i) {
    d.shift[i] = g->shift[i];
    d.ts[i] = g->tensor_stride[i] > 0 ? g->tensor_stride[i] : 1;
  }
  for (int i = 0; i <= ME_MAX_DIM; ++i) {
    d.sc_min[i] = g->sc_min[i];
    d.sc_dim[i] = g->sc_dim[i] > 0 ? g->sc_dim[i] : 1;
  }
  return d;
}

int64_t me_spatial_cells(const me_spatial_grid *g) {
  if (!g || g->ncol < 2 || g->ncol > ME_MAX_DIM + 1) return -1;
  int64_t m = 1;
  for (int d = 0; d < g->ncol; ++d) {
    if (g->sc_dim[d] <= 0) return -1;
    m *= g->sc_dim[d];
    if (m > (1ll << 31)) return -1;
  }
  return m;
}

static int64_t rs_blocks(int64_t n) { return rs_blocks_fwd(n); }

int64_t me_spatial_index_workspace_bytes(int64_t n, int64_t m) {
  if (n < 1) n = 1;
  if (m < 1) m = 1;
  const int64_t scan_items = 256 * rs_blocks(n) > m ? 256 * rs_blocks(n) : m;
  return 4 * align_up(n * 4, 256) + align_up(256 * rs_blocks(n) * 4, 256) + 256 + scan_workspace_bytes(scan_items);
}

int me_spatial_index_build(const int32_t *coords, int64_t n, const me_spatial_grid *grid, int32_t *order,
                           int32_t *pos_of_row, int32_t *coords_sorted, uint32_t *dir_start, void *workspace,
                           int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t m = me_spatial_cells(grid);
  ME_CHECK(m >= 1 && m <= (1ll << 30), "supercell directory out of range");
  ME_CHECK(n >= 0 && n < (1ll << 31), "number of rows must fit in int32");
  ME_CHECK(workspace_bytes >= me_spatial_index_workspace_bytes(n, m), "workspace too small");
  const int ncol = grid->ncol;
  ME_CHECK(ncol != 4 || ((uintptr_t)coords % 16 == 0 && (uintptr_t)coords_sorted % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  if (n == 0) {
    ME_HIP(hipMemsetAsync(dir_start, 0, (size_t)(m + 1) * 4, stream));
    return 0;
  }
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t asz = align_up(n * 4, 256);
  uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(ws), reinterpret_cast<uint32_t *>(ws + asz)};
  uint32_t *vals[2] = {reinterpret_cast<uint32_t *>(ws + 2 * asz), reinterpret_cast<uint32_t *>(ws + 3 * asz)};
  const int64_t nblocks = rs_blocks(n);
  uint32_t *hist = reinterpret_cast<uint32_t *>(ws + 4 * asz);
  uint32_t *hist_total = reinterpret_cast<uint32_t *>(ws + 4 * asz + align_up(256 * nblocks * 4, 256));
  void *scan_ws = ws + 4 * asz + align_up(256 * nblocks * 4, 256) + 256;
  const int64_t scan_items = 256 * nblocks > m ? 256 * nblocks : m;
  const SpatialGrid g = to_device_grid(grid);
  const dim3 grid_n((unsigned)ceil_div(n, 256)), block(256);
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_sp_keys<NCOL>, grid_n, block, 0, stream, coords, n, g, keys[0], vals[0]));
  ME_LAUNCH_CHECK();
  // stable LSD radix sort of (supercell, row) by supercell: ceil(log2 m / 8) passes
  int bits = 0;
  while ((1ll << bits) < m) ++bits;
  int cur = 0;
  for (int shift = 0; shift < bits; shift += 8) {
    hipLaunchKernelGGL(k_rs_hist<uint32_t>, dim3((unsigned)nblocks), dim3(256), 0, stream, keys[cur], n, shift, nblocks, hist);
    ME_LAUNCH_CHECK();
    if (int rc = exclusive_scan_u32(hist, hist, 256 * nblocks, hist_total, scan_ws, scan_workspace_bytes(scan_items),
                                    stream))
      return rc;
    hipLaunchKernelGGL(k_rs_scatter<uint32_t>, dim3((unsigned)nblocks), dim3(64), 0, stream, keys[cur], vals[cur], n, shift,
                       nblocks, hist, keys[cur ^ 1], vals[cur ^ 1]);
    ME_LAUNCH_CHECK();
    cur ^= 1;
  }
  // directory: first position of every supercell in the sorted keys; dir_start[m] = n
  hipLaunchKernelGGL(k_sp_directory, dim3((unsigned)ceil_div(m + 1, 256)), dim3(256), 0, stream, keys[cur], n, m,
                     dir_start);
  ME_LAUNCH_CHECK();
  ME_DISPATCH_NCOL(ncol, hipLaunchKernelGGL(k_sp_finish<NCOL>, grid_n, block, 0, stream, coords, n, vals[cur], order,
                                            pos_of_row, coords_sorted));
  ME_LAUNCH_CHECK();
  return 0;
}

// halo (cells on either side) of a region on the lookup map's grid, per axis; -1 when it is not a whole number of cells
static bool region_halo(const me_region *rg, const me_spatial_grid *l, int32_t *halo) {
  const int D = rg->ncol - 1;
  for (int d = 0; d < D; ++d) {
    if (rg->tensor_stride[d] != l->tensor_stride[d]) return false;   // offsets step by the lookup map's cell
    const int ks = rg->kernel_size[d], dil = rg->dilation[d];
    const int reach = (ks % 2 == 0) ? (ks - 1) : (ks / 2);
    halo[d] = reach * dil;
  }
  return true;
}

static bool probe_lds_geometry(const me_region *rg, const me_spatial_grid *q, const me_spatial_grid *l, ProbeLds *pl,
                               int64_t *lds_bytes) {
  if (!rg || !q || !l || q->ncol != l->ncol || rg->ncol != q->ncol) return false;
  const int D = q->ncol - 1;
  if (rg->region_type != ME_REGION_HYPER_CUBE && rg->region_type != ME_REGION_HYPER_CROSS) return false;
  int32_t halo[ME_MAX_DIM];
  if (!region_halo(rg, l, halo)) return false;
  int64_t cells = 1, sc_cells = 1;
  for (int d = 0; d < D; ++d) {
    // both maps on the same cell size and the same supercell side; offsets reach at most one supercell away
    if (q->tensor_stride[d] != l->tensor_stride[d] || q->shift[d] != l->shift[d]) return false;
    const int side = 1 << q->shift[d];
    if (halo[d] > side) return false;
    pl->halo[d] = halo[d];
    pl->gdim[d] = side + 2 * halo[d];
    cells *= pl->gdim[d];
    sc_cells *= side;
    if (cells > (1 << 20)) return false;
  }
  for (int d = D; d < ME_MAX_DIM; ++d) pl->halo[d] = 0, pl->gdim[d] = 1;
  const int64_t volume = me_region_volume(rg);
  if (volume < 1) return false;
  pl->grid_cells = (int32_t)cells;
  pl->sc_cells = (int32_t)sc_cells;
  int64_t n_nb = 1;
  for (int d = 0; d < D; ++d) n_nb *= 3;
  *lds_bytes = (cells + sc_cells + volume + 2 * n_nb + 2) * 4;
  return *lds_bytes <= 160 * 1024 - 256;
}

int64_t me_kernel_map_probe_lds_bytes(const me_region *region, const me_spatial_grid *q_grid,
                                      const me_spatial_grid *l_grid) {
  ProbeLds pl;
  int64_t bytes = 0;
  return probe_lds_geometry(region, q_grid, l_grid, &pl, &bytes) ? bytes : -1;
}

int me_kernel_map_probe_lds(const me_spatial_grid *q_grid, const int32_t *q_coords_sorted, const uint32_t *q_dir,
                            int64_t n_q, const me_spatial_grid *l_grid, const int32_t *l_coords_sorted,
                            const int32_t *l_order, const uint32_t *l_dir, const me_region *region, int32_t *nbr_pos,
                            int64_t *k_offsets_dev, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ProbeLds pl;
  int64_t lds = 0;
  ME_CHECK(probe_lds_geometry(region, q_grid, l_grid, &pl, &lds),
           "this kernel region / map pair is not eligible for the LDS-bucketed probe (me_kernel_map_probe_lds_bytes)");
  const int64_t volume = me_region_volume(region);
  ME_CHECK(n_q >= 0 && n_q * volume < (1ll << 32), "n_q * volume must fit in 32 bits");
  ME_CHECK(k_offsets_dev == nullptr || workspace_bytes >= me_kernel_map_workspace_bytes(n_q, volume),
           "workspace too small");
  if (n_q == 0) {
    if (k_offsets_dev) ME_HIP(hipMemsetAsync(k_offsets_dev, 0, (size_t)(volume + 1) * 8, stream));
    return 0;
  }
  const int64_t m_q = me_spatial_cells(q_grid);
  ME_CHECK(m_q >= 1, "invalid query grid");
  // with k_offsets_dev: the per-(offset, 64-position chunk) pair counts are taken inside the probe, then scanned
  // (the workspace afterwards feeds me_kernel_map_compact_ordered, exactly as after me_kernel_map_probe)
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t nw = kmap_nw(n_q);
  uint32_t *wcount = k_offsets_dev ? reinterpret_cast<uint32_t *>(ws) : nullptr;
  if (wcount) ME_HIP(hipMemsetAsync(wcount, 0, (size_t)volume * nw * 4, stream));
  const int ncol = q_grid->ncol;
  ME_CHECK(ncol != 4 || ((uintptr_t)q_coords_sorted % 16 == 0 && (uintptr_t)l_coords_sorted % 16 == 0),
           "coordinates with 4 columns must be 16-byte aligned");
  const SpatialGrid gq = to_device_grid(q_grid), gl = to_device_grid(l_grid);
  const me_region rg = *region;
  int64_t blocks = m_q;
  const int64_t cap = (int64_t)4 * 256 * (lds > 40 * 1024 ? 1 : 2);   // a few supercells per CU in flight
  if (blocks > cap) blocks = cap;
#define ME_PROBE_LDS(NC)                                                                                        \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (lds > 48 * 1024 && !attr_set) {                                                                         \
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmap_probe_lds<NC>),                         \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                      \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL(k_kmap_probe_lds<NC>, dim3((unsigned)blocks), dim3(256), (size_t)lds, stream, gq,        \
                       q_coords_sorted, q_dir, n_q, m_q, gl, l_coords_sorted, l_order, l_dir, rg, pl,           \
                       (int32_t)volume, nbr_pos, wcount, nw);                                                   \
  } while (0)
  ME_DISPATCH_NCOL(ncol, ME_PROBE_LDS(NCOL));
#undef ME_PROBE_LDS
  ME_LAUNCH_CHECK();
  if (wcount) {
    uint32_t *total = reinterpret_cast<uint32_t *>(ws + kmap_counts_bytes(n_q, volume));
    void *scan_ws = ws + kmap_counts_bytes(n_q, volume) + 256 + align_up((volume + 1) * 8, 256);
    if (int rc = exclusive_scan_u32(wcount, wcount, volume * nw, total, scan_ws, scan_workspace_bytes(volume * nw),
                                    stream))
      return rc;
    hipLaunchKernelGGL(k_kmap_koffsets, dim3((unsigned)ceil_div(volume + 1, 256)), dim3(256), 0, stream, wcount,
                       total, nw, volume, k_offsets_dev);
    ME_LAUNCH_CHECK();
  }
  return 0;
}

int me_kernel_map_count(const int32_t *nbr, int64_t n_out, int64_t volume, int64_t *k_offsets,
                        int64_t *k_offsets_dev, void *workspace, int64_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(k_offsets_dev != nullptr, "k_offsets_dev must not be null");
  ME_CHECK(workspace_bytes >= me_kernel_map_workspace_bytes(n_out, volume), "workspace too small");
  if (n_out == 0) {
    if (k_offsets)
      for (int64_t k = 0; k <= volume; ++k) k_offsets[k] = 0;
    ME_HIP(hipMemsetAsync(k_offsets_dev, 0, (size_t)(volume + 1) * 8, stream));
    return 0;
  }
  char *ws = reinterpret_cast<char *>(workspace);
  const int64_t nw = kmap_nw(n_out);
  uint32_t *wcount = reinterpret_cast<uint32_t *>(ws);
  uint32_t *total = reinterpret_cast<uint32_t *>(ws + kmap_counts_bytes(n_out, volume));
  void *scan_ws = ws + kmap_counts_bytes(n_out, volume) + 256 + align_up((volume + 1) * 8, 256);
  const dim3 grid((unsigned)ceil_div(n_out, 256), (unsigned)volume), block(256);
  hipLaunchKernelGGL(k_kmap_count, grid, block, 0, stream, nbr, n_out, wcount, nw);
  ME_LAUNCH_CHECK();
  if (int rc = exclusive_scan_u32(wcount, wcount, volume * nw, total, scan_ws, scan_workspace_bytes(volume * nw),
                                  stream))
    return rc;
  hipLaunchKernelGGL(k_kmap_koffsets, dim3((unsigned)ceil_div(volume + 1, 256)), dim3(256), 0, stream, wcount, total,
                     nw, volume, k_offsets_dev);
  ME_LAUNCH_CHECK();
  if (k_offsets) {   // optional: the caller may instead copy k_offsets_dev asynchronously and read it when needed
    ME_HIP(hipMemcpyAsync(k_offsets, k_offsets_dev, (size_t)(volume + 1) * 8, hipMemcpyDeviceToHost, stream));
    ME_HIP(hipStreamSynchronize(stream));
  }
  return 0;
}

int me_kernel_map_compact_ordered(const int32_t *nbr, const int32_t *order, int64_t n_out, int64_t volume,
                                  int32_t *in_pairs, int32_t *out_pairs, void *workspace, int64_t workspace_bytes,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (order == nullptr)
    return me_kernel_map_compact(nbr, n_out, volume, in_pairs, out_pairs, workspace, workspace_bytes, stream_);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  ME_CHECK(workspace_bytes >= me_kernel_map_workspace_bytes(n_out, volume), "workspace too small");
  if (n_out == 0) return 0;
  const int64_t nw = kmap_nw(n_out);
  const uint32_t *woffs = reinterpret_cast<const uint32_t *>(workspace);
  const dim3 grid((unsigned)ceil_div(n_out, 256), (unsigned)volume), block(256);
  hipLaunchKernelGGL(k_kmap_compact_ordered, grid, block, 0, stream, nbr, order, n_out, woffs, nw, in_pairs,
                     out_pairs);
  ME_LAUNCH_CHECK();
  return 0;
}

int me_kernel_map_transpose_ordered(const int32_t *in_pairs, const int32_t *out_pairs, const int64_t *k_offsets_dev,
                                    int64_t volume, int64_t n_pairs_bound, int64_t n_in, const int32_t *pos_in,
                                    int32_t *nbrT, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (pos_in == nullptr)
    return me_kernel_map_transpose(in_pairs, out_pairs, k_offsets_dev, volume, n_pairs_bound, n_in, nbrT, stream_);
  ME_CHECK(volume >= 1 && volume <= 65535, "kernel volume out of range");
  if (n_in == 0) return 0;
  ME_HIP(hipMemsetAsync(nbrT, 0xff, (size_t)volume * n_in * 4, stream));
  if (n_pairs_bound == 0) return 0;
  int64_t bx = ceil_div(ceil_div(n_pairs_bound, volume), 256);
  if (bx < 1) bx = 1;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(k_kmap_transpose_ordered, dim3((unsigned)bx, (unsigned)volume), dim3(256), 0, stream, in_pairs,
                     out_pairs, k_offsets_dev, pos_in, n_in, nbrT);
  ME_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_coords(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_rs_hist<uint32_t>));
}

// Load every code object of the library now (ABI 1.5).  HIP loads a translation unit's device code when the first kernel
// of it is launched: the first backward pass of a process paid 88 ms for the weight-gradient / reduce / batch-norm units
// (BENCH_r04 cold_breakdown_ms.first_backward_plans).  Called by the hosts at the first map insert on a device; costs what those first launches cost.
extern "C" {
void me_preload_conv(void); void me_preload_conv_bf16(void); void me_preload_conv_bf16_ws(void); void me_preload_conv_f32x3(void);
void me_preload_conv_halo(void); void me_preload_coords(void); void me_preload_norm(void); void me_preload_pack(void);
void me_preload_f64(void); void me_preload_pool(void); void me_preload_conv_stem(void); void me_preload_conv_rowwise(void);
int me_preload(void) {
  me_preload_coords(); me_preload_conv(); me_preload_conv_bf16(); me_preload_conv_bf16_ws(); me_preload_conv_f32x3();
  me_preload_conv_halo(); me_preload_conv_stem(); me_preload_conv_rowwise(); me_preload_norm(); me_preload_pack(); me_preload_pool(); me_preload_f64();
  ME_HIP(hipGetLastError());
  return 0;
}
}
