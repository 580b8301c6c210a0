// bf16 sparse convolution (forward / dgrad), OUTPUT-STATIONARY on a source halo staged in LDS (round 5).
//
// The tile kernels of rounds 1 - 4 (k_conv_tile_bf16[_ws]) walk a plan of single-offset batches: gather <= 64 rows of ONE
// offset from L2 / HBM, multiply, add into an fp32 accumulator tile in LDS through target indices — one barrier and one
// chain of LDS round trips (indices -> old sums -> new sums) per batch, every source row fetched once per offset that
// touches it.  Their phase counters (docs/HISTORY.md 10.4, 11.7) name that chain, not the matrix pipe or the bandwidth,
// as the bound.  This kernel removes the three things the chain consists of:
//   * a tile's SOURCE NEIGHBOURHOOD (the union of its rows' neighbours over all offsets: 1.1 - 2.7 x the tile on
//     spatially ordered rows) is staged in LDS ONCE per channel chunk; every offset gathers from LDS;
//   * the accumulators live in REGISTERS: wave (wr, wc) owns rows [wr * R * 16, ...) x columns [wc * CB * 16, ...) of the
//     tile for the whole walk — no target indices, no read-modify-write, nothing to order between offsets;
//   * there is NO barrier inside the walk: the halo is read-only, the weights stream from the packed image (L2) into
//     registers one offset ahead, per wave.
// Absent neighbours multiply a zero row (slot 0 of the halo); a 16-row group without any neighbour at an offset is
// skipped (kmask).  So the matrix pipe does 1.6 - 2.5 x the useful work on MinkUNet's levels — it has a factor of ten in
// hand.  Per target row the fp32 sum runs over the offsets in ascending order and over the source
// channels in ascending 32-channel steps inside one MFMA accumulator; one rounding to bf16 at the end: the semantics of
// me_conv_target_bf16 (fp32 sums in a fixed order), NOT its bits (there every batch starts a new accumulator).
// Reference: src/convolution_kernel.cu:320-496 (gather - GEMM - scatter per offset), src/convolution_kernel.hpp:81-144.
#include "conv_common.hpp"
#include "conv_ws.hpp"
#include <stdlib.h>

namespace me {

// ---- halo plan -----------------------------------------------------------------------------------------------------------
// One workgroup per tile of T consecutive target positions: the distinct source rows of the tile over all offsets, sorted
// by row (bitonic sort of the volume * T table entries in LDS), are its halo; slot 0 of the LDS image is the zero row, so
//   lidx[tile][k][r] = 0 (no neighbour) | 1 + halo slot | 0xffff (the halo overflowed `s_cap`: the tile takes the
//   kernel's direct-gather path), kmask[tile][k] bit g = some row of 16-row group g has a neighbour at offset k.
constexpr uint32_t kHaloNone = 0xffffffffu;

template <int N>
__global__ __launch_bounds__(256) void k_halo_plan(const int32_t *__restrict__ tbl, const int32_t *__restrict__ col_order,
                                                   const int32_t *__restrict__ src_pos, const int32_t *__restrict__ src_order,
                                                   int64_t n_tgt, int volume, int T, int s_cap,
                                                   int32_t *__restrict__ halo_cnt, int32_t *__restrict__ halo_rows,
                                                   uint16_t *__restrict__ lidx, uint32_t *__restrict__ kmask) {
  __shared__ uint32_t s_key[N];
  __shared__ uint32_t s_uniq[N];
  __shared__ uint32_t s_mask[64];
  __shared__ int s_wsum[4];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int n_cand = volume * T;
  auto source_of = [&](int e) -> int32_t {
    const int k = e / T, r = e % T;
    const int64_t p = tile * T + r;
    if (p >= n_tgt) return -1;
    const int64_t col = col_order ? (int64_t)col_order[p] : p;
    return tbl[(int64_t)k * n_tgt + col];
  };
  if (tid < 64) s_mask[tid] = 0u;
  for (int e = tid; e < N; e += 256) {
    uint32_t key = kHaloNone;
    if (e < n_cand) {
      const int32_t s = source_of(e);
      if (s >= 0) key = (uint32_t)(src_pos ? src_pos[s] : s);     // slots in POSITION order of the source map when given
    }
    s_key[e] = key;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < N / 2; i += 256) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const uint32_t x = s_key[a], y = s_key[b];
        if ((x > y) == up) {
          s_key[a] = y;
          s_key[b] = x;
        }
      }
      __syncthreads();
    }
  }
  // distinct keys, in order: thread t owns elements [t * E, (t + 1) * E)
  constexpr int E = N / 256;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid * E + j;
    const uint32_t v = s_key[i];
    cnt += (v != kHaloNone && (i == 0 || s_key[i - 1] != v)) ? 1 : 0;
  }
  const int incl = (int)wave_inclusive_scan((uint32_t)cnt);
  if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
  __syncthreads();
  int base = incl - cnt;
  for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
  const int S = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid * E + j;
    const uint32_t v = s_key[i];
    if (v != kHaloNone && (i == 0 || s_key[i - 1] != v)) s_uniq[base++] = v;
  }
  __syncthreads();
  if (tid == 0) halo_cnt[tile] = S;
  for (int j = tid; j < min(S, s_cap); j += 256)
    halo_rows[tile * s_cap + j] = src_order ? src_order[s_uniq[j]] : (int32_t)s_uniq[j];
  for (int e = tid; e < n_cand; e += 256) {
    const int32_t s = source_of(e);
    uint16_t li = 0;
    if (s >= 0) {
      const uint32_t key = (uint32_t)(src_pos ? src_pos[s] : s);
      int lo = 0, hi = S;   // first slot with s_uniq[slot] >= key
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_uniq[mid] < key) lo = mid + 1;
        else hi = mid;
      }
      li = lo < s_cap ? (uint16_t)(lo + 1) : (uint16_t)0xffffu;
      atomicOr(&s_mask[e / T], 1u << ((e % T) >> 4));
    }
    lidx[tile * n_cand + e] = li;
  }
  __syncthreads();
  if (tid < volume) kmask[tile * volume + tid] = s_mask[tid];
}

// ---- the convolution --------------------------------------------------------------------------------------------------------
// T rows per tile; a workgroup is NWR x NWC waves: wave (wr, wc) owns R = T / 16 / NWR row groups x CB 16-column blocks.
// KC: source channels staged per pass over the offsets.  SKIP: skip the MFMAs of 16-row groups without a neighbour.
__host__ __device__ constexpr int halo_s_cap(int t) { return t <= 64 ? 319 : 383; }   // halo slots of a tile (+ the zero row)

// LDS image of the halo: slot r (0 = the zero row) is a row of KC bf16 padded by 32 bytes, its 16-byte pieces in order.
// A ds_read_b128 serves 16 lanes per LDS cycle — eight rows' piece of one 8-channel quarter and eight rows' piece of the
// next — over 16 bank slots of 16 bytes; the rows of a gather are unrelated, so what matters is that the slot is a good
// hash of (row, piece): with 128-byte rows a read of one 32-channel step could only ever touch HALF the slots (simulated on
// the MinkUNet scene, LDS cycles per read, conflict-free = 4: 13.6 for that layout, 10.0 for an XOR of the whole piece
// index, 9.1 for 32 bytes of padding, 7.6 with the slots also ordered by position: halo plan).  Padding keeps the step
// an immediate offset and the lane's part a plain add.
template <int KC>
struct HaloLayout {
  static constexpr int kRowBytes = KC * 2 + 32;
  static constexpr bool kXor = false;
  __host__ __device__ static constexpr int slot_word(int r) { return r * kRowBytes; }            // lane part: + q * 16
  __host__ __device__ static constexpr int piece_off(int r, int u) { return r * kRowBytes + u * 16; }
};
__host__ __device__ constexpr int halo_row_bytes(int kc) { return kc * 2 + 32; }
__host__ __device__ constexpr int halo_out_ld(int nc) { return nc + 8; }   // bf16 elements per row of the output image
__host__ __device__ constexpr int conv_halo_lds(int t, int nc, int kc, int s_cap, int volume, int nt) {
  const int halo = (s_cap + 1) * halo_row_bytes(kc);
  const int outb = t * halo_out_ld(nc) * 2 + (nt / (nc / 4)) * (nc / 4) * 8 * 4;
  return (halo > outb ? halo : outb) + volume * t * 4 + 64 * 4 + s_cap * 4;
}

// waves per SIMD the register budget is sized for: two workgroups per CU unless the two weight sets of a deep chunk need more
__host__ __device__ constexpr int halo_min_waves(int t, int kc) { return (t > 64 && kc > 64) ? 1 : 2; }

__device__ __forceinline__ bf16x8 zero8() {
  const __bf16 z = (__bf16)0.f;
  return bf16x8{z, z, z, z, z, z, z, z};
}

// Phase counters of a -DME_HALO_TIMING build (scripts/halo_sweep.py TIMING=1; s_memtime ticks of wave 0 of every workgroup):
// [0] prologue, [1] staging (+ its barriers), [2] walk over the offsets, [3] epilogue, [4] tiles, [5] offsets walked
#ifdef ME_HALO_TIMING
__device__ unsigned long long d_halo_timing[8];
#define ME_HT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ht_[i] += t_ - ht_prev_; ht_prev_ = t_; } while (0)
#else
#define ME_HT(i) do { } while (0)
#endif

template <int T, int CB, int NWC, int NWR, int KC, bool SKIP, int WD>
__global__ __launch_bounds__(64 * NWC * NWR, 1) void k_conv_halo_bf16(
    const __bf16 *__restrict__ src, int c_src, const bf16x8 *__restrict__ wp, int ksp, int nchp, int ncb, int c_dst,
    const int32_t *__restrict__ halo_cnt, const int32_t *__restrict__ halo_rows, const uint16_t *__restrict__ lidx,
    const uint32_t *__restrict__ kmask, const int32_t *__restrict__ tbl, const int32_t *__restrict__ col_order,
    const int32_t *__restrict__ out_order, __bf16 *__restrict__ dst, int64_t n_tgt, int volume, int s_cap,
    float *__restrict__ stat_mean, float *__restrict__ stat_m2) {
  typedef HaloLayout<KC> HL;
  constexpr int NT = 64 * NWC * NWR;
  constexpr int NC = CB * 16 * NWC;
  constexpr int R = T / 16 / NWR;
  constexpr int KS = KC / 32;
  constexpr int F8 = KC / 8;
  constexpr int RS = HL::kRowBytes;
  constexpr int OLD = halo_out_ld(NC);
  static_assert(T % (16 * NWR) == 0 && R >= 1 && R <= 16, "row groups per wave");
  static_assert(NT % (NC / 4) == 0, "threads per output row");

  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int halo_bytes = (s_cap + 1) * RS;
  constexpr int OUT_BYTES = T * OLD * 2 + (NT / (NC / 4)) * (NC / 4) * 8 * 4;
  const int region = halo_bytes > OUT_BYTES ? halo_bytes : OUT_BYTES;
  char *s_halo = smem;
  uint32_t *s_atab = reinterpret_cast<uint32_t *>(smem + region);            // [volume][T]: LDS address word of the slot
  uint32_t *s_kmask = reinterpret_cast<uint32_t *>(smem + region + volume * T * 4);   // [<= 64]
  int32_t *s_hrow = reinterpret_cast<int32_t *>(s_kmask + 64);                         // [s_cap] source rows of the halo

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % NWC, wr = wave / NWC;
  const int i16 = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x;
  const int rb = wr * R * 16;                      // first tile row of this wave
  const int cb0 = blockIdx.y * (NC / 16) + wc * CB;   // first 16-column block of this wave
  const int nchunks = c_src / KC;                  // (whole chunks: host-checked)
  const int s_tot = halo_cnt[tile];
  const bool ovf = s_tot > s_cap;                  // uniform: the direct-gather path
  const int S = min(s_tot, s_cap);
  constexpr uint32_t RM = R >= 32 ? 0xffffffffu : ((1u << R) - 1u);

  typedef __attribute__((address_space(3))) const char lds_char;
  const unsigned halo_addr = (unsigned)(uintptr_t)(lds_char *)reinterpret_cast<const char *>(s_halo);   // (a multiple of 128)
  {
    // slot -> the LDS address word of its row (halo base + row + permutation), once per tile: the walk's operand address
    // is this word ^ (q * 16) [+ q * 16 for the padded 96-channel rows] with the step as an immediate offset
    const uint16_t *g = lidx + tile * volume * T;
    for (int x = tid; x < volume * T; x += NT) {
#if defined(ME_HALO_ABL) && ME_HALO_ABL == 1   // timing ablation (results invalid): every lane reads the zero row — no bank conflicts
      s_atab[x] = halo_addr;
#else
      s_atab[x] = halo_addr + (unsigned)HL::slot_word((int)g[x]);
#endif
    }
    if (tid < 64) s_kmask[tid] = tid < volume ? kmask[tile * volume + tid] : 0u;
    for (int x = tid; x < S; x += NT) s_hrow[x] = halo_rows[tile * s_cap + x];
    if (tid < F8) *reinterpret_cast<bf16x8 *>(s_halo + HL::piece_off(0, tid)) = zero8();
  }
  __syncthreads();
  uint32_t act = 0u;   // offsets at which this wave's rows have a neighbour
  for (int k = 0; k < volume; ++k) act |= ((s_kmask[k] >> (wr * R)) & RM) ? (1u << k) : 0u;
  act = __builtin_amdgcn_readfirstlane(act);

#ifdef ME_HALO_TIMING
  unsigned long long ht_[4] = {0, 0, 0, 0};
  unsigned long long ht_prev_ = __builtin_amdgcn_s_memtime();
  unsigned long long ht_offsets_ = 0;
#endif
  f32x4 acc[R][CB];
#pragma unroll
  for (int g = 0; g < R; ++g) {
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const char *srcb = reinterpret_cast<const char *>(src);
  const int64_t row_bytes = (int64_t)c_src * 2;
  constexpr int NPB = 8;   // 16-byte pieces of the image a thread loads per round: piece x = x0 + j * NT + tid

  auto load_w = [&](int chunk, int k, bf16x8 (&wd)[KS][CB]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int a = chunk * KS + s;          // absolute 32-channel step -> (chunk, step) of the packed image
      const int cpk = a / ksp, v = a - cpk * ksp;
#pragma unroll
      for (int c = 0; c < CB; ++c)
        wd[s][c] = wp[((((int64_t)k * nchp + cpk) * ncb + min(cb0 + c, ncb - 1)) * ksp + v) * 64 + lane];
    }
  };

  if (!ovf) {
    // Every load of the walk is inline asm at a fixed place with counted waits: left to hipcc, the weight loads of the
    // NEXT offset sink to their first use (the kernel sits at its register budget) and every offset starts with a full
    // L2 round trip.  Per offset, in this order: the next offset's weights (KS * CB global loads) and halo slots (R LDS
    // reads), then `s_waitcnt vmcnt(KS * CB)` — this offset's weights, requested one offset ago — then operand reads
    // and MFMAs (compiler-scheduled: its own LDS waits stay correct next to the older asm reads, LDS returns in order).
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned atab_addr = (unsigned)(uintptr_t)(lds_char *)reinterpret_cast<const char *>(s_atab) + (unsigned)((rb + i16) * 4);
    const unsigned kmask_addr = (unsigned)(uintptr_t)(lds_char *)reinterpret_cast<const char *>(s_kmask);
    const unsigned qx = (unsigned)q * 16u;
    // weight image: byte offset of fragment (k, chunk, s, c) = k * w_stride_k + w_frag[s][c]; the k part goes into the scalar
    // base of the load, the rest (+ this lane's 16 bytes) is a per-chunk vector offset: no address arithmetic per load
    const int64_t w_stride_k = (int64_t)nchp * ncb * ksp * 1024;
    unsigned w_voff[KS][CB];
    auto set_chunk = [&](int chunk) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int a = chunk * KS + s;
        const int cpk = a / ksp, v = a - cpk * ksp;
#pragma unroll
        for (int c = 0; c < CB; ++c) w_voff[s][c] = lane16 + (unsigned)(((cpk * ncb + min(cb0 + c, ncb - 1)) * ksp + v) * 1024);
      }
    };
    auto issue_w = [&](int k, bf16x8 (&wd)[KS][CB]) {
      const char *base = reinterpret_cast<const char *>(wp) + (int64_t)k * w_stride_k;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wd[s][c]) : "v"(w_voff[s][c]), "s"(base) : "memory");
      }
    };
    // the next offset's slot words (R reads) and its group mask (one broadcast read)
    auto issue_li = [&](int k, unsigned (&li)[R], unsigned &mk) {
      const unsigned a0 = atab_addr + (unsigned)(k * T * 4);
#pragma unroll
      for (int g = 0; g < R; ++g) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(li[g]) : "v"(a0), "n"(g * 64) : "memory");
      const unsigned am = kmask_addr + (unsigned)(k * 4);
      asm volatile("ds_read_b32 %0, %1" : "=v"(mk) : "v"(am) : "memory");
    };
    auto wait_w = [&](bf16x8 (&w)[KS][CB], auto younger) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int c = 0; c < CB; ++c) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[s][c]) : "n"(decltype(younger)::value) : "memory");
      }
    };
    auto wait_li = [&](unsigned (&li)[R], unsigned &mk) {
#pragma unroll
      for (int g = 0; g < R; ++g) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(li[g]) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk) : : "memory");
    };
    auto lane_part = [&](unsigned (&li)[R]) {
#pragma unroll
      for (int g = 0; g < R; ++g) li[g] = HL::kXor ? (li[g] ^ qx) : (li[g] + qx);
    };
    auto group_mask = [&](unsigned mk) {
      return SKIP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((mk >> (wr * R)) & RM)) : RM;
    };
    auto issue_x = [&](const unsigned (&a0)[R], auto s_, bf16x8 (&x)[R]) {
#pragma unroll
      for (int g = 0; g < R; ++g)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[g]) : "v"(a0[g]), "n"(decltype(s_)::value * 64) : "memory");
    };
    auto wait_x = [&](bf16x8 (&x)[R], auto younger) {
#pragma unroll
      for (int g = 0; g < R; ++g) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x[g]) : "n"(decltype(younger)::value) : "memory");
    };
    using NW = std::integral_constant<int, KS * CB>;
    using NR = std::integral_constant<int, (R > 15 ? 15 : R)>;
    bf16x8 xb[2][R];
    // One offset: its operands of step 0 are in flight in xb[P0] on entry.  The next offset's weights and halo slots are
    // requested first; every step requests the operands of the step after it — the LAST one those of the next offset's
    // step 0, once its slots have landed — before it multiplies: an LDS round trip is always one step ahead of its use.
    using NWV = std::integral_constant<int, ((WD - 1) * KS * CB > 63 ? 63 : (WD - 1) * KS * CB)>;   // weight loads that may stay in flight
    auto offset_step = [&](auto p0_, int kn, int kw, const bf16x8 (&wC)[KS][CB], bf16x8 (&wN)[KS][CB], const unsigned (&aC)[R],
                           unsigned (&aN)[R], unsigned mC, unsigned &mN) {
      constexpr int P0 = decltype(p0_)::value;
      const uint32_t m = group_mask(mC);     // 16-row groups of this wave with a neighbour at this offset
      issue_w(kw, wN);
      if constexpr (KS == 1) issue_li(kn, aN, mN);
      auto step = [&](auto s_) {
        constexpr int s = decltype(s_)::value;
        bf16x8(&xc)[R] = xb[(P0 + s) & 1];
        bf16x8(&xn)[R] = xb[(P0 + s + 1) & 1];
        if constexpr (s + 1 < KS) {
          issue_x(aC, std::integral_constant<int, (s + 1 < KS ? s + 1 : 0)>{}, xn);
          if constexpr (s == 0) {
            issue_li(kn, aN, mN);       // (the youngest LDS requests: the wait below does not cover them)
            wait_w(const_cast<bf16x8(&)[KS][CB]>(wC), NWV{});
            wait_x(xc, std::integral_constant<int, (2 * R + 1 > 15 ? 15 : 2 * R + 1)>{});
          } else {
            wait_x(xc, NR{});
          }
        } else {
          wait_li(aN, mN);              // lgkmcnt(0): this step's operands and the next offset's slot words
          lane_part(aN);
          issue_x(aN, std::integral_constant<int, 0>{}, xn);
          if constexpr (s == 0) wait_w(const_cast<bf16x8(&)[KS][CB]>(wC), NWV{});
          wait_x(xc, NR{});
        }
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks this step's MFMAs behind the NEXT step's waits)
#pragma unroll
        for (int g = 0; g < R; ++g) {
          if (!SKIP || ((m >> g) & 1u)) {    // (uniform: groups without a neighbour at this offset multiply nothing)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
#if defined(ME_HALO_ABL) && ME_HALO_ABL == 2   // timing ablation (results invalid): no MFMAs
              asm volatile("" : "+v"(acc[g][c]) : "v"(wC[s][c]), "v"(xc[g]));
#else
              acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wC[s][c], xc[g], acc[g][c], 0, 0, 0);
#endif
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      step(std::integral_constant<int, 0>{});
      if constexpr (KS > 1) step(std::integral_constant<int, 1>{});
      if constexpr (KS > 2) step(std::integral_constant<int, 2>{});
      if constexpr (KS > 3) step(std::integral_constant<int, 3>{});
    };
    ME_HT(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk > 0) __syncthreads();   // every wave is done with the previous chunk's image
      // the image of this chunk: rounds of NPB unconditional loads per thread (clamped to the last halo row) — predicated
      // loads made hipcc wait for every one of them separately, NP dependent round trips per chunk
      for (int x0 = 0; x0 < S * F8; x0 += NPB * NT) {
        int rows[NPB];
        bf16x8 v[NPB];
#pragma unroll
        for (int j = 0; j < NPB; ++j) rows[j] = s_hrow[min((x0 + j * NT + tid) / F8, S - 1)];
#pragma unroll
        for (int j = 0; j < NPB; ++j)
          v[j] = *reinterpret_cast<const bf16x8 *>(srcb + (int64_t)rows[j] * row_bytes + (chunk * KC + ((x0 + j * NT + tid) % F8) * 8) * 2);
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
          const int x = x0 + j * NT + tid;
          if (x < S * F8) *reinterpret_cast<bf16x8 *>(s_halo + HL::piece_off(x / F8 + 1, x % F8)) = v[j];
        }
      }
      __syncthreads();
      ME_HT(1);
      if (act) {
        // WD weight sets: the weights of offset i + WD - 1 are requested while offset i is multiplied (one L2 round trip is
        // 1,000 - 2,000 cycles under load, an offset 500 - 1,000: with ONE offset of look-ahead the walk ran at the pace of
        // the round trips).  `rw` walks WD - 1 offsets ahead of `rn`; behind the last offset the last one is requested
        // again, so that the counted waits stay the same.
        bf16x8 wr[WD][KS][CB];
        unsigned aA[R], aB[R], mA, mB;
        uint32_t rn = act, rw = act;
        int k_last = 31 - __builtin_clz(act);
        auto pop = [&](uint32_t &r) {
          const int k = r ? __builtin_ctz(r) : k_last;
          r &= r - 1;
          return k;
        };
        int n_left = __builtin_popcount(act);
        pop(rn);   // (the first offset)
        set_chunk(chunk);
#pragma unroll
        for (int j = 0; j < WD - 1; ++j) issue_w(pop(rw), wr[j]);
        issue_li(__builtin_ctz(act), aA, mA);
        wait_li(aA, mA);
        lane_part(aA);
        issue_x(aA, std::integral_constant<int, 0>{}, xb[0]);
        static_assert(WD == 2 || WD == 4, "weight sets");
        for (;;) {
          offset_step(std::integral_constant<int, 0>{}, pop(rn), pop(rw), wr[0], wr[(0 + WD - 1) % WD], aA, aB, mA, mB);
          if (--n_left == 0) break;
          offset_step(std::integral_constant<int, KS & 1>{}, pop(rn), pop(rw), wr[1], wr[(1 + WD - 1) % WD], aB, aA, mB, mA);
          if (--n_left == 0) break;
          if constexpr (WD == 4) {
            offset_step(std::integral_constant<int, 0>{}, pop(rn), pop(rw), wr[2], wr[1], aA, aB, mA, mB);
            if (--n_left == 0) break;
            offset_step(std::integral_constant<int, KS & 1>{}, pop(rn), pop(rw), wr[3], wr[2], aB, aA, mB, mA);
            if (--n_left == 0) break;
          }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the surplus requests behind the last offset)
      }
      ME_HT(2);
#ifdef ME_HALO_TIMING
      ht_offsets_ += __builtin_popcount(act);
#endif
    }
  } else {
    // the halo did not fit in `s_cap` slots: rows straight off the neighbour table (correct for any map; not a fast path)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      for (uint32_t rem = act; rem; rem &= rem - 1) {
        const int k = __builtin_ctz(rem);
        bf16x8 w[KS][CB];
        load_w(chunk, k, w);
#pragma unroll 1
        for (int g = 0; g < R; ++g) {
          const int64_t p = tile * T + rb + g * 16 + i16;
          int32_t srow = -1;
          if (p < n_tgt) srow = tbl[(int64_t)k * n_tgt + (col_order ? (int64_t)col_order[p] : p)];
          f32x4 part[CB];
#pragma unroll
          for (int c = 0; c < CB; ++c) part[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            bf16x8 x = zero8();
            if (srow >= 0) x = *reinterpret_cast<const bf16x8 *>(srcb + (int64_t)srow * row_bytes + (chunk * KC + (s * 4 + q) * 8) * 2);
#pragma unroll
            for (int c = 0; c < CB; ++c) part[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][c], x, part[c], 0, 0, 0);
          }
          // (g is a run-time index here: the accumulators are updated through a select chain)
#pragma unroll
          for (int gg = 0; gg < R; ++gg) {
            if (gg == g) {
#pragma unroll
              for (int c = 0; c < CB; ++c) acc[gg][c] += part[c];
            }
          }
        }
      }
    }
  }

  // ---- epilogue: the tile rounded to bf16 (RNE) through LDS, every target row written once in whole-row pieces; the
  // tile's batch-norm statistics (mean and M2 of the STORED values, shifted by the tile's first row: the arithmetic of
  // k_conv_tile_bf16's epilogue) ride along ----
  __syncthreads();
  __bf16 *s_out = reinterpret_cast<__bf16 *>(smem);
  float *s_st = reinterpret_cast<float *>(smem + T * OLD * 2);
#pragma unroll
  for (int g = 0; g < R; ++g) {
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const f32x4 v = acc[g][c];
      *reinterpret_cast<bf16x4 *>(s_out + (rb + g * 16 + i16) * OLD + (wc * CB + c) * 16 + q * 4) =
          bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
  }
  __syncthreads();
  const int64_t row0 = tile * T;
  const int rows_here = (int)min((int64_t)T, n_tgt - row0);
  const int col_base = blockIdx.y * NC;
  constexpr int G4 = NC / 4;       // threads per tile row: four columns each
  constexpr int RPT = NT / G4;     // rows in flight
  const bool do_stats = stat_mean != nullptr;
  const int c4 = tid % G4;
  const int cc = col_base + c4 * 4;
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (do_stats) {
    const bf16x4 v0 = *reinterpret_cast<const bf16x4 *>(s_out + c4 * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[t] = (float)v0[t];
  }
  for (int row = tid / G4; row < rows_here; row += RPT) {
    const bf16x4 vb = *reinterpret_cast<const bf16x4 *>(s_out + row * OLD + c4 * 4);
    if (do_stats) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float d = (float)vb[t] - sh[t];
        st1[t] += d;
        st2[t] = fmaf(d, d, st2[t]);
      }
    }
    if (cc < c_dst) {
      const int64_t grow = out_order ? (int64_t)out_order[row0 + row] : row0 + row;
      *reinterpret_cast<bf16x4 *>(dst + grow * c_dst + cc) = vb;   // (c_dst % 16 == 0: host-checked)
    }
  }
  if (do_stats) {
    float *mine = s_st + ((tid / G4) * G4 + c4) * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mine[t] = st1[t];
      mine[4 + t] = st2[t];
    }
    __syncthreads();
    if (tid < NC && col_base + tid < c_dst) {
      const int g4 = tid >> 2, t = tid & 3;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) {
        a += s_st[(rr * G4 + g4) * 8 + t];
        b += s_st[(rr * G4 + g4) * 8 + 4 + t];
      }
      const float shift = (float)s_out[tid];
      const float cnt = (float)rows_here, m = a / cnt;
      stat_mean[tile * c_dst + col_base + tid] = shift + m;
      stat_m2[tile * c_dst + col_base + tid] = fmaxf(b - a * m, 0.f);
    }
  }
#ifdef ME_HALO_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  ME_HT(3);
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&d_halo_timing[i], ht_[i]);
    atomicAdd(&d_halo_timing[4], 1ull);
    atomicAdd(&d_halo_timing[5], ht_offsets_);
  }
#endif
}

// ---- host side -------------------------------------------------------------------------------------------------------------
int g_halo_mode = -1;    // me_debug_set_halo: -1 policy (ME_AMD_HALO), 0 never, 1 wherever the shape is instantiated
int g_halo_tile = 0;     // forced tile height (64 | 128), 0 = policy
int g_halo_kc = 0;       // forced channel chunk, 0 = policy
int g_halo_skip = 1;     // 0: walk every offset (no kmask test)
int g_halo_wd = 0;       // weight register sets (look-ahead + 1): 0 policy | 2 | 4
int g_halo_cb4 = 0;      // 1: 128-column slabs as 2 x 2 waves of 64 columns (four MFMAs per operand read) instead of 1 x 4 of 32

struct HaloShape {
  int t, cb, nwc, nwr, kc, s_cap;
};

// Which form of the kernel a channel shape takes: forced (me_debug_set_halo(1, ...) / ME_AMD_HALO=1: the tuning globals) or
// the auto policy's per-shape choice.  Depends on (c_src, c_dst) only, so that plan build and launch agree without sharing
// state.  The policy table (profiles/r05_halo_sweep*.log, MinkUNet34C scene, us per launch, tile-plan kernel -> halo):
//   192 -> 128 @80k   137 -> 98    T 128, 2 x 2 waves of 64 columns, group masks
//   256 -> 384 @21k   126 -> 103   (the input gradient of 384 -> 256) the same form
//    64 ->  32 @80k    33 -> 26    (the input gradient of 32 -> 64) T 64, one column wave, no masks
// Every other MinkUNet shape is faster on the tile-plan kernels and keeps them.
struct HaloVariant {
  int tile, cb4, skip, wd;
  bool listed;      // the auto policy has an entry for this shape
};
static int halo_env_mode() {
  static int env_mode = -2;
  if (env_mode == -2) {
    const char *e = getenv("ME_AMD_HALO");
    env_mode = (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : -1);
  }
  return env_mode;
}
static HaloVariant halo_variant(int c_src, int c_dst) {
  const int mode = g_halo_mode >= 0 ? g_halo_mode : halo_env_mode();
  if (mode == 1) return HaloVariant{g_halo_tile == 64 ? 64 : 128, g_halo_cb4, g_halo_skip, g_halo_wd, false};
  if ((c_src == 192 && c_dst == 128) || (c_src == 256 && c_dst == 384)) return HaloVariant{128, 1, 1, 2, true};
  if (c_src == 64 && c_dst == 32) return HaloVariant{64, 0, 0, 2, true};
  return HaloVariant{128, 0, 1, 2, false};
}

static bool halo_shape(int64_t volume, int c_src, int c_dst, HaloShape *hs) {
  if (volume < 2 || volume > 32 || c_src % 32 != 0 || c_dst % 16 != 0) return false;
  const HaloVariant hv = halo_variant(c_src, c_dst);
  HaloShape s;
  s.t = hv.tile;
  if (c_dst % 128 == 0 && hv.cb4) { s.cb = 4; s.nwc = 2; s.nwr = 2; }
  else if (c_dst % 128 == 0) { s.cb = 2; s.nwc = 4; s.nwr = 1; }
  else if (c_dst == 96) { s.cb = 2; s.nwc = 3; s.nwr = 1; }
  else if (c_dst == 64) { s.cb = 2; s.nwc = 2; s.nwr = 2; }
  else if (c_dst == 32) { s.cb = 2; s.nwc = 1; s.nwr = 4; }
  else return false;
  // channel chunks instantiated per wave shape (ME_HALO_T below)
  auto has = [&](int kc) {
    if (c_src % kc != 0) return false;
    if (s.nwc == 4) return kc == 32 || kc == 64 || kc == 96 || kc == 128;
    if (s.cb == 4) return kc == 64;
    if (s.nwc == 3) return kc == 32 || kc == 64 || kc == 96;
    return kc == 32 || kc == 64;
  };
  s.kc = has(64) ? 64 : (has(96) ? 96 : 32);
  if (g_halo_kc > 0 && has(g_halo_kc)) s.kc = g_halo_kc;
  if (volume * s.t > 4096) return false;
#ifndef ME_DEBUG_VARIANTS
  // the shipped library holds the two wave shapes the auto policy selects (halo_variant: 192 -> 128 / 256 -> 384 on
  // 128-row tiles, 64 -> 32 on 64-row tiles); every other shape / tile height / channel chunk — reachable only through the
  // forced mode of the parity tests and sweeps — is in the tuning build (-DME_DEBUG_VARIANTS: scripts/build_debug.sh)
  if (!((s.t == 128 && s.cb == 4 && s.nwc == 2 && s.nwr == 2 && s.kc == 64) ||
        (s.t == 64 && s.cb == 2 && s.nwc == 1 && s.nwr == 4 && s.kc == 64)))
    return false;
#endif
  s.s_cap = halo_s_cap(s.t);
  *hs = s;
  return true;
}

// auto policy: a listed shape (halo_variant) on a map like the ones it was measured on — 3^3 offsets, at least 10k target
// rows (fewer tiles than CUs otherwise), at least 6 pairs per row (the dense walk multiplies absent neighbours: on sparse
// maps the tile-plan kernels win by a wide margin)
static bool halo_policy(int64_t n_tgt, int64_t volume, int64_t n_pairs, int c_src, int c_dst) {
  if (!halo_variant(c_src, c_dst).listed) return false;
  return volume == 27 && n_tgt >= 10000 && n_pairs >= 6 * n_tgt;
}

}  // namespace me

using namespace me;

extern "C" void me_debug_set_halo(int mode, int tile_rows, int kc, int skip) {
  g_halo_mode = mode;
  g_halo_tile = tile_rows;
  g_halo_kc = kc;
  g_halo_skip = skip & 1;
  g_halo_wd = (skip >> 1) & 7;   // (tuning: skip = 1 + 2 * weight sets, e.g. 5 = skip + two sets, 9 = skip + four sets)
  g_halo_cb4 = (skip >> 4) & 1;  // (+ 16: the 2 x 2-wave shape of 64 columns per wave on 128-column slabs)
}
extern "C" int32_t me_debug_halo_mode(void) { return g_halo_mode; }
// phase counters of a -DME_HALO_TIMING build (zeros otherwise); reset != 0 clears them
extern "C" int me_debug_halo_timing(uint64_t *out8, int32_t reset) {
#ifdef ME_HALO_TIMING
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out8 != nullptr) {
    ME_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(me::d_halo_timing), sizeof(h)));
    for (int i = 0; i < 8; ++i) out8[i] = h[i];
  }
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ME_HIP(hipMemcpyToSymbol(HIP_SYMBOL(me::d_halo_timing), z, sizeof(z)));
  }
#else
  if (out8 != nullptr)
    for (int i = 0; i < 8; ++i) out8[i] = 0;
  (void)reset;
#endif
  return 0;
}

// 1: the halo kernel has an instantiation for (volume, c_src, c_dst); tile_rows / s_cap: its plan geometry
extern "C" int32_t me_conv_halo_config_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                                            int32_t *tile_rows, int32_t *s_cap) {
  HaloShape s;
  (void)n_tgt;
  (void)n_pairs;
  if (!halo_shape(volume, c_src, c_dst, &s)) return 0;
  if (tile_rows) *tile_rows = s.t;
  if (s_cap) *s_cap = s.s_cap;
  return 1;
}

// The ONE decision both hosts follow (csrc_host/manager.cpp, backend.py): 1 = this launch side runs on the halo kernel.
// me_debug_set_halo mode, else ME_AMD_HALO = 0 | 1 | auto (default).
extern "C" int32_t me_conv_halo_use_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst) {
  HaloShape s;
  if (n_tgt <= 0 || !halo_shape(volume, c_src, c_dst, &s)) return 0;
  const int mode = g_halo_mode >= 0 ? g_halo_mode : halo_env_mode();
  if (mode >= 0) return mode ? 1 : 0;
  return halo_policy(n_tgt, volume, n_pairs, c_src, c_dst) ? 1 : 0;
}

// A halo plan costs 75 - 110 us to build (a sort per tile) and saves 10 - 40 us per launch: it pays from the SECOND launch
// on the same kernel-map side (a scene that is reused — cached maps), and loses 0.3 ms per scene when every step brings a
// new one (profiles/r05_rocprof_kernel_stats_minkunet34c_bf16_fresh.csv).  The hosts build the plan at this launch count:
// 2 under the policy (the first launch runs on the tile-plan kernel), 1 when the kernel is forced (ME_AMD_HALO=1, tests).
extern "C" int32_t me_conv_halo_min_uses(void) {
  const int mode = g_halo_mode >= 0 ? g_halo_mode : halo_env_mode();
  return mode == 1 ? 1 : 2;
}

extern "C" int64_t me_halo_plan_num_tiles(int64_t n_tgt, int32_t tile_rows) { return tile_rows > 0 ? ceil_div(n_tgt, tile_rows) : 0; }

extern "C" int me_halo_plan_build(const int32_t *tbl_dev, const int32_t *col_order_dev, const int32_t *src_pos_dev,
                                  const int32_t *src_order_dev, int64_t n_tgt, int64_t volume,
                                  int32_t tile_rows, int32_t s_cap, int32_t *halo_cnt_dev, int32_t *halo_rows_dev,
                                  uint16_t *lidx_dev, uint32_t *kmask_dev, void *stream) {
  ME_CHECK(tbl_dev && halo_cnt_dev && halo_rows_dev && lidx_dev && kmask_dev, "null argument");
  ME_CHECK((src_pos_dev == nullptr) == (src_order_dev == nullptr), "halo plan: source positions and their inverse, or neither");
  ME_CHECK(volume >= 1 && volume <= 64 && tile_rows >= 16 && tile_rows % 16 == 0 && volume * tile_rows <= 4096,
           "halo plan: volume x tile_rows must be <= 4096");
  ME_CHECK(s_cap >= 1 && s_cap < 0xffff, "halo plan: s_cap");
  if (n_tgt <= 0) return 0;
  const int64_t tiles = ceil_div(n_tgt, tile_rows);
  const int64_t cand = volume * tile_rows;
  const dim3 grid((unsigned)tiles), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ME_HP(NV)                                                                                                        \
  hipLaunchKernelGGL((k_halo_plan<NV>), grid, block, 0, st, tbl_dev, col_order_dev, src_pos_dev, src_order_dev, n_tgt, (int)volume, (int)tile_rows, \
                     (int)s_cap, halo_cnt_dev, halo_rows_dev, lidx_dev, kmask_dev)
  if (cand <= 256) ME_HP(256);
  else if (cand <= 512) ME_HP(512);
  else if (cand <= 1024) ME_HP(1024);
  else if (cand <= 2048) ME_HP(2048);
  else ME_HP(4096);
#undef ME_HP
  ME_LAUNCH_CHECK();
  return 0;
}

namespace me {
template <int T, int CB, int NWC, int NWR, int KC>
static int launch_halo(const HaloShape &hs, const __bf16 *src, int c_src, const bf16x8 *wp, int ksp, int nchp, int ncb,
                       int c_dst, const int32_t *halo_cnt, const int32_t *halo_rows, const uint16_t *lidx,
                       const uint32_t *kmask, const int32_t *tbl, const int32_t *col_order, const int32_t *out_order,
                       __bf16 *dst, int64_t n_tgt, int volume, float *stat_mean, float *stat_m2, hipStream_t stream) {
  constexpr int NT = 64 * NWC * NWR, NC = CB * 16 * NWC;
  const int lds = conv_halo_lds(T, NC, KC, hs.s_cap, volume, NT);
  typedef void (*kernel_t)(const __bf16 *, int, const bf16x8 *, int, int, int, int, const int32_t *, const int32_t *,
                           const uint16_t *, const uint32_t *, const int32_t *, const int32_t *, const int32_t *, __bf16 *,
                           int64_t, int, int, float *, float *);
  const HaloVariant hv = halo_variant(c_src, c_dst);
  const int wd = (hv.wd == 2 || hv.wd == 4) ? hv.wd : 2;
  const int which = !hv.skip ? 2 : (wd == 4 ? 1 : 0);
  kernel_t fn = which == 2 ? &k_conv_halo_bf16<T, CB, NWC, NWR, KC, false, 2>
                           : (which == 1 ? &k_conv_halo_bf16<T, CB, NWC, NWR, KC, true, 4> : &k_conv_halo_bf16<T, CB, NWC, NWR, KC, true, 2>);
  static bool attr_set[3] = {false, false, false};
  if (lds > 48 * 1024 && !attr_set[which]) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_set[which] = true;
  }
  ME_CHECK(lds <= kLdsBudget, "halo kernel: LDS budget");
  const dim3 grid((unsigned)ceil_div(n_tgt, T), (unsigned)ceil_div(c_dst, NC));
  hipLaunchKernelGGL(fn, grid, dim3(NT), (size_t)lds, stream, src, c_src, wp, ksp, nchp, ncb, c_dst, halo_cnt, halo_rows, lidx,
                     kmask, tbl, col_order, out_order, dst, n_tgt, volume, hs.s_cap, stat_mean, stat_m2);
  ME_LAUNCH_CHECK();
  return 0;
}
}  // namespace me

// dst[t, :] = sum over offsets k (ascending) of src[tbl[k][t], :] @ w[k]  — the operation of me_conv_target_bf16 on the
// halo plan of me_halo_plan_build (same tbl / col_order / tile_rows / s_cap).  packed_w_dev: the image of
// me_conv_pack_weights_bf16 for (c_src, c_dst).  part_mean_dev / part_m2_dev: as me_conv_target_bf16_stats (or NULL).
extern "C" int me_conv_halo_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const uint16_t *packed_w_dev,
                                 int64_t volume, int32_t c_dst, const int32_t *halo_cnt_dev, const int32_t *halo_rows_dev,
                                 const uint16_t *lidx_dev, const uint32_t *kmask_dev, const int32_t *tbl_dev,
                                 const int32_t *col_order_dev, const int32_t *out_order_dev, uint16_t *dst_feat_dev,
                                 int64_t n_tgt, int32_t tile_rows, int32_t s_cap, float *part_mean_dev,
                                 float *part_m2_dev, void *stream) {
  ME_CHECK(src_feat_dev && packed_w_dev && halo_cnt_dev && halo_rows_dev && lidx_dev && kmask_dev && tbl_dev && dst_feat_dev,
           "null argument");
  ME_CHECK((part_mean_dev == nullptr) == (part_m2_dev == nullptr), "statistics: both partial arrays or none");
  HaloShape hs;
  ME_CHECK(halo_shape(volume, c_src, c_dst, &hs), "no halo kernel for this shape");
  ME_CHECK(hs.t == tile_rows && hs.s_cap == s_cap, "the halo plan was built for another geometry");
  ME_CHECK(n_src < (1ll << 31), "halo kernel: source rows");
  if (n_tgt <= 0) return 0;
  const int kcp = me_conv_pack_chunk_bf16(c_src, c_dst);
  const int ksp = kcp / 32, nchp = (int)ceil_div(c_src, kcp), ncb = (int)ceil_div(c_dst, 16);
  const __bf16 *src = reinterpret_cast<const __bf16 *>(src_feat_dev);
  const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(packed_w_dev);
  __bf16 *dst = reinterpret_cast<__bf16 *>(dst_feat_dev);
  hipStream_t st = (hipStream_t)stream;
#define ME_HALO(TV, CBV, NWCV, NWRV, KCV)                                                                                  \
  if (hs.t == TV && hs.cb == CBV && hs.nwc == NWCV && hs.nwr == NWRV && hs.kc == KCV)                                       \
  return launch_halo<TV, CBV, NWCV, NWRV, KCV>(hs, src, c_src, wp, ksp, nchp, ncb, c_dst, halo_cnt_dev, halo_rows_dev,      \
                                               lidx_dev, kmask_dev, tbl_dev, col_order_dev, out_order_dev, dst, n_tgt,     \
                                               (int)volume, part_mean_dev, part_m2_dev, st)
#ifndef ME_DEBUG_VARIANTS
  ME_HALO(128, 4, 2, 2, 64);
  ME_HALO(64, 2, 1, 4, 64);
#else
#define ME_HALO_T(TV)            \
  ME_HALO(TV, 2, 4, 1, 32);      \
  ME_HALO(TV, 2, 4, 1, 64);      \
  ME_HALO(TV, 2, 4, 1, 96);      \
  ME_HALO(TV, 2, 4, 1, 128);     \
  ME_HALO(TV, 2, 3, 1, 32);      \
  ME_HALO(TV, 2, 3, 1, 64);      \
  ME_HALO(TV, 2, 3, 1, 96);      \
  ME_HALO(TV, 2, 2, 2, 32);      \
  ME_HALO(TV, 2, 2, 2, 64);      \
  ME_HALO(TV, 2, 1, 4, 32);      \
  ME_HALO(TV, 2, 1, 4, 64);      \
  ME_HALO(TV, 4, 2, 2, 64)
  ME_HALO_T(128);
  ME_HALO_T(64);
#undef ME_HALO_T
#endif
#undef ME_HALO
  ME_FAIL("no halo kernel instantiation for this shape");
}

// code-object preload (me_preload, coords.hip): resolving one kernel of this translation unit makes the runtime load the
// unit's whole code object now instead of at the first launch from it
extern "C" __attribute__((visibility("hidden"))) void me_preload_conv_halo(void) {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&me::k_halo_plan<256>));
}
