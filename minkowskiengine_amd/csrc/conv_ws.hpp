// Shared by the wave-specialised tile kernels (conv_f32x3.hip: fp32 features on the bf16 matrix pipe; conv_bf16_ws.hip:
// bf16 features): the LDS stage layout of a plane of gathered rows and the multiplier waves' batch step.
#pragma once
#include <type_traits>
#include "conv_common.hpp"

namespace me {

// LDS stage layout of one plane: 64 rows x KC bf16.  The LDS serves a ds_read_b128 in four fixed lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), i.e. 16 (row, 8-channel piece) accesses per cycle that must fall on 16
// distinct 16-byte bank slots.  For KC = 32 / 64 / 128 an XOR swizzle of the piece index by a function of the row does
// that without padding (checked exhaustively, scripts/check_stage_layout.py); KC = 96 uses rows padded by 32 bytes.
// (16 bytes of padding — the round-1 bf16 layout — left every group 2-way conflicting.)
template <int KC>
struct StageLayout {
  static constexpr bool kSwizzled = KC == 32 || KC == 64 || KC == 128;
  static constexpr int kLd = kSwizzled ? KC : KC + 16;   // bf16 elements per row
  __host__ __device__ static constexpr int swz(int row) {
    return KC == 128 ? (row & 15) : KC == 64 ? ((row >> 1) & 7) : KC == 32 ? ((row >> 1) & 3) : 0;
  }
  // element offset of the 8-channel piece u of stage row r
  __host__ __device__ static constexpr int off(int r, int u) { return r * kLd + ((u ^ swz(r)) * 8); }
};
__host__ __device__ constexpr int x3_stage_ld(int kc) { return (kc == 32 || kc == 64 || kc == 128) ? kc : kc + 16; }

// One batch in a multiplier wave: groups 0 .. RP0 - 1 (pass 0) and 2 .. 2 + RP1 - 1 (pass 1).  EVERY LDS operation of
// the batch is inline asm in a fixed order with counted waits: left to the compiler, the plain loads of the target
// indices and of the old accumulator values float between the MFMAs and bring `s_waitcnt lgkmcnt(0)` with them —
// 200 - 500 idle matrix cycles each time.  Order (the in-order lgkm counter is what the waits count against):
//   pass 0: indices, operands of step 0 | old accumulators | per step: next operands (last step: pass 1's indices and
//   first operands), wait, MFMAs | pass 1: old accumulators, stores of pass 0 | per step as before | stores.
// PL: operand planes — 3: fp32 rows split into three bf16 terms, six MFMAs per product (conv_f32x3.hip); 1: bf16 rows,
// one MFMA per product (conv_bf16_ws.hip, round 4).  The sums of a target row run over the steps of a batch in the same
// order either way.
template <int RP0, int RP1, int CB, int KC, int PL, int ABL = 0, typename NextWeights>
__device__ __forceinline__ void consume_batch_ws(const __bf16 *__restrict__ rowp, const int (&pofs)[KC / 32],
                                                 const bf16x8 (&w)[CB][PL][KC / 32],
                                                 const int32_t *__restrict__ dstp, float *__restrict__ accp,
                                                 int acc_ld, NextWeights &&load_next_weights) {
  static_assert(PL == 1 || PL == 3, "one plane (bf16 rows) or three (split fp32 rows)");
  typedef __attribute__((address_space(3))) const char lds_char;
  constexpr int KS = KC / 32;
  constexpr int LD = StageLayout<KC>::kLd;
  constexpr int PLANE = ME_MAX_BATCH_GROUPS * 16 * LD;
  constexpr int RPM = RP0 > RP1 ? RP0 : RP1;
  constexpr int R1 = RP1 > 0 ? RP1 : 1;
  const unsigned row_addr = (unsigned)(uintptr_t)(lds_char *)rowp;
  const unsigned dst_addr = (unsigned)(uintptr_t)(lds_char *)dstp;
  const unsigned acc_addr = (unsigned)(uintptr_t)(lds_char *)accp;
  bf16x8 a[2][RPM][PL];          // operands: steps alternate between the two halves
  bf16x8 a1[R1][PL];             // first operands of pass 1
  f32x4 acc[RPM][CB];
  unsigned op_addr[2][KS];      // byte address of this lane's operand piece: [pass][step]
#pragma unroll
  for (int sx = 0; sx < KS; ++sx) {
    op_addr[0][sx] = row_addr + (unsigned)pofs[sx] * 2u;
    op_addr[1][sx] = op_addr[0][sx] + (unsigned)(2 * 16 * LD * 2);
  }
  // ABL (timing ablations, results invalid): 1 = no MFMAs, 2 = no operand reads, 3 = no accumulator reads / stores
  auto read_ops = [](unsigned addr, auto n_, bf16x8 (*dst)[PL]) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
      for (int p = 0; p < PL; ++p) {
        if constexpr (ABL == 2) asm volatile("" : "=v"(dst[r][p]) : "v"(addr));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[r][p]) : "v"(addr), "n"((p * PLANE + r * 16 * LD) * 2) : "memory");
      }
    }
  };
  auto read_idx = [](unsigned addr, auto n_, int32_t *d) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r)
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[r]) : "v"(addr), "n"(r * 64) : "memory");
  };
  // wait until at most `younger` LDS operations are outstanding; the listed registers are tied to the wait
  auto wait_idx = [](auto n_, int32_t *d, auto younger) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d[r]) : "n"(decltype(younger)::value) : "memory");
  };
  auto wait_ops = [](auto n_, bf16x8 (*dst)[PL], f32x4 (*ac)[CB], auto younger) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      if constexpr (PL == 3)
        asm volatile("s_waitcnt lgkmcnt(%4)"
                     : "+v"(dst[r][0]), "+v"(dst[r][1]), "+v"(dst[r][2]), "+v"(ac[r][0])
                     : "n"(decltype(younger)::value)
                     : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(dst[r][0]), "+v"(ac[r][0]) : "n"(decltype(younger)::value) : "memory");
    }
  };
  auto read_old = [](auto n_, const unsigned *addr, f32x4 (*old)[CB]) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        if constexpr (ABL == 3) asm volatile("" : "=v"(old[r][c]) : "v"(addr[r]));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(old[r][c]) : "v"(addr[r]), "n"(c * 64) : "memory");
      }
    }
  };
  auto store_acc = [](auto n_, const unsigned *addr, f32x4 (*old)[CB], f32x4 (*ac)[CB]) {
    constexpr int N = decltype(n_)::value;
#pragma unroll
    for (int r = 0; r < N; ++r) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old[r][c]) : : "memory");   // (long since arrived; ties the value)
        const f32x4 v = old[r][c] + ac[r][c];
        if constexpr (ABL == 3) asm volatile("" : : "v"(addr[r]), "v"(v));
        else asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr[r]), "v"(v), "n"(c * 64) : "memory");
      }
    }
  };
  auto mfmas = [&w](auto n_, auto s_, bf16x8 (*ops)[PL], f32x4 (*ac)[CB]) {
    constexpr int N = decltype(n_)::value;
    constexpr int S = decltype(s_)::value;
    constexpr int NT_ = PL == 3 ? 6 : 1;
    constexpr int WP[6] = {PL == 3 ? 2 : 0, 0, PL == 3 ? 1 : 0, PL == 3 ? 1 : 0, 0, 0};   // (weight plane, row plane) by
    constexpr int AP[6] = {0, PL == 3 ? 2 : 0, PL == 3 ? 1 : 0, 0, PL == 3 ? 1 : 0, 0};   //  ascending magnitude
#pragma unroll
    for (int t = 0; t < NT_; ++t) {
#pragma unroll
      for (int c = 0; c < CB; ++c) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
          if constexpr (ABL == 1) asm volatile("" : "+v"(ac[r][c]) : "v"(w[c][WP[t]][S]), "v"(ops[r][AP[t]]));
          else ac[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][WP[t]][S], ops[r][AP[t]], ac[r][c], 0, 0, 0);
        }
      }
    }
  };
  using IC0 = std::integral_constant<int, RP0>;
  using IC1 = std::integral_constant<int, RP1>;
  auto cap = [](int v) constexpr { return v > 15 ? 15 : v; };
  // ---- pass 0 ----
  int32_t d0[RP0], d1[R1];
  unsigned addr0[RP0], addr1[R1];
  f32x4 old0[RP0][CB], old1[R1][CB];
  read_idx(dst_addr, IC0{}, d0);
  read_ops(op_addr[0][0], IC0{}, a[0]);
  // the next batch's weight loads (12 vector-memory issues) go out here: their issue time hides behind the LDS
  // round trip that the first MFMA has to wait for anyway, instead of standing in front of it
  load_next_weights();
  wait_idx(IC0{}, d0, std::integral_constant<int, PL * RP0>{});
#pragma unroll
  for (int r = 0; r < RP0; ++r) addr0[r] = acc_addr + __umul24((unsigned)d0[r], (unsigned)acc_ld) * 4u;
  read_old(IC0{}, addr0, old0);
#pragma unroll
  for (int r = 0; r < RP0; ++r) {
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto step0 = [&](auto s_) {
    constexpr int S = decltype(s_)::value;
    constexpr int kOld = S == 0 ? RP0 * CB : 0;
    if constexpr (S + 1 < KS) {
      read_ops(op_addr[0][S + 1 < KS ? S + 1 : 0], IC0{}, a[(S + 1) & 1]);
      wait_ops(IC0{}, a[S & 1], acc, std::integral_constant<int, cap(kOld + PL * RP0)>{});
    } else if constexpr (RP1 > 0) {
      read_idx(dst_addr + 2 * 64, IC1{}, d1);   // last step of pass 0: pass 1's indices and first operands go out now
      read_ops(op_addr[1][0], IC1{}, a1);
      wait_ops(IC0{}, a[S & 1], acc, std::integral_constant<int, cap(kOld + (PL + 1) * RP1)>{});
    } else {
      wait_ops(IC0{}, a[S & 1], acc, std::integral_constant<int, kOld>{});
    }
    mfmas(IC0{}, s_, a[S & 1], acc);
  };
  step0(std::integral_constant<int, 0>{});
  if constexpr (KS > 1) step0(std::integral_constant<int, 1>{});
  if constexpr (KS > 2) step0(std::integral_constant<int, 2>{});
  if constexpr (KS > 3) step0(std::integral_constant<int, 3>{});
  if constexpr (RP1 == 0) {
    store_acc(IC0{}, addr0, old0, acc);
  } else {
    // ---- pass 1 (groups 2 ..: other rows than pass 0 — the groups of a batch share their offset) ----
    wait_idx(IC1{}, d1, std::integral_constant<int, PL * RP1>{});
#pragma unroll
    for (int r = 0; r < RP1; ++r) addr1[r] = acc_addr + __umul24((unsigned)d1[r], (unsigned)acc_ld) * 4u;
    read_old(IC1{}, addr1, old1);
    {
      // stores of pass 0 (its old values arrived before pass 1's indices); no wait inside: counted below
#pragma unroll
      for (int r = 0; r < RP0; ++r) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          asm volatile("" : "+v"(old0[r][c]));
          const f32x4 v = old0[r][c] + acc[r][c];
          if constexpr (ABL == 3) asm volatile("" : : "v"(addr0[r]), "v"(v));
          else asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr0[r]), "v"(v), "n"(c * 64) : "memory");
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RP1; ++r) {
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto step1 = [&](auto s_) {
      constexpr int S = decltype(s_)::value;
      bf16x8(*cur)[PL] = S == 0 ? a1 : a[S & 1];
      constexpr int kFirst = S == 0 ? (RP1 + RP0) * CB : 0;   // pass 1's old reads + pass 0's stores, issued behind a1
      if constexpr (S + 1 < KS) {
        read_ops(op_addr[1][S + 1 < KS ? S + 1 : 0], IC1{}, a[(S + 1) & 1]);
        wait_ops(IC1{}, cur, acc, std::integral_constant<int, cap(kFirst + PL * RP1)>{});
      } else {
        wait_ops(IC1{}, cur, acc, std::integral_constant<int, cap(kFirst)>{});
      }
      mfmas(IC1{}, s_, cur, acc);
    };
    step1(std::integral_constant<int, 0>{});
    if constexpr (KS > 1) step1(std::integral_constant<int, 1>{});
    if constexpr (KS > 2) step1(std::integral_constant<int, 2>{});
    if constexpr (KS > 3) step1(std::integral_constant<int, 3>{});
    store_acc(IC1{}, addr1, old1, acc);
  }
}


// One SUPER-BATCH in a multiplier wave (sparse maps: multi-offset batches, see k_conv_tile_bf16's batch fusion): either
// `nsub` > 1 single-group batches of DIFFERENT offsets staged together — group r multiplies with its own offset's weights
// w[r] and the accumulator tile is updated in batch order (a later group may hit a row of an earlier one; LDS operations of a
// wave execute in order) — or one batch of `ng` groups of one offset (weights w[0]; its rows are distinct).  `refill` (the
// caller requests the next super-batch's weights into its other register set) first, then per 32-channel step the operand
// reads and the MFMAs, then the accumulator updates.  Plain loads and stores: the compiler schedules this path (the multi-group batches of dense maps
// take consume_batch_ws).  Same sums in the same order as one consume_batch_ws per batch.
template <int MAXSUB, int CB, int KC, int PL, typename Refill>
__device__ __forceinline__ void consume_super_ws(const __bf16 *__restrict__ rowp, const int (&pofs)[KC / 32],
                                                 const bf16x8 (&w)[MAXSUB][CB][PL][KC / 32], int nsub, int ng,
                                                 const int32_t *__restrict__ dstp, float *__restrict__ accp, int acc_ld,
                                                 Refill &&refill) {
  constexpr int KS = KC / 32;
  constexpr int LD = StageLayout<KC>::kLd;
  constexpr int PLANE = ME_MAX_BATCH_GROUPS * 16 * LD;
  constexpr int G = ME_MAX_BATCH_GROUPS;
  constexpr int NT_ = PL == 3 ? 6 : 1;
  constexpr int WP[6] = {PL == 3 ? 2 : 0, 0, PL == 3 ? 1 : 0, PL == 3 ? 1 : 0, 0, 0};
  constexpr int AP[6] = {0, PL == 3 ? 2 : 0, PL == 3 ? 1 : 0, 0, PL == 3 ? 1 : 0, 0};
  int d[G];
  f32x4 acc[G][CB];
#pragma unroll
  for (int r = 0; r < G; ++r) {
    d[r] = (int)__umul24((unsigned)dstp[r * 16], (unsigned)acc_ld);
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  refill();         // (the next super-batch's weights, into the caller's OTHER register set — or nothing)
  // operands of ONE 32-channel step at a time (all groups, all planes): G * PL registers of 16 bytes, whatever KC
  if (nsub > 1) {   // wave-uniform
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      bf16x8 a[MAXSUB][PL];
#pragma unroll
      for (int r = 0; r < MAXSUB; ++r) {
#pragma unroll
        for (int p = 0; p < PL; ++p) a[r][p] = *reinterpret_cast<const bf16x8 *>(rowp + p * PLANE + r * 16 * LD + pofs[sx]);
      }
#pragma unroll
      for (int t = 0; t < NT_; ++t) {
#pragma unroll
        for (int r = 0; r < MAXSUB; ++r) {
          if (r < nsub) {
#pragma unroll
            for (int c = 0; c < CB; ++c)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[r][c][WP[t]][sx], a[r][AP[t]], acc[r][c], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < MAXSUB; ++r) {
      if (r < nsub) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          float *o = accp + d[r] + c * 16;
          const f32x4 old = *reinterpret_cast<const f32x4 *>(o);
          *reinterpret_cast<f32x4 *>(o) = old + acc[r][c];
        }
      }
    }
  } else {
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      bf16x8 a[G][PL];
#pragma unroll
      for (int r = 0; r < G; ++r) {
#pragma unroll
        for (int p = 0; p < PL; ++p) a[r][p] = *reinterpret_cast<const bf16x8 *>(rowp + p * PLANE + r * 16 * LD + pofs[sx]);
      }
#pragma unroll
      for (int t = 0; t < NT_; ++t) {
#pragma unroll
        for (int r = 0; r < G; ++r) {
          if (r < ng) {
#pragma unroll
            for (int c = 0; c < CB; ++c)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0][c][WP[t]][sx], a[r][AP[t]], acc[r][c], 0, 0, 0);
          }
        }
      }
    }
    f32x4 old[G][CB];
#pragma unroll
    for (int r = 0; r < G; ++r) {
#pragma unroll
      for (int c = 0; c < CB; ++c) old[r][c] = *reinterpret_cast<const f32x4 *>(accp + d[r] + c * 16);
    }
#pragma unroll
    for (int r = 0; r < G; ++r) {
      if (r < ng) {
#pragma unroll
        for (int c = 0; c < CB; ++c) *reinterpret_cast<f32x4 *>(accp + d[r] + c * 16) = old[r][c] + acc[r][c];
      }
    }
  }
}

}  // namespace me
