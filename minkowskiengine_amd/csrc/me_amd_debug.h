/* Test and tuning hooks of libme_amd.so.  NOT part of the drop-in boundary (include/me_amd.h): these switches are
 * process-global, not thread-safe, and exist so that tests/ can compare alternate kernels that must agree bit for
 * bit (64-bit addresses, ping-pong vs wave-specialised split kernels, weight-gradient pipes) and scripts/ can time
 * ablations.  A host integrator never calls them. */
#ifndef ME_AMD_DEBUG_H
#define ME_AMD_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* 1 when the library was built with -DME_DEBUG_VARIANTS (phase counters, timing ablations whose RESULTS ARE INVALID,
 * the LDS-DMA experiment family); the default build does not contain those kernels. */
int32_t me_debug_variants_compiled(void);
/* Kernel selection for me_conv_target_* (0 = shipped).  Default build: 6 (64-bit gather addresses), 7 (no batch
 * fusion), 9 (32-wide passes for 96 channels), 30 / 31 (ping-pong / four-multiplier split kernels) — all with valid,
 * bit-identical results; any other code returns an error unless me_debug_variants_compiled(). */
int me_debug_set_conv_variant(int variant);
/* Phase cycle counters of the instrumented builds (variants 256 / 257, ME_DEBUG_VARIANTS only): barrier A, stage
 * write + wait, barrier B, load issue, multiply, prologue, epilogue (s_memtime cycles summed over wave 0 of every
 * workgroup), batches. */
int me_debug_conv_timing(uint64_t *out8, int32_t reset);
int me_debug_conv_timing_f32x3(uint64_t *out8, int32_t reset);
/* bf16 tile kernel: columns per workgroup (32 / 64 / 128) and source-channel chunk (32 / 64 / 96 / 128 / 256) instead
 * of the policy of conv_variant_bf16 (0 = policy); plans and packed weights follow (set it before the first call). */
void me_debug_set_bf16_shape(int nc, int kc);
/* deep (two batches ahead) pipeline of the eight-wave bf16 tile kernels: -1 policy, 0 never, 1 wherever instantiated */
void me_debug_set_bf16_deep(int deep);
/* two stage buffers + one barrier per batch in the deep-pipeline bf16 tile kernels: -1 policy, 0 never, 1 wherever the
 * deep pipeline runs and the LDS holds the second buffer (bit-identical results) */
void me_debug_set_bf16_twobuf(int mode);
// the wave-specialised bf16 tile kernel (conv_bf16_ws.hip): -1 policy (default), 0 never, 1 wherever instantiated
// fp32 multi-offset launches (sparse maps) on the bf16 pipe (conv_f32x3_fused.hip): 1 = where instantiated (tuning build), 0 never (default)
void me_debug_set_f32_fused_split(int mode);
void me_debug_set_bf16_ws(int mode);
void me_debug_set_bf16_ws_fuse(int mode);    // 1: multi-offset batches (sparse maps) on the wave-specialised kernel too (tuning build; default 0)
void me_debug_set_bf16_ws_depth(int depth);
void me_debug_set_bf16_ws_ncw(int ncw);       // 0 policy | 4 | 8 multiplier waves of the wave-specialised kernel on 128-column slabs
// phase counters of k_conv_tile_bf16_ws in a -DME_WS_TIMING build (see conv_bf16_ws.hip); zeros otherwise
int me_debug_ws_timing(uint64_t *out8, int32_t reset);   // 2 | 4 register sets of gathered rows per producer thread (default 4)
/* bf16 forward / dgrad schedule: 0 column-split k_conv_tile_bf16, 1 offset-synchronous k_conv_off_bf16 where eligible
 * (row-split waves, weights through LDS, one barrier per offset; bit-identical results), 2: its other wave shape.
 * Changes the pack chunk of 256-channel layers: set it before weights are packed and plans are made. */
void me_debug_set_bf16_offsync(int mode);
/* split-K of the eight-wave bf16 tile kernels: -1 policy, 0 / 1 never, G = 2..8: G offset groups wherever eligible
 * (set it before the plans are made: me_conv_plan_config_bf16_ex answers with it) */
void me_debug_set_bf16_splitk(int g);
/* 1: forced offset groups keep the tile height of the unsplit plan (G times the workgroups: occupancy instead of weight
 * bytes); 0 (default): G-times taller tiles */
void me_debug_set_bf16_splitk_mode(int same_tiles);
/* phase counters of k_conv_tile_bf16 (all zero unless the library was compiled with -DME_BF16_TIMING): 20 uint64,
 * slots 0-9 deep pipeline / 10-19 plain loop: barrier A, stage write + wait, barrier B, load issue, multiply, refill +
 * descriptors, prologue, epilogue (s_memtime cycles of wave 0 of every workgroup), batches, workgroups */
int me_debug_bf16_timing(uint64_t *out20, int32_t reset);
/* Weight-gradient kernels: depth 4 / 8 = prefetch ring depth of k_wgrad_f32 (steps of 4 pairs); depth -1 = bf16 rows
 * through the fp32-MFMA kernel instead of k_wgrad_bf16; -2 = fp32 rows through the LDS-staged kernel; -3 / -4 = fp32
 * rows through the fp32-MFMA / the split kernel; wgs_per_cu = resident workgroups per CU the ranges are sized for;
 * 0 = shipped defaults. */
void me_debug_set_wgrad_config(int depth, int wgs_per_cu);
/* 0 (default): ranges of the same list fraction go to the same XCD (WgRangeOrder, conv.hip); -1: launch order */
void me_debug_set_wgrad_order(int mode);
// dispatch order of the tiles of plans built from now on: 0 = heaviest first (default), 1 = contiguous tile chunks per XCD
void me_debug_set_tile_dispatch(int mode);
/* k_wgrad_bf16: input-channel blocks of 16 per workgroup — 0 policy (8 where c_in >= 192 and c_out > 64), 4, 8 */
void me_debug_set_wgrad_mb(int mb);
// bf16 weight gradient: 0 = k_wgrad_bf16 (default), 1 / 2 = the wave-specialised kernel with four / two row register sets (tuning build)
void me_debug_set_wgrad_ws(int mode);

/* halo kernel (conv_halo.hip): mode -1 policy / 0 never / 1 wherever instantiated; tile_rows 0 | 64 | 128; kc 0 | 32 | 64 |
 * 96 | 128 (channels staged per pass); skip 0 = multiply every 16-row group.  Set before plans are made. */
void me_debug_set_halo(int mode, int tile_rows, int kc, int skip);
int32_t me_debug_halo_mode(void);
int me_debug_halo_timing(uint64_t *out8, int32_t reset);   /* phase counters of a -DME_HALO_TIMING build */
/* stacked-offset kernel (conv_stem.hip): -1 policy / 0 never / 1 wherever the shape is supported */
void me_debug_set_stem(int mode, int groups);   /* groups: 16-row groups per wave, 0 policy | 1 | 2 | 4 */
/* insert_and_map: 1 (default) = fused resolve / rank / emit kernels (round 6), 0 = resolve + scan + finalize + bbox */
void me_debug_set_insert_fused(int on);
/* row-wise kernel (conv_rowwise.hip): 16-row groups per wave, 0 policy | 1 | 2 */
void me_debug_set_rowwise_groups(int groups);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
