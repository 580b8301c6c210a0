/*
 * me_amd.h — C ABI of the MI355X (gfx950) sparse-convolution hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / pybind types.  Every pointer
 * named *_dev is DEVICE memory owned by the caller (in the PyTorch-ROCm host layer these are
 * tensor.data_ptr() values, so the torch caching allocator plays the role of the reference's
 * c10 allocator, src/allocators.cuh:74-103).  `stream` is a hipStream_t passed as void*; all
 * work is enqueued on it and nothing synchronises the device except the functions documented
 * as "SYNC" (they must return a count to the host, exactly where the reference synchronises).
 *
 * Each entry point cites the reference interface (path:line under the MinkowskiEngine tree) that
 * it replaces.  The reference reaches those through pybind11 (pybind/extern.hpp); INTEGRATION.md
 * shows the binding a maintainer would add.
 *
 * Return value: 0 on success, non-zero on error; me_last_error() describes the last failure of
 * the calling thread (mirrors the reference's ASSERT -> std::runtime_error, src/utils.hpp:141-150).
 */
#ifndef ME_AMD_H
#define ME_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libme_amd.so is built with -fvisibility=hidden: the entry points declared in this header (and the test hooks of
 * csrc/me_amd_debug.h) are its ONLY exported symbols — no C++ internals, no device stubs. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define ME_MAX_DIM 7 /* spatial dimensions D; coordinates carry D+1 int32 (batch index first) */

/* region types: src/types.hpp:148-152 (RegionType::HYPER_CUBE / HYPER_CROSS / CUSTOM) */
#define ME_REGION_HYPER_CUBE 0
#define ME_REGION_HYPER_CROSS 1

/* Kernel geometry of one layer; replaces gpu_kernel_region (src/kernel_region.hpp:283-398).
 * Neighbour offsets follow kernel_region.hpp:198-247 (axis 0 fastest; odd sizes centred, even
 * sizes start at 0; multiplied by dilation * tensor_stride). */
typedef struct me_region {
  int32_t ncol;                    /* D + 1 */
  int32_t region_type;             /* ME_REGION_* */
  int32_t kernel_size[ME_MAX_DIM]; /* per spatial axis */
  int32_t dilation[ME_MAX_DIM];
  int32_t tensor_stride[ME_MAX_DIM]; /* tensor stride of the map that is LOOKED UP (the "in" map) */
} me_region;

/* ---- library ------------------------------------------------------------------------------- */
/* 100 * major + 10 * minor.  Changelog:
 *   1.0 (100)  round 1: coordinate maps, kernel maps, tile plans, fp32 / bf16 convolution, pooling
 *   1.1        round 2: tile_bptr_dev grew to 2 * num_tiles + 1 int32 (dispatch order behind the batch ranges) — an
 *              ABI change without a size argument; LDS-bucketed kernel-map build; split fp32 kernels
 *   1.2 (120)  round 3: me_plan_tile_bptr_elems sizes that buffer; me_debug_* hooks left this header
 *              (csrc/me_amd_debug.h: tests / tuning only)
 *   1.3 (130)  round 3: batch-norm statistics in the convolution's epilogue (me_conv_target_bf16_stats,
 *              me_conv_stats_supported_bf16, me_bn_stats_from_tiles)
 *   1.4 (140)  round 4: split-K launches of the bf16 convolution for small coordinate maps
 *              (me_conv_plan_config_bf16_ex, me_conv_splitk_workspace_bytes, me_conv_target_bf16_ex); float64
 *              features (me_conv_target_f64, me_conv_wgrad_f64, me_pool_*_f64, me_global_pool_f64, me_broadcast_f64);
 *              all tile plans of a scene in four launches (me_plan_job, me_plan_jobs_init, me_plan_build_multi)
 *   1.5 (150)  round 5: bf16 convolution on an LDS-staged source halo with register accumulators
 *              (me_conv_halo_config_bf16, me_halo_plan_build, me_conv_halo_bf16); stacked-offset kernel for layers
 *              with at most 8 source channels (me_conv_stem_use_bf16, me_conv_stem_tile_rows, me_conv_stem_bf16)
 *   1.6 (160)  round 6: row-wise launches for kernel-map sides with exactly one pair per target row — K = 1 layers
 *              (the reference's `input.F.mm(kernel)`) and the fine side of kernel_size == stride maps
 *              (me_conv_rowwise_supported_bf16, me_conv_rowwise_bf16); Z-order of a map by the library's own radix sort
 *              (me_coords_zorder) */
int me_version(void);
const char *me_last_error(void);
/* Load the device code of every translation unit of the library now (needs a GPU; ABI 1.5): HIP loads a unit's code object
 * at the first launch from it, which put 88 ms into the first backward pass of a process.  The hosts call it at the first
 * map insert on a device, under that device's guard (code objects are per device) — not at import. */
int me_preload(void);
/* kernel volume of a region: src/kernel_region.hpp:250-270 (set_volume) */
int64_t me_region_volume(const me_region *region);

/* ---- coordinate hash map (replaces CoordinateMapGPU, src/coordinate_map_gpu.cuh:47-223) ------ */

/* Number of 8-byte slots for n keys: power of two >= 2n (<= 50 % load; the reference uses 25-50 %,
 * src/coordinate_map_manager.hpp:139-155). */
int64_t me_hash_capacity(int64_t n);
/* Scratch bytes needed by me_coords_insert_and_map for n rows. */
int64_t me_insert_workspace_bytes(int64_t n);

/* insert + dedup + row maps.  Replaces CoordinateMapGPU::insert<true>
 * (src/coordinate_map_gpu.cu:196-278) but with the CPU path's deterministic semantics
 * (CoordinateMapCPU::insert_and_map, src/coordinate_map_cpu.hpp:353-380): the FIRST occurrence of
 * a coordinate wins and unique rows keep first-occurrence order.
 *   coords_dev      int32 [n, ncol] contiguous (16-byte aligned when ncol == 4)
 *   table_dev       uint64 [capacity]          (out) open-addressing table {hash tag, row}
 *   coords_unique   int32 [n, ncol]            (out) first n_unique rows valid
 *   unique_map_dev  int64 [n]                  (out) first n_unique valid: input row of each unique row
 *   inverse_map_dev int64 [n]                  (out) unique row of each input row
 *   n_unique        host int64                 (out)  — SYNC (one 8-byte D2H copy)
 */
int me_coords_insert_and_map(const int32_t *coords_dev, int64_t n, int32_t ncol, uint64_t *table_dev,
                             int64_t capacity, int32_t *coords_unique_dev, int64_t *unique_map_dev,
                             int64_t *inverse_map_dev, int64_t *n_unique, void *workspace_dev,
                             int64_t workspace_bytes, void *stream);

/* out[i] = floor(c / ts_out) * ts_out on the spatial columns, batch column copied.
 * Replaces stride_copy (src/coordinate_map_gpu.cu:363-397; CPU detail::stride_coordinate,
 * src/coordinate_map.hpp:58-66).  Integer floor division (== the reference's float floor for
 * |c| < 2^24). */
int me_coords_stride(const int32_t *coords_dev, int64_t n, int32_t ncol,
                     const int32_t *out_tensor_stride /* host [ncol-1] */, int32_t *out_coords_dev,
                     void *stream);

/* keys[i] = Z-order (Morton) key of coordinate row i: batch index on top, then the bit-interleaved
 * spatial coordinates in units of the tensor stride.  Sorting target rows by this key gives the tile
 * plan spatially compact tiles (their gathers then hit the L2 instead of HBM).  No reference
 * counterpart (the reference never reorders work); the row order of the maps is NOT changed. */
int me_coords_spatial_keys(const int32_t *coords_dev, int64_t n, int32_t ncol,
                           const int32_t *tensor_stride /* host [ncol-1] */, int64_t *keys_dev, void *stream);
/* Rows in Z-order (round 6): order_dev int32 [n] = the STABLE argsort of me_coords_spatial_keys' keys, by the library's
 * own LSD radix sort (the reference sorts with thrust on its map path, src/coordinate_map_gpu.cu:766-772).  bbox: host
 * ints [2 * ncol], column minima then maxima of the rows as me_coords_insert_and_map_bbox returns them, or NULL (then
 * every key byte is sorted: eight passes instead of three for a 70^3 scene).  The tile plans of the fp32 kernels on
 * row-space tables and the halo plans are cut from this order. */
int64_t me_coords_zorder_workspace_bytes(int64_t n);
int me_coords_zorder(const int32_t *coords_dev, int64_t n, int32_t ncol, const int32_t *tensor_stride,
                     const int32_t *bbox, int32_t *order_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);

/* rows[q] = row of query q in the map, or -1.  Replaces CoordinateMapGPU::find
 * (src/coordinate_map_gpu.cu:284-361). */
int me_coords_find(const uint64_t *table_dev, int64_t capacity, const int32_t *map_coords_dev,
                   int32_t ncol, const int32_t *queries_dev, int64_t nq, int32_t *rows_dev,
                   void *stream);

/* ---- kernel map (replaces CoordinateMapGPU::kernel_map, src/coordinate_map_gpu.cu:1546-1745, and
 *      gpu_kernel_map::decompose, src/kernel_map.cuh:313-405) ----------------------------------- */

/* Scratch bytes for the two kernel-map passes. */
int64_t me_kernel_map_workspace_bytes(int64_t n_out, int64_t volume);

/* Pass 1: neighbour table + per-offset pair counts.
 *   nbr_dev   int32 [volume, n_out] (out): in-map row of (out row u, offset k) or -1
 *   k_offsets host int64 [volume + 1] (out): exclusive prefix of pair counts — SYNC.  NULL (ABI 1.2): no read-back and
 *             no synchronisation; k_offsets_dev is then required and the caller allocates the pair lists at their
 *             upper bound n_out * volume (me_kernel_map_compact writes only the first k_offsets[volume] entries)
 *   k_offsets_dev int64 [volume + 1] on the device (out, may be NULL): the same prefix, for the kernels that
 *             take it as a device array (transpose, wgrad) — saves the caller a host-to-device copy
 * The iteration direction is the reference's: iterate OUTPUT coordinates, look up the INPUT map
 * (src/coordinate_map_cpu.hpp:626-649). */
int me_kernel_map_probe(const uint64_t *in_table_dev, int64_t in_capacity,
                        const int32_t *in_coords_dev, const int32_t *out_coords_dev, int64_t n_out,
                        const me_region *region, int32_t *nbr_dev, int64_t *k_offsets, int64_t *k_offsets_dev,
                        void *workspace_dev, int64_t workspace_bytes, void *stream);

/* Pass 2: wavefront ballot/prefix-sum compaction of the table into the reference's per-offset
 * pair lists (kernel_map.hpp:40-53): pairs of offset k live at [k_offsets[k], k_offsets[k+1]),
 * sorted by output row (deterministic).  Must follow me_kernel_map_probe with the same workspace. */
int me_kernel_map_compact(const int32_t *nbr_dev, int64_t n_out, int64_t volume,
                          int32_t *in_pairs_dev, int32_t *out_pairs_dev, void *workspace_dev,
                          int64_t workspace_bytes, void *stream);

/* Transposed neighbour table: nbrT[k, i] = out row paired with in row i under offset k, or -1.
 * (each (k, in row) occurs in at most one pair).   k_offsets_dev: int64 [volume+1] on device. */
int me_kernel_map_transpose(const int32_t *in_pairs_dev, const int32_t *out_pairs_dev,
                            const int64_t *k_offsets_dev, int64_t volume, int64_t n_pairs,
                            int64_t n_in, int32_t *nbrT_dev, void *stream);

/* ---- spatial index + LDS-bucketed kernel map (round 2) ---------------------------------------------------------
 * The north_star's "LDS-bucketed open-address hashing with coalesced HBM reads": the bucket of a kernel-map probe
 * is SPATIAL.  A map's rows are ordered by the supercell that contains them (<= 4096 cells: 16^3 for D = 3); a
 * workgroup owns one supercell of the query map, stages the rows of the 3^D neighbouring supercells of the lookup
 * map (contiguous ranges of its sorted coordinate array: coalesced reads, the only HBM reads of the build) into a
 * dense halo grid in LDS and answers all rows x volume probes from LDS; the neighbour table is written in POSITION
 * space (position = rank in supercell order; `order` maps positions back to rows), where a supercell is one
 * contiguous, coalesced range.  Replaces CoordinateMapGPU::kernel_map (src/coordinate_map_gpu.cu:1546-1745), whose
 * probes each walk the global table (src/3rdparty/concurrent_unordered_map.cuh:304-360). */
typedef struct me_spatial_grid {
  int32_t ncol;                        /* D + 1 */
  int32_t shift[ME_MAX_DIM];           /* log2 of the supercell side (cells of one tensor stride) per axis */
  int32_t sc_min[ME_MAX_DIM + 1];      /* [0] smallest batch index, [1 + d] smallest supercell coordinate of axis d */
  int32_t sc_dim[ME_MAX_DIM + 1];      /* extents of the dense supercell directory: batch indices, supercells per axis */
  int32_t tensor_stride[ME_MAX_DIM];   /* cell size */
} me_spatial_grid;

/* me_coords_insert_and_map that also returns the bounding box of the coordinates (host int32 [2 * ncol]: the minima
 * of the ncol columns, then the maxima) on the SAME read-back: it sizes the supercell directory without a
 * synchronisation of its own. */
int me_coords_insert_and_map_bbox(const int32_t *coords_dev, int64_t n, int32_t ncol, uint64_t *table_dev,
                                  int64_t capacity, int32_t *coords_unique_dev, int64_t *unique_map_dev,
                                  int64_t *inverse_map_dev, int64_t *n_unique, int32_t *bbox /* host, may be NULL */,
                                  void *workspace_dev, int64_t workspace_bytes, void *stream);

/* number of supercells m of the directory (product of sc_dim), -1 if invalid / too large */
int64_t me_spatial_cells(const me_spatial_grid *grid);
int64_t me_spatial_index_workspace_bytes(int64_t n, int64_t m);
/* order [n] (position -> row), pos_of_row [n], coords_sorted [n, ncol] (rows in position order), dir_start uint32
 * [m + 1] (first position of every supercell).  One key pass, ceil(log2 m / 8) stable radix passes: no host sync. */
int me_spatial_index_build(const int32_t *coords_dev, int64_t n, const me_spatial_grid *grid, int32_t *order_dev,
                           int32_t *pos_of_row_dev, int32_t *coords_sorted_dev, uint32_t *dir_start_dev,
                           void *workspace_dev, int64_t workspace_bytes, void *stream);

/* LDS bytes of the probe for this region and map pair, or -1 when it is not eligible (maps of different tensor
 * stride or supercell side, offsets reaching beyond one supercell, halo grid larger than the LDS): the caller then
 * uses me_kernel_map_probe. */
int64_t me_kernel_map_probe_lds_bytes(const me_region *region, const me_spatial_grid *query_grid,
                                      const me_spatial_grid *lookup_grid);
/* nbr_pos int32 [volume, n_q] (out): ROW of the lookup map paired with (offset k, query POSITION p), or -1.
 * k_offsets_dev int64 [volume + 1] (out, device; may be NULL): exclusive prefix of the per-offset pair counts — the
 * counts are taken inside the probe (wave ballots) and scanned into the workspace (me_kernel_map_workspace_bytes),
 * which then feeds me_kernel_map_compact_ordered.  NO host synchronisation: copy k_offsets_dev asynchronously and
 * read it when a host value is first needed. */
int me_kernel_map_probe_lds(const me_spatial_grid *query_grid, const int32_t *q_coords_sorted_dev,
                            const uint32_t *q_dir_start_dev, int64_t n_q, const me_spatial_grid *lookup_grid,
                            const int32_t *l_coords_sorted_dev, const int32_t *l_order_dev,
                            const uint32_t *l_dir_start_dev, const me_region *region, int32_t *nbr_pos_dev,
                            int64_t *k_offsets_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);
/* Per-offset pair counts of an existing neighbour table (+ their prefix in the workspace for the compaction):
 * k_offsets_dev int64 [volume + 1] on the device; k_offsets (host, may be NULL: then NO synchronisation — copy
 * k_offsets_dev asynchronously and read it when first needed). */
int me_kernel_map_count(const int32_t *nbr_dev, int64_t n_out, int64_t volume, int64_t *k_offsets /* host or NULL */,
                        int64_t *k_offsets_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);
/* me_kernel_map_compact for a position-space table: the target row of position p is order[p] (NULL: p itself). */
int me_kernel_map_compact_ordered(const int32_t *nbr_dev, const int32_t *order_dev, int64_t n_out, int64_t volume,
                                  int32_t *in_pairs_dev, int32_t *out_pairs_dev, void *workspace_dev,
                                  int64_t workspace_bytes, void *stream);
/* me_kernel_map_transpose into the POSITION space of the in map: nbrT[k, pos_in[in row]] = out row (pos_in NULL: the
 * row itself).  n_pairs_bound: any upper bound of the pair count (launch geometry only). */
int me_kernel_map_transpose_ordered(const int32_t *in_pairs_dev, const int32_t *out_pairs_dev,
                                    const int64_t *k_offsets_dev, int64_t volume, int64_t n_pairs_bound, int64_t n_in,
                                    const int32_t *pos_in_dev, int32_t *nbrT_dev, void *stream);

/* ---- tile plan for the target-stationary convolution ----------------------------------------- */
/* A plan cuts the target rows into tiles of `tile_rows` consecutive rows (any value in
 * [ME_GROUP_ROWS, ME_MAX_TILE_ROWS]; me_conv_plan_config picks it so that tiles x column slabs
 * fill the GPU's workgroup slots evenly).  Per tile and kernel offset k ("item") the valid
 * (k, source row) entries are padded to groups of 16 (one MFMA tile) and the groups are cut into
 * batches of at most `batch_groups` groups (dealt evenly: 5 groups -> 3 + 2); a batch is what the convolution
 * kernel stages in LDS at once and never mixes offsets. */
#define ME_GROUP_ROWS 16
#define ME_MAX_TILE_ROWS 256
#define ME_MAX_BATCH_GROUPS 4
int64_t me_plan_num_tiles(int64_t n_tgt, int32_t tile_rows);
/* int32 elements of tile_bptr_dev: 2 * num_tiles + 1 since ABI 1.2 (the tiles' batch ranges, then their dispatch
 * order).  Size the buffer with this call, not with num_tiles + 1 (the ABI 1.0 size): me_plan_build writes and every
 * me_conv_target_* reads the dispatch order behind the ranges. */
int64_t me_plan_tile_bptr_elems(int64_t n_tgt, int32_t tile_rows);
/* upper bound on the number of groups (and of batches) for a table with n_pairs valid entries; includes the four
 * groups (64 slots) that me_plan_build fills BEHIND the last group of the plan with {source row 0, dummy target row}:
 * the convolution kernels read the 64-slot index window of a batch to its end without clamping.  plan_src_dev /
 * plan_dst_dev must therefore really hold 16 * me_plan_max_groups entries. */
int64_t me_plan_max_groups(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t tile_rows);
int64_t me_plan_workspace_bytes(int64_t n_tgt, int64_t volume, int32_t tile_rows);
/*   tbl_dev        int32 [volume, n_tgt]  neighbour table (nbr for forward, nbrT for dgrad)
 *   order_dev      int32 [n_tgt] or NULL   target rows in tile order: tile t owns the rows
 *                                          order[t*tile_rows .. (t+1)*tile_rows)  (NULL = identity;
 *                                          pass the argsort of me_coords_spatial_keys for compact tiles)
 *   plan_src_dev   int32 [16 * max_groups] (out) source row per slot, -1 = padding (the kernels gather row 0 for it;
 *                                                its products land in the dummy accumulator row)
 *   plan_dst_dev   int32 [16 * max_groups] (out) target row local to its tile; padding slots point at
 *                                                the dummy row `tile_rows`; the global target row of
 *                                                (tile t, local row d) is order[t * tile_rows + d]
 *   batch_desc_dev int32 [2 * max_groups]  (out) per batch {first group, (k << 8) | number of groups}
 *   tile_bptr_dev  int32 [me_plan_tile_bptr_elems = 2 * num_tiles + 1] (out) [0, num_tiles]: batch range of each tile; behind it the DISPATCH
 *                  ORDER of the tiles (a permutation, heaviest tile first): workgroup b of the convolution kernels
 *                  takes tile tile_bptr[num_tiles + 1 + b]
 *   item_gptr_dev  int32 [num_tiles * volume + 1] (out) first group of each (tile, k) item
 */
int me_plan_build(const int32_t *tbl_dev, const int32_t *order_dev, int64_t n_tgt, int64_t volume,
                  int32_t tile_rows, int32_t batch_groups, int32_t *plan_src_dev, int32_t *plan_dst_dev,
                  int32_t *batch_desc_dev, int32_t *tile_bptr_dev, int32_t *item_gptr_dev,
                  void *workspace_dev, int64_t workspace_bytes, void *stream);

/* Every plan of a scene at once (ABI 1.4).  A network on a NEW scene builds ~45 plans (MinkUNet34C: 10 kernel maps x
 * forward / input gradient x tile geometries); me_plan_build is 4 launches + 2 memsets each, on mostly idle hardware.
 * me_plan_build_multi walks a table of jobs in 4 launches: the arrays it writes are those of me_plan_build, bit for bit.
 * What the reference does at this point: one thrust sort + scan per kernel map (src/kernel_map.cuh:313-405).
 *   me_plan_jobs_init       fills n_tiles / n_items / item_base of every job (host) -> total items, -1: invalid geometry
 *   jobs_host / jobs_dev    the SAME initialised table in host memory (launch geometry, argument checks) and in device
 *                           memory (read by the kernels; uploaded by the caller on `stream`)
 *   workspace_dev           me_plan_multi_workspace_bytes(total items) */
typedef struct me_plan_job {
  const int32_t *tbl;       /* device: [volume, n_tgt], as me_plan_build */
  const int32_t *order;     /* device: [n_tgt] or NULL */
  int64_t n_tgt, volume;
  int32_t tile_rows, batch_groups;
  int32_t *plan_src, *plan_dst, *batch_desc, *tile_bptr, *item_gptr;   /* device (out), sized as for me_plan_build */
  int64_t n_tiles, n_items, item_base;                                  /* out of me_plan_jobs_init */
} me_plan_job;
int64_t me_plan_jobs_init(me_plan_job *jobs_host, int32_t n_jobs);
int64_t me_plan_multi_workspace_bytes(int64_t total_items);
int me_plan_build_multi(const me_plan_job *jobs_host, const me_plan_job *jobs_dev, int32_t n_jobs, void *workspace_dev,
                        int64_t workspace_bytes, void *stream);

/* ---- convolution feature kernels (replace ConvolutionForwardKernelGPU / BackwardKernelGPU,
 *      src/convolution_kernel.cu:320-496, 553-757; CPU twins src/convolution_kernel.hpp:33-144) -- */

/* Weights are handed to the convolution kernel PACKED: the register image of the MFMA operand
 * (one 16-byte element per lane and k-step quad, zero-padded to the kernel's channel tiling; layout
 * documented at k_pack_weights in csrc/conv.hip).  Pack once per call (~1 us for 0.9 MB):
 *   transposed = 0: w is [volume, c_src, c_dst]                     (forward: w = kernel)
 *   transposed = 1: w is [volume, c_dst, c_src], i.e. the FORWARD kernel when computing dgrad with
 *                   c_src = Cout, c_dst = Cin (so no separate transpose pass is needed). */
int64_t me_conv_packed_weight_elems(int64_t volume, int32_t c_src, int32_t c_dst); /* floats */
int me_conv_pack_weights_f32(const float *w_dev, int64_t volume, int32_t c_src, int32_t c_dst,
                             int32_t transposed, float *packed_dev, void *stream);

/* Target-stationary gather -> LDS -> MFMA(fp32 16x16x4) -> LDS accumulate -> one coalesced store.
 *   dst[t, :] = sum over plan entries (k, s) of tile(t):  src[s, :] @ w[k]      (w[k]: [c_src, c_dst])
 * Forward: src = in_feat, packed kernel, plan from nbr.  dgrad: src = grad_out, kernel packed with
 * transposed = 1, plan from nbrT.  Every target row is written (rows without entries get zeros).
 * The plan must have been built with the same tile_rows / batch_groups (me_conv_plan_config). */
int me_conv_target_f32(const float *src_feat_dev, int64_t n_src, int32_t c_src,
                       const float *packed_w_dev, int64_t volume, int32_t c_dst,
                       const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                       const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                       const int32_t *order_dev /* as given to me_plan_build, or NULL */, float *dst_feat_dev,
                       int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream);

/* The same launch with MULTI-OFFSET BATCHES (round 3): runs of single-group batches of consecutive offsets — what a
 * sparse map's plan consists of — are staged and multiplied together, up to four offsets per barrier pair, each group
 * with its own offset's weight slice.  For maps with fewer than ~24 pairs per (tile, offset) item; fp32 results in a
 * fixed order (not bit-identical to me_conv_target_f32: the partial sums of a fused batch are added to the tile as a
 * whole).  Shapes without a fused instantiation (slabs of 96 columns, chunks of 96 channels, >= 4 GiB sources) run
 * the plain kernel. */
int me_conv_target_f32_fused(const float *src_feat_dev, int64_t n_src, int32_t c_src, const float *packed_w_dev,
                             int64_t volume, int32_t c_dst, const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                             const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev, const int32_t *order_dev,
                             float *dst_feat_dev, int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream);

/* Plan geometry for a (target rows, channels) problem: the tile height is chosen so that tiles x column
 * slabs is just below a multiple of the GPU's resident-workgroup slots (a 100k-voxel layer is only
 * ~2 workgroup rounds long, so an unlucky tile count can idle a third of the chip) while the
 * accumulator tile + the stage buffer of batch_groups groups fit the 160 KiB LDS; n_pairs (density)
 * steers the trade-off against the 16-row group padding and sizes the batch. */
int me_conv_plan_config(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                        int32_t *tile_rows, int32_t *batch_groups);

/* wt[k, j, i] = w[k, i, j] */
int me_transpose_kernel_f32(const float *w_dev, int64_t volume, int32_t c_in, int32_t c_out,
                            float *wt_dev, void *stream);

/* Weight gradient: grad_w[k] = sum over pairs e of offset k of  x[in[e], :]^T (outer) dy[out[e], :]
 * (src/convolution_kernel.hpp:128-142).  The pair list is cut into equal ranges (one per workgroup,
 * regardless of offset boundaries); MFMA fp32 16x16x4 fed straight from global memory; partial
 * register images to the workspace; deterministic second-pass reduction in range order.
 *   k_offsets: host int64 [volume+1] (sizes the grid), k_offsets_dev: the same values on the device;
 *   workspace bytes from me_conv_wgrad_workspace_bytes. */
int64_t me_conv_wgrad_workspace_bytes(const int64_t *k_offsets, int64_t volume, int32_t c_in,
                                      int32_t c_out);
int me_conv_wgrad_f32(const float *x_dev, int64_t n_in /* rows of x */, int32_t c_in, const float *dy_dev,
                      int64_t n_out /* rows of dy */, int32_t c_out,
                      const int32_t *in_pairs_dev, const int32_t *out_pairs_dev,
                      const int64_t *k_offsets /* host */, const int64_t *k_offsets_dev,
                      int64_t volume, float *grad_w_dev, void *workspace_dev,
                      int64_t workspace_bytes, void *stream);

/* ---- bf16 features (fp32 accumulation) --------------------------------------------------------------
 * The reference computes in float / double only (AT_DISPATCH_FLOATING_TYPES, src/convolution_gpu.cu:137-155);
 * BASELINE configs[2] (MinkUNet34C, bf16) asks for a reduced-precision path.  Feature matrices are bf16
 * (uint16_t bit patterns, torch.bfloat16), [n, c] row-major; the SAME tile plans / pair lists are used.
 * Semantics: weights rounded to bf16 (RNE) at pack time, exact products, fp32 sums in the plan's fixed
 * order, one rounding of the result to bf16 — i.e. the fp32 convolution of the rounded operands.
 *   me_conv_pack_weights_bf16: w_is_f32 != 0: w_dev holds fp32 master weights, else bf16; layouts and
 *                              `transposed` as for me_conv_pack_weights_f32.
 *   me_conv_target_bf16:       forward / dgrad on v_mfma_f32_16x16x32_bf16 (arguments as me_conv_target_f32).
 *   me_conv_wgrad_bf16:        grad_w (fp32 out) from bf16 x / dy on v_mfma_f32_16x16x32_bf16 (rows staged in
 *                              LDS once per workgroup, operands read back transposed with
 *                              ds_read_b64_tr_b16); workspace bytes from me_conv_wgrad_workspace_bytes_bf16. */
int me_conv_plan_config_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                             int32_t *tile_rows, int32_t *batch_groups);  /* plan geometry of the bf16 kernel */
int64_t me_conv_packed_weight_elems_bf16(int64_t volume, int32_t c_src, int32_t c_dst); /* bf16 elements */
int me_conv_pack_weights_bf16(const void *w_dev, int32_t w_is_f32, int64_t volume, int32_t c_src, int32_t c_dst,
                              int32_t transposed, uint16_t *packed_dev, void *stream);
int me_conv_target_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src,
                        const uint16_t *packed_w_dev, int64_t volume, int32_t c_dst,
                        const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                        const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                        const int32_t *order_dev, uint16_t *dst_feat_dev,
                        int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream);
/* The same operation, arguments and results (bit-identical) with BATCH FUSION: consecutive small batches of a tile —
 * up to four kernel offsets — are staged and multiplied per barrier pair.  For sparse maps, where most (tile, offset)
 * items hold a single 16-row group (a dense layer is 20 - 40 % faster on me_conv_target_bf16). */
int me_conv_target_bf16_fused(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src,
                              const uint16_t *packed_w_dev, int64_t volume, int32_t c_dst,
                              const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                              const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                              const int32_t *order_dev, uint16_t *dst_feat_dev,
                              int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream);
/* The same launch (fused != 0: the batch-fusion instantiation) that ALSO leaves the batch-norm statistics of its output
 * behind: per tile and output channel the mean and M2 = sum (y - mean)^2 of the rounded values it stores, in
 * part_mean_dev / part_m2_dev [ceil(n_tgt / tile_rows)][c_dst] floats (slot = tile index, every slot written).  In the
 * reference batch norm is a separate torch operator on the feature matrix (MinkowskiEngine/MinkowskiNormalization.py:
 * 51-98); me_bn_stats_from_tiles turns these partials into its mean / rstd without reading the matrix again.
 * me_conv_stats_supported_bf16: 1 when the tile shape chosen for (c_src, c_dst) has the epilogue. */
int32_t me_conv_stats_supported_bf16(int32_t c_src, int32_t c_dst);
int me_conv_target_bf16_stats(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src,
                              const uint16_t *packed_w_dev, int64_t volume, int32_t c_dst,
                              const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                              const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                              const int32_t *order_dev, uint16_t *dst_feat_dev,
                              int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, int32_t fused,
                              float *part_mean_dev, float *part_m2_dev, void *stream);
/* Split-K launches (round 4).  The reference runs one GEMM per kernel offset and adds into the output with atomics
 * (src/convolution_kernel.cu:320-496); here a workgroup owns a tile of target rows and walks all offsets — which on a
 * SMALL coordinate map (MinkUNet's 5k-voxel level) means 128 tiles of 39 rows that each stream the whole packed weight
 * tensor.  me_conv_plan_config_bf16_ex is me_conv_plan_config_bf16 that may also answer split_k = G > 1 with G-times
 * taller tiles: the launch then has G offset groups per tile, every workgroup walks K / G offsets, the groups' fp32
 * tiles go through `workspace_dev` (me_conv_splitk_workspace_bytes) and a second kernel adds them in group order,
 * rounds once and stores (same semantics: fp32 sums in a fixed order, one rounding; NOT the bits of the unsplit launch).
 * me_conv_target_bf16_ex is the one entry point of the bf16 forward / dgrad launch: fused / statistics / split-K by
 * argument (split_k = 1, workspace NULL, part_* NULL: me_conv_target_bf16). */
int me_conv_plan_config_bf16_ex(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                                int32_t *tile_rows, int32_t *batch_groups, int32_t *split_k);
int64_t me_conv_splitk_workspace_bytes(int64_t n_tgt, int32_t tile_rows, int32_t c_dst, int32_t split_k);
int me_conv_target_bf16_ex(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src,
                           const uint16_t *packed_w_dev, int64_t volume, int32_t c_dst,
                           const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                           const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                           const int32_t *order_dev, uint16_t *dst_feat_dev,
                           int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, int32_t fused, int32_t split_k,
                           void *workspace_dev, float *part_mean_dev, float *part_m2_dev, void *stream);
/* Row-wise launches (round 6, csrc/conv_rowwise.hip).  For a kernel-map side on which EVERY target row has EXACTLY ONE
 * pair, dst[tgt_rows[e]] = src[src_rows[e]] @ W[k(e)] for every pair e — no sum, hence no tile plan, no LDS accumulator:
 * the x rows stream from global memory into the MFMA operand registers, W[k] comes from the packed image of
 * me_conv_pack_weights_bf16 / me_conv_pack_weights_multi (the SAME image the tile-plan launch of the layer uses; the
 * transposed image for an input gradient).  Replaces, for those sides, what me_conv_target_bf16 does:
 *   - kernel volume 1, stride 1: the reference's dense `input.F.mm(kernel)` (MinkowskiEngine/MinkowskiConvolution.py:304-308)
 *     and its input gradient;
 *   - kernel_size == stride maps (src/coordinate_map_manager.cpp:402-429 out maps): forward of
 *     ConvolutionTransposeForwardGPU (pybind/extern.hpp:99-143) and the input gradient of ConvolutionBackwardGPU
 *     (src/convolution_gpu.cu:161-244) — the fine side of the map.
 * src_rows_dev / tgt_rows_dev: the kernel map's pair lists (kernel_map.hpp:40-53 layout: pairs of an offset contiguous)
 * with the side's SOURCE rows in src_rows; k_offsets_dev int64 [volume + 1] on the device; n_pairs_bound >= the pair
 * count (the launch is sized by it; for such a side the pair count IS n_tgt).  The caller guarantees the one-pair-per-
 * row property (a row without a pair is not written, a row with two pairs is written twice).  Semantics of the tile-plan
 * launch: fp32 sums over the channels, one rounding to bf16.
 * me_conv_rowwise_supported_bf16: 1 when (volume <= 64, c_src % 8 == 0, c_dst % 8 == 0, W[k] of one 128-column slab
 * within 64 KB of LDS) the kernel takes the shape.
 * No batch-norm statistics epilogue: a layer whose statistics are wanted either stays on me_conv_target_bf16_stats or
 * is followed by the ordinary pass over its output (me_bn_stats). */
int32_t me_conv_rowwise_supported_bf16(int64_t volume, int32_t c_src, int32_t c_dst);
int me_conv_rowwise_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const uint16_t *packed_w_dev,
                         int64_t volume, int32_t c_dst, const int32_t *src_rows_dev, const int32_t *tgt_rows_dev,
                         const int64_t *k_offsets_dev, int64_t n_pairs_bound, uint16_t *dst_feat_dev, int64_t n_tgt,
                         void *stream);
int64_t me_conv_wgrad_workspace_bytes_bf16(const int64_t *k_offsets, int64_t volume, int32_t c_in, int32_t c_out);
int me_conv_wgrad_bf16(const uint16_t *x_dev, int64_t n_in, int32_t c_in, const uint16_t *dy_dev, int64_t n_out,
                       int32_t c_out,
                       const int32_t *in_pairs_dev, const int32_t *out_pairs_dev,
                       const int64_t *k_offsets /* host */, const int64_t *k_offsets_dev,
                       int64_t volume, float *grad_w_dev, void *workspace_dev,
                       int64_t workspace_bytes, void *stream);

/* ---- float64 features (round 4) ----------------------------------------------------------------------------------------
 * The reference instantiates its feature operators for float and double (AT_DISPATCH_FLOATING_TYPES,
 * src/convolution_gpu.cu:137-155; the pooling / broadcast units alike) and its own tests are float64 gradchecks
 * (MinkowskiEngine/utils/gradcheck.py:34-57).  csrc/f64.hip is that instantiation: plain double FMAs, one thread per
 * output element on the dense neighbour tables / pair lists, deterministic (offsets ascending, then channels) —
 * a yardstick, not a hot path.
 *   me_conv_target_f64: dst[t] = sum_k src[tbl[k][t]] @ W_k on the neighbour table tbl int32 [volume, n_tgt] (source row
 *                       of (offset, target row) or -1); w_dev is the kernel [K, c_src, c_dst], or with transposed != 0
 *                       the FORWARD kernel [K, c_dst, c_src] read transposed (input gradient).
 *   me_conv_wgrad_f64:  grad_w [K, c_in, c_out] from the pair lists (k_offsets_dev: device prefix).
 *   me_pool_*_f64, me_global_pool_f64 (no workspace), me_broadcast_f64: arguments as their _f32 twins, double rows;
 *                       counts stay float32, argmax / masks int32. */
int me_conv_target_f64(const double *src_feat_dev, int64_t n_src, int32_t c_src, const double *w_dev, int32_t transposed,
                       int64_t volume, int32_t c_dst, const int32_t *tbl_dev, double *dst_feat_dev, int64_t n_tgt,
                       void *stream);
int me_conv_wgrad_f64(const double *x_dev, int32_t c_in, const double *dy_dev, int32_t c_out, const int32_t *in_pairs_dev,
                      const int32_t *out_pairs_dev, const int64_t *k_offsets_dev, int64_t volume, double *grad_w_dev,
                      void *stream);
int me_pool_sum_f64(const double *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                    const float *src_count_dev, int32_t average, double *dst_dev, float *dst_count_dev, void *stream);
int me_pool_max_f64(const double *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                    double *dst_dev, int32_t *mask_dev, void *stream);
int me_pool_max_backward_f64(const double *grad_out_dev, int32_t c, const int32_t *tbl_in_dev, int64_t n_in,
                             int64_t volume, const int32_t *mask_dev, double *grad_in_dev, void *stream);
int me_global_pool_f64(const double *src_dev, const double *src2_dev, int32_t c, const int32_t *batch_row_dev, int64_t n,
                       int32_t n_batch, int32_t mode, double *dst_dev, int32_t *dst_arg_dev, float *dst_count_dev,
                       void *stream);
int me_broadcast_f64(const double *in_dev, const double *glob_dev, const int32_t *batch_row_dev, int64_t n, int32_t c,
                     int32_t multiply, double *out_dev, void *stream);

/* Output-stationary bf16 convolution (k_conv_gather_bf16): the target rows' fp32 accumulators stay in registers for
 * all kernel offsets, absent neighbours multiply as zero rows (bf16 MFMAs cost 1/16 of fp32 ones: the waste is cheaper
 * than the plan kernel's LDS accumulator traffic).  Works on a neighbour table directly — tbl int32 [volume, n_tgt]:
 * source ROW of (offset k, target position p) or -1; order (may be NULL): position -> target row — for forward (table
 * of the out side) and dgrad (table of the in side, weights packed transposed).  No tile plan.  Same semantics as
 * me_conv_target_bf16 (fp32 sums in offset order, one rounding at the store).  Channel counts: multiples of 32
 * (me_conv_gather_supported_bf16), else use me_conv_target_bf16. */
int32_t me_conv_gather_supported_bf16(int32_t c_src, int32_t c_dst);
int64_t me_conv_gather_weight_elems_bf16(int64_t volume, int32_t c_src, int32_t c_dst);
int me_conv_gather_pack_weights_bf16(const void *w_dev, int32_t w_is_f32, int64_t volume, int32_t c_src, int32_t c_dst,
                                     int32_t transposed, uint16_t *packed_dev, void *stream);
int me_conv_gather_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const uint16_t *packed_dev,
                        int64_t volume, int32_t c_dst, const int32_t *tbl_dev, const int32_t *order_dev,
                        uint16_t *dst_feat_dev, int64_t n_tgt, void *stream);

/* ---- bf16 convolution on a source halo staged in LDS (csrc/conv_halo.hip, ABI 1.5) ----------------------------
 * The same operation as me_conv_target_bf16 — dst[t, :] = sum over offsets k of src[tbl[k][t], :] @ w[k], fp32 sums in a
 * fixed order (per 64-channel chunk: ascending k, ascending source channel), one rounding to bf16 — on another schedule: a tile of
 * `tile_rows` target positions keeps its fp32 sums in REGISTERS for all offsets; the distinct source rows the tile
 * touches (its halo, <= s_cap of them) are staged in LDS once per channel chunk and every offset gathers from LDS;
 * absent neighbours multiply a zero row, 16-row groups without any neighbour at an offset are skipped; no barrier and
 * no read-modify-write inside the walk over the offsets.  The reference gathers, multiplies and scatter-adds once per
 * offset (src/convolution_kernel.cu:320-496).  Results are NOT the bits of me_conv_target_bf16 (there every batch
 * starts a new accumulator), they are bitwise reproducible.
 *   me_conv_halo_use_bf16     the policy: 1 when a host should run this launch side on the halo kernel (0: the tile-plan
 *                             kernel).  ME_AMD_HALO=0 | 1 | auto.  Both hosts of this repository follow it — from the
 *                             me_conv_halo_min_uses()-th launch on the same kernel-map side on (the plan costs more
 *                             than one launch saves: scenes that are not reused never build it).
 *   me_conv_halo_config_bf16  1 when the kernel is instantiated for (volume, c_src, c_dst): volume in [2, 32],
 *                             c_src % 32 == 0, c_dst in {32, 64, 96} or a multiple of 128; answers the plan geometry.
 *   me_halo_plan_build        tbl_dev int32 [volume, n_tgt] (source row of (offset, table column) or -1; nbr for forward,
 *                             nbrT for dgrad), col_order_dev int32 [n_tgt] or NULL: table column of tile position p
 *                             (NULL: p); src_pos_dev int32 [n_src] position of every SOURCE row in a spatial order of
 *                             the source map and src_order_dev int32 [n_src] its inverse (both or neither): the halo
 *                             slots are then in position order (fewer LDS bank conflicts among the rows a gather
 *                             touches together), else in row order.  Out: halo_cnt_dev int32 [tiles] distinct source
 *                             rows of each tile; halo_rows_dev int32 [tiles * s_cap] its first s_cap of them in slot
 *                             order; lidx_dev uint16
 *                             [tiles * volume * tile_rows]: 0 = no neighbour, 1 + halo slot, 0xffff = beyond s_cap;
 *                             kmask_dev uint32 [tiles * volume]: bit g = 16-row group g has a neighbour at the offset.
 *                             A tile whose halo exceeds s_cap is served by direct gathers off tbl_dev (any map works).
 *   me_conv_halo_bf16         packed_w_dev: the image of me_conv_pack_weights_bf16 for (c_src, c_dst) (transposed = 1
 *                             for dgrad); out_order_dev int32 [n_tgt] or NULL: target ROW of tile position p;
 *                             part_mean_dev / part_m2_dev [tiles][c_dst] or NULL: the batch-norm partials of
 *                             me_conv_target_bf16_stats with tile_rows = the halo tile. */
int32_t me_conv_halo_use_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst);
int32_t me_conv_halo_min_uses(void);   /* the launch count on one kernel-map side at which a host builds the halo plan (2; forced: 1) */
int32_t me_conv_halo_config_bf16(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                                 int32_t *tile_rows, int32_t *s_cap);
int64_t me_halo_plan_num_tiles(int64_t n_tgt, int32_t tile_rows);
int me_halo_plan_build(const int32_t *tbl_dev, const int32_t *col_order_dev, const int32_t *src_pos_dev,
                       const int32_t *src_order_dev, int64_t n_tgt, int64_t volume, int32_t tile_rows, int32_t s_cap, int32_t *halo_cnt_dev, int32_t *halo_rows_dev,
                       uint16_t *lidx_dev, uint32_t *kmask_dev, void *stream);
int me_conv_halo_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const uint16_t *packed_w_dev,
                      int64_t volume, int32_t c_dst, const int32_t *halo_cnt_dev, const int32_t *halo_rows_dev,
                      const uint16_t *lidx_dev, const uint32_t *kmask_dev, const int32_t *tbl_dev,
                      const int32_t *col_order_dev, const int32_t *out_order_dev, uint16_t *dst_feat_dev,
                      int64_t n_tgt, int32_t tile_rows, int32_t s_cap, float *part_mean_dev, float *part_m2_dev,
                      void *stream);

/* ---- bf16 convolution with the offsets stacked into the MFMA reduction (csrc/conv_stem.hip, ABI 1.5) -------------
 * For layers with AT MOST 8 source channels (a network's stem: 3 -> 32 channels over 5^3 offsets): one MFMA step
 * multiplies FOUR offsets (4 x 8 channels = its 32-deep reduction), every neighbour row is one 16-byte load, a wave keeps
 * 32 target rows x all output columns in registers over all offsets; the weights are read from the layer's OWN kernel
 * tensor (no packed image) and kept in LDS as MFMA fragments.  The reference runs one gather - GEMM - scatter launch per
 * offset (src/convolution_kernel.cu:320-496).  Semantics of me_conv_target_bf16 (weights rounded to bf16, exact products,
 * fp32 sums in a fixed order, one rounding); results are bitwise reproducible, not the bits of the tile-plan kernel.
 *   me_conv_stem_use_bf16   the policy both hosts follow: c_src <= 8, c_dst in {16, 32, 64}, volume >= 8 (ME_AMD_STEM=0 | 1)
 *   me_conv_stem_tile_rows  rows per tile (256): the tile of the batch-norm partials
 *   me_conv_stem_bf16       src_feat_dev bf16 [n_src, 8] (rows padded with zeros to 8 channels, 16-byte aligned; c_src
 *                           must be 8), w_dev the kernel [volume, 8, c_dst] (transposed != 0: [volume, c_dst, 8]) in
 *                           fp32 (w_is_f32) or bf16; tbl_dev / col_order_dev / out_order_dev as me_conv_halo_bf16;
 *                            part_mean_dev / part_m2_dev [ceil(n_tgt / tile rows)][c_dst] or NULL. */
int32_t me_conv_stem_use_bf16(int64_t n_tgt, int64_t volume, int32_t c_src, int32_t c_dst);
int32_t me_conv_stem_tile_rows(void);
int me_conv_stem_bf16(const uint16_t *src_feat_dev, int64_t n_src, int32_t c_src, const void *w_dev, int32_t w_is_f32,
                      int32_t transposed, int64_t volume, int32_t c_dst, const int32_t *tbl_dev,
                      const int32_t *col_order_dev, const int32_t *out_order_dev, uint16_t *dst_feat_dev, int64_t n_tgt,
                      float *part_mean_dev, float *part_m2_dev, void *stream);

/* ---- fp32 convolution on the bf16 matrix pipe ("bf16x6" split, csrc/conv_f32x3.hip) ---------------------
 * Same contract as me_conv_target_f32 (fp32 features / weights in, fp32 out, the reference's
 * ConvolutionForwardKernelGPU / BackwardKernelGPU input gradient, src/convolution_kernel.cu:320-757, in fp32 as
 * AT_DISPATCH_FLOATING_TYPES computes it), different arithmetic unit: every operand is split exactly into three
 * bf16 terms (a = a1 + a2 + a3) and the product is rebuilt from six v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation; the dropped terms are < 2^-23 |a b| per product, i.e. below one fp32 rounding.  Fixed summation
 * order (bitwise reproducible).  An infinite input yields NaN (fp32 arithmetic would keep the infinity).
 * Needs c_src % 8 == 0 (me_conv_f32x3_supported); plans come from me_plan_build with the geometry of
 * me_conv_plan_config_f32x3; weights packed by me_conv_pack_weights_f32x3 (three bf16 planes). */
int32_t me_conv_f32x3_supported(int32_t c_src, int32_t c_dst);
int me_conv_plan_config_f32x3(int64_t n_tgt, int64_t volume, int64_t n_pairs, int32_t c_src, int32_t c_dst,
                              int32_t *tile_rows, int32_t *batch_groups);
int64_t me_conv_packed_weight_elems_f32x3(int64_t volume, int32_t c_src, int32_t c_dst); /* bf16 elements */
int me_conv_pack_weights_f32x3(const float *w_dev, int64_t volume, int32_t c_src, int32_t c_dst,
                               int32_t transposed, uint16_t *packed_dev, void *stream);
int me_conv_target_f32x3(const float *src_feat_dev, int64_t n_src, int32_t c_src,
                         const uint16_t *packed_w_dev, int64_t volume, int32_t c_dst,
                         const int32_t *plan_src_dev, const int32_t *plan_dst_dev,
                         const int32_t *batch_desc_dev, const int32_t *tile_bptr_dev,
                         const int32_t *order_dev, float *dst_feat_dev,
                         int64_t n_tgt, int32_t tile_rows, int32_t batch_groups, void *stream);

/* ---- weights of many layers packed by ONE launch (csrc/pack.hip) ------------------------------------------
 * The packed images above only change when the weights do (the optimizer step), not per activation: a network packs
 * all its (layer, direction) images at once instead of twice per layer and step (MinkUNet34C: 126 launches -> 1).
 * Images are bit-identical to me_conv_pack_weights_bf16 / me_conv_pack_weights_f32x3.
 *   host:   fill w / wp / volume / c_src / c_dst / transposed / w_is_f32 / mode of each job, call
 *           me_conv_pack_job_init (sets kc, nchunks, ncb, threads), build the exclusive prefix of `threads`
 *           (n_jobs + 1 int64), copy jobs and prefix to the device;
 *   device: me_conv_pack_weights_multi(jobs_dev, n_jobs, prefix_dev, prefix[n_jobs], stream).
 * c_src / c_dst are the channels of the LAUNCH the image is for (forward: Cin, Cout; input gradient: Cout, Cin with
 * transposed = 1, `w` always the reference-layout kernel[K, Cin, Cout]). */
#define ME_PACK_BF16 0    /* image of me_conv_pack_weights_bf16 (w_is_f32: fp32 master weights are rounded RNE) */
#define ME_PACK_F32X3 1   /* image of me_conv_pack_weights_f32x3 (fp32 weights, three bf16 planes) */
typedef struct me_pack_job {
  const void *w;        /* device: kernel[K, Cin, Cout] */
  void *wp;             /* device: packed image, me_conv_packed_weight_elems_{bf16,f32x3} bf16 elements, 16-byte aligned */
  int64_t volume;
  int64_t threads;      /* out: 16-byte elements (bf16) / element triples (f32x3) of the image */
  int32_t c_src, c_dst, transposed, w_is_f32, mode;
  int32_t kc, nchunks, ncb;   /* out: source-channel chunk of the tile kernel, chunks, 16-column blocks */
} me_pack_job;
int32_t me_conv_pack_chunk_bf16(int32_t c_src, int32_t c_dst);
int32_t me_conv_pack_chunk_f32x3(int32_t c_src, int32_t c_dst);
int me_conv_pack_job_init(me_pack_job *job);
int me_conv_pack_weights_multi(const me_pack_job *jobs_dev, int32_t n_jobs, const int64_t *thread_prefix_dev,
                               int64_t total_threads, void *stream);

/* ---- pooling / broadcast (replace src/pooling_avg_kernel.cu, src/pooling_max_kernel.cu,
 *      src/broadcast_kernel.cu; CPU twins src/pooling_avg_kernel.hpp:41-150,
 *      src/pooling_max_kernel.hpp:36-117, src/broadcast_kernel.hpp:35-160) ------------------------
 * All target-stationary on the dense neighbour tables of a kernel map (tbl[k, target row] -> source row
 * or -1): no atomics, no zero-fill, summation in ascending k (the reference CPU order). */

/* Local sum / average pooling, forward AND backward:
 *   dst[t, :] = sum_k src[tbl[k, t], :]                              (src_count_dev == NULL)
 *   dst[t, :] = sum_k src[s, :] / src_count[s],  s = tbl[k, t]       (average-pooling backward:
 *               src = grad_out, tbl = transposed table, src_count = num_nonzero of the forward pass)
 * `average` != 0 divides by the number of summed rows; that number is written to dst_count_dev (float
 * [n_tgt], may be NULL) — the reference's num_nonzero (src/local_pooling_cpu.cpp:104-122). */
int me_pool_sum_f32(const float *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                    const float *src_count_dev, int32_t average, float *dst_dev, float *dst_count_dev,
                    void *stream);
/* Local max pooling forward: dst = max over k, mask[t, c] = flat index (source row * c + channel) of the
 * first maximum in k order; rows without neighbours get -FLT_MAX / -1 (src/pooling_max_kernel.hpp:73-96). */
int me_pool_max_f32(const float *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                    float *dst_dev, int32_t *mask_dev, void *stream);
/* Local max pooling backward: grad_in[i, c] = sum of grad_out[o, c] over the outputs o = tbl_in[k, i] whose
 * mask names (i, c)  (src/pooling_max_kernel.hpp:98-117).  tbl_in: transposed table [volume, n_in]. */
int me_pool_max_backward_f32(const float *grad_out_dev, int32_t c, const int32_t *tbl_in_dev, int64_t n_in,
                             int64_t volume, const int32_t *mask_dev, float *grad_in_dev, void *stream);

/* Global pooling over the rows of each batch index (src/global_pooling_cpu.cpp:43-238).
 *   batch_row_dev int32 [n]: row of the origin map (output row) of every input row
 *   mode 0 sum, 1 average, 2 max;  src2_dev (may be NULL): multiplied element-wise before the reduction
 *   (the gradient of a broadcast multiplication);  dst [n_batch, c];  dst_arg int32 [n_batch, c] (max: flat
 *   index row * c + channel);  dst_count float [n_batch] (sum / average; may be NULL). */
int64_t me_global_pool_workspace_bytes(int64_t n, int32_t n_batch, int32_t c);
int me_global_pool_f32(const float *src_dev, const float *src2_dev, int32_t c, const int32_t *batch_row_dev,
                       int64_t n, int32_t n_batch, int32_t mode, float *dst_dev, int32_t *dst_arg_dev,
                       float *dst_count_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);

/* Broadcast (src/broadcast_kernel.hpp:35-160): out[i, :] = in[i, :] (+ | *) glob[batch_row[i], :];
 * in_dev == NULL: out[i, :] = glob[batch_row[i], :] (the gradient of global sum pooling). */
int me_broadcast_f32(const float *in_dev, const float *glob_dev, const int32_t *batch_row_dev, int64_t n,
                     int32_t c, int32_t multiply, float *out_dev, void *stream);

/* bf16 feature rows (uint16_t = raw bfloat16 bits) on the same kernels: values are widened to fp32, summed /
 * compared in fp32 in the same order and rounded once at the store.  Counts and argmax masks keep their types;
 * global pooling returns fp32 [n_batch, c] (a handful of rows; the caller rounds).  The reference has no
 * reduced-precision pooling (pybind/extern.hpp:187-392 dispatches float / double only). */
int me_pool_sum_bf16(const uint16_t *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                     const float *src_count_dev, int32_t average, uint16_t *dst_dev, float *dst_count_dev,
                     void *stream);
int me_pool_max_bf16(const uint16_t *src_dev, int32_t c, const int32_t *tbl_dev, int64_t n_tgt, int64_t volume,
                     uint16_t *dst_dev, int32_t *mask_dev, void *stream);
int me_pool_max_backward_bf16(const uint16_t *grad_out_dev, int32_t c, const int32_t *tbl_in_dev, int64_t n_in,
                              int64_t volume, const int32_t *mask_dev, uint16_t *grad_in_dev, void *stream);
int me_global_pool_bf16(const uint16_t *src_dev, const uint16_t *src2_dev, int32_t c, const int32_t *batch_row_dev,
                        int64_t n, int32_t n_batch, int32_t mode, float *dst_dev, int32_t *dst_arg_dev,
                        float *dst_count_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);
int me_broadcast_bf16(const uint16_t *in_dev, const uint16_t *glob_dev, const int32_t *batch_row_dev, int64_t n,
                      int32_t c, int32_t multiply, uint16_t *out_dev, void *stream);

/* Generative / expanding convolutions (CoordinateMapCPU::stride_region, src/coordinate_map_cpu.hpp:446-487;
 * manager: src/coordinate_map_manager.cpp:436-466): candidate output coordinates = every kernel offset of the
 * region around every input coordinate, candidate (row, k) at out[row * volume + k]; a following
 * me_coords_insert_and_map removes the duplicates (first occurrence wins).  region->tensor_stride holds the
 * OUTPUT tensor stride, as the reference's kernel region does.  align_stride (may be NULL) with `aligned`
 * (uint8 [n * volume]): marks the candidates whose spatial coordinates are multiples of align_stride (the
 * non-transposed expand_coordinates case keeps only those, :478-483). */
int me_coords_expand_region(const int32_t *coords_dev, int64_t n, int32_t ncol, const me_region *region,
                            const int32_t *align_stride, int32_t *out_coords_dev, uint8_t *aligned_dev,
                            void *stream);

/* ---- input pipeline on the device --------------------------------------------------
 * Voxelisation = me_coords_insert_and_map (unique_map / inverse_map) plus these two reductions.
 * Labels (src/quantization.cpp:140-196, quantize_label): colabels[u] = label of the voxel's first point,
 * or ignore_label if any point of the voxel carries a different label.  (The reference writes the
 * ignore label to colabels[inverse_mapping[u]] instead of colabels[u] — src/quantization.cpp:189 — which
 * marks the wrong voxel whenever duplicates precede row u; the intended semantics are implemented.) */
int me_coords_quantize_labels(const int64_t *unique_map_dev, int64_t n_unique, const int64_t *inverse_map_dev,
                              const int32_t *labels_dev, int64_t n, int32_t ignore_label,
                              int32_t *colabels_dev, void *stream);
/* Features of duplicate coordinates (SparseTensorQuantizationMode UNWEIGHTED_SUM / UNWEIGHTED_AVERAGE,
 * MinkowskiSparseTensor.py:317-341; the reference goes through cuSPARSE coo_spmm, src/spmm.cu):
 * dst[s, :] = sum (or mean) of src[perm[i], :] for i in [seg_offsets[s], seg_offsets[s+1]) in that order. */
int me_segment_sum_f32(const float *src_dev, int32_t c, const int64_t *perm_dev, const int64_t *seg_offsets_dev,
                       int64_t n_seg, int32_t average, float *dst_dev, void *stream);

/* ---- batch normalisation over feature rows (MinkowskiBatchNorm = torch.nn.BatchNorm1d on the feature matrix,
 *      MinkowskiEngine/MinkowskiNormalization.py:35-82; 62 of them per MinkUNet34C pass) ----
 * x / y / dy / dx: [n, c] row-major, fp32 (is_bf16 = 0) or bf16 (is_bf16 = 1); statistics and parameters fp32.
 *   me_bn_stats:    per-channel mean and rstd = 1 / sqrt(biased variance + eps) of the batch; when running_mean /
 *                   running_var are given they are updated as torch does (momentum, unbiased variance).
 *                   Two-level reduction in a fixed order (Chan's combination of per-chunk mean / M2): bitwise
 *                   reproducible.
 *   me_bn_apply:    y = (x - mean) * rstd * gamma + beta        (gamma / beta may be NULL); relu != 0: y = max(y, 0)
 *   me_bn_backward: grad_beta = sum dy, grad_gamma = sum dy * xhat,
 *                   dx = gamma * rstd * (dy - grad_beta / n - xhat * grad_gamma / n)   (training-mode gradient);
 *                   relu != 0: dy is first masked where the forward output (recomputed from x) was not positive
 *                   (batch norm + ReLU fused: MinkowskiBatchNorm followed by MinkowskiReLU in the reference)
 * workspace bytes for stats and backward: me_bn_workspace_bytes(n, c). */
int64_t me_bn_workspace_bytes(int64_t n, int32_t c);
int me_bn_stats(const void *x_dev, int32_t is_bf16, int64_t n, int32_t c, float eps, float momentum,
                float *mean_dev, float *rstd_dev, float *running_mean_dev, float *running_var_dev,
                int64_t *num_batches_tracked_dev /* may be NULL; += 1 (torch's BatchNorm buffer) */,
                void *workspace_dev, int64_t workspace_bytes, void *stream);
int me_bn_apply(const void *x_dev, int32_t is_bf16, int64_t n, int32_t c, const float *mean_dev,
                const float *rstd_dev, const float *gamma_dev, const float *beta_dev, int32_t relu, void *y_dev,
                void *stream);
int me_bn_backward(const void *x_dev, const void *dy_dev, int32_t is_bf16, int64_t n, int32_t c,
                   const float *mean_dev, const float *rstd_dev, const float *gamma_dev, const float *beta_dev,
                   int32_t relu, void *dx_dev, float *grad_gamma_dev, float *grad_beta_dev, void *workspace_dev,
                   int64_t workspace_bytes, void *stream);
/* Statistics of a matrix whose per-tile (mean, M2) partials a convolution already wrote (me_conv_target_bf16_stats):
 * tile g = rows [g * tile_rows, min((g + 1) * tile_rows, n)), partials [ceil(n / tile_rows)][c] floats each.  Same
 * outputs and running-statistics update as me_bn_stats, without reading the matrix (one small launch). */
int me_bn_stats_from_tiles(const float *part_mean_dev, const float *part_m2_dev, int64_t n, int32_t c,
                           int32_t tile_rows, float eps, float momentum, float *mean_dev, float *rstd_dev,
                           float *running_mean_dev, float *running_var_dev, int64_t *num_batches_tracked_dev,
                           void *stream);
/* Residual form (a ResNet block's `relu(bn(conv(x)) + skip)`): y = [relu] (T(x * a + b) + skip) in one pass, bit-identical
 * to me_bn_apply followed by an addition and a ReLU; the backward pass masks dy where the stored output `yout` is not
 * positive (relu != 0), writes the masked gradient to dskip (the residual branch's gradient; may be NULL) and dx. */
int me_bn_apply_residual(const void *x_dev, const void *skip_dev, int32_t is_bf16, int64_t n, int32_t c,
                         const float *mean_dev, const float *rstd_dev, const float *gamma_dev, const float *beta_dev,
                         int32_t relu, void *y_dev, void *stream);
int me_bn_backward_residual(const void *x_dev, const void *dy_dev, const void *yout_dev, int32_t is_bf16, int64_t n,
                            int32_t c, const float *mean_dev, const float *rstd_dev, const float *gamma_dev,
                            const float *beta_dev, int32_t relu, void *dx_dev, void *dskip_dev, float *grad_gamma_dev,
                            float *grad_beta_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);

/* Plain VALU + atomics versions on the pair lists (debug cross-check only; never the default).
 * out / grad_in / grad_w must be zero-filled by the caller. */
int me_conv_forward_naive_f32(const float *in_feat_dev, int32_t c_in, const float *w_dev, int32_t c_out,
                              const int32_t *in_pairs_dev, const int32_t *out_pairs_dev,
                              const int64_t *k_offsets_dev, int64_t volume, int64_t n_pairs,
                              float *out_feat_dev, void *stream);
int me_conv_backward_naive_f32(const float *in_feat_dev, int32_t c_in, const float *grad_out_dev,
                               int32_t c_out, const float *w_dev, const int32_t *in_pairs_dev,
                               const int32_t *out_pairs_dev, const int64_t *k_offsets_dev,
                               int64_t volume, int64_t n_pairs, float *grad_in_dev,
                               float *grad_w_dev, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ME_AMD_H */
