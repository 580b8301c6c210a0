// Native host layer of the MI355X sparse-convolution hot path: the operator module the reference builds as
// `MinkowskiEngineBackend._C` (pybind/minkowski.cu:36-68, pybind/extern.hpp:515-838), written over the C ABI of
// include/me_amd.h.  C++ twin of minkowskiengine_amd/backend.py (which stays as the ctypes test harness): the same
// keys, caches, launch policies and C-ABI calls, without the Python interpreter between the launches.
//
//   host.hpp      shared types: CoordinateMapKey, coordinate maps, kernel maps (+ tile plans), the manager
//   manager.cpp   CoordinateMapManager (src/coordinate_map_manager.{hpp,cpp,cu}), kernel-map / plan builders
//   ops.cpp       Convolution / ConvolutionTranspose / pooling / broadcast / pruning / batch-norm operators
//   bind.cpp      pybind11 module + torch::autograd functions (the backward pass never enters Python)
//
// torch is plumbing here as in the reference's own extension: tensors for device memory (the caching allocator plays
// the reference's c10 allocator, src/allocators.cuh:74-103), the current HIP stream, autograd.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/me_amd.h"

namespace meh {

using at::Tensor;
typedef std::vector<int32_t> ivec;

// ---- errors: std::runtime_error, like the reference's ASSERT (src/utils.hpp:141-150) -------------------------------
[[noreturn]] void fail(const std::string &msg);
inline void check(bool cond, const char *msg) {
  if (!cond) fail(msg);
}
inline void check(bool cond, const std::string &msg) {
  if (!cond) fail(msg);
}
void me_ok(int rc);   // throws me_last_error() when a C-ABI call failed

inline void *stream_of(const c10::Device &dev) {
  return (void *)c10::hip::getCurrentHIPStream(dev.index()).stream();
}
inline Tensor workspace(int64_t nbytes, const c10::Device &dev) {
  return at::empty({nbytes < 256 ? 256 : nbytes}, at::TensorOptions().dtype(at::kByte).device(dev));
}
inline Tensor empty_i32(at::IntArrayRef shape, const c10::Device &dev) {
  return at::empty(shape, at::TensorOptions().dtype(at::kInt).device(dev));
}
template <typename T>
inline T *ptr(const Tensor &t) {
  return t.defined() ? reinterpret_cast<T *>(t.data_ptr()) : nullptr;
}
inline void *vptr(const Tensor &t) { return t.defined() ? t.data_ptr() : nullptr; }

// environment switches shared with backend.py (same names, same defaults)
struct Policy {
  int spatial_maps;        // ME_AMD_SPATIAL_MAPS: 1 / 0 / -1 (auto)
  bool rowwise = true;     // ME_AMD_ROWWISE=0: one-pair-per-row sides stay on the tile-plan kernels
  // a K = 1 FORWARD launch whose batch-norm statistics are wanted (training, bf16) takes the row-wise kernel — which has no
  // statistics epilogue: the batch norm then reads the output once more — only on maps of at least this many rows
  // (measured per layer, gpurun_out/r06_layers_rowwise_*.log: 128 -> 96 on 161k / 200k rows 43 / 51 -> 36 / 42 us with the
  // extra pass, 192 -> 128 on 80k even, 32 -> 64 on 80k and everything smaller slower)
  int64_t rowwise_min_rows_with_stats = 100000;   // ME_AMD_ROWWISE_STATS_ROWS
  std::string tile_order;  // ME_AMD_TILE_ORDER: auto | rows | spatial
  int64_t tile_spatial_src_bytes = 28ll << 20;  // ME_AMD_TILE_SPATIAL_SRC_MB: bf16 sources of at least this size take spatial tiles
  std::string bf16_fuse;   // ME_AMD_BF16_FUSE: auto | 1 | 0
  int f32_split;           // ME_AMD_F32_SPLIT: 1 / 0 / -1 (auto)
  int tile_rows, batch_groups;   // ME_AMD_TILE_ROWS / ME_AMD_BATCH_GROUPS overrides (0 = plan config)
  bool pack_cache;         // ME_AMD_PACK_CACHE
  bool f32_fuse;           // ME_AMD_F32_FUSE
  bool conv_bn_stats;      // ME_AMD_CONV_BN_STATS: batch-norm statistics in the bf16 convolution's epilogue
  bool roctx;              // ME_AMD_ROCTX: roctx ranges around insert / stride / kernel map / plan / forward / dgrad / wgrad
  static const Policy &get();
  static void set(const std::string &name, int64_t value);   // tests / tuning scripts (plans already cached keep theirs)
};

// roctx range of a scope (ranges around the phases of the path, for `rocprofv3 --marker-trace`): the
// marker library is resolved at run time (librocprofiler-sdk-roctx.so, else libroctx64.so) and only when ME_AMD_ROCTX=1
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *name);
  ~RoctxRange();
};

// ---- CoordinateMapKey (src/coordinate_map_key.hpp:44-157) -----------------------------------------------------------
typedef std::pair<ivec, std::string> KeyT;   // coordinate_map_key_type (src/types.hpp:77-78)

struct CoordinateMapKey {
  int coordinate_size;
  bool key_set;
  KeyT key;
  explicit CoordinateMapKey(int coordinate_size_) : coordinate_size(coordinate_size_), key_set(false) {}
  CoordinateMapKey(const ivec &tensor_stride, const std::string &string_id)
      : coordinate_size((int)tensor_stride.size() + 1), key_set(true), key(tensor_stride, string_id) {}
  void set_key(const ivec &tensor_stride, const std::string &string_id);
  const KeyT &get() const {
    check(key_set, "Key not set");
    return key;
  }
  bool equals(const CoordinateMapKey &o) const { return key_set && o.key_set && key == o.key; }
  std::string repr() const;
};

// ---- device objects --------------------------------------------------------------------------------------------------
struct SpatialIndex {     // rows of a coordinate map in supercell order (csrc/coords.hip me_spatial_index_build)
  me_spatial_grid grid;
  int64_t m;
  Tensor order, pos_of_row, coords_sorted, dir_start;
};

struct CoordMap {         // one coordinate map resident in HBM (replaces CoordinateMapGPU, src/coordinate_map_gpu.cuh:47-223)
  Tensor coords, table;
  int64_t capacity = 0, n = 0;
  ivec tensor_stride;
  ivec bbox;              // column minima, then maxima; empty: none
  int spatial_state = 0;  // 0: not tried, 1: built, -1: not available
  std::shared_ptr<SpatialIndex> sp;
  std::shared_ptr<SpatialIndex> spatial();
  Tensor zorder_rows;     // rows in Z-order (me_coords_spatial_keys + argsort), built on first use: the halo kernel's tiles
  Tensor zorder();
  Tensor zorder_inverse;  // row -> its position in zorder()
  Tensor zorder_inv();
};

struct InsertResult {
  std::shared_ptr<CoordMap> map;
  Tensor unique_map, inverse_map;
};
InsertResult insert_coords(const Tensor &coords, const ivec &tensor_stride);

// per-offset pair prefix read back WITHOUT stalling the build (pinned copy + event; waited for on first host use)
struct LazyOffsets {
  Tensor pinned;
  void *event = nullptr;   // hipEvent_t
  std::vector<int64_t> values;
  bool ready = false;
  explicit LazyOffsets(const Tensor &k_offsets_dev);
  explicit LazyOffsets(std::vector<int64_t> v) : values(std::move(v)), ready(true) {}
  ~LazyOffsets();
  const std::vector<int64_t> &get();
};

struct Plan {
  Tensor plan_src, plan_dst, batch_desc, tile_bptr, item_gptr;
  bool failed = false;   // a deferred plan whose batched build threw: out of the caches, never launched (manager.cpp PlanBatch)
};
// plan of the output-stationary bf16 kernel (csrc/conv_halo.hip, me_halo_plan_build): tiles of `tile_rows` target
// positions, the distinct source rows of each (its halo, staged in LDS), local slots and group masks per offset
struct HaloPlan {
  int tile_rows = 0, s_cap = 0;
  Tensor halo_cnt, halo_rows, lidx, kmask;
  Tensor tbl, col_order, out_order;   // neighbour table; table column / target row of a tile position (may be undefined)
};

// Build requests a manager served, in order (the "recipe" a new scene's manager replays: CoordinateMapManager::prefetch).
// Locked: a loader thread may read the previous scene's log (to replay it on the next scene) while the training thread's
// layers still append to it.
struct RecipeLog {
  std::mutex mu;
  std::vector<std::string> v;
  void push_back(const std::string &r) {
    std::lock_guard<std::mutex> g(mu);
    v.push_back(r);
  }
  std::vector<std::string> snapshot() {
    std::lock_guard<std::mutex> g(mu);
    return v;
  }
};

struct ConvCfg {          // launch geometry of a (kernel map side, channel shape, dtype)
  int tile_rows, batch_groups;
  std::shared_ptr<Plan> plan;
  Tensor order;           // may be undefined
  int64_t elems;
  bool fuse, split;
  int split_k = 1;        // offset groups of a split-K launch (bf16 features on small maps; 1: not split)
  bool rowwise = false;   // the side has exactly one pair per target row: no plan, csrc/conv_rowwise.hip (round 6)
  std::shared_ptr<HaloPlan> halo;   // set: the launch runs on the halo kernel (me_conv_halo_use_bf16)
  int halo_at_use = 0;    // > 0: the policy wants the halo kernel here from that launch count on (me_conv_halo_min_uses)
  int uses = 0;           // launches that asked for this configuration (recipe replays do not count)
};
struct WgradCfg {
  std::vector<int64_t> koffs;
  int64_t ws_bytes;
};

struct KernelMapStore {   // buffers shared by a kernel map and its swapped view
  std::map<std::string, Tensor> t;
  std::map<std::string, std::shared_ptr<Plan>> plans;
  std::map<std::string, std::shared_ptr<HaloPlan>> halos;   // (a null entry: no spatial order for this side)
};

struct KernelMap : std::enable_shared_from_this<KernelMap> {
  int64_t volume, n_in, n_out;
  std::shared_ptr<CoordMap> in_map, out_map;
  std::shared_ptr<LazyOffsets> offsets;
  Tensor k_offsets_dev, in_pairs_buf, out_pairs_buf;
  std::shared_ptr<KernelMapStore> store;
  bool flip = false;
  // sides on which a row has AT MOST one pair by construction (bit 0: "in" rows, bit 1: "out" rows): the 1x1 identity map
  // (both), the fine side of a kernel_size == stride map.  With n_pairs == rows of the side: exactly one -> row-wise launch
  int one_pair_sides = 0;
  std::map<std::string, ConvCfg> conv_cfgs;
  std::map<std::string, WgradCfg> wgrad_cfgs;
  std::weak_ptr<RecipeLog> log;   // the owning manager's request log (build recipe) ...
  std::string log_key;                           // ... and this map's serialised cache key in it

  const std::vector<int64_t> &k_offsets() { return offsets->get(); }
  int64_t n_pairs() { return k_offsets().back(); }
  c10::Device device() const { return k_offsets_dev.device(); }
  std::string name(const std::string &kind, const std::string &target) const;
  std::shared_ptr<KernelMap> swapped();
  // (table, order): table [volume, n_tgt] of source ROWS indexed by target POSITION; order undefined when positions are rows
  std::pair<Tensor, Tensor> table_pos(const std::string &target);
  Tensor table(const std::string &target);      // row-space view
  std::string tile_order(const std::string &target, bool matrix_bound, int64_t src_bytes = 0);
  Tensor flat_order(const std::string &target, const std::string &tile_order);
  Tensor order(const std::string &target, const std::string &tile_order);
  std::shared_ptr<Plan> plan(const std::string &target, int tile_rows, int batch_groups, const std::string &tile_order);
  std::shared_ptr<HaloPlan> halo_plan(const std::string &target, int tile_rows, int s_cap);
  const ConvCfg &conv_cfg(const std::string &target, int64_t n_tgt, int c_src, int c_dst, bool bf16, bool no_rowwise = false);
  const WgradCfg &wgrad_cfg(int c_in, int c_out, bool bf16);
  pybind11::dict to_dict();
};

std::shared_ptr<KernelMap> build_kernel_map(const std::shared_ptr<CoordMap> &in_map,
                                            const std::shared_ptr<CoordMap> &out_map, const me_region &region);
me_region make_region(int ncol, int region_type, const ivec &kernel_size, const ivec &dilation, const ivec &tensor_stride);

// ---- CoordinateMapManager (src/coordinate_map_manager.{hpp,cpp,cu}) ---------------------------------------------------
typedef std::tuple<KeyT, KeyT, ivec, ivec, ivec, int, bool, bool> KernelMapKeyT;   // src/types.hpp:183-192

struct CoordinateMapManager {
  int algorithm, num_threads;
  std::map<KeyT, std::shared_ptr<CoordMap>> maps;
  std::vector<KeyT> map_order;      // insertion order (origin(): "any key"; __repr__)
  std::map<KernelMapKeyT, std::shared_ptr<KernelMap>> kernel_maps;
  std::map<KeyT, Tensor> origin_rows_cache;
  std::map<std::pair<KeyT, KeyT>, Tensor> prune_rows;
  std::map<std::pair<KeyT, KeyT>, std::pair<Tensor, Tensor>> stride_maps;

  CoordinateMapManager(int algorithm_ = 0, int num_threads_ = 0) : algorithm(algorithm_), num_threads(num_threads_) {}

  bool exists(const KeyT &k) const { return maps.count(k) != 0; }
  std::shared_ptr<CoordMap> get(const KeyT &k) const;
  void put(const KeyT &k, const std::shared_ptr<CoordMap> &m);
  KeyT random_string_id(const ivec &tensor_stride, const std::string &string_id);
  KeyT register_map(const ivec &ts, const std::shared_ptr<CoordMap> &m, const std::string &string_id);

  std::tuple<KeyT, Tensor, Tensor> insert_and_map(Tensor coordinates, const ivec &tensor_stride, const std::string &string_id);
  KeyT stride(const KeyT &in_key, const ivec &kernel_stride, const std::string &string_id);
  std::pair<Tensor, Tensor> stride_map(const KeyT &in_key, const KeyT &strided_key);
  std::pair<KeyT, bool> stride_region(const KeyT &in_key, const ivec &kernel_size, const ivec &kernel_dilation,
                                      int region_type, const ivec &out_tensor_stride, bool expand_coordinates,
                                      bool is_transpose, const ivec *region_tensor_stride);
  KeyT prune(const KeyT &in_key, const Tensor &keep);
  Tensor pruning_rows(const KeyT &in_key, const KeyT &out_key);
  std::vector<Tensor> union_map(const std::vector<KeyT> &in_keys, CoordinateMapKey *out_key);
  KeyT origin();
  Tensor origin_rows(const KeyT &in_key);
  std::shared_ptr<KernelMap> kernel_map(const KeyT &in_key, const KeyT &out_key, const ivec &kernel_size,
                                        const ivec &kernel_stride, const ivec &kernel_dilation, int region_type,
                                        bool is_transpose, bool is_pool);
  // every strided map, kernel map, tile plan and weight-gradient geometry a network asks for, built in ONE call for a
  // new scene (the replay of another scene's request log: docs/HISTORY.md 9.8, round-3 build recipe)
  std::shared_ptr<RecipeLog> recipe_log = std::make_shared<RecipeLog>();   // serialised requests
  int64_t prefetch(const std::vector<std::string> &recipe);
  void log_request(const std::string &r) { recipe_log->push_back(r); }
  std::vector<Tensor> device_tensors();
  std::string repr() const;
};

// ---- operators (ops.cpp) ----------------------------------------------------------------------------------------------
Tensor conv_forward_km(const Tensor &in_feat, const Tensor &kernel, KernelMap &km);
std::pair<Tensor, Tensor> conv_backward_km(const Tensor &in_feat, Tensor grad_out, const Tensor &kernel, KernelMap &km,
                                           bool need_grad_in);
std::shared_ptr<KernelMap> prepare_conv(const Tensor &in_feat, const Tensor &kernel, const ivec &kernel_size,
                                        const ivec &kernel_stride, const ivec &kernel_dilation, int region_type,
                                        bool expand_coordinates, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                                        CoordinateMapManager *manager, bool transpose);

std::pair<Tensor, Tensor> local_pooling_forward(const Tensor &in_feat, const ivec &ks, const ivec &st, const ivec &dl,
                                                int region_type, int pooling_mode, CoordinateMapKey *in_key,
                                                CoordinateMapKey *out_key, CoordinateMapManager *mgr);
Tensor local_pooling_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, const ivec &ks,
                              const ivec &st, const ivec &dl, int region_type, int pooling_mode,
                              CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr);
std::pair<Tensor, Tensor> local_pooling_transpose_forward(const Tensor &in_feat, const ivec &ks, const ivec &st,
                                                          const ivec &dl, int region_type, bool generate_new_coordinates,
                                                          int pooling_mode, CoordinateMapKey *in_key,
                                                          CoordinateMapKey *out_key, CoordinateMapManager *mgr);
Tensor local_pooling_transpose_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, const ivec &ks,
                                        const ivec &st, const ivec &dl, int region_type, int pooling_mode,
                                        CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr);
std::pair<Tensor, Tensor> global_pooling_forward(const Tensor &in_feat, int pooling_mode, CoordinateMapKey *in_key,
                                                 CoordinateMapKey *out_key, CoordinateMapManager *mgr);
Tensor global_pooling_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, int pooling_mode,
                               CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr);
Tensor broadcast_forward(const Tensor &in_feat, const Tensor &in_feat_glob, int broadcast_mode, CoordinateMapKey *in_key,
                         CoordinateMapKey *glob_key, CoordinateMapManager *mgr);
std::pair<Tensor, Tensor> broadcast_backward(const Tensor &in_feat, const Tensor &in_feat_glob, Tensor grad_out,
                                             int broadcast_mode, CoordinateMapKey *in_key, CoordinateMapKey *glob_key,
                                             CoordinateMapManager *mgr);
Tensor pruning_forward(const Tensor &in_feat, const Tensor &keep, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                       CoordinateMapManager *mgr);
Tensor pruning_backward(const Tensor &grad_out, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                        CoordinateMapManager *mgr);

// gradient destinations (ops.cpp; distributed.GradientArena): parameter -> the buffer its gradient is written into
void set_grad_destination(const Tensor &param, const Tensor &dest);   // dest undefined: forget it
void clear_grad_destinations();
void debug_fail_next_plan_batch();   // the next batched tile-plan build throws (tests)
void arm_grad_destinations();          // every entry may be used once (again)
void arm_grad_destinations_for(const std::vector<int64_t> &param_addresses);   // ... only these parameters' entries
void drop_grad_destinations(const std::vector<int64_t> &param_addresses);
Tensor grad_destination(const Tensor &param, at::IntArrayRef shape);

// batch normalisation over feature rows (csrc/norm.hip)
void conv_bn_stats_hint(bool flag);   // the convolution module's training flag (a training-mode batch norm may follow)
void set_conv_bn_stats(int enabled);  // -1: ME_AMD_CONV_BN_STATS, 0 / 1: override (tests)
std::pair<Tensor, Tensor> bn_stats(const Tensor &x, double eps, double momentum, const Tensor &running_mean,
                                   const Tensor &running_var, const Tensor &num_batches_tracked);
Tensor bn_apply(const Tensor &x, const Tensor &mean, const Tensor &rstd, const Tensor &gamma, const Tensor &beta,
                bool relu, const Tensor &skip);
std::tuple<Tensor, Tensor, Tensor, Tensor> bn_backward(const Tensor &x, Tensor dy, const Tensor &yout, const Tensor &mean,
                                                       const Tensor &rstd, const Tensor &gamma, const Tensor &beta,
                                                       bool relu, bool residual, bool need_dskip);

// optional per-launch timing with HIP events on the launch stream (bench.py: average duration of each hot kernel inside
// the timed region) — C++ twin of backend.KernelTimer
void timing_enable(bool on);
std::vector<std::tuple<std::string, double, double>> timing_records(bool clear);   // (name, ms, algorithmic flops)

// packed weight images cached per (weight tensor, direction) — C++ twin of backend._WeightPacker
Tensor packed_weights(const Tensor &kernel, int mode, bool transposed, int c_src, int c_dst, int64_t elems);
// every cached image is repacked at its next use: for weight updates the version counter cannot see (`p.data` writes)
void invalidate_packed_weights();
void invalidate_packed_weights_for(const std::vector<int64_t> &sorted_ptrs);
std::pair<int64_t, int64_t> invalidate_packed_weights_in(const std::vector<int64_t> &begins, const std::vector<int64_t> &ends);

}  // namespace meh
