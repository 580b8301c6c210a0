// CoordinateMapManager, coordinate maps, kernel maps and tile plans of the native host layer (see host.hpp).
// C++ twin of the corresponding classes of minkowskiengine_amd/backend.py; reference: src/coordinate_map_manager.cpp
// (insert_and_map :349-399, stride :402-429, kernel_map :655-823), src/coordinate_map_gpu.cu, src/kernel_map.cuh.
#include <cstring>
#include "host.hpp"
#include <functional>

#include <dlfcn.h>

#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <random>
#include <set>
#include <sstream>

namespace meh {

void fail(const std::string &msg) { throw std::runtime_error("assertion failed. " + msg); }

void me_ok(int rc) {
  if (rc != 0) {
    const char *m = me_last_error();
    throw std::runtime_error(m && *m ? std::string(m) : "libme_amd call failed with code " + std::to_string(rc));
  }
}

static std::string env_str(const char *name, const char *dflt) {
  const char *v = std::getenv(name);
  return v ? std::string(v) : std::string(dflt);
}

const Policy &Policy::get() {
  static Policy p = [] {
    Policy q;
    const std::string sm = env_str("ME_AMD_SPATIAL_MAPS", "auto");
    q.spatial_maps = sm == "1" ? 1 : (sm == "0" ? 0 : -1);
    q.rowwise = env_str("ME_AMD_ROWWISE", "1") != "0";
    q.rowwise_min_rows_with_stats = std::atoll(env_str("ME_AMD_ROWWISE_STATS_ROWS", "100000").c_str());
    q.tile_order = env_str("ME_AMD_TILE_ORDER", "auto");
    q.tile_spatial_src_bytes = (int64_t)std::atoi(env_str("ME_AMD_TILE_SPATIAL_SRC_MB", "28").c_str()) << 20;
    q.bf16_fuse = env_str("ME_AMD_BF16_FUSE", "auto");
    const std::string fs = env_str("ME_AMD_F32_SPLIT", "auto");
    q.f32_split = fs == "1" ? 1 : (fs == "0" ? 0 : -1);
    q.tile_rows = std::atoi(env_str("ME_AMD_TILE_ROWS", "0").c_str());
    q.batch_groups = std::atoi(env_str("ME_AMD_BATCH_GROUPS", "0").c_str());
    q.pack_cache = env_str("ME_AMD_PACK_CACHE", "1") != "0";
    q.f32_fuse = env_str("ME_AMD_F32_FUSE", "1") != "0";
    q.conv_bn_stats = env_str("ME_AMD_CONV_BN_STATS", "1") != "0";
    q.roctx = env_str("ME_AMD_ROCTX", "0") != "0";
    return q;
  }();
  return p;
}

// tuning / test hook: the integer policies by name (the environment is read once, at the first use)
void Policy::set(const std::string &name, int64_t value) {
  Policy &p = const_cast<Policy &>(get());
  if (name == "tile_spatial_src_bytes") p.tile_spatial_src_bytes = value;
  else if (name == "spatial_maps") p.spatial_maps = (int)value;
  else if (name == "rowwise") p.rowwise = value != 0;
  else if (name == "rowwise_min_rows_with_stats") p.rowwise_min_rows_with_stats = value;
  else if (name == "f32_split") p.f32_split = (int)value;
  else if (name == "tile_rows") p.tile_rows = (int)value;
  else if (name == "batch_groups") p.batch_groups = (int)value;
  else check(false, "unknown policy: " + name);
}

namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)();
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool roctx_resolve() {
  static const bool ok = [] {
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
        g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
        g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return true;
      }
    }
    return false;
  }();
  return ok;
}
}  // namespace

RoctxRange::RoctxRange(const char *name) : on(Policy::get().roctx && roctx_resolve()) {
  if (on) g_roctx_push(name);
}
RoctxRange::~RoctxRange() {
  if (on) g_roctx_pop();
}

// ---- CoordinateMapKey -------------------------------------------------------------------------------------------------
void CoordinateMapKey::set_key(const ivec &tensor_stride, const std::string &string_id) {
  check(coordinate_size - 1 == (int)tensor_stride.size(), "Invalid tensor_stride size");
  key = KeyT(tensor_stride, string_id);
  key_set = true;
}

static std::string ivec_str(const ivec &v) {
  std::ostringstream o;
  o << "[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i];
  o << "]";
  return o.str();
}

std::string CoordinateMapKey::repr() const {
  if (!key_set) return "coordinate map key: (unset)";
  std::string s = "coordinate map key:" + ivec_str(key.first);
  if (!key.second.empty()) s += ":" + key.second;
  return s;
}

me_region make_region(int ncol, int region_type, const ivec &kernel_size, const ivec &dilation,
                      const ivec &tensor_stride) {
  me_region rg;
  rg.ncol = ncol;
  rg.region_type = region_type;
  for (int d = 0; d < ME_MAX_DIM; ++d) {
    rg.kernel_size[d] = d < ncol - 1 ? kernel_size[d] : 1;
    rg.dilation[d] = d < ncol - 1 ? dilation[d] : 1;
    rg.tensor_stride[d] = d < ncol - 1 ? tensor_stride[d] : 1;
  }
  return rg;
}

// ---- coordinate maps --------------------------------------------------------------------------------------------------
static int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

std::shared_ptr<SpatialIndex> CoordMap::spatial() {
  if (spatial_state != 0) return sp;
  spatial_state = -1;
  static const int kShift[8] = {0, 12, 6, 4, 3, 2, 2, 1};      // log2 side: <= 4096 cells per supercell
  static const int kMinShift[8] = {0, 6, 4, 3, 2, 1, 1, 1};
  const int ncol = (int)coords.size(1), D = ncol - 1;
  if (n == 0 || bbox.empty() || D < 1 || D > 7) return nullptr;
  for (int t : tensor_stride)
    if (t <= 0) return nullptr;
  me_spatial_grid g;
  std::memset(&g, 0, sizeof(g));
  g.ncol = ncol;
  g.sc_min[0] = bbox[0];
  g.sc_dim[0] = bbox[ncol] - bbox[0] + 1;
  // supercell side: the largest (<= 4096 cells) that still gives the probe kernel a few workgroups per CU
  int sh = kShift[D];
  int64_t m;
  while (true) {
    m = g.sc_dim[0];
    for (int d = 0; d < D; ++d) {
      const int ts = tensor_stride[d];
      const int lo = floordiv(bbox[1 + d], ts) >> sh;   // arithmetic shift of the floored quotient
      const int hi = floordiv(bbox[ncol + 1 + d], ts) >> sh;
      g.shift[d] = sh;
      g.tensor_stride[d] = ts;
      g.sc_min[1 + d] = lo;
      g.sc_dim[1 + d] = hi - lo + 1;
      m *= hi - lo + 1;
    }
    if (m >= 1024 || sh <= kMinShift[D]) break;
    --sh;
  }
  m = me_spatial_cells(&g);
  if (m < 1 || m > (1ll << 22)) return nullptr;
  const c10::Device dev = coords.device();
  auto s = std::make_shared<SpatialIndex>();
  s->grid = g;
  s->m = m;
  s->order = empty_i32({n}, dev);
  s->pos_of_row = empty_i32({n}, dev);
  s->coords_sorted = empty_i32({n, ncol}, dev);
  s->dir_start = empty_i32({m + 1}, dev);
  Tensor ws = workspace(me_spatial_index_workspace_bytes(n, m), dev);
  c10::DeviceGuard guard(dev);
  me_ok(me_spatial_index_build(ptr<int32_t>(coords), n, &g, ptr<int32_t>(s->order), ptr<int32_t>(s->pos_of_row),
                               ptr<int32_t>(s->coords_sorted), ptr<uint32_t>(s->dir_start), vptr(ws), ws.numel(),
                               stream_of(dev)));
  sp = s;
  spatial_state = 1;
  return sp;
}

InsertResult insert_coords(const Tensor &coords, const ivec &tensor_stride) {
  const c10::Device dev = coords.device();
  const int64_t n = coords.size(0);
  const int ncol = (int)coords.size(1);
  const int64_t cap = me_hash_capacity(n);
  Tensor table = at::empty({cap}, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor cu = empty_i32({n > 0 ? n : 1, ncol}, dev);
  Tensor um = at::empty({n > 0 ? n : 1}, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor im = at::empty({n > 0 ? n : 1}, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor ws = workspace(me_insert_workspace_bytes(n), dev);
  int64_t n_unique = 0;
  int32_t bbox[2 * (ME_MAX_DIM + 1)];
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_coords_insert_and_map_bbox(ptr<int32_t>(coords), n, ncol, ptr<uint64_t>(table), cap, ptr<int32_t>(cu),
                                        ptr<int64_t>(um), ptr<int64_t>(im), &n_unique, bbox, vptr(ws), ws.numel(),
                                        stream_of(dev)));
  }
  InsertResult r;
  r.map = std::make_shared<CoordMap>();
  r.map->coords = cu.narrow(0, 0, n_unique);
  r.map->table = table;
  r.map->capacity = cap;
  r.map->n = n_unique;
  r.map->tensor_stride = tensor_stride;
  if (n > 0) r.map->bbox.assign(bbox, bbox + 2 * ncol);
  r.unique_map = um.narrow(0, 0, n_unique);
  r.inverse_map = im.narrow(0, 0, n);
  return r;
}

// ---- lazy per-offset prefix -----------------------------------------------------------------------------------------
LazyOffsets::LazyOffsets(const Tensor &k_offsets_dev) {
  pinned = at::empty({k_offsets_dev.numel()}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
  pinned.copy_(k_offsets_dev, /*non_blocking=*/true);
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) fail("hipEventCreate failed");
  (void)hipEventRecord(ev, (hipStream_t)stream_of(k_offsets_dev.device()));
  event = ev;
}
LazyOffsets::~LazyOffsets() {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
}
const std::vector<int64_t> &LazyOffsets::get() {
  if (!ready) {
    (void)hipEventSynchronize((hipEvent_t)event);
    const int64_t *p = pinned.data_ptr<int64_t>();
    values.assign(p, p + pinned.numel());
    (void)hipEventDestroy((hipEvent_t)event);
    event = nullptr;
    pinned = Tensor();
    ready = true;
  }
  return values;
}

// ---- kernel maps --------------------------------------------------------------------------------------------------------
std::string KernelMap::name(const std::string &kind, const std::string &target) const {
  std::string t = target;
  if (flip) t = target == "out" ? "in" : "out";
  return kind + "_" + t;
}

std::shared_ptr<KernelMap> KernelMap::swapped() {
  auto k = std::make_shared<KernelMap>();
  k->volume = volume;
  k->n_in = n_out;
  k->n_out = n_in;
  k->in_map = out_map;
  k->out_map = in_map;
  k->offsets = offsets;
  k->k_offsets_dev = k_offsets_dev;
  k->in_pairs_buf = out_pairs_buf;
  k->out_pairs_buf = in_pairs_buf;
  k->store = store;
  k->flip = !flip;
  k->one_pair_sides = ((one_pair_sides & 1) << 1) | ((one_pair_sides >> 1) & 1);
  return k;
}

static Tensor store_get(KernelMapStore &s, const std::string &key) {
  auto it = s.t.find(key);
  return it == s.t.end() ? Tensor() : it->second;
}

std::pair<Tensor, Tensor> KernelMap::table_pos(const std::string &target) {
  const std::string nm = name("nbr", target);
  Tensor order_t = store_get(*store, name("order", target));
  if (!store->t.count(nm)) {
    // the missing table is the transpose of the existing one: scatter the pair lists
    const c10::Device dev = device();
    const int64_t n_tgt = target == "out" ? n_out : n_in;
    const Tensor &src_pairs = target == "out" ? in_pairs_buf : out_pairs_buf;   // values stored
    const Tensor &tgt_pairs = target == "out" ? out_pairs_buf : in_pairs_buf;   // rows indexed
    Tensor pos = store_get(*store, name("pos", target));
    Tensor tbl = empty_i32({volume, n_tgt > 0 ? n_tgt : 1}, dev);
    const int64_t bound = (target == "in" ? n_out : n_in) * volume;
    c10::DeviceGuard guard(dev);
    me_ok(me_kernel_map_transpose_ordered(ptr<int32_t>(tgt_pairs), ptr<int32_t>(src_pairs), ptr<int64_t>(k_offsets_dev),
                                          volume, bound, n_tgt, ptr<int32_t>(pos), ptr<int32_t>(tbl), stream_of(dev)));
    store->t[nm] = tbl;
  }
  return {store->t[nm], order_t};
}

Tensor KernelMap::table(const std::string &target) {
  auto tp = table_pos(target);
  if (!tp.second.defined()) return tp.first;
  const std::string nm = name("nbrrow", target);
  if (!store->t.count(nm)) {
    Tensor row = at::empty_like(tp.first);
    row.index_put_({at::indexing::Slice(), tp.second.to(at::kLong)},
                   tp.first.index({at::indexing::Slice(), at::indexing::Slice(0, tp.second.numel())}));
    store->t[nm] = row;
  }
  return store->t[nm];
}

std::string KernelMap::tile_order(const std::string &target, bool matrix_bound, int64_t src_bytes) {
  const Policy &p = Policy::get();
  if (p.tile_order != "auto") return p.tile_order;
  // The fp32 kernels on the bf16 matrix pipe take tiles in Z-ORDER of the target map where the neighbour table is in row
  // space (flat-table maps: the headline layer): a run of 128 Morton-ordered rows touches ~235 distinct source rows, a
  // run of the supercell order ~321 — config 2, round 6 (profiles/r06_headline_tile_order.log): input
  // gradient 116.0 -> 114.3 us and 3.06x -> 2.44x its compulsory traffic, forward 1.67x -> 1.52x.  A position-space table
  // (LDS-bucketed build) keeps its own supercell order.
  if (matrix_bound && !store_get(*store, name("order", target)).defined()) {
    auto cm = target == "out" ? out_map : in_map;
    if (cm && cm->n > 0) return "zorder";
  }
  // (bf16 launches whose source matrix no longer fits the eight L2s: see backend.py _TILE_SPATIAL_MIN_SRC_BYTES)
  if (matrix_bound || (p.tile_spatial_src_bytes > 0 && src_bytes >= p.tile_spatial_src_bytes)) return "spatial";
  const int64_t n_tgt = target == "out" ? n_out : n_in;
  return volume * n_tgt * 4 <= (32ll << 20) ? "rows" : "spatial";
}

Tensor KernelMap::flat_order(const std::string &target, const std::string &tile_order_) {
  auto cmap = target == "out" ? out_map : in_map;
  if (tile_order_ == "zorder" && cmap && cmap->n > 0) return cmap->zorder();
  if (tile_order_ == "spatial" && cmap && cmap->n > 0) {
    auto s = cmap->spatial();
    if (s) return s->order;
  }
  return Tensor();
}

Tensor KernelMap::order(const std::string &target, const std::string &tile_order_) {
  Tensor native = store_get(*store, name("order", target));
  if (native.defined()) return tile_order_ == "spatial" ? native : Tensor();
  return flat_order(target, tile_order_);
}

// Plans requested while a batch is open (CoordinateMapManager::prefetch) are allocated and cached at once but BUILT
// together when the batch closes: me_plan_build_multi, four launches for all of them (round 4)
struct PlanBatch {
  std::vector<me_plan_job> jobs;
  std::vector<Tensor> keep;      // neighbour tables / tile orders / plan arrays of the jobs
  std::vector<std::function<void()>> undo;   // takes a deferred plan (and the configurations holding it) out of the caches
  c10::Device dev = c10::Device(c10::kCPU);
};
static thread_local PlanBatch *g_plan_batch = nullptr;

static void build_plan_batch(PlanBatch &b);

// A deferred plan is in the cache BEFORE it is built.  If the build fails (allocation, geometry) the cached plans of the
// batch hold uninitialised arrays: they — and the launch configurations that point at them — leave the caches before the
// error travels on, so that a caller who catches it (an out-of-memory retry) rebuilds them on the next use.
static void flush_plan_batch(PlanBatch &b) {
  if (b.jobs.empty()) return;
  try {
    build_plan_batch(b);
  } catch (...) {
    for (auto &u : b.undo) u();
    b.jobs.clear();
    b.keep.clear();
    b.undo.clear();
    throw;
  }
  b.undo.clear();
}

static bool g_fail_next_plan_batch = false;   // debug_fail_next_plan_batch(): fault injection for the test of the undo path
void debug_fail_next_plan_batch() { g_fail_next_plan_batch = true; }

static void build_plan_batch(PlanBatch &b) {
  RoctxRange rx("me:tile_plans_multi");
  if (g_fail_next_plan_batch) {
    g_fail_next_plan_batch = false;
    check(false, "injected failure of a batched tile-plan build (debug_fail_next_plan_batch)");
  }
  const int64_t total = me_plan_jobs_init(b.jobs.data(), (int32_t)b.jobs.size());
  check(total >= 0, "invalid plan geometry in a batch of plans");
  const int64_t bytes = (int64_t)(b.jobs.size() * sizeof(me_plan_job));
  Tensor host = at::empty({bytes}, at::TensorOptions().dtype(at::kByte).pinned_memory(true));
  std::memcpy(host.data_ptr(), b.jobs.data(), (size_t)bytes);
  c10::DeviceGuard guard(b.dev);
  Tensor jobs_dev = at::empty({bytes}, at::TensorOptions().dtype(at::kByte).device(b.dev));
  jobs_dev.copy_(host, /*non_blocking=*/true);
  Tensor ws = workspace(me_plan_multi_workspace_bytes(total), b.dev);
  me_ok(me_plan_build_multi(b.jobs.data(), reinterpret_cast<const me_plan_job *>(jobs_dev.data_ptr()),
                            (int32_t)b.jobs.size(), vptr(ws), ws.numel(), stream_of(b.dev)));
  b.jobs.clear();
  b.keep.clear();
}

std::shared_ptr<Plan> KernelMap::plan(const std::string &target, int tile_rows, int batch_groups,
                                      const std::string &tile_order_) {
  const std::string nm = name("plan", target) + "_" + std::to_string(tile_rows) + "_" + std::to_string(batch_groups) +
                         "_" + tile_order_;
  auto it = store->plans.find(nm);
  if (it != store->plans.end()) return it->second;
  RoctxRange rx("me:tile_plan");
  const c10::Device dev = device();
  const int64_t n_tgt = target == "out" ? n_out : n_in;
  auto tp = table_pos(target);
  // (pair lists built without a read-back hold n_pairs only on the device: the plan is sized by the upper bound)
  const int64_t pairs_bound = offsets->ready ? n_pairs() : std::min<int64_t>(in_pairs_buf.numel(), n_tgt * volume);
  const int64_t max_groups = me_plan_max_groups(n_tgt, volume, pairs_bound, tile_rows);
  const int64_t n_tiles = me_plan_num_tiles(n_tgt, tile_rows);
  auto p = std::make_shared<Plan>();
  p->plan_src = empty_i32({max_groups * ME_GROUP_ROWS}, dev);
  p->plan_dst = empty_i32({max_groups * ME_GROUP_ROWS}, dev);
  p->batch_desc = empty_i32({2 * max_groups}, dev);
  p->tile_bptr = empty_i32({me_plan_tile_bptr_elems(n_tgt, tile_rows)}, dev);
  p->item_gptr = empty_i32({n_tiles * volume + 1}, dev);
  Tensor gather_order;
  if (tp.second.defined()) gather_order = tile_order_ == "spatial" ? Tensor() : store_get(*store, name("pos", target));
  else gather_order = flat_order(target, tile_order_);
  if (g_plan_batch != nullptr && (g_plan_batch->jobs.empty() || g_plan_batch->dev == dev)) {
    // (a deferred plan is in the cache before it is built: refuse what me_plan_build_multi would refuse NOW)
    check(volume >= 1 && volume <= 65535 && tile_rows >= ME_GROUP_ROWS && tile_rows <= ME_MAX_TILE_ROWS &&
              batch_groups >= 1 && batch_groups <= ME_MAX_BATCH_GROUPS && n_tiles * volume < (1ll << 31),
          "invalid tile-plan geometry");
    me_plan_job j;
    std::memset(&j, 0, sizeof(j));
    j.tbl = ptr<int32_t>(tp.first);
    j.order = ptr<int32_t>(gather_order);
    j.n_tgt = n_tgt;
    j.volume = volume;
    j.tile_rows = tile_rows;
    j.batch_groups = batch_groups;
    j.plan_src = ptr<int32_t>(p->plan_src);
    j.plan_dst = ptr<int32_t>(p->plan_dst);
    j.batch_desc = ptr<int32_t>(p->batch_desc);
    j.tile_bptr = ptr<int32_t>(p->tile_bptr);
    j.item_gptr = ptr<int32_t>(p->item_gptr);
    g_plan_batch->dev = dev;
    g_plan_batch->jobs.push_back(j);
    g_plan_batch->keep.push_back(tp.first);
    if (gather_order.defined()) g_plan_batch->keep.push_back(gather_order);
    store->plans[nm] = p;
    {
      std::shared_ptr<KernelMapStore> st = store;
      KernelMap *self = this;     // (alive for the batch: the manager holds the kernel maps while it replays a recipe)
      g_plan_batch->undo.push_back([st, self, nm, p]() {
        p->failed = true;           // (configurations of other views of this map that hold it: see conv_cfg)
        st->plans.erase(nm);
        self->conv_cfgs.clear();
      });
    }
    return p;
  }
  Tensor ws = workspace(me_plan_workspace_bytes(n_tgt, volume, tile_rows), dev);
  c10::DeviceGuard guard(dev);
  me_ok(me_plan_build(ptr<int32_t>(tp.first), ptr<int32_t>(gather_order), n_tgt, volume, tile_rows, batch_groups,
                      ptr<int32_t>(p->plan_src), ptr<int32_t>(p->plan_dst), ptr<int32_t>(p->batch_desc),
                      ptr<int32_t>(p->tile_bptr), ptr<int32_t>(p->item_gptr), vptr(ws), ws.numel(), stream_of(dev)));
  store->plans[nm] = p;
  return p;
}

// Rows in Z-order (Morton keys of the coordinates in units of the tensor stride, batch index on top): runs of consecutive
// entries are spatially compact at EVERY length — the supercell order of spatial() only down to a supercell.
Tensor CoordMap::zorder() {
  if (!zorder_rows.defined() && n > 0) {
    const c10::Device dev = coords.device();
    const int32_t ncol = (int32_t)coords.size(1);
    std::vector<int32_t> ts(tensor_stride.begin(), tensor_stride.end());
    std::vector<int32_t> bb(bbox.begin(), bbox.end());
    Tensor order = empty_i32({n}, dev);
    Tensor ws = workspace(me_coords_zorder_workspace_bytes(n), dev);
    c10::DeviceGuard guard(dev);
    // (the library's own radix sort over the key bytes that can differ inside the map's bounding box: no torch sort on
    // the map path — round 6)
    me_ok(me_coords_zorder(ptr<int32_t>(coords), n, ncol, ts.data(), (int64_t)bb.size() == 2 * ncol ? bb.data() : nullptr,
                           ptr<int32_t>(order), vptr(ws), ws.numel(), stream_of(dev)));
    zorder_rows = order;
  }
  return zorder_rows;
}

Tensor CoordMap::zorder_inv() {
  if (!zorder_inverse.defined() && n > 0) {
    Tensor z = zorder();
    zorder_inverse = at::empty_like(z);
    zorder_inverse.index_put_({z.to(at::kLong)}, at::arange(n, z.options()));
  }
  return zorder_inverse;
}

// Halo plan of a launch side (csrc/conv_halo.hip): tiles are runs of target rows in the Z-order of the target's
// coordinate map (small halos); null when the side has no coordinate map attached.
std::shared_ptr<HaloPlan> KernelMap::halo_plan(const std::string &target, int tile_rows, int s_cap) {
  const std::string nm = name("halo", target) + "_" + std::to_string(tile_rows) + "_" + std::to_string(s_cap);
  auto it = store->halos.find(nm);
  if (it != store->halos.end()) return it->second;
  RoctxRange rx("me:halo_plan");
  const c10::Device dev = device();
  const int64_t n_tgt = target == "out" ? n_out : n_in;
  auto tp = table_pos(target);
  auto h = std::make_shared<HaloPlan>();
  h->tile_rows = tile_rows;
  h->s_cap = s_cap;
  h->tbl = tp.first;
  auto cmap = target == "out" ? out_map : in_map;
  if (cmap) h->out_order = cmap->zorder();
  if (!h->out_order.defined() || h->out_order.numel() != n_tgt) {
    store->halos[nm] = nullptr;
    return nullptr;
  }
  // a position-space table (LDS-bucketed map build) is read through pos_of_row
  h->col_order = tp.second.defined() ? store_get(*store, name("pos", target)).index({h->out_order.to(at::kLong)}).contiguous()
                                     : h->out_order;
  // the halo slots in the Z-order of the SOURCE map: rows that are gathered together sit in neighbouring slots
  auto smap = target == "out" ? in_map : out_map;
  Tensor src_order = smap ? smap->zorder() : Tensor(), src_pos = smap ? smap->zorder_inv() : Tensor();
  const int64_t tiles = me_halo_plan_num_tiles(n_tgt, tile_rows);
  h->halo_cnt = empty_i32({tiles}, dev);
  h->halo_rows = empty_i32({tiles * s_cap}, dev);
  h->lidx = at::empty({tiles * volume * tile_rows}, at::TensorOptions().dtype(at::kShort).device(dev));
  h->kmask = empty_i32({tiles * volume}, dev);
  c10::DeviceGuard guard(dev);
  me_ok(me_halo_plan_build(ptr<int32_t>(h->tbl), ptr<int32_t>(h->col_order), ptr<int32_t>(src_pos), ptr<int32_t>(src_order), n_tgt,
                           volume, tile_rows, s_cap,
                           ptr<int32_t>(h->halo_cnt), ptr<int32_t>(h->halo_rows), ptr<uint16_t>(h->lidx),
                           ptr<uint32_t>(h->kmask), stream_of(dev)));
  store->halos[nm] = h;
  return h;
}

static bool use_split(int c_src, int c_dst) {
  const Policy &p = Policy::get();
  if (p.f32_split == 0 || !me_conv_f32x3_supported(c_src, c_dst)) return false;
  return p.f32_split == 1 ? true : (int64_t)c_src * c_dst >= 8192;
}

const ConvCfg &KernelMap::conv_cfg(const std::string &target, int64_t n_tgt, int c_src, int c_dst, bool bf16,
                                   bool no_rowwise) {
  const Policy &pol = Policy::get();
  const bool split = !bf16 && use_split(c_src, c_dst);
  const std::string ck = target + "/" + std::to_string(c_src) + "/" + std::to_string(c_dst) + (bf16 ? "/b" : "/f") +
                         (split ? "s" : "-") + (no_rowwise ? "/p" : "");
  auto it = conv_cfgs.find(ck);
  if (it != conv_cfgs.end()) {
    if (!(it->second.plan && it->second.plan->failed)) {
      ConvCfg &c = it->second;
      if (c.halo_at_use > 0 && g_plan_batch == nullptr && ++c.uses >= c.halo_at_use) {
        // the side is launched again (a reused scene): now the halo plan pays (me_conv_halo_min_uses)
        c.halo_at_use = 0;
        int32_t ht = 0, hc = 0;
        std::shared_ptr<HaloPlan> h;
        if (me_conv_halo_config_bf16(n_tgt, volume, n_pairs(), c_src, c_dst, &ht, &hc)) h = halo_plan(target, ht, hc);
        if (h) {
          c.halo = h;
          c.tile_rows = ht;
          c.batch_groups = 0;
          c.split = false;
          c.fuse = false;
          c.split_k = 1;
          c.elems = me_conv_packed_weight_elems_bf16(volume, c_src, c_dst);
        }
      }
      return c;
    }
    conv_cfgs.erase(it);   // (held a plan of a batch whose build failed: configured again)
  }
  // the plan geometry depends on the pair count (density): the one host value a first launch on a new map waits for
  const int64_t np = n_pairs();
  if (bf16 && pol.rowwise && !no_rowwise && (one_pair_sides & (target == "in" ? 1 : 2)) && np == n_tgt &&
      me_conv_rowwise_supported_bf16(volume, c_src, c_dst)) {
    // every target row has exactly one pair (at most one by construction, and as many pairs as rows): no sum, no plan —
    // out[t] = src[s(t)] @ W[k(t)] straight off the pair lists (csrc/conv_rowwise.hip)
    ConvCfg c;
    c.tile_rows = 0;
    c.batch_groups = 0;
    c.split = false;
    c.fuse = false;
    c.rowwise = true;
    c.elems = me_conv_packed_weight_elems_bf16(volume, c_src, c_dst);
    if (auto lg = log.lock())
      lg->push_back("conv_cfg;" + log_key + ";" + target + ";" + std::to_string(c_src) + ";" + std::to_string(c_dst) + ";1");
    return conv_cfgs.emplace(ck, std::move(c)).first->second;
  }
  int32_t t = 0, g = 0, sk = 1;
  if (bf16 && !pol.tile_rows && !pol.batch_groups)
    me_ok(me_conv_plan_config_bf16_ex(n_tgt, volume, np, c_src, c_dst, &t, &g, &sk));   // may answer a split-K geometry
  else
    me_ok((bf16 ? me_conv_plan_config_bf16 : (split ? me_conv_plan_config_f32x3 : me_conv_plan_config))(
        n_tgt, volume, np, c_src, c_dst, &t, &g));
  ConvCfg c;
  const bool halo_wanted = bf16 && me_conv_halo_use_bf16(n_tgt, volume, np, c_src, c_dst);
  const int halo_min_uses = halo_wanted ? me_conv_halo_min_uses() : 0;
  if (halo_wanted && halo_min_uses > 1) {
    // ... from the halo_min_uses-th launch on: the plan costs more than one launch saves (a scene used once never builds it)
    c.halo_at_use = halo_min_uses;
    c.uses = g_plan_batch == nullptr ? 1 : 0;
  }
  if (halo_wanted && halo_min_uses <= 1) {
    // libme_amd's policy sends this launch side to the output-stationary kernel on an LDS-staged halo
    int32_t ht = 0, hc = 0;
    if (me_conv_halo_config_bf16(n_tgt, volume, np, c_src, c_dst, &ht, &hc)) c.halo = halo_plan(target, ht, hc);
    if (c.halo) {
      c.tile_rows = ht;
      c.batch_groups = 0;
      c.split = false;
      c.fuse = false;
      c.elems = me_conv_packed_weight_elems_bf16(volume, c_src, c_dst);
      if (auto lg = log.lock())
        lg->push_back("conv_cfg;" + log_key + ";" + target + ";" + std::to_string(c_src) + ";" + std::to_string(c_dst) + ";1");
      return conv_cfgs.emplace(ck, std::move(c)).first->second;
    }
  }
  c.tile_rows = pol.tile_rows ? pol.tile_rows : t;
  c.batch_groups = pol.batch_groups ? pol.batch_groups : g;
  c.split = split;
  c.split_k = sk;
  const int64_t n_src = target == "out" ? n_in : n_out;
  const std::string to = tile_order(target, split, bf16 ? n_src * c_src * 2 : 0);
  c.plan = plan(target, c.tile_rows, c.batch_groups, to);
  c.elems = (bf16 ? me_conv_packed_weight_elems_bf16
                  : (split ? me_conv_packed_weight_elems_f32x3 : me_conv_packed_weight_elems))(volume, c_src, c_dst);
  c.order = order(target, to);
  const int64_t n_tiles = (n_tgt + c.tile_rows - 1) / c.tile_rows;
  const double per_item = volume > 1 ? (double)(np - std::min(n_in, n_out)) / std::max<int64_t>(1, (volume - 1) * n_tiles)
                                     : 1e9;
  c.fuse = pol.bf16_fuse == "1" ? true : (pol.bf16_fuse == "0" ? false : per_item < 24.0);
  if (c.split_k > 1) c.fuse = false;
  if (auto lg = log.lock())
    lg->push_back("conv_cfg;" + log_key + ";" + target + ";" + std::to_string(c_src) + ";" + std::to_string(c_dst) + ";" +
                  (bf16 ? "1" : "0") + (no_rowwise ? ";p" : ""));   // (";p": the tile plan of a side that could run row-wise)
  return conv_cfgs.emplace(ck, std::move(c)).first->second;
}

const WgradCfg &KernelMap::wgrad_cfg(int c_in, int c_out, bool bf16) {
  const std::string ck = std::to_string(c_in) + "/" + std::to_string(c_out) + (bf16 ? "/b" : "/f");
  auto it = wgrad_cfgs.find(ck);
  if (it != wgrad_cfgs.end()) return it->second;
  WgradCfg w;
  w.koffs = k_offsets();
  w.ws_bytes = (bf16 ? me_conv_wgrad_workspace_bytes_bf16 : me_conv_wgrad_workspace_bytes)(w.koffs.data(), volume, c_in,
                                                                                           c_out);
  if (auto lg = log.lock())
    lg->push_back("wgrad_cfg;" + log_key + ";" + std::to_string(c_in) + ";" + std::to_string(c_out) + ";" +
                  (bf16 ? "1" : "0"));
  return wgrad_cfgs.emplace(ck, std::move(w)).first->second;
}

pybind11::dict KernelMap::to_dict() {
  pybind11::dict out;
  const auto &ko = k_offsets();
  for (int64_t k = 0; k < volume; ++k) {
    const int64_t b = ko[k], e = ko[k + 1];
    if (e > b) out[pybind11::int_(k)] = at::stack({in_pairs_buf.narrow(0, b, e - b), out_pairs_buf.narrow(0, b, e - b)});
  }
  return out;
}

// LDS-bucketed build (csrc/coords.hip k_kmap_probe_lds): position-space table, no host synchronisation
static std::shared_ptr<KernelMap> build_kernel_map_lds(const std::shared_ptr<CoordMap> &in_map,
                                                       const std::shared_ptr<CoordMap> &out_map, const me_region &region,
                                                       int64_t volume) {
  const Policy &pol = Policy::get();
  if (pol.spatial_maps == 0 || in_map->tensor_stride != out_map->tensor_stride || in_map->n == 0 || out_map->n == 0)
    return nullptr;
  if (pol.spatial_maps == -1 && volume < 64 && out_map->n * volume < (1ll << 24)) return nullptr;
  auto sq = out_map->spatial(), sl = in_map->spatial();
  if (!sq || !sl) return nullptr;
  if (me_kernel_map_probe_lds_bytes(&region, &sq->grid, &sl->grid) < 0) return nullptr;
  const c10::Device dev = in_map->coords.device();
  const int64_t n_out = out_map->n, n_in = in_map->n;
  Tensor nbr = empty_i32({volume, n_out}, dev);
  Tensor ws = workspace(me_kernel_map_workspace_bytes(n_out, volume), dev);
  Tensor koffs = at::empty({volume + 1}, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor in_pairs = empty_i32({n_out * volume}, dev), out_pairs = empty_i32({n_out * volume}, dev);
  auto km = std::make_shared<KernelMap>();
  {
    c10::DeviceGuard guard(dev);
    void *st = stream_of(dev);
    me_ok(me_kernel_map_probe_lds(&sq->grid, ptr<int32_t>(sq->coords_sorted), ptr<uint32_t>(sq->dir_start), n_out,
                                  &sl->grid, ptr<int32_t>(sl->coords_sorted), ptr<int32_t>(sl->order),
                                  ptr<uint32_t>(sl->dir_start), &region, ptr<int32_t>(nbr), ptr<int64_t>(koffs), vptr(ws),
                                  ws.numel(), st));
    km->offsets = std::make_shared<LazyOffsets>(koffs);
    me_ok(me_kernel_map_compact_ordered(ptr<int32_t>(nbr), ptr<int32_t>(sq->order), n_out, volume,
                                        ptr<int32_t>(in_pairs), ptr<int32_t>(out_pairs), vptr(ws), ws.numel(), st));
  }
  km->volume = volume;
  km->n_in = n_in;
  km->n_out = n_out;
  km->in_map = in_map;
  km->out_map = out_map;
  km->k_offsets_dev = koffs;
  km->in_pairs_buf = in_pairs;
  km->out_pairs_buf = out_pairs;
  km->store = std::make_shared<KernelMapStore>();
  km->store->t["nbr_out"] = nbr;
  km->store->t["order_out"] = sq->order;
  km->store->t["pos_out"] = sq->pos_of_row;
  km->store->t["order_in"] = sl->order;
  km->store->t["pos_in"] = sl->pos_of_row;
  return km;
}

// iterate OUT coordinates, look up the IN map (src/coordinate_map_cpu.hpp:569-670)
std::shared_ptr<KernelMap> build_kernel_map(const std::shared_ptr<CoordMap> &in_map,
                                            const std::shared_ptr<CoordMap> &out_map, const me_region &region) {
  const int64_t volume = me_region_volume(&region);
  check(volume > 0, "invalid kernel region");
  RoctxRange rx("me:kernel_map");
  if (auto km = build_kernel_map_lds(in_map, out_map, region, volume)) return km;
  const c10::Device dev = in_map->coords.device();
  const int64_t n_out = out_map->n, n_in = in_map->n;
  Tensor nbr = empty_i32({volume, n_out > 0 ? n_out : 1}, dev);
  Tensor ws = workspace(me_kernel_map_workspace_bytes(n_out, volume), dev);
  Tensor koffs = at::empty({volume + 1}, at::TensorOptions().dtype(at::kLong).device(dev));
  // flat-table probe WITHOUT the read-back of the pair counts (k_offsets = NULL, ABI 1.2): the pair lists are
  // allocated at their bound and the prefix reaches the host lazily, as in the LDS-bucketed build
  const int64_t bound = std::max<int64_t>(n_out * volume, 1);
  Tensor in_pairs = empty_i32({bound}, dev), out_pairs = empty_i32({bound}, dev);
  auto km = std::make_shared<KernelMap>();
  {
    c10::DeviceGuard guard(dev);
    void *st = stream_of(dev);
    me_ok(me_kernel_map_probe(ptr<uint64_t>(in_map->table), in_map->capacity, ptr<int32_t>(in_map->coords),
                              ptr<int32_t>(out_map->coords), n_out, &region, ptr<int32_t>(nbr), nullptr,
                              ptr<int64_t>(koffs), vptr(ws), ws.numel(), st));
    km->offsets = std::make_shared<LazyOffsets>(koffs);
    me_ok(me_kernel_map_compact(ptr<int32_t>(nbr), n_out, volume, ptr<int32_t>(in_pairs), ptr<int32_t>(out_pairs),
                                vptr(ws), ws.numel(), st));
  }
  km->volume = volume;
  km->n_in = n_in;
  km->n_out = n_out;
  km->in_map = in_map;
  km->out_map = out_map;
  km->k_offsets_dev = koffs;
  km->in_pairs_buf = in_pairs;
  km->out_pairs_buf = out_pairs;
  km->store = std::make_shared<KernelMapStore>();
  km->store->t["nbr_out"] = nbr;
  return km;
}

// ---- manager ------------------------------------------------------------------------------------------------------------
std::shared_ptr<CoordMap> CoordinateMapManager::get(const KeyT &k) const {
  auto it = maps.find(k);
  check(it != maps.end(), "coordinate map not found " + ivec_str(k.first) + ":" + k.second);
  return it->second;
}

void CoordinateMapManager::put(const KeyT &k, const std::shared_ptr<CoordMap> &m) {
  if (!maps.count(k)) map_order.push_back(k);
  maps[k] = m;
}

KeyT CoordinateMapManager::random_string_id(const ivec &tensor_stride, const std::string &string_id) {
  // src/coordinate_map_manager.hpp:473-485
  static const char cs[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz";
  static std::mt19937 rng(std::random_device{}());
  while (true) {
    std::string rnd;
    for (int i = 0; i < 5; ++i) rnd += cs[rng() % 62];
    KeyT key(tensor_stride, string_id.empty() ? rnd : string_id + "-" + rnd);
    if (!maps.count(key)) return key;
  }
}

KeyT CoordinateMapManager::register_map(const ivec &ts, const std::shared_ptr<CoordMap> &m, const std::string &string_id) {
  KeyT key(ts, string_id);
  if (maps.count(key)) key = random_string_id(ts, string_id);
  put(key, m);
  return key;
}

// Code objects of every translation unit of libme_amd, loaded on `dev` at the first map insert there (once per device of
// the process, under the device's guard) — not at import: a rank imports the package before it selects its device, and
// code objects are per device.  ME_AMD_PRELOAD=0 keeps HIP's lazy loading.
void preload_device(const c10::Device &dev) {
  static std::mutex mu;
  static std::set<int> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.insert((int)dev.index()).second) return;
  }
  if (env_str("ME_AMD_PRELOAD", "1") == "0") return;
  c10::DeviceGuard guard(dev);
  me_ok(me_preload());
}

std::tuple<KeyT, Tensor, Tensor> CoordinateMapManager::insert_and_map(Tensor coordinates, const ivec &tensor_stride,
                                                                       const std::string &string_id) {
  RoctxRange rx("me:insert_and_map");
  check(coordinates.dim() == 2, "coordinates must be 2-D");
  check(coordinates.is_contiguous(), "coordinates must be contiguous");
  check(coordinates.scalar_type() == at::kInt, "coordinates must be int32");
  check(coordinates.is_cuda(), "coordinates must be on the GPU (the MI355X path has no CPU map)");
  check(coordinates.size(1) - 1 == (int64_t)tensor_stride.size(),
        "The coordinate dimension (coordinate_size - 1) must match the size of tensor stride");
  preload_device(coordinates.device());
  KeyT key(tensor_stride, string_id);
  if (maps.count(key)) key = random_string_id(tensor_stride, string_id);
  if ((uintptr_t)coordinates.data_ptr() % 16 != 0) coordinates = coordinates.clone();
  InsertResult r = insert_coords(coordinates, tensor_stride);
  put(key, r.map);
  return {key, r.unique_map, r.inverse_map};
}

static std::string key_ser(const KeyT &k) {
  std::string s;
  for (size_t i = 0; i < k.first.size(); ++i) s += (i ? "," : "") + std::to_string(k.first[i]);
  return s + "|" + k.second;
}
static std::string vec_ser(const ivec &v) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]);
  return s;
}

KeyT CoordinateMapManager::stride(const KeyT &in_key, const ivec &kernel_stride, const std::string &string_id) {
  RoctxRange rx("me:stride");
  check(maps.count(in_key), "coordinate map not found");
  check(kernel_stride.size() == in_key.first.size(), "stride size mismatch.");
  ivec out_ts(in_key.first.size());
  for (size_t i = 0; i < out_ts.size(); ++i) {
    check(kernel_stride[i] > 0, "Invalid stride");
    out_ts[i] = in_key.first[i] * kernel_stride[i];
  }
  KeyT ok(out_ts, string_id.empty() ? in_key.second : string_id);
  if (!maps.count(ok)) {
    auto in_map = maps[in_key];
    const c10::Device dev = in_map->coords.device();
    const int ncol = (int)out_ts.size() + 1;
    Tensor strided = empty_i32({in_map->n > 0 ? in_map->n : 1, ncol}, dev);
    {
      c10::DeviceGuard guard(dev);
      me_ok(me_coords_stride(ptr<int32_t>(in_map->coords), in_map->n, ncol, out_ts.data(), ptr<int32_t>(strided),
                             stream_of(dev)));
    }
    InsertResult r = insert_coords(strided.narrow(0, 0, in_map->n), out_ts);
    put(ok, r.map);
    log_request("stride;" + key_ser(in_key) + ";" + vec_ser(kernel_stride) + ";" + string_id);
  }
  return ok;
}

std::pair<Tensor, Tensor> CoordinateMapManager::stride_map(const KeyT &in_key, const KeyT &strided_key) {
  check(maps.count(in_key) && maps.count(strided_key), "coordinate map not found");
  for (size_t i = 0; i < in_key.first.size(); ++i)
    check(in_key.first[i] > 0 && strided_key.first[i] % in_key.first[i] == 0,
          "The tensor stride of the strided map must be divisible by the tensor stride of the input map.");
  auto it = stride_maps.find({in_key, strided_key});
  if (it != stride_maps.end()) return it->second;
  auto in_map = maps[in_key], out_map = maps[strided_key];
  const c10::Device dev = in_map->coords.device();
  const int ncol = (int)strided_key.first.size() + 1;
  Tensor strided = empty_i32({in_map->n > 0 ? in_map->n : 1, ncol}, dev);
  Tensor rows = empty_i32({in_map->n > 0 ? in_map->n : 1}, dev);
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_coords_stride(ptr<int32_t>(in_map->coords), in_map->n, ncol, strided_key.first.data(), ptr<int32_t>(strided),
                           stream_of(dev)));
    me_ok(me_coords_find(ptr<uint64_t>(out_map->table), out_map->capacity, ptr<int32_t>(out_map->coords), ncol,
                         ptr<int32_t>(strided), in_map->n, ptr<int32_t>(rows), stream_of(dev)));
  }
  Tensor r64 = rows.narrow(0, 0, in_map->n).to(at::kLong);
  Tensor in_rows = at::arange(in_map->n, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor found = r64.ge(0);
  if (!found.all().item<bool>()) {
    in_rows = in_rows.index({found});
    r64 = r64.index({found});
  }
  stride_maps[{in_key, strided_key}] = {in_rows, r64};
  return {in_rows, r64};
}

std::pair<KeyT, bool> CoordinateMapManager::stride_region(const KeyT &in_key, const ivec &kernel_size,
                                                          const ivec &kernel_dilation, int region_type,
                                                          const ivec &out_tensor_stride, bool expand_coordinates,
                                                          bool is_transpose, const ivec *region_tensor_stride) {
  // src/coordinate_map_manager.cpp:436-466, src/coordinate_map_cpu.hpp:446-487
  check(maps.count(in_key), "coordinate map not found");
  check(region_type != 2, "Not implemented yet.");
  KeyT ok(out_tensor_stride, "");
  if (maps.count(ok) && !expand_coordinates) return {ok, false};
  auto in_map = maps[in_key];
  const c10::Device dev = in_map->coords.device();
  const int ncol = (int)out_tensor_stride.size() + 1;
  const ivec &rts = region_tensor_stride ? *region_tensor_stride : out_tensor_stride;
  me_region region = make_region(ncol, region_type, kernel_size, kernel_dilation, rts);
  const int64_t volume = me_region_volume(&region);
  const int64_t total = std::max<int64_t>(in_map->n * volume, 1);
  Tensor cand = empty_i32({total, ncol}, dev);
  Tensor aligned;
  if (!is_transpose) aligned = at::empty({total}, at::TensorOptions().dtype(at::kByte).device(dev));
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_coords_expand_region(ptr<int32_t>(in_map->coords), in_map->n, ncol, &region,
                                  is_transpose ? nullptr : out_tensor_stride.data(), ptr<int32_t>(cand),
                                  ptr<uint8_t>(aligned), stream_of(dev)));
  }
  cand = cand.narrow(0, 0, in_map->n * volume);
  if (aligned.defined()) cand = cand.index({aligned.narrow(0, 0, in_map->n * volume).to(at::kBool)}).contiguous();
  InsertResult r = insert_coords(cand, out_tensor_stride);
  return {register_map(out_tensor_stride, r.map, ""), true};
}

KeyT CoordinateMapManager::prune(const KeyT &in_key, const Tensor &keep) {
  check(maps.count(in_key), "coordinate map not found");
  auto in_map = maps[in_key];
  check(keep.dim() == 1 && keep.numel() == in_map->n, "Invalid range for pruning");
  Tensor rows = at::nonzero(keep.to(in_map->coords.device()).to(at::kBool)).squeeze(1);
  InsertResult r = insert_coords(in_map->coords.index({rows}).contiguous(), in_key.first);
  KeyT key = register_map(in_key.first, r.map, "pruned");
  prune_rows[{in_key, key}] = rows.to(at::kInt);
  return key;
}

Tensor CoordinateMapManager::pruning_rows(const KeyT &in_key, const KeyT &out_key) {
  auto it = prune_rows.find({in_key, out_key});
  if (it != prune_rows.end()) return it->second;
  auto in_map = get(in_key), out_map = get(out_key);
  const c10::Device dev = in_map->coords.device();
  Tensor rows = empty_i32({out_map->n > 0 ? out_map->n : 1}, dev);
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_coords_find(ptr<uint64_t>(in_map->table), in_map->capacity, ptr<int32_t>(in_map->coords),
                         (int)in_map->coords.size(1), ptr<int32_t>(out_map->coords), out_map->n, ptr<int32_t>(rows),
                         stream_of(dev)));
  }
  rows = rows.narrow(0, 0, out_map->n);
  check(rows.ge(0).all().item<bool>(), "the pruned map is not a subset of the input map");
  prune_rows[{in_key, out_key}] = rows;
  return rows;
}

std::vector<Tensor> CoordinateMapManager::union_map(const std::vector<KeyT> &in_keys, CoordinateMapKey *out_key) {
  check(in_keys.size() > 1, "Number of input coordinate keys must be > 1");
  std::vector<Tensor> cs;
  std::vector<std::shared_ptr<CoordMap>> ms;
  for (const KeyT &k : in_keys) {
    check(maps.count(k), "coordinate map not found");
    check(k.first == in_keys[0].first, "Invalid tensor stride");
    ms.push_back(maps[k]);
    cs.push_back(maps[k]->coords);
  }
  Tensor allc = at::cat(cs, 0).contiguous();
  InsertResult r = insert_coords(allc, in_keys[0].first);
  if (!out_key->key_set) {
    KeyT key = register_map(in_keys[0].first, r.map, "union");
    out_key->set_key(key.first, key.second);
  } else {
    put(out_key->get(), r.map);
  }
  std::vector<Tensor> out;
  int64_t s0 = 0;
  for (auto &m : ms) {
    Tensor rows = at::arange(m->n, at::TensorOptions().dtype(at::kLong).device(allc.device()));
    out.push_back(at::stack({rows, r.inverse_map.narrow(0, s0, m->n)}));
    s0 += m->n;
  }
  return out;
}

KeyT CoordinateMapManager::origin() {
  check(!maps.empty(), "origin() needs at least one coordinate map");
  const size_t D = map_order.front().first.size();
  KeyT okey(ivec(D, 0), "");
  if (!maps.count(okey)) {
    // the map with the smallest tensor stride holds every batch index
    const KeyT *best = nullptr;
    int64_t best_sum = 0;
    for (const auto &kv : maps) {
      bool pos = true;
      int64_t sum = 0;
      for (int t : kv.first.first) {
        pos = pos && t > 0;
        sum += t;
      }
      if (!pos) continue;
      if (!best || sum < best_sum || (sum == best_sum && kv.first.second < best->second)) {
        best = &kv.first;
        best_sum = sum;
      }
    }
    check(best != nullptr, "origin() needs a coordinate map with a positive tensor stride");
    auto base = maps[*best];
    Tensor batches = std::get<0>(at::_unique(base->coords.select(1, 0), /*sorted=*/true));
    Tensor oc = at::zeros({batches.numel(), base->coords.size(1)}, base->coords.options());
    oc.select(1, 0).copy_(batches);
    InsertResult r = insert_coords(oc.contiguous(), okey.first);   // unique rows: insertion order = sorted order
    put(okey, r.map);
  }
  return okey;
}

Tensor CoordinateMapManager::origin_rows(const KeyT &in_key) {
  const KeyT okey = origin();
  auto it = origin_rows_cache.find(in_key);
  if (it != origin_rows_cache.end()) return it->second;
  auto in_map = get(in_key), omap = get(okey);
  const c10::Device dev = in_map->coords.device();
  Tensor q = at::zeros_like(in_map->coords);
  q.select(1, 0).copy_(in_map->coords.select(1, 0));
  Tensor rows = empty_i32({in_map->n > 0 ? in_map->n : 1}, dev);
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_coords_find(ptr<uint64_t>(omap->table), omap->capacity, ptr<int32_t>(omap->coords), (int)q.size(1),
                         ptr<int32_t>(q), in_map->n, ptr<int32_t>(rows), stream_of(dev)));
  }
  rows = rows.narrow(0, 0, in_map->n);
  check(in_map->n == 0 || rows.ge(0).all().item<bool>(),
        "the origin map does not contain every batch index of this coordinate map");
  origin_rows_cache[in_key] = rows;
  return rows;
}

// Sides of a hyper-cube map (looked-up map `fine_ts`, iterated map `coarse_ts`) on which a row has at most ONE pair by
// construction: windows of kernel_size voxels that tile space without overlap (coarse stride = kernel_size x fine stride,
// no dilation) hold every fine voxel exactly once -> bit 0 (the "in" = fine side).  A single-offset kernel pairs a row
// with at most one row on BOTH sides.
static int one_pair_sides_of(const ivec &kernel_size, const ivec &dilation, int region_type, const ivec &fine_ts,
                             const ivec &coarse_ts, bool) {
  bool single = true, tiling = region_type == 0 && fine_ts.size() == kernel_size.size() && coarse_ts.size() == kernel_size.size();
  for (size_t i = 0; i < kernel_size.size(); ++i) {
    single = single && kernel_size[i] == 1;
    tiling = tiling && dilation[i] == 1 && coarse_ts[i] == fine_ts[i] * kernel_size[i];
  }
  return single ? 3 : (tiling ? 1 : 0);
}

std::shared_ptr<KernelMap> CoordinateMapManager::kernel_map(const KeyT &in_key, const KeyT &out_key,
                                                            const ivec &kernel_size, const ivec &kernel_stride,
                                                            const ivec &kernel_dilation, int region_type,
                                                            bool is_transpose, bool is_pool) {
  // src/coordinate_map_manager.cpp:655-823 (cached; transposed maps reuse the swapped forward map :763-774)
  check(region_type != 2, "Not implemented yet.");
  check(kernel_size.size() == kernel_stride.size() && kernel_size.size() == kernel_dilation.size(), "kernel size mismatch");
  KernelMapKeyT key(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, is_transpose, is_pool);
  auto it = kernel_maps.find(key);
  if (it != kernel_maps.end()) return it->second;
  auto in_map = get(in_key), out_map = get(out_key);
  check((int64_t)kernel_size.size() + 1 == in_map->coords.size(1), "kernel size mismatch");
  std::shared_ptr<KernelMap> km;
  bool all_one = true;
  for (int k : kernel_size) all_one = all_one && k == 1;
  if (in_key == out_key && all_one) {
    // a 1x1 kernel on one map: every row is paired with itself (src/coordinate_map_cpu.hpp:605-616)
    const int64_t n = in_map->n;
    const c10::Device dev = in_map->coords.device();
    Tensor rows = at::arange(n, at::TensorOptions().dtype(at::kInt).device(dev));
    km = std::make_shared<KernelMap>();
    km->volume = 1;
    km->n_in = km->n_out = n;
    km->in_map = in_map;
    km->out_map = out_map;
    km->offsets = std::make_shared<LazyOffsets>(std::vector<int64_t>{0, n});
    km->k_offsets_dev = at::tensor(std::vector<int64_t>{0, n}, at::TensorOptions().dtype(at::kLong)).to(dev);
    km->in_pairs_buf = km->out_pairs_buf = rows;
    km->store = std::make_shared<KernelMapStore>();
    km->store->t["nbr_out"] = n ? rows.view({1, n}) : empty_i32({1, 1}, dev);
    km->one_pair_sides = 3;
    kernel_maps[key] = km;
    return km;
  }
  if (!is_transpose) {
    me_region region = make_region((int)kernel_size.size() + 1, region_type, kernel_size, kernel_dilation,
                                   in_map->tensor_stride);
    km = build_kernel_map(in_map, out_map, region);
    km->one_pair_sides = one_pair_sides_of(kernel_size, kernel_dilation, region_type, in_map->tensor_stride,
                                           out_map->tensor_stride, false);
  } else {
    KernelMapKeyT swapped_key(out_key, in_key, kernel_size, kernel_stride, kernel_dilation, region_type, false, is_pool);
    auto fit = kernel_maps.find(swapped_key);
    std::shared_ptr<KernelMap> fwd;
    if (fit != kernel_maps.end()) {
      fwd = fit->second;
    } else {
      // out -> in map with the (finer) out tensor stride, then swap
      me_region region = make_region((int)kernel_size.size() + 1, region_type, kernel_size, kernel_dilation,
                                     out_map->tensor_stride);
      fwd = build_kernel_map(out_map, in_map, region);
      fwd->one_pair_sides = one_pair_sides_of(kernel_size, kernel_dilation, region_type, out_map->tensor_stride,
                                              in_map->tensor_stride, false);
    }
    km = fwd->swapped();
  }
  kernel_maps[key] = km;
  km->log_key = key_ser(in_key) + ";" + key_ser(out_key) + ";" + vec_ser(kernel_size) + ";" + vec_ser(kernel_stride) +
                ";" + vec_ser(kernel_dilation) + ";" + std::to_string(region_type) + ";" + (is_transpose ? "1" : "0") +
                ";" + (is_pool ? "1" : "0");
  km->log = recipe_log;
  log_request("kernel_map;" + km->log_key);
  return km;
}

// ---- build recipe: replay of another scene's request log ----------------------------------------------------------------
static std::vector<std::string> split(const std::string &s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) {
    if (c == sep) {
      out.push_back(cur);
      cur.clear();
    } else {
      cur += c;
    }
  }
  out.push_back(cur);
  return out;
}
static ivec vec_de(const std::string &s) {
  ivec v;
  if (s.empty()) return v;
  for (const auto &p : split(s, ',')) v.push_back(std::atoi(p.c_str()));
  return v;
}
static KeyT key_de(const std::string &s) {
  const size_t bar = s.find('|');
  return KeyT(vec_de(s.substr(0, bar)), s.substr(bar + 1));
}

int64_t CoordinateMapManager::prefetch(const std::vector<std::string> &recipe) {
  // Every strided coordinate map and kernel map of the network first (their builds enqueue kernels and asynchronous
  // copies of the per-offset prefixes, no read-back apart from the n_unique of each strided map), THEN the tile plans
  // and weight-gradient geometries, which need the pair counts on the host: by then every prefix copy is in flight
  // and the first wait covers them all — one drain of the queue per scene instead of one per layer.
  // (The replay logs every request it serves exactly once, on its cache miss — as the layers would have — and requests
  // that hit log nothing: the manager's own recipe stays complete, so the NEXT scene can be prefetched from this one.)
  int64_t done = 0;
  std::vector<std::pair<std::shared_ptr<KernelMap>, std::vector<std::string>>> cfgs;
  for (const std::string &line : recipe) {
    const auto f = split(line, ';');
    if (f[0] == "stride" && f.size() >= 4) {
      const KeyT ik = key_de(f[1]);
      if (maps.count(ik)) {
        stride(ik, vec_de(f[2]), f[3]);
        ++done;
      }
    } else if (f[0] == "kernel_map" && f.size() >= 9) {
      const KeyT ik = key_de(f[1]), ok = key_de(f[2]);
      if (maps.count(ik) && maps.count(ok)) {
        kernel_map(ik, ok, vec_de(f[3]), vec_de(f[4]), vec_de(f[5]), std::atoi(f[6].c_str()), f[7] == "1", f[8] == "1");
        ++done;
      }
    } else if ((f[0] == "conv_cfg" && f.size() >= 13) || (f[0] == "wgrad_cfg" && f.size() >= 12)) {
      const KeyT ik = key_de(f[1]), ok = key_de(f[2]);
      KernelMapKeyT key(ik, ok, vec_de(f[3]), vec_de(f[4]), vec_de(f[5]), std::atoi(f[6].c_str()), f[7] == "1",
                        f[8] == "1");
      auto it = kernel_maps.find(key);
      if (it != kernel_maps.end()) cfgs.push_back({it->second, f});
    }
  }
  // (the plans of all these configurations are built together: see PlanBatch)
  PlanBatch batch;
  struct BatchScope {
    PlanBatch *prev;
    explicit BatchScope(PlanBatch *b) : prev(g_plan_batch) { g_plan_batch = b; }
    ~BatchScope() { g_plan_batch = prev; }
  };
  try {
    BatchScope scope(&batch);
    for (auto &c : cfgs) {
      const auto &f = c.second;
      KernelMap &km = *c.first;
      if (f[0] == "conv_cfg") {
        const std::string target = f[9];
        km.conv_cfg(target, target == "out" ? km.n_out : km.n_in, std::atoi(f[10].c_str()), std::atoi(f[11].c_str()),
                    f[12] == "1", f.size() >= 14 && f[13] == "p");
      } else {
        km.wgrad_cfg(std::atoi(f[9].c_str()), std::atoi(f[10].c_str()), f[11] == "1");
      }
      ++done;
    }
  } catch (...) {
    flush_plan_batch(batch);   // the plans requested so far are in the cache: they must exist
    throw;
  }
  flush_plan_batch(batch);
  return done;
}

static void collect(std::vector<Tensor> &out, const Tensor &t) {
  if (t.defined() && t.is_cuda()) out.push_back(t);
}

std::vector<Tensor> CoordinateMapManager::device_tensors() {
  std::vector<Tensor> out;
  for (auto &kv : maps) {
    collect(out, kv.second->coords);
    collect(out, kv.second->table);
    if (kv.second->sp) {
      collect(out, kv.second->sp->order);
      collect(out, kv.second->sp->pos_of_row);
      collect(out, kv.second->sp->coords_sorted);
      collect(out, kv.second->sp->dir_start);
    }
  }
  for (auto &kv : kernel_maps) {
    KernelMap &km = *kv.second;
    collect(out, km.k_offsets_dev);
    collect(out, km.in_pairs_buf);
    collect(out, km.out_pairs_buf);
    for (auto &t : km.store->t) collect(out, t.second);
    for (auto &p : km.store->plans) {
      collect(out, p.second->plan_src);
      collect(out, p.second->plan_dst);
      collect(out, p.second->batch_desc);
      collect(out, p.second->tile_bptr);
      collect(out, p.second->item_gptr);
    }
    for (auto &h : km.store->halos) {
      if (!h.second) continue;
      collect(out, h.second->halo_cnt);
      collect(out, h.second->halo_rows);
      collect(out, h.second->lidx);
      collect(out, h.second->kmask);
    }
  }
  for (auto &kv : origin_rows_cache) collect(out, kv.second);
  for (auto &kv : prune_rows) collect(out, kv.second);
  for (auto &kv : stride_maps) {
    collect(out, kv.second.first);
    collect(out, kv.second.second);
  }
  return out;
}

std::string CoordinateMapManager::repr() const {
  std::ostringstream s;
  s << "CoordinateMapManagerGPU_c10(\n";
  for (const KeyT &k : map_order) {
    auto it = maps.find(k);
    s << "\t" << ivec_str(k.first) << (k.second.empty() ? "" : ":" + k.second) << ":\tCoordinateMapGPU:" << it->second->n
      << "x" << it->second->coords.size(1) << "\n";
  }
  for (const auto &kv : kernel_maps)
    s << "\t" << ivec_str(std::get<0>(kv.first).first) << "->" << ivec_str(std::get<1>(kv.first).first)
      << ":\tgpu_kernel_map: number of unique maps:" << kv.second->volume << "\n";
  static const char *algo[] = {"DEFAULT", "MEMORY_EFFICIENT", "SPEED_OPTIMIZED"};
  s << "\talgorithm=" << algo[algorithm >= 0 && algorithm <= 2 ? algorithm : 0] << "\n)";
  return s.str();
}

}  // namespace meh
