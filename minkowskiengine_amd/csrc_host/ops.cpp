// Operators of the native host layer (see host.hpp): C++ twin of the operator functions of minkowskiengine_amd/backend.py.
// Reference entry points: src/convolution_gpu.cu:45-244, src/convolution_transpose_gpu.cu, src/local_pooling_gpu.cu,
// src/global_pooling_gpu.cu, src/broadcast_gpu.cu, src/pruning_gpu.cu (signatures pybind/extern.hpp:53-392).
#include "host.hpp"

#include <c10/util/intrusive_ptr.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <mutex>

namespace meh {

static void check_feat(const char *name, const Tensor &t) {
  check(t.is_contiguous(), std::string(name) + " must be contiguous");
  check(t.is_cuda(), std::string(name) + " must be CUDA (ROCm) — the MI355X path has no CPU implementation");
  check(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kDouble,
        std::string(name) + " must be float32, bfloat16 or float64");
}

// ---- packed weight images: cached per (weight storage, direction), validated by the tensor's version counter ------------
// (C++ twin of backend._WeightPacker; every stale image of the device is repacked by ONE launch, csrc/pack.hip)
namespace {
struct PackEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> ref{c10::weak_intrusive_ptr<c10::TensorImpl>(
      c10::intrusive_ptr<c10::TensorImpl>())};
  const void *ptr = nullptr;
  std::vector<int64_t> shape;
  at::ScalarType dtype = at::kFloat;
  int64_t version = -1;
  int64_t epoch = -1;          // g_pack_epoch at pack time (invalidate_packed_weights)
  bool packed_once = false;
  Tensor packed;
  me_pack_job job;
};
typedef std::tuple<const void *, int, bool> PackKey;

struct Packer {
  std::map<PackKey, std::shared_ptr<PackEntry>> entries;
  std::vector<PackEntry *> table_key;
  Tensor table_jobs, table_prefix;
  int64_t table_total = 0;
  std::vector<std::shared_ptr<PackEntry>> table_entries;
};
std::map<int, Packer> g_packers;
// The version counter does not see writes through `.data` (p.data.add_(1) leaves p._version alone): optimizers and
// utilities that update weights that way (Apex / DeepSpeed style, EMA, clipping) call invalidate_packed_weights(), and
// the package does so itself after every torch.optim step (a global step post-hook, __init__.py).  An image is valid
// for (version, epoch); the epoch only grows.  Backward runs on autograd worker threads: one mutex for the packers.
std::atomic<int64_t> g_pack_epoch{0};
std::mutex g_packers_mu;

// version counter and liveness of the tensor an entry was made for (a full view is anchored at its base)
bool entry_alive(const PackEntry &e, int64_t *version) {
  auto strong = e.ref.lock();
  if (!strong) return false;
  if (strong->storage().data() == nullptr) return false;
  const char *p = static_cast<const char *>(strong->storage().data()) + strong->storage_offset() * strong->itemsize();
  if (p != e.ptr) return false;
  *version = (int64_t)strong->version_counter().current_version();
  return true;
}

void pack_one(PackEntry &e, const c10::Device &dev) {
  c10::DeviceGuard guard(dev);
  const me_pack_job &j = e.job;
  if (j.mode == ME_PACK_BF16)
    me_ok(me_conv_pack_weights_bf16(j.w, j.w_is_f32, j.volume, j.c_src, j.c_dst, j.transposed, (uint16_t *)j.wp,
                                    stream_of(dev)));
  else
    me_ok(me_conv_pack_weights_f32x3((const float *)j.w, j.volume, j.c_src, j.c_dst, j.transposed, (uint16_t *)j.wp,
                                     stream_of(dev)));
}
}  // namespace

void invalidate_packed_weights() { g_pack_epoch.fetch_add(1); }

// only the images of the weights at these storage addresses (the optimizer-step hook passes the stepping optimizer's
// own parameters: frozen / teacher networks keep their images)
void invalidate_packed_weights_for(const std::vector<int64_t> &ptrs) {
  std::lock_guard<std::mutex> lk(g_packers_mu);
  for (auto &pk : g_packers)
    for (auto &kv : pk.second.entries)
      if (std::binary_search(ptrs.begin(), ptrs.end(), (int64_t)(intptr_t)kv.second->ptr)) kv.second->epoch = -1;
}

// ... of the weights that live inside one of the byte ranges [begins[i], ends[i]) (sorted, disjoint or nested starts):
// a kernel that is an offset view of a stepping parameter is matched too.  -> (images matched, images cached): a caller
// that matched nothing holds master copies, not the model's tensors, and bumps the global epoch
std::pair<int64_t, int64_t> invalidate_packed_weights_in(const std::vector<int64_t> &begins,
                                                         const std::vector<int64_t> &ends) {
  std::lock_guard<std::mutex> lk(g_packers_mu);
  int64_t matched = 0, total = 0;
  for (auto &pk : g_packers)
    for (auto &kv : pk.second.entries) {
      ++total;
      const int64_t a = (int64_t)(intptr_t)kv.second->ptr;
      auto it = std::upper_bound(begins.begin(), begins.end(), a);
      if (it == begins.begin()) continue;
      const size_t i = (size_t)(it - begins.begin()) - 1;
      if (i < ends.size() && a < ends[i]) {
        kv.second->epoch = -1;
        ++matched;
      }
    }
  return {matched, total};
}

Tensor packed_weights(const Tensor &kernel, int mode, bool transposed, int c_src, int c_dst, int64_t elems) {
  const c10::Device dev = kernel.device();
  std::lock_guard<std::mutex> lk(g_packers_mu);
  const int64_t epoch = g_pack_epoch.load();
  Packer &pk = g_packers[dev.index()];
  const PackKey key(kernel.data_ptr(), mode, transposed);
  const int64_t cur_version = (int64_t)kernel.unsafeGetTensorImpl()->version_counter().current_version();
  std::shared_ptr<PackEntry> ent;
  auto it = pk.entries.find(key);
  int64_t v = 0;
  if (it != pk.entries.end() && entry_alive(*it->second, &v) && it->second->shape == kernel.sizes().vec() &&
      it->second->dtype == kernel.scalar_type()) {
    ent = it->second;
    if (ent->packed_once && ent->version == cur_version && ent->epoch == epoch) return ent->packed;
  } else {
    if (pk.entries.size() > 4096) {
      for (auto i = pk.entries.begin(); i != pk.entries.end();) {
        int64_t vv;
        i = entry_alive(*i->second, &vv) ? std::next(i) : pk.entries.erase(i);
      }
    }
    ent = std::make_shared<PackEntry>();
    // anchor: the base of a full view (a 1x1 convolution passes kernel.unsqueeze(0)), else the tensor itself
    c10::intrusive_ptr<c10::TensorImpl> anchor = kernel.getIntrusivePtr();
    if (kernel.is_view()) {
      const at::TensorBase &base = kernel._base();
      if (base.defined() && base.data_ptr() == kernel.data_ptr() && base.numel() == kernel.numel())
        anchor = base.getIntrusivePtr();
    }
    ent->ref = c10::weak_intrusive_ptr<c10::TensorImpl>(anchor);
    ent->ptr = kernel.data_ptr();
    ent->shape = kernel.sizes().vec();
    ent->dtype = kernel.scalar_type();
    ent->packed = at::empty({elems}, at::TensorOptions().dtype(at::kBFloat16).device(dev));
    std::memset(&ent->job, 0, sizeof(ent->job));
    ent->job.w = kernel.data_ptr();
    ent->job.wp = ent->packed.data_ptr();
    ent->job.volume = kernel.size(0);
    ent->job.c_src = c_src;
    ent->job.c_dst = c_dst;
    ent->job.transposed = transposed ? 1 : 0;
    ent->job.w_is_f32 = kernel.scalar_type() == at::kFloat ? 1 : 0;
    ent->job.mode = mode;
    me_ok(me_conv_pack_job_init(&ent->job));
    pk.entries[key] = ent;
  }
  // a NEW entry is packed on its own, so that the set of established images (and the cached job table) repeats
  if (!ent->packed_once) {
    pack_one(*ent, dev);
    ent->packed_once = true;
    ent->version = cur_version;
    ent->epoch = epoch;
  }
  std::vector<std::pair<std::shared_ptr<PackEntry>, int64_t>> stale;
  for (auto &kv : pk.entries) {
    PackEntry &e = *kv.second;
    if (!e.packed_once) continue;
    int64_t ev = 0;
    if (kv.second == ent) ev = cur_version;
    else if (!entry_alive(e, &ev)) continue;
    if (ev != e.version || e.epoch != epoch) stale.push_back({kv.second, ev});
  }
  if (stale.empty()) return ent->packed;
  if (stale.size() > 1024) {   // (the wanted image is never the one that is dropped)
    const bool wanted_stale = ent->version != cur_version || ent->epoch != epoch;
    stale.erase(stale.begin(), stale.end() - 1024);
    if (wanted_stale && std::none_of(stale.begin(), stale.end(), [&](auto &s) { return s.first == ent; }))
      stale.front() = {ent, cur_version};
  }
  if (stale.size() == 1) {
    pack_one(*stale[0].first, dev);
    stale[0].first->version = stale[0].second;
    stale[0].first->epoch = epoch;
    return ent->packed;
  }
  std::vector<PackEntry *> tk;
  for (auto &s : stale) tk.push_back(s.first.get());
  if (tk != pk.table_key) {
    std::vector<me_pack_job> jobs;
    std::vector<int64_t> prefix{0};
    for (auto &s : stale) {
      jobs.push_back(s.first->job);
      prefix.push_back(prefix.back() + s.first->job.threads);
    }
    Tensor jb = at::empty({(int64_t)(jobs.size() * sizeof(me_pack_job))}, at::TensorOptions().dtype(at::kByte));
    std::memcpy(jb.data_ptr(), jobs.data(), jobs.size() * sizeof(me_pack_job));
    pk.table_jobs = jb.to(dev);
    pk.table_prefix = at::tensor(prefix, at::TensorOptions().dtype(at::kLong)).to(dev);
    pk.table_total = prefix.back();
    pk.table_key = tk;
    pk.table_entries.clear();
    for (auto &s : stale) pk.table_entries.push_back(s.first);
  }
  {
    c10::DeviceGuard guard(dev);
    me_ok(me_conv_pack_weights_multi((const me_pack_job *)pk.table_jobs.data_ptr(), (int)stale.size(),
                                     ptr<int64_t>(pk.table_prefix), pk.table_total, stream_of(dev)));
  }
  for (auto &s : stale) {
    s.first->version = s.second;
    s.first->epoch = epoch;
  }
  return ent->packed;
}

// ---- per-launch timing (bench.py) ---------------------------------------------------------------------------------------------
namespace {
struct TimedLaunch {
  std::string name;
  hipEvent_t a, b;
  double flops;
};
bool g_timing = false;
std::vector<TimedLaunch> g_timed;

struct ScopedTimer {   // records an event pair around the launches of its scope when timing is on; a roctx range always
  bool on;                // (when ME_AMD_ROCTX=1: "me:conv_forward" / "me:conv_dgrad" / "me:conv_wgrad")
  TimedLaunch t;
  hipStream_t st;
  std::string rx_name;
  RoctxRange rx;
  ScopedTimer(const char *name, double flops, void *stream)
      : on(g_timing), st((hipStream_t)stream), rx_name(std::string("me:") + name), rx(rx_name.c_str()) {
    if (!on) return;
    t.name = name;
    t.flops = flops;
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, st);
  }
  ~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(t.b, st);
    g_timed.push_back(t);
  }
};
}  // namespace

void timing_enable(bool on) { g_timing = on; }

std::vector<std::tuple<std::string, double, double>> timing_records(bool clear) {
  std::vector<std::tuple<std::string, double, double>> out;
  for (auto &t : g_timed) {
    (void)hipEventSynchronize(t.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    out.emplace_back(t.name, (double)ms, t.flops);
  }
  if (clear) {
    for (auto &t : g_timed) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    g_timed.clear();
  }
  return out;
}

// ---- statistics a convolution leaves behind for the batch norm that follows (me_conv_target_bf16_stats) ----------------
// Keyed by the output's storage address; valid for THAT TensorImpl at THAT version only (a recycled address, a view, an
// in-place update or a channel slice misses and bn_stats reads the matrix).  Consumed by the first bn_stats on it.
namespace {
struct BnPartials {
  c10::weak_intrusive_ptr<c10::TensorImpl> owner;
  uint32_t version;
  Tensor part;   // [2][tiles][c] float
  int tile_rows;
};
std::unordered_map<const void *, BnPartials> g_bn_partials;
std::mutex g_bn_partials_mu;
thread_local bool g_conv_bn_stats_hint = false;   // per thread: a loader thread (ScenePrefetcher) never sees the training thread's flag
int g_conv_bn_stats_enabled = -1;   // -1: Policy (ME_AMD_CONV_BN_STATS), 0 / 1: set_conv_bn_stats (tests)

void bn_partials_put(const Tensor &out, const Tensor &part, int tile_rows) {
  std::lock_guard<std::mutex> lk(g_bn_partials_mu);
  if (g_bn_partials.size() > 32) {    // outputs that were never normalised (their tensors are long gone)
    for (auto it = g_bn_partials.begin(); it != g_bn_partials.end();)
      it = it->second.owner.expired() ? g_bn_partials.erase(it) : std::next(it);
  }
  g_bn_partials.erase(out.data_ptr());
  g_bn_partials.emplace(out.data_ptr(), BnPartials{c10::weak_intrusive_ptr<c10::TensorImpl>(out.getIntrusivePtr()),
                                                   out._version(), part, tile_rows});
}

bool bn_partials_take(const Tensor &x, Tensor &part, int &tile_rows) {
  std::lock_guard<std::mutex> lk(g_bn_partials_mu);
  auto it = g_bn_partials.find(x.data_ptr());
  if (it == g_bn_partials.end()) return false;
  BnPartials e = std::move(it->second);
  g_bn_partials.erase(it);
  auto owner = e.owner.lock();
  if (!owner || owner.get() != x.unsafeGetTensorImpl() || x._version() != e.version) return false;
  if (e.part.size(2) != x.size(1)) return false;
  part = e.part;
  tile_rows = e.tile_rows;
  return true;
}
}  // namespace

void conv_bn_stats_hint(bool flag) { g_conv_bn_stats_hint = flag; }
void set_conv_bn_stats(int enabled) { g_conv_bn_stats_enabled = enabled; }
static bool conv_bn_stats_enabled() {
  return g_conv_bn_stats_enabled >= 0 ? g_conv_bn_stats_enabled != 0 : Policy::get().conv_bn_stats;
}

// ---- gradient destinations (distributed.GradientArena) ---------------------------------------------------------------------
// A parameter's gradient can be BORN in a caller-provided buffer — a slice of one flat all-reduce buffer per dtype — instead
// of a fresh allocation: the weight-gradient / batch-norm backward kernels write there, autograd's AccumulateGrad takes the
// returned alias as p.grad without a copy, and the data-parallel exchange is one RCCL all-reduce over the flat buffer (no
// per-parameter hooks, no bucket copies: torch DDP costs this host-bound step 1.7 ms on ONE rank).  Keyed by the
// parameter's storage address; fp32 only (the dtype the kernels accumulate in).
// An entry is used ONCE per arming (GradientArena.zero_grad arms all of them): a second backward pass before the next
// zero_grad — a gradient-accumulation window — gets fresh tensors, which autograd then ADDS to p.grad (the slice); writing
// into the slice again would destroy what it has accumulated.
namespace {
struct GradDest {
  Tensor dest;
  bool armed;
};
std::unordered_map<const void *, GradDest> g_grad_dest;
std::mutex g_grad_dest_mu;
}  // namespace

void set_grad_destination(const Tensor &param, const Tensor &dest) {
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  if (dest.defined()) g_grad_dest[param.data_ptr()] = GradDest{dest, false};
  else g_grad_dest.erase(param.data_ptr());
}
void arm_grad_destinations() {
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  for (auto &kv : g_grad_dest) kv.second.armed = true;
}
void arm_grad_destinations_for(const std::vector<int64_t> &ptrs) {   // one arena's own parameters
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  for (int64_t a : ptrs) {
    auto it = g_grad_dest.find(reinterpret_cast<const void *>((uintptr_t)a));
    if (it != g_grad_dest.end()) it->second.armed = true;
  }
}
void drop_grad_destinations(const std::vector<int64_t> &ptrs) {
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  for (int64_t a : ptrs) g_grad_dest.erase(reinterpret_cast<const void *>((uintptr_t)a));
}
void clear_grad_destinations() {
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  g_grad_dest.clear();
}
// a FRESH alias (its own TensorImpl: AccumulateGrad steals a gradient only when nobody else holds it) of the registered
// buffer in `shape`, or an undefined tensor
Tensor grad_destination(const Tensor &param, at::IntArrayRef shape) {
  if (!param.defined()) return Tensor();
  std::lock_guard<std::mutex> lk(g_grad_dest_mu);
  auto it = g_grad_dest.find(param.data_ptr());
  if (it == g_grad_dest.end() || !it->second.armed) return Tensor();
  const Tensor &d = it->second.dest;
  int64_t n = 1;
  for (auto v : shape) n *= v;
  if (d.scalar_type() != at::kFloat || d.numel() != n || d.device() != param.device()) return Tensor();
  it->second.armed = false;
  return d.view(shape);
}

// ---- convolution ----------------------------------------------------------------------------------------------------------
// dst[t] = sum over plan entries of src[s] @ W[k]; transposed (dgrad): W[k] = kernel[k]^T
static Tensor conv_target(const Tensor &src, const Tensor &kernel, KernelMap &km, const std::string &target, int64_t n_tgt,
                          bool transposed) {
  const c10::Device dev = src.device();
  const int64_t volume = kernel.size(0);
  const int c_src = (int)(transposed ? kernel.size(2) : kernel.size(1));
  const int c_dst = (int)(transposed ? kernel.size(1) : kernel.size(2));
  const bool bf16 = src.scalar_type() == at::kBFloat16;
  Tensor out = at::empty({n_tgt, c_dst}, src.options());
  if (n_tgt == 0) return out;
  if (src.scalar_type() == at::kDouble) {
    // float64 features (the reference's double instantiation, src/convolution_gpu.cu:137-155 — its gradcheck dtype):
    // csrc/f64.hip, plain double FMAs on the neighbour table; no plan, no packed weights
    check(kernel.scalar_type() == at::kDouble, "float64 features need a float64 kernel");
    Tensor tbl = km.table(target);
    Tensor w = kernel.contiguous();
    c10::DeviceGuard guard(dev);
    void *st = stream_of(dev);
    ScopedTimer tm(transposed ? "conv_dgrad" : "conv_forward", g_timing ? 2.0 * (double)km.n_pairs() * c_src * c_dst : 0.0, st);
    me_ok(me_conv_target_f64(ptr<double>(src), src.size(0), c_src, ptr<double>(w), transposed ? 1 : 0, volume, c_dst,
                             ptr<int32_t>(tbl), ptr<double>(out), n_tgt, st));
    return out;
  }
  if (bf16 && c_src == 8 && src.size(0) < (1ll << 28) && me_conv_stem_use_bf16(n_tgt, volume, c_src, c_dst)) {
    // at most 8 source channels (a stem): four offsets per MFMA step straight off the neighbour table and the layer's own
    // kernel tensor — no tile plan, no packed image (csrc/conv_stem.hip)
    check(kernel.scalar_type() == at::kFloat || kernel.scalar_type() == at::kBFloat16, "kernel must be float32 or bfloat16");
    auto tp = km.table_pos(target);
    Tensor w = kernel.contiguous();
    c10::DeviceGuard guard(dev);
    void *st = stream_of(dev);
    const int tile_rows = me_conv_stem_tile_rows();
    const bool want_stats = !transposed && target == "out" && g_conv_bn_stats_hint && conv_bn_stats_enabled();
    Tensor part;
    int64_t tiles = 0;
    if (want_stats) {
      tiles = (n_tgt + tile_rows - 1) / tile_rows;
      part = at::empty({2, tiles, (int64_t)c_dst}, at::TensorOptions().dtype(at::kFloat).device(dev));
    }
    {
      ScopedTimer tm(transposed ? "conv_dgrad" : "conv_forward", g_timing ? 2.0 * (double)km.n_pairs() * c_src * c_dst : 0.0, st);
      me_ok(me_conv_stem_bf16(ptr<uint16_t>(src), src.size(0), c_src, w.data_ptr(), w.scalar_type() == at::kFloat ? 1 : 0,
                              transposed ? 1 : 0, volume, c_dst, ptr<int32_t>(tp.first), nullptr, ptr<int32_t>(tp.second),
                              ptr<uint16_t>(out), n_tgt, want_stats ? ptr<float>(part) : nullptr,
                              want_stats ? ptr<float>(part) + tiles * c_dst : nullptr, st));
    }
    if (want_stats) bn_partials_put(out, part, tile_rows);
    return out;
  }
  // (a forward launch whose batch-norm statistics are wanted keeps the tile-plan kernel — which leaves them behind — on
  // small maps: ConvCfg::rowwise_min_rows_with_stats)
  const bool stats_wanted = bf16 && !transposed && target == "out" && g_conv_bn_stats_hint && conv_bn_stats_enabled();
  const ConvCfg &cfg0 = km.conv_cfg(target, n_tgt, c_src, c_dst, bf16);
  const ConvCfg &cfg = (cfg0.rowwise && stats_wanted && n_tgt < Policy::get().rowwise_min_rows_with_stats)
                           ? km.conv_cfg(target, n_tgt, c_src, c_dst, bf16, /*no_rowwise=*/true)
                           : cfg0;
  c10::DeviceGuard guard(dev);
  void *st = stream_of(dev);
  auto timed_name = transposed ? "conv_dgrad" : "conv_forward";
  const double flops = g_timing ? 2.0 * (double)km.n_pairs() * c_src * c_dst : 0.0;
  if (cfg.rowwise) {
    // exactly one pair per target row (K = 1 layers — the reference's input.F.mm(kernel) —, the fine side of a
    // kernel_size == stride map): rows stream through the MFMA registers, no plan (csrc/conv_rowwise.hip)
    check(kernel.scalar_type() == at::kFloat || kernel.scalar_type() == at::kBFloat16, "kernel must be float32 or bfloat16");
    Tensor packed;
    if (Policy::get().pack_cache) {
      packed = packed_weights(kernel, ME_PACK_BF16, transposed, c_src, c_dst, cfg.elems);
    } else {
      packed = at::empty({cfg.elems}, at::TensorOptions().dtype(at::kBFloat16).device(dev));
      me_ok(me_conv_pack_weights_bf16(kernel.data_ptr(), kernel.scalar_type() == at::kFloat ? 1 : 0, volume, c_src, c_dst,
                                      transposed ? 1 : 0, ptr<uint16_t>(packed), st));
    }
    const Tensor &src_rows = target == "out" ? km.in_pairs_buf : km.out_pairs_buf;
    const Tensor &tgt_rows = target == "out" ? km.out_pairs_buf : km.in_pairs_buf;
    ScopedTimer tm(timed_name, flops, st);
    me_ok(me_conv_rowwise_bf16(ptr<uint16_t>(src), src.size(0), c_src, ptr<uint16_t>(packed), km.volume, c_dst,
                               ptr<int32_t>(src_rows), ptr<int32_t>(tgt_rows), ptr<int64_t>(km.k_offsets_dev), n_tgt,
                               ptr<uint16_t>(out), n_tgt, st));
    return out;
  }
  if (cfg.halo) {
    // output-stationary launch on the LDS-staged source halo (csrc/conv_halo.hip): the same packed weights; the
    // batch-norm partials per halo tile
    check(kernel.scalar_type() == at::kFloat || kernel.scalar_type() == at::kBFloat16, "kernel must be float32 or bfloat16");
    const HaloPlan &h = *cfg.halo;
    Tensor packed;
    if (Policy::get().pack_cache) {
      packed = packed_weights(kernel, ME_PACK_BF16, transposed, c_src, c_dst, cfg.elems);
    } else {
      packed = at::empty({cfg.elems}, at::TensorOptions().dtype(at::kBFloat16).device(dev));
      me_ok(me_conv_pack_weights_bf16(kernel.data_ptr(), kernel.scalar_type() == at::kFloat ? 1 : 0, volume, c_src, c_dst,
                                      transposed ? 1 : 0, ptr<uint16_t>(packed), st));
    }
    const bool want_stats = !transposed && target == "out" && g_conv_bn_stats_hint && conv_bn_stats_enabled();
    Tensor part;
    int64_t tiles = 0;
    if (want_stats) {
      tiles = (n_tgt + h.tile_rows - 1) / h.tile_rows;
      part = at::empty({2, tiles, (int64_t)c_dst}, at::TensorOptions().dtype(at::kFloat).device(dev));
    }
    {
      ScopedTimer tm(timed_name, flops, st);
      me_ok(me_conv_halo_bf16(ptr<uint16_t>(src), src.size(0), c_src, ptr<uint16_t>(packed), km.volume, c_dst,
                              ptr<int32_t>(h.halo_cnt), ptr<int32_t>(h.halo_rows), ptr<uint16_t>(h.lidx),
                              ptr<uint32_t>(h.kmask), ptr<int32_t>(h.tbl), ptr<int32_t>(h.col_order),
                              ptr<int32_t>(h.out_order), ptr<uint16_t>(out), n_tgt, h.tile_rows, h.s_cap,
                              want_stats ? ptr<float>(part) : nullptr, want_stats ? ptr<float>(part) + tiles * c_dst : nullptr,
                              st));
    }
    if (want_stats) bn_partials_put(out, part, h.tile_rows);
    return out;
  }
  const Plan &p = *cfg.plan;
  if (bf16) {
    check(kernel.scalar_type() == at::kFloat || kernel.scalar_type() == at::kBFloat16, "kernel must be float32 or bfloat16");
    Tensor packed;
    if (Policy::get().pack_cache) {
      packed = packed_weights(kernel, ME_PACK_BF16, transposed, c_src, c_dst, cfg.elems);
    } else {
      packed = at::empty({cfg.elems}, at::TensorOptions().dtype(at::kBFloat16).device(dev));
      me_ok(me_conv_pack_weights_bf16(kernel.data_ptr(), kernel.scalar_type() == at::kFloat ? 1 : 0, volume, c_src, c_dst,
                                      transposed ? 1 : 0, ptr<uint16_t>(packed), st));
    }
    // One entry point for the bf16 forward / dgrad launch (me_conv_target_bf16_ex): batch fusion, the batch-norm
    // statistics of the output (forward launches only: target "out", weights not transposed — the input gradient feeds
    // no batch norm) and split-K (small maps: offset groups through an fp32 workspace) by argument.
    const bool want_stats = !transposed && target == "out" && g_conv_bn_stats_hint && conv_bn_stats_enabled() &&
                            me_conv_stats_supported_bf16(c_src, c_dst);
    Tensor part, ws;
    int64_t tiles = 0;
    if (want_stats) {
      // the tiles' (mean, M2) partials ride along with the output (bn_stats consumes them: bn_partials_take)
      tiles = (n_tgt + cfg.tile_rows - 1) / cfg.tile_rows;
      part = at::empty({2, tiles, (int64_t)c_dst}, at::TensorOptions().dtype(at::kFloat).device(dev));
    }
    if (cfg.split_k > 1)
      ws = at::empty({me_conv_splitk_workspace_bytes(n_tgt, cfg.tile_rows, c_dst, cfg.split_k) / 4},
                     at::TensorOptions().dtype(at::kFloat).device(dev));
    {
      ScopedTimer tm(timed_name, flops, st);
      me_ok(me_conv_target_bf16_ex(
          ptr<uint16_t>(src), src.size(0), c_src, ptr<uint16_t>(packed), km.volume, c_dst, ptr<int32_t>(p.plan_src),
          ptr<int32_t>(p.plan_dst), ptr<int32_t>(p.batch_desc), ptr<int32_t>(p.tile_bptr), ptr<int32_t>(cfg.order),
          ptr<uint16_t>(out), n_tgt, cfg.tile_rows, cfg.batch_groups, cfg.fuse ? 1 : 0, cfg.split_k, vptr(ws),
          want_stats ? ptr<float>(part) : nullptr, want_stats ? ptr<float>(part) + tiles * c_dst : nullptr, st));
    }
    if (want_stats) bn_partials_put(out, part, cfg.tile_rows);
    return out;
  }
  check(kernel.scalar_type() == at::kFloat, "float32 features need a float32 kernel");
  if (cfg.split) {
    Tensor packed;
    if (Policy::get().pack_cache) {
      packed = packed_weights(kernel, ME_PACK_F32X3, transposed, c_src, c_dst, cfg.elems);
    } else {
      packed = at::empty({cfg.elems}, at::TensorOptions().dtype(at::kBFloat16).device(dev));
      me_ok(me_conv_pack_weights_f32x3(ptr<float>(kernel), volume, c_src, c_dst, transposed ? 1 : 0, ptr<uint16_t>(packed),
                                       st));
    }
    ScopedTimer tm(timed_name, flops, st);
    me_ok(me_conv_target_f32x3(ptr<float>(src), src.size(0), c_src, ptr<uint16_t>(packed), km.volume, c_dst,
                               ptr<int32_t>(p.plan_src), ptr<int32_t>(p.plan_dst), ptr<int32_t>(p.batch_desc),
                               ptr<int32_t>(p.tile_bptr), ptr<int32_t>(cfg.order), ptr<float>(out), n_tgt, cfg.tile_rows,
                               cfg.batch_groups, st));
    return out;
  }
  Tensor packed = at::empty({cfg.elems}, at::TensorOptions().dtype(at::kFloat).device(dev));
  me_ok(me_conv_pack_weights_f32(ptr<float>(kernel), volume, c_src, c_dst, transposed ? 1 : 0, ptr<float>(packed), st));
  ScopedTimer tm(timed_name, flops, st);
  me_ok((cfg.fuse && Policy::get().f32_fuse ? me_conv_target_f32_fused : me_conv_target_f32)(ptr<float>(src), src.size(0), c_src, ptr<float>(packed), km.volume, c_dst,
                           ptr<int32_t>(p.plan_src), ptr<int32_t>(p.plan_dst), ptr<int32_t>(p.batch_desc),
                           ptr<int32_t>(p.tile_bptr), ptr<int32_t>(cfg.order), ptr<float>(out), n_tgt, cfg.tile_rows,
                           cfg.batch_groups, st));
  return out;
}

Tensor conv_forward_km(const Tensor &in_feat, const Tensor &kernel, KernelMap &km) {
  return conv_target(in_feat, kernel, km, "out", km.n_out, false);
}

std::pair<Tensor, Tensor> conv_backward_km(const Tensor &in_feat, Tensor grad_out, const Tensor &kernel, KernelMap &km,
                                           bool need_grad_in) {
  const c10::Device dev = in_feat.device();
  const int64_t volume = kernel.size(0);
  const int c_in = (int)kernel.size(1), c_out = (int)kernel.size(2);
  if (grad_out.scalar_type() != in_feat.scalar_type()) grad_out = grad_out.to(in_feat.scalar_type());
  const bool bf16 = in_feat.scalar_type() == at::kBFloat16;
  // dgrad: the same target-stationary kernel; the weights are packed transposed per offset
  Tensor grad_in;
  if (need_grad_in) grad_in = conv_target(grad_out, kernel, km, "in", km.n_in, true);
  if (in_feat.scalar_type() == at::kDouble) {   // csrc/f64.hip: sequential double sums over the pairs of each offset
    check(kernel.scalar_type() == at::kDouble, "float64 features need a float64 kernel");
    Tensor gw = at::empty(kernel.sizes(), kernel.options());
    Tensor x = in_feat.contiguous(), dy = grad_out.contiguous();
    c10::DeviceGuard guard(dev);
    me_ok(me_conv_wgrad_f64(ptr<double>(x), c_in, ptr<double>(dy), c_out, ptr<int32_t>(km.in_pairs_buf),
                            ptr<int32_t>(km.out_pairs_buf), ptr<int64_t>(km.k_offsets_dev), volume, ptr<double>(gw),
                            stream_of(dev)));
    return {grad_in, gw};
  }
  // wgrad: always accumulated and reduced in fp32; handed back in the kernel's dtype
  Tensor grad_w = kernel.scalar_type() == at::kFloat ? grad_destination(kernel, kernel.sizes()) : Tensor();
  if (!grad_w.defined()) grad_w = at::empty(kernel.sizes(), at::TensorOptions().dtype(at::kFloat).device(dev));
  const WgradCfg &w = km.wgrad_cfg(c_in, c_out, bf16);
  Tensor ws = workspace(w.ws_bytes, dev);
  {
    c10::DeviceGuard guard(dev);
    void *st = stream_of(dev);
    ScopedTimer tm("conv_wgrad", g_timing ? 2.0 * (double)km.n_pairs() * c_in * c_out : 0.0, st);
    if (bf16)
      me_ok(me_conv_wgrad_bf16(ptr<uint16_t>(in_feat), in_feat.size(0), c_in, ptr<uint16_t>(grad_out), grad_out.size(0),
                               c_out, ptr<int32_t>(km.in_pairs_buf), ptr<int32_t>(km.out_pairs_buf), w.koffs.data(),
                               ptr<int64_t>(km.k_offsets_dev), volume, ptr<float>(grad_w), vptr(ws), ws.numel(), st));
    else
      me_ok(me_conv_wgrad_f32(ptr<float>(in_feat), in_feat.size(0), c_in, ptr<float>(grad_out), grad_out.size(0), c_out,
                              ptr<int32_t>(km.in_pairs_buf), ptr<int32_t>(km.out_pairs_buf), w.koffs.data(),
                              ptr<int64_t>(km.k_offsets_dev), volume, ptr<float>(grad_w), vptr(ws), ws.numel(), st));
  }
  if (kernel.scalar_type() != at::kFloat) grad_w = grad_w.to(kernel.scalar_type());
  return {grad_in, grad_w};
}

std::shared_ptr<KernelMap> prepare_conv(const Tensor &in_feat, const Tensor &kernel, const ivec &kernel_size,
                                        const ivec &kernel_stride, const ivec &kernel_dilation, int region_type,
                                        bool expand_coordinates, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                                        CoordinateMapManager *manager, bool transpose) {
  check_feat("in_feat", in_feat);
  check_feat("kernel", kernel);
  check(in_feat.dim() == 2, "in_feat.dim() must be 2");
  check(kernel.dim() == 3, "kernel.dim() must be 3");
  check(in_feat.size(1) == kernel.size(1), "Input feature size and kernel size mismatch");
  const KeyT &ik = in_key->get();
  check(manager->exists(ik), "coordinate map not found");
  check(in_feat.size(0) == manager->get(ik)->n, "Invalid in_feat size");
  if (!out_key->key_set) {
    const ivec &ts = ik.first;
    if (!transpose) {
      if (expand_coordinates) {
        // src/convolution_cpu.cpp:79-103: every kernel offset around every input voxel that falls on the output grid
        ivec out_ts(ts.size());
        for (size_t i = 0; i < ts.size(); ++i) out_ts[i] = ts[i] * kernel_stride[i];
        auto r = manager->stride_region(ik, kernel_size, kernel_dilation, region_type, out_ts, true, false, &ts);
        out_key->set_key(r.first.first, r.first.second);
      } else {
        KeyT ok = manager->stride(ik, kernel_stride, "");
        out_key->set_key(ok.first, ok.second);
      }
    } else {
      // src/convolution_transpose_cpu.cpp:76-97: out tensor stride = in / stride
      ivec out_ts(ts.size());
      for (size_t i = 0; i < ts.size(); ++i) {
        check(kernel_stride[i] > 0 && ts[i] % kernel_stride[i] == 0, "Invalid up stride on tensor stride");
        out_ts[i] = ts[i] / kernel_stride[i];
      }
      auto r = manager->stride_region(ik, kernel_size, kernel_dilation, region_type, out_ts, expand_coordinates, true,
                                      nullptr);
      out_key->set_key(r.first.first, r.first.second);
    }
  }
  return manager->kernel_map(ik, out_key->get(), kernel_size, kernel_stride, kernel_dilation, region_type, transpose, false);
}

// ---- pooling / broadcast ----------------------------------------------------------------------------------------------------
enum { LOCAL_SUM = 0, LOCAL_AVG = 1, LOCAL_MAX = 2 };
static int global_mode(int pooling_mode) {   // -> 0 sum / 1 avg / 2 max (PoolingMode values 3..11)
  check(pooling_mode >= 3 && pooling_mode <= 11, "Invalid pooling mode");
  return (pooling_mode - 3) % 3;
}

static std::pair<Tensor, Tensor> pool_sum(const Tensor &src, const Tensor &tbl, int64_t n_tgt, int64_t volume,
                                          const Tensor &src_count, bool average, bool want_count) {
  const c10::Device dev = src.device();
  const int c = (int)src.size(1);
  Tensor out = at::empty({n_tgt, c}, src.options());
  Tensor cnt;
  if (want_count) cnt = at::empty({n_tgt > 0 ? n_tgt : 1}, at::TensorOptions().dtype(at::kFloat).device(dev)).narrow(0, 0, n_tgt);
  if (src_count.defined()) check(src_count.scalar_type() == at::kFloat, "num_nonzero must be float32");
  c10::DeviceGuard guard(dev);
  if (src.scalar_type() == at::kDouble)
    me_ok(me_pool_sum_f64(ptr<double>(src), c, ptr<int32_t>(tbl), n_tgt, volume, ptr<float>(src_count), average ? 1 : 0,
                          ptr<double>(out), ptr<float>(cnt), stream_of(dev)));
  else if (src.scalar_type() == at::kBFloat16)
    me_ok(me_pool_sum_bf16(ptr<uint16_t>(src), c, ptr<int32_t>(tbl), n_tgt, volume, ptr<float>(src_count), average ? 1 : 0,
                           ptr<uint16_t>(out), ptr<float>(cnt), stream_of(dev)));
  else
    me_ok(me_pool_sum_f32(ptr<float>(src), c, ptr<int32_t>(tbl), n_tgt, volume, ptr<float>(src_count), average ? 1 : 0,
                          ptr<float>(out), ptr<float>(cnt), stream_of(dev)));
  return {out, cnt};
}

static void prepare_pool(const Tensor &in_feat, const ivec &kernel_stride, CoordinateMapKey *in_key,
                         CoordinateMapKey *out_key, CoordinateMapManager *mgr, bool transpose) {
  check_feat("in_feat", in_feat);
  check(in_feat.dim() == 2, "in_feat.dim() must be 2");
  const KeyT &ik = in_key->get();
  check(mgr->exists(ik), "coordinate map not found");
  check(in_feat.size(0) == mgr->get(ik)->n, "Invalid in_feat size");
  if (out_key->key_set) return;
  if (!transpose) {
    KeyT ok = mgr->stride(ik, kernel_stride, "");
    out_key->set_key(ok.first, ok.second);
  } else {
    ivec out_ts(ik.first.size());
    for (size_t i = 0; i < out_ts.size(); ++i) {
      check(kernel_stride[i] > 0 && ik.first[i] % kernel_stride[i] == 0, "Invalid up stride on tensor stride");
      out_ts[i] = ik.first[i] / kernel_stride[i];
    }
    check(mgr->exists(KeyT(out_ts, "")), "pooling transpose needs an existing output map");
    out_key->set_key(out_ts, "");
  }
}

std::pair<Tensor, Tensor> local_pooling_forward(const Tensor &in_feat, const ivec &ks, const ivec &st, const ivec &dl,
                                                int region_type, int pooling_mode, CoordinateMapKey *in_key,
                                                CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  prepare_pool(in_feat, st, in_key, out_key, mgr, false);
  auto km = mgr->kernel_map(in_key->get(), out_key->get(), ks, st, dl, region_type, false, true);
  if (pooling_mode == LOCAL_MAX) {
    const c10::Device dev = in_feat.device();
    const int c = (int)in_feat.size(1);
    Tensor out = at::empty({km->n_out, c}, in_feat.options());
    Tensor mask = empty_i32({km->n_out, c}, dev);
    Tensor tbl = km->table("out");
    c10::DeviceGuard guard(dev);
    if (in_feat.scalar_type() == at::kDouble)
      me_ok(me_pool_max_f64(ptr<double>(in_feat), c, ptr<int32_t>(tbl), km->n_out, km->volume, ptr<double>(out),
                            ptr<int32_t>(mask), stream_of(dev)));
    else if (in_feat.scalar_type() == at::kBFloat16)
      me_ok(me_pool_max_bf16(ptr<uint16_t>(in_feat), c, ptr<int32_t>(tbl), km->n_out, km->volume, ptr<uint16_t>(out),
                             ptr<int32_t>(mask), stream_of(dev)));
    else
      me_ok(me_pool_max_f32(ptr<float>(in_feat), c, ptr<int32_t>(tbl), km->n_out, km->volume, ptr<float>(out),
                            ptr<int32_t>(mask), stream_of(dev)));
    return {out, mask};
  }
  check(pooling_mode == LOCAL_SUM || pooling_mode == LOCAL_AVG, "Invalid pooling mode");
  const bool avg = pooling_mode == LOCAL_AVG;
  auto r = pool_sum(in_feat, km->table("out"), km->n_out, km->volume, Tensor(), avg, avg);
  if (!r.second.defined()) r.second = at::empty({0}, at::TensorOptions().dtype(at::kFloat).device(in_feat.device()));
  return r;
}

Tensor local_pooling_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, const ivec &ks,
                              const ivec &st, const ivec &dl, int region_type, int pooling_mode,
                              CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  check_feat("in_feat", in_feat);
  grad_out = grad_out.contiguous();
  check_feat("grad_out_feat", grad_out);
  if (grad_out.scalar_type() != in_feat.scalar_type()) grad_out = grad_out.to(in_feat.scalar_type());
  auto km = mgr->kernel_map(in_key->get(), out_key->get(), ks, st, dl, region_type, false, true);
  check(grad_out.size(0) == km->n_out, "Invalid grad_out size");
  if (pooling_mode == LOCAL_MAX) {
    const c10::Device dev = in_feat.device();
    const int c = (int)in_feat.size(1);
    check(num_nonzero.scalar_type() == at::kInt, "the max-pooling mask must be int32");
    Tensor grad_in = at::empty({km->n_in, c}, in_feat.options());
    Tensor tbl = km->table("in");
    c10::DeviceGuard guard(dev);
    if (in_feat.scalar_type() == at::kDouble)
      me_ok(me_pool_max_backward_f64(ptr<double>(grad_out), c, ptr<int32_t>(tbl), km->n_in, km->volume,
                                     ptr<int32_t>(num_nonzero), ptr<double>(grad_in), stream_of(dev)));
    else if (in_feat.scalar_type() == at::kBFloat16)
      me_ok(me_pool_max_backward_bf16(ptr<uint16_t>(grad_out), c, ptr<int32_t>(tbl), km->n_in, km->volume,
                                      ptr<int32_t>(num_nonzero), ptr<uint16_t>(grad_in), stream_of(dev)));
    else
      me_ok(me_pool_max_backward_f32(ptr<float>(grad_out), c, ptr<int32_t>(tbl), km->n_in, km->volume,
                                     ptr<int32_t>(num_nonzero), ptr<float>(grad_in), stream_of(dev)));
    return grad_in;
  }
  const bool avg = pooling_mode == LOCAL_AVG;
  return pool_sum(grad_out, km->table("in"), km->n_in, km->volume, avg ? num_nonzero : Tensor(), false, false).first;
}

std::pair<Tensor, Tensor> local_pooling_transpose_forward(const Tensor &in_feat, const ivec &ks, const ivec &st,
                                                          const ivec &dl, int region_type, bool generate_new_coordinates,
                                                          int pooling_mode, CoordinateMapKey *in_key,
                                                          CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  (void)pooling_mode;
  check(!generate_new_coordinates, "generate_new_coordinates (stride_region) is not part of the hot path yet");
  prepare_pool(in_feat, st, in_key, out_key, mgr, true);
  auto km = mgr->kernel_map(in_key->get(), out_key->get(), ks, st, dl, region_type, true, true);
  return pool_sum(in_feat, km->table("out"), km->n_out, km->volume, Tensor(), false, true);
}

Tensor local_pooling_transpose_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, const ivec &ks,
                                        const ivec &st, const ivec &dl, int region_type, int pooling_mode,
                                        CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  (void)num_nonzero;
  (void)pooling_mode;
  grad_out = grad_out.contiguous();
  check_feat("grad_out_feat", grad_out);
  if (grad_out.scalar_type() != in_feat.scalar_type()) grad_out = grad_out.to(in_feat.scalar_type());
  auto km = mgr->kernel_map(in_key->get(), out_key->get(), ks, st, dl, region_type, true, true);
  return pool_sum(grad_out, km->table("in"), km->n_in, km->volume, Tensor(), false, false).first;
}

// mode 0 sum / 1 avg / 2 max over the rows of each origin row -> (out, argmax | undefined, count | undefined)
static std::tuple<Tensor, Tensor, Tensor> global_pool(const Tensor &src, const Tensor &src2, const Tensor &rows,
                                                      int64_t n_batch, int mode) {
  const c10::Device dev = src.device();
  const int64_t n = src.size(0);
  const int c = (int)src.size(1);
  if (src2.defined()) check(src2.scalar_type() == src.scalar_type(), "the second factor must have the dtype of the input features");
  const bool f64 = src.scalar_type() == at::kDouble;
  Tensor out = at::empty({n_batch, c}, at::TensorOptions().dtype(f64 ? at::kDouble : at::kFloat).device(dev));
  Tensor arg, cnt;
  if (mode == 2) arg = empty_i32({n_batch, c}, dev);
  else cnt = at::empty({n_batch}, at::TensorOptions().dtype(at::kFloat).device(dev));
  if (f64) {
    c10::DeviceGuard guard(dev);
    me_ok(me_global_pool_f64(ptr<double>(src), ptr<double>(src2), c, ptr<int32_t>(rows), n, (int)n_batch, mode,
                             ptr<double>(out), ptr<int32_t>(arg), ptr<float>(cnt), stream_of(dev)));
    return {out, arg, cnt};
  }
  Tensor ws = workspace(me_global_pool_workspace_bytes(n, (int)n_batch, c), dev);
  {
    c10::DeviceGuard guard(dev);
    if (src.scalar_type() == at::kBFloat16)
      me_ok(me_global_pool_bf16(ptr<uint16_t>(src), ptr<uint16_t>(src2), c, ptr<int32_t>(rows), n, (int)n_batch, mode,
                                ptr<float>(out), ptr<int32_t>(arg), ptr<float>(cnt), vptr(ws), ws.numel(), stream_of(dev)));
    else
      me_ok(me_global_pool_f32(ptr<float>(src), ptr<float>(src2), c, ptr<int32_t>(rows), n, (int)n_batch, mode,
                               ptr<float>(out), ptr<int32_t>(arg), ptr<float>(cnt), vptr(ws), ws.numel(), stream_of(dev)));
  }
  if (src.scalar_type() != at::kFloat) out = out.to(src.scalar_type());
  return {out, arg, cnt};
}

std::pair<Tensor, Tensor> global_pooling_forward(const Tensor &in_feat, int pooling_mode, CoordinateMapKey *in_key,
                                                 CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  check_feat("in_feat", in_feat);
  check(in_feat.dim() == 2, "Invalid in_feat.dim()");
  const KeyT &ik = in_key->get();
  check(mgr->exists(ik), "coordinate map not found");
  check(in_feat.size(0) == mgr->get(ik)->n, "Invalid in_feat size");
  const int m = global_mode(pooling_mode);
  if (!out_key->key_set) {
    KeyT ok = mgr->origin();
    out_key->set_key(ok.first, ok.second);
  }
  Tensor rows = mgr->origin_rows(ik);
  const int64_t n_batch = mgr->get(out_key->get())->n;
  auto r = global_pool(in_feat, Tensor(), rows, n_batch, m);
  return {std::get<0>(r), m == 2 ? std::get<1>(r) : std::get<2>(r)};
}

static Tensor broadcast(const Tensor &in_feat, const Tensor &glob, const Tensor &rows, int64_t n, int c, bool multiply) {
  const c10::Device dev = glob.device();
  if (in_feat.defined()) check(in_feat.scalar_type() == glob.scalar_type(), "in_feat must have the dtype of the input features");
  Tensor out = at::empty({n, c}, glob.options());
  c10::DeviceGuard guard(dev);
  if (glob.scalar_type() == at::kDouble)
    me_ok(me_broadcast_f64(ptr<double>(in_feat), ptr<double>(glob), ptr<int32_t>(rows), n, c, multiply ? 1 : 0,
                           ptr<double>(out), stream_of(dev)));
  else if (glob.scalar_type() == at::kBFloat16)
    me_ok(me_broadcast_bf16(ptr<uint16_t>(in_feat), ptr<uint16_t>(glob), ptr<int32_t>(rows), n, c, multiply ? 1 : 0,
                            ptr<uint16_t>(out), stream_of(dev)));
  else
    me_ok(me_broadcast_f32(ptr<float>(in_feat), ptr<float>(glob), ptr<int32_t>(rows), n, c, multiply ? 1 : 0,
                           ptr<float>(out), stream_of(dev)));
  return out;
}

Tensor global_pooling_backward(const Tensor &in_feat, Tensor grad_out, const Tensor &num_nonzero, int pooling_mode,
                               CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
  (void)out_key;
  check_feat("in_feat", in_feat);
  grad_out = grad_out.contiguous();
  check_feat("grad_out_feat", grad_out);
  if (grad_out.scalar_type() != in_feat.scalar_type()) grad_out = grad_out.to(in_feat.scalar_type());
  const int m = global_mode(pooling_mode);
  const int64_t n = in_feat.size(0);
  const int c = (int)in_feat.size(1);
  if (m == 2) {
    Tensor grad_in = at::zeros({n, c}, in_feat.options());
    Tensor flat = num_nonzero.reshape({-1});
    Tensor valid = flat.ge(0);
    grad_in.view({-1}).index_put_({flat.index({valid}).to(at::kLong)}, grad_out.reshape({-1}).index({valid}));
    return grad_in;
  }
  Tensor g = grad_out;
  if (m == 1)
    g = ((g.scalar_type() == at::kDouble ? g : g.to(at::kFloat)) / num_nonzero.clamp_min(1.0).unsqueeze(1))
            .to(in_feat.scalar_type()).contiguous();
  Tensor rows = mgr->origin_rows(in_key->get());
  return broadcast(Tensor(), g, rows, n, c, false);
}

Tensor broadcast_forward(const Tensor &in_feat, const Tensor &in_feat_glob, int broadcast_mode, CoordinateMapKey *in_key,
                         CoordinateMapKey *glob_key, CoordinateMapManager *mgr) {
  check_feat("in_feat", in_feat);
  check_feat("in_feat_glob", in_feat_glob);
  check(in_feat_glob.scalar_type() == in_feat.scalar_type(), "in_feat_glob must have the dtype of the input features");
  check(in_feat.size(1) == in_feat_glob.size(1), "feature sizes must match");
  check(in_feat.size(0) == mgr->get(in_key->get())->n, "Invalid in_feat size");
  check(in_feat_glob.size(0) == mgr->get(glob_key->get())->n, "Invalid in_feat_glob size");
  Tensor rows = mgr->origin_rows(in_key->get());
  return broadcast(in_feat, in_feat_glob, rows, in_feat.size(0), (int)in_feat.size(1), broadcast_mode == 1);
}

std::pair<Tensor, Tensor> broadcast_backward(const Tensor &in_feat, const Tensor &in_feat_glob, Tensor grad_out,
                                             int broadcast_mode, CoordinateMapKey *in_key, CoordinateMapKey *glob_key,
                                             CoordinateMapManager *mgr) {
  (void)glob_key;
  check_feat("in_feat", in_feat);
  check_feat("in_feat_glob", in_feat_glob);
  grad_out = grad_out.contiguous();
  check_feat("grad_out_feat", grad_out);
  if (grad_out.scalar_type() != in_feat.scalar_type()) grad_out = grad_out.to(in_feat.scalar_type());
  Tensor rows = mgr->origin_rows(in_key->get());
  const int64_t n_batch = in_feat_glob.size(0);
  if (broadcast_mode == 0) {
    Tensor grad_glob = std::get<0>(global_pool(grad_out, Tensor(), rows, n_batch, 0));
    return {grad_out.clone(), grad_glob};
  }
  Tensor grad_in = broadcast(grad_out, in_feat_glob, rows, in_feat.size(0), (int)in_feat.size(1), true);
  Tensor grad_glob = std::get<0>(global_pool(grad_out, in_feat, rows, n_batch, 0));
  return {grad_in, grad_glob};
}

// ---- pruning (src/pruning_cpu.cpp:40-150, src/pruning_gpu.cu) ------------------------------------------------------------------
Tensor pruning_forward(const Tensor &in_feat, const Tensor &keep, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                       CoordinateMapManager *mgr) {
  check_feat("in_feat", in_feat);
  check(keep.scalar_type() == at::kBool || keep.scalar_type() == at::kByte, "keep must be a boolean tensor");
  check(in_feat.dim() == 2 && keep.dim() == 1, "in_feat.dim() must be 2, keep.dim() 1");
  check(in_feat.size(0) == keep.size(0), "Input feature size and keep size mismatch");
  const KeyT &ik = in_key->get();
  check(mgr->exists(ik), "coordinate map not found");
  check(in_feat.size(0) == mgr->get(ik)->n, "Invalid in_feat size");
  if (!out_key->key_set) {
    KeyT ok = mgr->prune(ik, keep);
    out_key->set_key(ok.first, ok.second);
  }
  Tensor rows = mgr->pruning_rows(ik, out_key->get());
  return in_feat.index_select(0, rows.to(at::kLong));
}

Tensor pruning_backward(const Tensor &grad_out, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                        CoordinateMapManager *mgr) {
  check_feat("grad_out_feat", grad_out);
  Tensor rows = mgr->pruning_rows(in_key->get(), out_key->get());
  Tensor grad_in = at::zeros({mgr->get(in_key->get())->n, grad_out.size(1)}, grad_out.options());
  grad_in.index_copy_(0, rows.to(at::kLong), grad_out);
  return grad_in;
}

// ---- batch normalisation over feature rows (csrc/norm.hip) ----------------------------------------------------------------------
static void bn_check(const Tensor &x) {
  check(x.is_cuda() && x.is_contiguous() && x.dim() == 2, "batch norm input must be a contiguous GPU matrix");
  check(x.scalar_type() == at::kFloat || x.scalar_type() == at::kBFloat16, "batch norm input must be float32 or bfloat16");
  check(x.size(0) > 0, "batch norm needs at least one row");
}

// The kernels read and write the running statistics as float32 [c] and the batch counter as int64 (norm.hip).  Buffers
// of another dtype / layout (after `model.bfloat16()` the running statistics are bf16: c floats written into a 2c-byte
// buffer would corrupt its neighbours) go through float32 temporaries that are copied back; a counter that is not an
// int64 GPU scalar is incremented by torch.
static bool bn_stat_ok(const Tensor &t, int c) {
  return !t.defined() || (t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat && t.numel() == c);
}

static void bn_check_vec(const char *name, const Tensor &t, int c) {
  check(!t.defined() || (t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat && t.numel() == c),
        std::string(name) + " must be a contiguous float32 GPU vector of one value per channel");
}

std::pair<Tensor, Tensor> bn_stats(const Tensor &x, double eps, double momentum, const Tensor &running_mean_,
                                   const Tensor &running_var_, const Tensor &num_batches_tracked_) {
  bn_check(x);
  const c10::Device dev = x.device();
  const int64_t n = x.size(0);
  const int c = (int)x.size(1);
  check(running_mean_.defined() == running_var_.defined(), "running_mean and running_var: both or none");
  check(!running_mean_.defined() || (running_mean_.numel() == c && running_var_.numel() == c),
        "running statistics must hold one value per channel");
  const bool direct = bn_stat_ok(running_mean_, c) && bn_stat_ok(running_var_, c);
  Tensor running_mean = running_mean_, running_var = running_var_;
  if (!direct) {
    running_mean = running_mean_.to(dev, at::kFloat).contiguous().clone();
    running_var = running_var_.to(dev, at::kFloat).contiguous().clone();
  }
  Tensor num_batches_tracked = num_batches_tracked_;
  if (num_batches_tracked.defined() &&
      !(num_batches_tracked.is_cuda() && num_batches_tracked.scalar_type() == at::kLong && num_batches_tracked.numel() == 1)) {
    num_batches_tracked_.add_(1);
    num_batches_tracked = Tensor();
  }
  Tensor mean = at::empty({c}, at::TensorOptions().dtype(at::kFloat).device(dev));
  Tensor rstd = at::empty({c}, at::TensorOptions().dtype(at::kFloat).device(dev));
  c10::DeviceGuard guard(dev);
  Tensor part;
  int tile_rows = 0;
  if (bn_partials_take(x, part, tile_rows)) {   // the producing convolution already summed its tiles
    const int64_t tiles = part.size(1);
    me_ok(me_bn_stats_from_tiles(ptr<float>(part), ptr<float>(part) + tiles * c, n, c, tile_rows, (float)eps,
                                 (float)momentum, ptr<float>(mean), ptr<float>(rstd), ptr<float>(running_mean),
                                 ptr<float>(running_var), ptr<int64_t>(num_batches_tracked), stream_of(dev)));
  } else {
    Tensor ws = workspace(me_bn_workspace_bytes(n, c), dev);
    me_ok(me_bn_stats(x.data_ptr(), x.scalar_type() == at::kBFloat16 ? 1 : 0, n, c, (float)eps, (float)momentum,
                      ptr<float>(mean), ptr<float>(rstd), ptr<float>(running_mean), ptr<float>(running_var),
                      ptr<int64_t>(num_batches_tracked), vptr(ws), ws.numel(), stream_of(dev)));
  }
  if (!direct) {
    running_mean_.copy_(running_mean);
    running_var_.copy_(running_var);
  }
  return {mean, rstd};
}

Tensor bn_apply(const Tensor &x, const Tensor &mean, const Tensor &rstd, const Tensor &gamma, const Tensor &beta,
                bool relu, const Tensor &skip) {
  bn_check(x);
  const c10::Device dev = x.device();
  const int cc = (int)x.size(1);
  bn_check_vec("mean", mean, cc);
  bn_check_vec("rstd", rstd, cc);
  bn_check_vec("gamma", gamma, cc);
  bn_check_vec("beta", beta, cc);
  check(mean.defined() && rstd.defined(), "batch norm needs mean and rstd");
  Tensor y = at::empty_like(x);
  c10::DeviceGuard guard(dev);
  const int bf = x.scalar_type() == at::kBFloat16 ? 1 : 0;
  if (skip.defined()) {
    check(skip.sizes() == x.sizes() && skip.scalar_type() == x.scalar_type() && skip.is_contiguous(),
          "residual branch must match x");
    me_ok(me_bn_apply_residual(x.data_ptr(), skip.data_ptr(), bf, x.size(0), (int)x.size(1), ptr<float>(mean),
                               ptr<float>(rstd), ptr<float>(gamma), ptr<float>(beta), relu ? 1 : 0, y.data_ptr(),
                               stream_of(dev)));
  } else {
    me_ok(me_bn_apply(x.data_ptr(), bf, x.size(0), (int)x.size(1), ptr<float>(mean), ptr<float>(rstd), ptr<float>(gamma),
                      ptr<float>(beta), relu ? 1 : 0, y.data_ptr(), stream_of(dev)));
  }
  return y;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> bn_backward(const Tensor &x, Tensor dy, const Tensor &yout, const Tensor &mean,
                                                       const Tensor &rstd, const Tensor &gamma, const Tensor &beta,
                                                       bool relu, bool residual, bool need_dskip) {
  bn_check(x);
  const c10::Device dev = x.device();
  const int64_t n = x.size(0);
  const int c = (int)x.size(1);
  bn_check_vec("mean", mean, c);
  bn_check_vec("rstd", rstd, c);
  bn_check_vec("gamma", gamma, c);
  bn_check_vec("beta", beta, c);
  check(mean.defined() && rstd.defined(), "batch norm backward needs mean and rstd");
  if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
  dy = dy.contiguous();
  Tensor dx = at::empty_like(x);
  Tensor gg = grad_destination(gamma, {(int64_t)c}), gb = grad_destination(beta, {(int64_t)c});
  if (!gg.defined()) gg = at::empty({c}, at::TensorOptions().dtype(at::kFloat).device(dev));
  if (!gb.defined()) gb = at::empty({c}, at::TensorOptions().dtype(at::kFloat).device(dev));
  Tensor ws = workspace(me_bn_workspace_bytes(n, c), dev);
  Tensor dskip;
  const int bf = x.scalar_type() == at::kBFloat16 ? 1 : 0;
  c10::DeviceGuard guard(dev);
  if (residual) {
    if (need_dskip && relu) dskip = at::empty_like(x);
    me_ok(me_bn_backward_residual(x.data_ptr(), dy.data_ptr(), vptr(yout), bf, n, c, ptr<float>(mean), ptr<float>(rstd),
                                  ptr<float>(gamma), ptr<float>(beta), relu ? 1 : 0, dx.data_ptr(), vptr(dskip),
                                  ptr<float>(gg), ptr<float>(gb), vptr(ws), ws.numel(), stream_of(dev)));
    if (need_dskip && !relu) dskip = dy;
  } else {
    me_ok(me_bn_backward(x.data_ptr(), dy.data_ptr(), bf, n, c, ptr<float>(mean), ptr<float>(rstd), ptr<float>(gamma),
                         ptr<float>(beta), relu ? 1 : 0, dx.data_ptr(), ptr<float>(gg), ptr<float>(gb), vptr(ws),
                         ws.numel(), stream_of(dev)));
  }
  return {dx, dskip, gg, gb};
}

}  // namespace meh
