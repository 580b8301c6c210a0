// pybind11 module of the native host layer: the names, argument order and argument meaning of the reference's
// `MinkowskiEngineBackend._C` (pybind/extern.hpp:515-838) for the hot path, plus torch::autograd functions so that a
// training step's backward pass runs without entering Python (the reference's autograd Functions are Python,
// MinkowskiConvolution.py:42-121; this is where its per-layer host time goes).
#include <torch/csrc/autograd/custom_function.h>

#include <sstream>

#include "host.hpp"

namespace py = pybind11;
using namespace meh;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

int to_int(const py::object &o) { return py::cast<int>(py::int_(o)); }   // pybind enums, Python IntEnums, ints

KeyT keyt(const CoordinateMapKey *k) { return k->get(); }

CoordinateMapKey *new_key(const KeyT &k) { return new CoordinateMapKey(k.first, k.second); }

// ---- autograd: convolution ----------------------------------------------------------------------------------------------------
struct KmHolder : torch::CustomClassHolder {
  std::shared_ptr<KernelMap> km;
  explicit KmHolder(std::shared_ptr<KernelMap> k) : km(std::move(k)) {}
};

struct ConvFn : public torch::autograd::Function<ConvFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &in_feat_, const Tensor &kernel,
                        const std::shared_ptr<KernelMap> &km) {
    Tensor in_feat = in_feat_.contiguous();
    ctx->save_for_backward({in_feat, kernel});
    ctx->saved_data["km"] = c10::IValue::make_capsule(c10::make_intrusive<KmHolder>(km));
    return conv_forward_km(in_feat, kernel, *km);
  }
  static variable_list backward(AutogradContext *ctx, variable_list grad_outputs) {
    auto saved = ctx->get_saved_variables();
    auto holder = c10::static_intrusive_pointer_cast<KmHolder>(ctx->saved_data["km"].toCapsule());
    Tensor gy = grad_outputs[0].contiguous();
    auto r = conv_backward_km(saved[0], gy, saved[1], *holder->km, ctx->needs_input_grad(0));
    return {r.first, r.second, Tensor()};
  }
};

Tensor conv_autograd(const Tensor &in_feat, const Tensor &kernel, const ivec &kernel_size, const ivec &kernel_stride,
                     const ivec &kernel_dilation, const py::object &region_type, bool expand_coordinates,
                     CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *manager, bool transpose) {
  auto km = prepare_conv(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, to_int(region_type),
                         expand_coordinates, in_key, out_key, manager, transpose);
  return ConvFn::apply(in_feat, kernel, km);
}

// ---- autograd: training-mode batch norm (+ residual add + ReLU) -------------------------------------------------------------------
struct BnFn : public torch::autograd::Function<BnFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const c10::optional<Tensor> &skip_o,
                        const c10::optional<Tensor> &weight_o, const c10::optional<Tensor> &bias_o,
                        const c10::optional<Tensor> &rm_o, const c10::optional<Tensor> &rv_o, double momentum, double eps,
                        bool relu, const c10::optional<Tensor> &nbt_o) {
    const Tensor skip_ = skip_o.value_or(Tensor()), weight = weight_o.value_or(Tensor()), bias = bias_o.value_or(Tensor());
    const Tensor running_mean = rm_o.value_or(Tensor()), running_var = rv_o.value_or(Tensor());
    const Tensor num_batches_tracked = nbt_o.value_or(Tensor());
    Tensor x = x_.contiguous();
    Tensor skip = skip_.defined() ? skip_.contiguous() : Tensor();
    Tensor w32 = weight.defined() ? weight.to(at::kFloat) : Tensor();
    Tensor b32 = bias.defined() ? bias.to(at::kFloat) : Tensor();
    auto st = bn_stats(x, eps, momentum, running_mean, running_var, num_batches_tracked);
    Tensor y = bn_apply(x, st.first, st.second, w32, b32, relu, skip);
    const bool residual = skip.defined();
    ctx->save_for_backward({x, (residual && relu) ? y : Tensor(), st.first, st.second, w32, b32});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["residual"] = residual;
    ctx->saved_data["wdtype"] = weight.defined() ? (int64_t)weight.scalar_type() : (int64_t)-1;
    ctx->saved_data["has_bias"] = bias.defined();
    return y;
  }
  static variable_list backward(AutogradContext *ctx, variable_list grad_outputs) {
    auto s = ctx->get_saved_variables();
    const bool relu = ctx->saved_data["relu"].toBool(), residual = ctx->saved_data["residual"].toBool();
    auto r = bn_backward(s[0], grad_outputs[0], s[1], s[2], s[3], s[4], s[5], relu, residual,
                         residual && ctx->needs_input_grad(1));
    const int64_t wd = ctx->saved_data["wdtype"].toInt();
    Tensor gw = wd >= 0 ? std::get<2>(r).to((at::ScalarType)wd) : Tensor();
    Tensor gb = ctx->saved_data["has_bias"].toBool() ? std::get<3>(r).to((at::ScalarType)wd) : Tensor();
    return {std::get<0>(r), std::get<1>(r), gw, gb, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor opt_tensor(const py::object &o) { return o.is_none() ? Tensor() : py::cast<Tensor>(o); }
c10::optional<Tensor> opt_opt(const py::object &o) {
  return o.is_none() ? c10::optional<Tensor>() : c10::optional<Tensor>(py::cast<Tensor>(o));
}

Tensor batch_norm_train(const Tensor &x, const py::object &skip, const py::object &weight, const py::object &bias,
                        const py::object &running_mean, const py::object &running_var, double momentum, double eps,
                        bool relu, const py::object &num_batches_tracked) {
  return BnFn::apply(x, opt_opt(skip), opt_opt(weight), opt_opt(bias), opt_opt(running_mean), opt_opt(running_var),
                     momentum, eps, relu, opt_opt(num_batches_tracked));
}

py::object opt_out(const Tensor &t) { return t.defined() ? py::cast(t) : py::none(); }

// enums of pybind/extern.hpp:669-741
enum GPUMemoryAllocatorType { PYTORCH = 0, CUDA = 1 };
enum CUDAKernelMapMode { KM_MEMORY_EFFICIENT = 0, KM_SPEED_OPTIMIZED = 1 };
enum MinkowskiAlgorithm { DEFAULT = 0, MEMORY_EFFICIENT = 1, SPEED_OPTIMIZED = 2 };
enum CoordinateMapType { MAP_CPU = 0, MAP_CUDA = 1 };
enum RegionType { HYPER_CUBE = 0, HYPER_CROSS = 1, CUSTOM = 2 };
enum PoolingMode {
    LOCAL_SUM_POOLING = 0, LOCAL_AVG_POOLING, LOCAL_MAX_POOLING, GLOBAL_SUM_POOLING_DEFAULT, GLOBAL_AVG_POOLING_DEFAULT,
    GLOBAL_MAX_POOLING_DEFAULT, GLOBAL_SUM_POOLING_KERNEL, GLOBAL_AVG_POOLING_KERNEL, GLOBAL_MAX_POOLING_KERNEL,
    GLOBAL_SUM_POOLING_PYTORCH_INDEX, GLOBAL_AVG_POOLING_PYTORCH_INDEX, GLOBAL_MAX_POOLING_PYTORCH_INDEX
  };
enum BroadcastMode { ELEMENTWISE_ADDITON = 0, ELEMENTWISE_MULTIPLICATION = 1 };
enum ConvolutionMode { CONV_DEFAULT = 0, DIRECT_GEMM = 1, COPY_GEMM = 2 };

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "MI355X-native operator module (MinkowskiEngineBackend._C for the sparse-convolution hot path)";
  m.attr("_host") = "native";

  // ---- enums (pybind/extern.hpp:669-741) ----
  py::enum_<GPUMemoryAllocatorType>(m, "GPUMemoryAllocatorType").value("PYTORCH", PYTORCH).value("CUDA", CUDA);
  py::enum_<CUDAKernelMapMode>(m, "CUDAKernelMapMode")
      .value("MEMORY_EFFICIENT", KM_MEMORY_EFFICIENT)
      .value("SPEED_OPTIMIZED", KM_SPEED_OPTIMIZED);
  py::enum_<MinkowskiAlgorithm>(m, "MinkowskiAlgorithm")
      .value("DEFAULT", DEFAULT)
      .value("MEMORY_EFFICIENT", MEMORY_EFFICIENT)
      .value("SPEED_OPTIMIZED", SPEED_OPTIMIZED);
  py::enum_<CoordinateMapType>(m, "CoordinateMapType").value("CPU", MAP_CPU).value("CUDA", MAP_CUDA);
  py::enum_<RegionType>(m, "RegionType").value("HYPER_CUBE", HYPER_CUBE).value("HYPER_CROSS", HYPER_CROSS).value("CUSTOM", CUSTOM);
  py::enum_<PoolingMode>(m, "PoolingMode")
      .value("LOCAL_SUM_POOLING", LOCAL_SUM_POOLING)
      .value("LOCAL_AVG_POOLING", LOCAL_AVG_POOLING)
      .value("LOCAL_MAX_POOLING", LOCAL_MAX_POOLING)
      .value("GLOBAL_SUM_POOLING_DEFAULT", GLOBAL_SUM_POOLING_DEFAULT)
      .value("GLOBAL_AVG_POOLING_DEFAULT", GLOBAL_AVG_POOLING_DEFAULT)
      .value("GLOBAL_MAX_POOLING_DEFAULT", GLOBAL_MAX_POOLING_DEFAULT)
      .value("GLOBAL_SUM_POOLING_KERNEL", GLOBAL_SUM_POOLING_KERNEL)
      .value("GLOBAL_AVG_POOLING_KERNEL", GLOBAL_AVG_POOLING_KERNEL)
      .value("GLOBAL_MAX_POOLING_KERNEL", GLOBAL_MAX_POOLING_KERNEL)
      .value("GLOBAL_SUM_POOLING_PYTORCH_INDEX", GLOBAL_SUM_POOLING_PYTORCH_INDEX)
      .value("GLOBAL_AVG_POOLING_PYTORCH_INDEX", GLOBAL_AVG_POOLING_PYTORCH_INDEX)
      .value("GLOBAL_MAX_POOLING_PYTORCH_INDEX", GLOBAL_MAX_POOLING_PYTORCH_INDEX);
  py::enum_<BroadcastMode>(m, "BroadcastMode")
      .value("ELEMENTWISE_ADDITON", ELEMENTWISE_ADDITON)
      .value("ELEMENTWISE_MULTIPLICATION", ELEMENTWISE_MULTIPLICATION);
  py::enum_<ConvolutionMode>(m, "ConvolutionMode")
      .value("DEFAULT", CONV_DEFAULT)
      .value("DIRECT_GEMM", DIRECT_GEMM)
      .value("COPY_GEMM", COPY_GEMM);

  m.def("conv_bn_stats_hint", &conv_bn_stats_hint, py::arg("flag"));
  m.def("set_conv_bn_stats", &set_conv_bn_stats, py::arg("enabled"));
  m.def("invalidate_packed_weights", &invalidate_packed_weights);
  m.def("invalidate_packed_weights_for", &invalidate_packed_weights_for, py::arg("sorted_ptrs"));
  m.def("invalidate_packed_weights_in", &invalidate_packed_weights_in, py::arg("begins"), py::arg("ends"));
  m.def("set_grad_destination", [](const Tensor &param, const py::object &dest) { set_grad_destination(param, opt_tensor(dest)); });
  m.def("clear_grad_destinations", &clear_grad_destinations);
  m.def("debug_fail_next_plan_batch", &debug_fail_next_plan_batch);
  m.def("arm_grad_destinations", &arm_grad_destinations);
  m.def("arm_grad_destinations_for", &arm_grad_destinations_for);
  m.def("drop_grad_destinations", &drop_grad_destinations);
  m.def("set_policy", &Policy::set, "integer policies of the native host by name (tests, tuning scripts)");
  m.def("timing_enable", &timing_enable);
  m.def("timing_records", &timing_records, py::arg("clear") = true);
  m.def("is_cuda_available", [] { return true; });
  m.def("cuda_version", [] { return -1; });
  m.def("cudart_version", [] { return -1; });
  m.def("get_gpu_memory_info", [] {
    py::object mem = py::module_::import("torch").attr("cuda").attr("mem_get_info")();
    return mem;
  });

  // ---- CoordinateMapKey (pybind/extern.hpp:744-764) ----
  py::class_<CoordinateMapKey>(m, "CoordinateMapKey")
      .def(py::init<int>())
      .def(py::init<ivec, std::string>(), py::arg("tensor_stride"), py::arg("string_id") = "")
      .def("get_coordinate_size", [](const CoordinateMapKey &k) { return k.coordinate_size; })
      .def("is_key_set", [](const CoordinateMapKey &k) { return k.key_set; })
      .def("set_key", [](CoordinateMapKey &k, const ivec &ts, const std::string &sid) { k.set_key(ts, sid); })
      .def("set_key", [](CoordinateMapKey &k, const std::pair<ivec, std::string> &p) { k.set_key(p.first, p.second); })
      .def("get_key", [](const CoordinateMapKey &k) { return std::make_pair(k.get().first, k.get().second); })
      .def("get_tensor_stride", [](const CoordinateMapKey &k) { return k.get().first; })
      .def("__eq__", [](const CoordinateMapKey &a, const py::object &b) {
        if (!py::isinstance<CoordinateMapKey>(b)) return false;
        return a.equals(py::cast<const CoordinateMapKey &>(b));
      })
      .def("__hash__", [](const CoordinateMapKey &k) {
        const KeyT &kk = k.get();
        size_t h = std::hash<std::string>()(kk.second);
        for (int v : kk.first) h = h * 1000003u + (size_t)v;
        return (int64_t)(h & 0x7fffffffffffffffull);
      })
      .def("__repr__", &CoordinateMapKey::repr);

  // ---- CoordinateMapManager (pybind/extern.hpp:767-806) ----
  auto mgr = py::class_<CoordinateMapManager>(m, "CoordinateMapManagerGPU_c10");
  mgr.def(py::init([](const py::object &algorithm, int num_threads) {
            return new CoordinateMapManager(algorithm.is_none() ? 0 : to_int(algorithm), num_threads);
          }),
          py::arg("algorithm") = py::none(), py::arg("num_threads") = 0)
      .def("exists", [](CoordinateMapManager &s, const CoordinateMapKey *k) { return k->key_set && s.exists(k->key); })
      .def("insert_and_map",
           [](CoordinateMapManager &s, const Tensor &coords, const ivec &ts, const std::string &sid) {
             auto r = [&] {
               py::gil_scoped_release nogil;   // (see prefetch)
               return s.insert_and_map(coords, ts, sid);
             }();
             return py::make_tuple(py::cast(new_key(std::get<0>(r)), py::return_value_policy::take_ownership),
                                   py::make_tuple(std::get<1>(r), std::get<2>(r)));
           },
           py::arg("coordinates"), py::arg("tensor_stride"), py::arg("string_id") = "")
      .def("stride",
           [](CoordinateMapManager &s, const CoordinateMapKey *k, const ivec &stride, const std::string &sid) {
             return new_key(s.stride(keyt(k), stride, sid));
           },
           py::arg("in_key"), py::arg("kernel_stride"), py::arg("string_id") = "", py::return_value_policy::take_ownership)
      .def("stride_map",
           [](CoordinateMapManager &s, const CoordinateMapKey *a, const CoordinateMapKey *b) {
             auto r = s.stride_map(keyt(a), keyt(b));
             return py::make_tuple(r.first, r.second);
           })
      .def("origin", [](CoordinateMapManager &s) { return new_key(s.origin()); }, py::return_value_policy::take_ownership)
      .def("origin_map_size", [](CoordinateMapManager &s) { return s.get(s.origin())->n; })
      .def("origin_map",
           [](CoordinateMapManager &s, const CoordinateMapKey *k) {
             Tensor rows = s.origin_rows(keyt(k));
             py::dict d;
             d[py::int_(0)] = at::stack({at::arange(rows.numel(), rows.options()), rows});
             return d;
           })
      .def("_origin_rows", [](CoordinateMapManager &s, const CoordinateMapKey *k) { return s.origin_rows(keyt(k)); })
      .def("get_coordinates", [](CoordinateMapManager &s, const CoordinateMapKey *k) { return s.get(keyt(k))->coords; })
      .def("size", [](CoordinateMapManager &s, const CoordinateMapKey *k) { return s.get(keyt(k))->n; })
      .def("get_random_string_id",
           [](CoordinateMapManager &s, const ivec &ts, const std::string &sid) {
             KeyT k = s.random_string_id(ts, sid);
             return std::make_pair(k.first, k.second);
           })
      .def("get_coordinate_map_keys",
           [](CoordinateMapManager &s, const ivec &ts) {
             py::list out;
             for (const KeyT &k : s.map_order)
               if (k.first == ts) out.append(py::cast(new_key(k), py::return_value_policy::take_ownership));
             return out;
           })
      .def("union_map",
           [](CoordinateMapManager &s, const std::vector<CoordinateMapKey *> &in_keys, CoordinateMapKey *out_key) {
             std::vector<KeyT> ks;
             for (auto *k : in_keys) ks.push_back(keyt(k));
             return s.union_map(ks, out_key);
           })
      .def("prune", [](CoordinateMapManager &s, const CoordinateMapKey *k, const Tensor &keep) {
             return new_key(s.prune(keyt(k), keep));
           }, py::return_value_policy::take_ownership)
      .def("kernel_map",
           [](CoordinateMapManager &s, const CoordinateMapKey *ik, const CoordinateMapKey *ok, const ivec &ks,
              const ivec &st, const ivec &dl, const py::object &region_type, const py::object & /*offset*/,
              bool is_transpose, bool is_pool) {
             return s.kernel_map(keyt(ik), keyt(ok), ks, st, dl, to_int(region_type), is_transpose, is_pool)->to_dict();
           })
      .def("kernel_map_pairs",    // (not in the reference) number of pairs of a cached / built kernel map
           [](CoordinateMapManager &s, const CoordinateMapKey *ik, const CoordinateMapKey *ok, const ivec &ks,
              const ivec &st, const ivec &dl, const py::object &region_type, bool is_transpose, bool is_pool) {
             return s.kernel_map(keyt(ik), keyt(ok), ks, st, dl, to_int(region_type), is_transpose, is_pool)->n_pairs();
           })
      .def("_conv_cfg",           // (bench / tests) builds the tile plan of a launch side -> (tile_rows, batch_groups)
           [](CoordinateMapManager &s, const CoordinateMapKey *ik, const CoordinateMapKey *ok, const ivec &ks,
              const ivec &st, const ivec &dl, const py::object &region_type, bool is_transpose, const std::string &target,
              int c_src, int c_dst, bool bf16) {
             auto km = s.kernel_map(keyt(ik), keyt(ok), ks, st, dl, to_int(region_type), is_transpose, false);
             const ConvCfg &c = km->conv_cfg(target, target == "out" ? km->n_out : km->n_in, c_src, c_dst, bf16);
             return py::make_tuple(c.tile_rows, c.batch_groups);
           })
      .def("recipe", [](CoordinateMapManager &s) { return s.recipe_log->snapshot(); })
      // (no GIL while the maps and plans of a scene are built: a loader thread can prepare the next scene — on its own
      // stream — while the training thread launches the current step)
      .def("prefetch", [](CoordinateMapManager &s, const std::vector<std::string> &r) { return s.prefetch(r); },
           py::call_guard<py::gil_scoped_release>())
      .def("record_stream",
           [](CoordinateMapManager &s, const py::object &stream) {
             // (torch.cuda.Stream -> c10::Stream, then no Python object and no GIL: a loader thread hands a scene's
             // ~300 map tensors to the training stream without stalling the training thread)
             const c10::Stream st = c10::Stream::unpack3(stream.attr("stream_id").cast<int64_t>(),
                                                         (c10::DeviceIndex)stream.attr("device_index").cast<int64_t>(),
                                                         (c10::DeviceType)stream.attr("device_type").cast<int64_t>());
             py::gil_scoped_release nogil;
             for (const Tensor &t : s.device_tensors()) t.record_stream(st);
           })
      .def("print_coordinate_map",   // manager_type::to_string(key): pybind/extern.hpp:777-779
           [](CoordinateMapManager &s, const CoordinateMapKey *key) {
             const KeyT &k = keyt(key);
             auto m = s.get(k);
             std::ostringstream o;
             o << "[";
             for (size_t i = 0; i < k.first.size(); ++i) o << (i ? ", " : "") << k.first[i];
             o << "]" << (k.second.empty() ? "" : ":" + k.second) << " : CoordinateMapGPU:" << m->n << "x" << m->coords.size(1);
             return o.str();
           })
      .def("__repr__", &CoordinateMapManager::repr);
  m.attr("CoordinateMapManagerGPU_default") = m.attr("CoordinateMapManagerGPU_c10");

  // ---- operators with the reference's signatures (pybind/extern.hpp:53-392, 515-646) ----
  m.def("ConvolutionForwardGPU",
        [](const Tensor &in_feat, const Tensor &kernel, const ivec &ks, const ivec &st, const ivec &dl,
           const py::object &region_type, const py::object & /*offset*/, bool expand_coordinates,
           const py::object & /*convolution_mode*/, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
           CoordinateMapManager *mgr) {
          auto km = prepare_conv(in_feat, kernel, ks, st, dl, to_int(region_type), expand_coordinates, in_key, out_key, mgr,
                                 false);
          return conv_forward_km(in_feat, kernel, *km);
        });
  m.def("ConvolutionTransposeForwardGPU",
        [](const Tensor &in_feat, const Tensor &kernel, const ivec &ks, const ivec &st, const ivec &dl,
           const py::object &region_type, const py::object & /*offset*/, bool expand_coordinates,
           const py::object & /*convolution_mode*/, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
           CoordinateMapManager *mgr) {
          auto km = prepare_conv(in_feat, kernel, ks, st, dl, to_int(region_type), expand_coordinates, in_key, out_key, mgr,
                                 true);
          return conv_forward_km(in_feat, kernel, *km);
        });
  auto conv_bwd = [](bool transpose) {
    return [transpose](const Tensor &in_feat, Tensor grad_out, const Tensor &kernel, const ivec &ks, const ivec &st,
                       const ivec &dl, const py::object &region_type, const py::object & /*offset*/,
                       const py::object & /*convolution_mode*/, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
                       CoordinateMapManager *mgr, bool need_grad_in) {
      check(in_feat.size(1) == kernel.size(1), "Input feature size and kernel size mismatch");
      check(grad_out.size(1) == kernel.size(2), "Output feature size and kernel size mismatch");
      grad_out = grad_out.contiguous();
      auto km = mgr->kernel_map(in_key->get(), out_key->get(), ks, st, dl, to_int(region_type), transpose, false);
      check(grad_out.size(0) == km->n_out, "Invalid grad_out size");
      auto r = conv_backward_km(in_feat.contiguous(), grad_out, kernel, *km, need_grad_in);
      return py::make_tuple(opt_out(r.first), r.second);
    };
  };
  m.def("ConvolutionBackwardGPU", conv_bwd(false), py::arg("in_feat"), py::arg("grad_out_feat"), py::arg("kernel"),
        py::arg("kernel_size"), py::arg("kernel_stride"), py::arg("kernel_dilation"), py::arg("region_type"),
        py::arg("offset"), py::arg("convolution_mode"), py::arg("in_key"), py::arg("out_key"), py::arg("manager"),
        py::arg("need_grad_in") = true);
  m.def("ConvolutionTransposeBackwardGPU", conv_bwd(true), py::arg("in_feat"), py::arg("grad_out_feat"), py::arg("kernel"),
        py::arg("kernel_size"), py::arg("kernel_stride"), py::arg("kernel_dilation"), py::arg("region_type"),
        py::arg("offset"), py::arg("convolution_mode"), py::arg("in_key"), py::arg("out_key"), py::arg("manager"),
        py::arg("need_grad_in") = true);

  m.def("LocalPoolingForwardGPU",
        [](const Tensor &in_feat, const ivec &ks, const ivec &st, const ivec &dl, const py::object &region_type,
           const py::object & /*offset*/, const py::object &pooling_mode, CoordinateMapKey *in_key,
           CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
          auto r = local_pooling_forward(in_feat, ks, st, dl, to_int(region_type), to_int(pooling_mode), in_key, out_key, mgr);
          return py::make_tuple(r.first, r.second);
        });
  m.def("LocalPoolingBackwardGPU",
        [](const Tensor &in_feat, const Tensor &grad_out, const Tensor &num_nonzero, const ivec &ks, const ivec &st,
           const ivec &dl, const py::object &region_type, const py::object & /*offset*/, const py::object &pooling_mode,
           CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
          return local_pooling_backward(in_feat, grad_out, num_nonzero, ks, st, dl, to_int(region_type),
                                        to_int(pooling_mode), in_key, out_key, mgr);
        });
  m.def("LocalPoolingTransposeForwardGPU",
        [](const Tensor &in_feat, const ivec &ks, const ivec &st, const ivec &dl, const py::object &region_type,
           const py::object & /*offset*/, bool generate_new_coordinates, const py::object &pooling_mode,
           CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
          auto r = local_pooling_transpose_forward(in_feat, ks, st, dl, to_int(region_type), generate_new_coordinates,
                                                   to_int(pooling_mode), in_key, out_key, mgr);
          return py::make_tuple(r.first, r.second);
        });
  m.def("LocalPoolingTransposeBackwardGPU",
        [](const Tensor &in_feat, const Tensor &grad_out, const Tensor &num_nonzero, const ivec &ks, const ivec &st,
           const ivec &dl, const py::object &region_type, const py::object & /*offset*/, const py::object &pooling_mode,
           CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
          return local_pooling_transpose_backward(in_feat, grad_out, num_nonzero, ks, st, dl, to_int(region_type),
                                                  to_int(pooling_mode), in_key, out_key, mgr);
        });
  m.def("GlobalPoolingForwardGPU",
        [](const Tensor &in_feat, const py::object &pooling_mode, CoordinateMapKey *in_key, CoordinateMapKey *out_key,
           CoordinateMapManager *mgr) {
          auto r = global_pooling_forward(in_feat, to_int(pooling_mode), in_key, out_key, mgr);
          return py::make_tuple(r.first, r.second);
        });
  m.def("GlobalPoolingBackwardGPU",
        [](const Tensor &in_feat, const Tensor &grad_out, const Tensor &num_nonzero, const py::object &pooling_mode,
           CoordinateMapKey *in_key, CoordinateMapKey *out_key, CoordinateMapManager *mgr) {
          return global_pooling_backward(in_feat, grad_out, num_nonzero, to_int(pooling_mode), in_key, out_key, mgr);
        });
  m.def("BroadcastForwardGPU",
        [](const Tensor &in_feat, const Tensor &glob, const py::object &mode, CoordinateMapKey *in_key,
           CoordinateMapKey *glob_key, CoordinateMapManager *mgr) {
          return broadcast_forward(in_feat, glob, to_int(mode), in_key, glob_key, mgr);
        });
  m.def("BroadcastBackwardGPU",
        [](const Tensor &in_feat, const Tensor &glob, const Tensor &grad_out, const py::object &mode,
           CoordinateMapKey *in_key, CoordinateMapKey *glob_key, CoordinateMapManager *mgr) {
          auto r = broadcast_backward(in_feat, glob, grad_out, to_int(mode), in_key, glob_key, mgr);
          return py::make_tuple(r.first, r.second);
        });
  m.def("PruningForwardGPU", &pruning_forward);
  m.def("PruningBackwardGPU", &pruning_backward);

  // ---- batch norm over feature rows (not in the reference's module: its MinkowskiBatchNorm is torch's BatchNorm1d) ----
  m.def("bn_stats",
        [](const Tensor &x, double eps, double momentum, const py::object &rm, const py::object &rv, const py::object &nbt) {
          auto r = bn_stats(x, eps, momentum, opt_tensor(rm), opt_tensor(rv), opt_tensor(nbt));
          return py::make_tuple(r.first, r.second);
        },
        py::arg("x"), py::arg("eps"), py::arg("momentum"), py::arg("running_mean") = py::none(),
        py::arg("running_var") = py::none(), py::arg("num_batches_tracked") = py::none());
  m.def("bn_apply",
        [](const Tensor &x, const Tensor &mean, const Tensor &rstd, const py::object &gamma, const py::object &beta,
           bool relu) { return bn_apply(x, mean, rstd, opt_tensor(gamma), opt_tensor(beta), relu, Tensor()); },
        py::arg("x"), py::arg("mean"), py::arg("rstd"), py::arg("gamma"), py::arg("beta"), py::arg("relu") = false);

  // ---- autograd entry points (the Python modules call these; backward never enters Python) ----
  m.def("conv_autograd", &conv_autograd, py::arg("in_feat"), py::arg("kernel"), py::arg("kernel_size"),
        py::arg("kernel_stride"), py::arg("kernel_dilation"), py::arg("region_type"), py::arg("expand_coordinates"),
        py::arg("in_key"), py::arg("out_key"), py::arg("manager"), py::arg("transpose"));
  m.def("batch_norm_train", &batch_norm_train, py::arg("x"), py::arg("skip"), py::arg("weight"), py::arg("bias"),
        py::arg("running_mean"), py::arg("running_var"), py::arg("momentum"), py::arg("eps"), py::arg("relu"),
        py::arg("num_batches_tracked"));
}
