// A sparse convolution layer — coordinate insert, kernel map, tile plans, forward, input gradient, weight gradient —
// driven from plain C++ through the C ABI of libme_amd.so (include/me_amd.h): no Python, no torch, no pybind.  This
// is the call sequence a C++ host such as the reference's ConvolutionForwardGPU / ConvolutionBackwardGPU
// (src/convolution_gpu.cu:45-244) issues when it binds this library; device memory is plain
// hipMalloc.
//
//   conv_layer <in.bin> <out.bin>
//   in.bin : int64 {n, ncol, c_in, c_out, kernel_size}, int32 coords[n * ncol] (batch index first),
//            float feats[n * c_in], float kernel[K * c_in * c_out] (K = kernel_size^(ncol - 1)), float grad_out[n * c_out]
//   out.bin: int64 {n_unique, n_pairs, used_split_kernel}, float out[n_unique * c_out], float grad_in[n_unique * c_in],
//            float grad_kernel[K * c_in * c_out]
// (stride 1, dilation 1, hyper-cube region; the input must hold no duplicate coordinates so that rows keep their
// order.)  tests/test_gpu_cpp_host.py runs it against the oracle.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "me_amd.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)
#define ME_OK(x)                                                                       \
  do {                                                                                 \
    if ((x) != 0) {                                                                    \
      std::fprintf(stderr, "%s:%d me_amd: %s\n", __FILE__, __LINE__, me_last_error()); \
      std::exit(3);                                                                    \
    }                                                                                  \
  } while (0)

template <typename T>
static T *dev_alloc(int64_t count) {
  void *p = nullptr;
  HIP_OK(hipMalloc(&p, (size_t)(count > 0 ? count : 1) * sizeof(T)));
  return static_cast<T *>(p);
}
template <typename T>
static T *to_device(const std::vector<T> &h) {
  T *d = dev_alloc<T>((int64_t)h.size());
  HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <typename T>
static void read_vec(std::FILE *f, std::vector<T> &v, int64_t count) {
  v.resize((size_t)count);
  if (std::fread(v.data(), sizeof(T), (size_t)count, f) != (size_t)count) {
    std::fprintf(stderr, "short read\n");
    std::exit(1);
  }
}

struct Plan {
  int32_t tile_rows = 0, batch_groups = 0;
  int32_t *src = nullptr, *dst = nullptr, *desc = nullptr, *bptr = nullptr, *gptr = nullptr;
};

static Plan build_plan(const int32_t *tbl, int64_t n_tgt, int64_t volume, int64_t n_pairs, int c_src, int c_dst,
                       bool split, hipStream_t stream) {
  Plan p;
  ME_OK((split ? me_conv_plan_config_f32x3 : me_conv_plan_config)(n_tgt, volume, n_pairs, c_src, c_dst, &p.tile_rows,
                                                                  &p.batch_groups));
  const int64_t groups = me_plan_max_groups(n_tgt, volume, n_pairs, p.tile_rows);
  const int64_t tiles = me_plan_num_tiles(n_tgt, p.tile_rows);
  p.src = dev_alloc<int32_t>(16 * groups);
  p.dst = dev_alloc<int32_t>(16 * groups);
  p.desc = dev_alloc<int32_t>(2 * groups);
  p.bptr = dev_alloc<int32_t>(me_plan_tile_bptr_elems(n_tgt, p.tile_rows));
  p.gptr = dev_alloc<int32_t>(tiles * volume + 1);
  const int64_t wsb = me_plan_workspace_bytes(n_tgt, volume, p.tile_rows);
  char *ws = dev_alloc<char>(wsb);
  ME_OK(me_plan_build(tbl, nullptr, n_tgt, volume, p.tile_rows, p.batch_groups, p.src, p.dst, p.desc, p.bptr, p.gptr,
                      ws, wsb, stream));
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipFree(ws));
  return p;
}

// dst = conv over the plan (forward: transposed = 0; input gradient: source = grad_out, transposed = 1)
static void conv_target(const float *src, int64_t n_src, int c_src, const float *kernel, int64_t volume, int c_dst,
                        const Plan &p, float *dst, int64_t n_tgt, int transposed, bool split, hipStream_t stream) {
  if (split) {
    uint16_t *packed = dev_alloc<uint16_t>(me_conv_packed_weight_elems_f32x3(volume, c_src, c_dst));
    ME_OK(me_conv_pack_weights_f32x3(kernel, volume, c_src, c_dst, transposed, packed, stream));
    ME_OK(me_conv_target_f32x3(src, n_src, c_src, packed, volume, c_dst, p.src, p.dst, p.desc, p.bptr, nullptr, dst,
                               n_tgt, p.tile_rows, p.batch_groups, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipFree(packed));
  } else {
    float *packed = dev_alloc<float>(me_conv_packed_weight_elems(volume, c_src, c_dst));
    ME_OK(me_conv_pack_weights_f32(kernel, volume, c_src, c_dst, transposed, packed, stream));
    ME_OK(me_conv_target_f32(src, n_src, c_src, packed, volume, c_dst, p.src, p.dst, p.desc, p.bptr, nullptr, dst, n_tgt,
                             p.tile_rows, p.batch_groups, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipFree(packed));
  }
}

int main(int argc, char **argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]);
    return 1;
  }
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) {
    std::perror(argv[1]);
    return 1;
  }
  std::vector<int64_t> hdr;
  read_vec(f, hdr, 5);
  const int64_t n = hdr[0];
  const int ncol = (int)hdr[1], c_in = (int)hdr[2], c_out = (int)hdr[3], ksize = (int)hdr[4];
  me_region region = {};
  region.ncol = ncol;
  region.region_type = ME_REGION_HYPER_CUBE;
  for (int d = 0; d < ncol - 1; ++d) {
    region.kernel_size[d] = ksize;
    region.dilation[d] = 1;
    region.tensor_stride[d] = 1;
  }
  const int64_t volume = me_region_volume(&region);
  std::vector<int32_t> h_coords;
  std::vector<float> h_feats, h_kernel, h_gout;
  read_vec(f, h_coords, n * ncol);
  read_vec(f, h_feats, n * c_in);
  read_vec(f, h_kernel, volume * c_in * c_out);
  read_vec(f, h_gout, n * c_out);
  std::fclose(f);

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const auto t0 = std::chrono::steady_clock::now();

  // ---- coordinate map: hash insert + dedup (CoordinateMapManager::insert_and_map) --------------------------------
  int32_t *coords_in = to_device(h_coords);
  const int64_t capacity = me_hash_capacity(n);
  uint64_t *table = dev_alloc<uint64_t>(capacity);
  int32_t *coords = dev_alloc<int32_t>(n * ncol);
  int64_t *unique_map = dev_alloc<int64_t>(n), *inverse_map = dev_alloc<int64_t>(n);
  int64_t n_unique = 0;
  {
    const int64_t wsb = me_insert_workspace_bytes(n);
    char *ws = dev_alloc<char>(wsb);
    ME_OK(me_coords_insert_and_map(coords_in, n, ncol, table, capacity, coords, unique_map, inverse_map, &n_unique, ws,
                                   wsb, stream));
    HIP_OK(hipFree(ws));
  }
  if (n_unique != n) {
    std::fprintf(stderr, "the example expects unique coordinates (%lld of %lld are)\n", (long long)n_unique, (long long)n);
    return 1;
  }

  // ---- kernel map (CoordinateMapManager::kernel_map): neighbour table, pair lists, transposed table ---------------
  int32_t *nbr = dev_alloc<int32_t>(volume * n), *nbr_t = dev_alloc<int32_t>(volume * n);
  std::vector<int64_t> k_offsets((size_t)volume + 1);
  int64_t *k_offsets_dev = dev_alloc<int64_t>(volume + 1);
  const int64_t kwsb = me_kernel_map_workspace_bytes(n, volume);
  char *kws = dev_alloc<char>(kwsb);
  ME_OK(me_kernel_map_probe(table, capacity, coords, coords, n, &region, nbr, k_offsets.data(), k_offsets_dev, kws, kwsb,
                            stream));
  const int64_t n_pairs = k_offsets[(size_t)volume];
  int32_t *in_pairs = dev_alloc<int32_t>(n_pairs), *out_pairs = dev_alloc<int32_t>(n_pairs);
  ME_OK(me_kernel_map_compact(nbr, n, volume, in_pairs, out_pairs, kws, kwsb, stream));
  ME_OK(me_kernel_map_transpose(in_pairs, out_pairs, k_offsets_dev, volume, n_pairs, n, nbr_t, stream));

  // ---- tile plans + the three feature kernels --------------------------------------------------------------------
  const bool split = me_conv_f32x3_supported(c_in, c_out) && me_conv_f32x3_supported(c_out, c_in) &&
                     (int64_t)c_in * c_out >= 8192;   // the dispatch rule of minkowskiengine_amd/backend.py
  const Plan fwd = build_plan(nbr, n, volume, n_pairs, c_in, c_out, split, stream);
  const Plan bwd = build_plan(nbr_t, n, volume, n_pairs, c_out, c_in, split, stream);
  float *feats = to_device(h_feats), *kernel = to_device(h_kernel), *gout = to_device(h_gout);
  float *out = dev_alloc<float>(n * c_out), *gin = dev_alloc<float>(n * c_in);
  float *gw = dev_alloc<float>(volume * c_in * c_out);
  conv_target(feats, n, c_in, kernel, volume, c_out, fwd, out, n, 0, split, stream);
  conv_target(gout, n, c_out, kernel, volume, c_in, bwd, gin, n, 1, split, stream);
  {
    const int64_t wsb = me_conv_wgrad_workspace_bytes(k_offsets.data(), volume, c_in, c_out);
    char *ws = dev_alloc<char>(wsb);
    ME_OK(me_conv_wgrad_f32(feats, n, c_in, gout, n, c_out, in_pairs, out_pairs, k_offsets.data(), k_offsets_dev, volume,
                            gw, ws, wsb, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipFree(ws));
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  std::vector<float> h_out((size_t)(n * c_out)), h_gin((size_t)(n * c_in)), h_gw((size_t)(volume * c_in * c_out));
  HIP_OK(hipMemcpy(h_out.data(), out, h_out.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_gin.data(), gin, h_gin.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_gw.data(), gw, h_gw.size() * 4, hipMemcpyDeviceToHost));
  std::FILE *g = std::fopen(argv[2], "wb");
  if (!g) {
    std::perror(argv[2]);
    return 1;
  }
  const int64_t ohdr[3] = {n_unique, n_pairs, split ? 1 : 0};
  std::fwrite(ohdr, sizeof(int64_t), 3, g);
  std::fwrite(h_out.data(), 4, h_out.size(), g);
  std::fwrite(h_gin.data(), 4, h_gin.size(), g);
  std::fwrite(h_gw.data(), 4, h_gw.size(), g);
  std::fclose(g);
  std::printf("me_amd %d: %lld voxels, K = %lld, %lld pairs, %d -> %d channels, %s kernels, tiles of %d / %d rows, "
              "cold layer (maps + plans + forward + backward) %.2f ms\n",
              me_version(), (long long)n, (long long)volume, (long long)n_pairs, c_in, c_out,
              split ? "bf16x6 split" : "fp32 MFMA", fwd.tile_rows, bwd.tile_rows, ms);
  return 0;
}
