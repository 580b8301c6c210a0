"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF: the reference's CPU
operators (ConvolutionForwardCPU, ConvolutionBackwardCPU, ConvolutionTranspose*CPU,
CoordinateMapManagerCPU — /root/reference/src/convolution_cpu.cpp, convolution_transpose_cpu.cpp,
coordinate_map_manager.cpp) compiled unmodified into oracle/_ref/_C.so by oracle/build_ref.py.

Run in the authoring container (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py
The .npz files are committed; tests never need /root/reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref  # noqa: E402

C = ref.load()
EMPTY = torch.IntTensor()


def kmap_to_arrays(km):
    ks = sorted(km.keys())
    return (np.array(ks, np.int32), np.array([km[k].shape[1] for k in ks], np.int64),
            np.concatenate([km[k].numpy() for k in ks], axis=1).astype(np.int32) if ks else np.zeros((2, 0), np.int32))


def conv_case(name, coords, cin, cout, kernel_size, stride=1, dilation=1, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    D = coords.shape[1] - 1
    aslist = lambda v: [int(v)] * D if isinstance(v, int) else [int(x) for x in v]
    ks, st, dl = aslist(kernel_size), aslist(stride), aslist(dilation)
    K = int(np.prod(ks))
    mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 4)
    in_key, (umap, imap) = mgr.insert_and_map(coords.contiguous().int(), [1] * D, "")
    n_in = mgr.size(in_key)
    feats = torch.rand(n_in, cin, generator=g, dtype=dtype)
    kernel = (torch.rand(K, cin, cout, generator=g, dtype=dtype) - 0.5)
    out_key = C.CoordinateMapKey(D + 1)
    out = C.ConvolutionForwardCPU(feats, kernel, ks, st, dl, C.RegionType.HYPER_CUBE, EMPTY, False,
                                  C.ConvolutionMode.DEFAULT, in_key, out_key, mgr)
    grad_out = torch.rand(out.shape, generator=g, dtype=dtype)
    grad_in, grad_kernel = C.ConvolutionBackwardCPU(feats, grad_out, kernel, ks, st, dl, C.RegionType.HYPER_CUBE,
                                                    EMPTY, C.ConvolutionMode.DEFAULT, in_key, out_key, mgr)
    km = mgr.kernel_map(in_key, out_key, ks, st, dl, C.RegionType.HYPER_CUBE, EMPTY, False, False)
    kk, kn, kp = kmap_to_arrays(km)
    data = dict(coords=coords.numpy().astype(np.int32), unique_map=umap.numpy(), inverse_map=imap.numpy(),
                in_coords=mgr.get_coordinates(in_key).numpy(), out_coords=mgr.get_coordinates(out_key).numpy(),
                out_tensor_stride=np.array(out_key.get_tensor_stride(), np.int32),
                kernel_size=np.array(ks, np.int32), stride=np.array(st, np.int32), dilation=np.array(dl, np.int32),
                feats=feats.numpy(), kernel=kernel.numpy(), out=out.numpy(), grad_out=grad_out.numpy(),
                grad_in=grad_in.numpy(), grad_kernel=grad_kernel.numpy(), kmap_k=kk, kmap_n=kn, kmap_pairs=kp)
    # transposed convolution back to the input map (the MinkUNet up path): only for strided cases
    if any(s > 1 for s in st):
        kernel_t = (torch.rand(K, cout, cin, generator=g, dtype=dtype) - 0.5)
        back_key = C.CoordinateMapKey([1] * D, "")
        up = C.ConvolutionTransposeForwardCPU(out, kernel_t, ks, st, dl, C.RegionType.HYPER_CUBE, EMPTY, False,
                                              C.ConvolutionMode.DEFAULT, out_key, back_key, mgr)
        gup = torch.rand(up.shape, generator=g, dtype=dtype)
        t_gin, t_gk = C.ConvolutionTransposeBackwardCPU(out, gup, kernel_t, ks, st, dl, C.RegionType.HYPER_CUBE,
                                                        EMPTY, C.ConvolutionMode.DEFAULT, out_key, back_key, mgr)
        data.update(kernel_t=kernel_t.numpy(), up=up.numpy(), up_grad_out=gup.numpy(), up_grad_in=t_gin.numpy(),
                    up_grad_kernel=t_gk.numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "n_in", n_in, "n_out", out.shape[0], "pairs", int(kn.sum()))


def pool_case(name, coords, c, kernel_size, stride, seed=0):
    """Local sum / avg / max pooling, global pooling and broadcast through the reference's CPU operators
    (src/local_pooling_cpu.cpp, src/global_pooling_cpu.cpp, src/broadcast_cpu.cpp)."""
    g = torch.Generator().manual_seed(seed)
    rp = ref.RefPool(coords, kernel_size, stride)
    n_in = rp.in_coordinates().shape[0]
    n_out = rp.out_coordinates().shape[0]
    feats = torch.rand(n_in, c, generator=g) - 0.5
    grad_out = torch.rand(n_out, c, generator=g) - 0.5
    kk, kn, kp = kmap_to_arrays(rp.pool_kernel_map())
    data = dict(coords=coords.numpy().astype(np.int32), in_coords=rp.in_coordinates().numpy(),
                out_coords=rp.out_coordinates().numpy(), kernel_size=np.array(rp.kernel_size, np.int32),
                stride=np.array(rp.stride, np.int32), feats=feats.numpy(), grad_out=grad_out.numpy(),
                kmap_k=kk, kmap_n=kn, kmap_pairs=kp)
    for mode in ("sum", "avg", "max"):
        out, aux = rp.pool_forward(feats, mode)
        gin = rp.pool_backward(feats, grad_out, aux, mode)
        data.update({mode + "_out": out.numpy(), mode + "_aux": aux.numpy(), mode + "_grad_in": gin.numpy()})
    for mode in ("sum", "avg", "max"):
        out, aux = rp.global_forward(feats, mode)
        gglob = torch.rand(out.shape, generator=g) - 0.5
        gin = rp.global_backward(feats, gglob, aux, mode)
        data.update({"g" + mode + "_out": out.numpy(), "g" + mode + "_aux": aux.numpy(),
                     "g" + mode + "_grad_out": gglob.numpy(), "g" + mode + "_grad_in": gin.numpy()})
    data["glob_coords"] = rp.glob_coordinates().numpy()
    glob = torch.rand(data["gsum_out"].shape, generator=g) - 0.5
    gb = torch.rand(n_in, c, generator=g) - 0.5
    for nm, mul in (("badd", False), ("bmul", True)):
        out = rp.broadcast_forward(feats, glob, mul)
        gi, gg = rp.broadcast_backward(feats, glob, gb, mul)
        data.update({nm + "_out": out.numpy(), nm + "_grad_in": gi.numpy(), nm + "_grad_glob": gg.numpy()})
    data.update(bglob=glob.numpy(), bgrad_out=gb.numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "n_in", n_in, "n_out", n_out, "pairs", int(kn.sum()))


def cloud(n, extent, D, seed, batch=1, dup=0):
    g = torch.Generator().manual_seed(seed)
    parts = []
    for b in range(batch):
        pts = torch.randint(-extent // 2, extent - extent // 2, (2 * n, D), generator=g)
        pts = torch.unique(pts, dim=0)
        pts = pts[torch.randperm(pts.shape[0], generator=g)][:n]
        parts.append(torch.cat([torch.full((pts.shape[0], 1), b, dtype=torch.long), pts], 1))
    c = torch.cat(parts, 0)
    if dup:
        c = torch.cat([c, c[torch.randint(0, c.shape[0], (dup,), generator=g)]], 0)
        c = c[torch.randperm(c.shape[0], generator=g)]
    return c.int()


def data_loader_fixture(batch_size=2):
    """The ASCII-art fixture of the reference tests (tests/python/common.py:57-78), re-typed."""
    rows = ["   X   ", "  X X  ", " XXXXX "]
    pts = [[r, c] for r, line in enumerate(rows) for c, ch in enumerate(line) if ch != " "]
    coords = []
    for b in range(batch_size):
        coords += [[b] + p for p in pts]
    return torch.IntTensor(coords)


def make_pool_cases():
    pool_case("pool_3d_k2s2_c16", cloud(700, 12, 3, 11, batch=3), 16, 2, 2)
    pool_case("pool_3d_k3s1_c8", cloud(500, 10, 3, 12, batch=2), 8, 3, 1)
    pool_case("pool_3d_k3s2_c5", cloud(500, 10, 3, 13, batch=2), 5, 3, 2)


if __name__ == "__main__":
    make_pool_cases()
    if "--pool-only" in sys.argv:
        sys.exit(0)
    # the reference's own fixture: conv k=3 s=2 -> 26 pairs / 10 output voxels (probe in SURVEY.md §8c)
    conv_case("ref_fixture2d_k3s2", data_loader_fixture(), 2, 3, 3, stride=2)
    # 1-D analytic case of tests/python/convolution.py:226-245
    mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 1)
    co = torch.IntTensor([[0, 0], [0, 1], [0, 2]])
    ik, _ = mgr.insert_and_map(co, [1], "")
    ok = C.CoordinateMapKey(2)
    feats = torch.FloatTensor([[0, 1], [1, 0], [1, 1]])
    W = torch.FloatTensor([[[1, 2], [2, 1]], [[0, 1], [1, 0]]])
    out = C.ConvolutionForwardCPU(feats, W, [2], [1], [1], C.RegionType.HYPER_CUBE, EMPTY, False,
                                  C.ConvolutionMode.DEFAULT, ik, ok, mgr)
    np.savez_compressed(os.path.join(HERE, "ref_analytic1d.npz"), coords=co.numpy(), feats=feats.numpy(),
                        kernel=W.numpy(), out=out.numpy())
    print("analytic1d", out.tolist())
    conv_case("ref_3d_k3s1_c4x8", cloud(600, 12, 3, 1, batch=2, dup=40), 4, 8, 3)
    conv_case("ref_3d_k3s1_c64x128", cloud(1500, 16, 3, 2), 64, 128, 3)
    conv_case("ref_3d_k2s2_c16x32", cloud(900, 14, 3, 3, batch=2), 16, 32, 2, stride=2)
    conv_case("ref_3d_k5s1_c3x32", cloud(500, 12, 3, 4), 3, 32, 5)
    conv_case("ref_3d_k3s2_c8x24", cloud(700, 14, 3, 5), 8, 24, 3, stride=2)
    conv_case("ref_3d_k3d2_c8x8", cloud(700, 12, 3, 6), 8, 8, 3, dilation=2)
    conv_case("ref_4d_k3s1_c32x64", cloud(800, 8, 4, 7), 32, 64, 3)
    conv_case("ref_2d_k32_c5x7", cloud(300, 20, 2, 8), 5, 7, [3, 2])
