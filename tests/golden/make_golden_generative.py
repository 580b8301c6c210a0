"""Golden fixture for SURVEY 8f rank 4 FROM THE REFERENCE's CPU operators (oracle/_ref/_C.so): generative
transposed convolution (ConvolutionTransposeForwardCPU with generate_new_coordinates = true ->
CoordinateMapCPU::stride_region), expanding convolution (ConvolutionForwardCPU with expand_coordinates),
pruning (PruningForwardCPU / BackwardCPU) and union maps (CoordinateMapManager::union_map).

    python tests/golden/make_golden_generative.py      (authoring container, needs /root/reference)
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
MODE = C.ConvolutionMode.DEFAULT
CUBE = C.RegionType.HYPER_CUBE


def cloud(n, extent, seed, stride=1):
    g = torch.Generator().manual_seed(seed)
    pts = torch.unique(torch.randint(-extent // 2, extent // 2, (n * 2, 3), generator=g), dim=0)
    pts = pts[torch.randperm(pts.shape[0], generator=g)][:n] * stride
    b = torch.randint(0, 2, (pts.shape[0], 1), generator=g)
    return torch.cat([b, pts], 1).int().contiguous()


if __name__ == "__main__":
    g = torch.Generator().manual_seed(11)
    data = {}
    # ---- generative transposed convolution: tensor stride 2 -> 1, k = 2 and k = 3 ------------------------
    for name, ks, st in (("gen_k2s2", 2, 2), ("gen_k3s1", 3, 1)):
        ts_in = 2 if st == 2 else 1
        coords = cloud(400, 12, 3, stride=ts_in)
        mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 4)
        in_key, _ = mgr.insert_and_map(coords, [ts_in] * 3, "")
        n = mgr.size(in_key)
        feats = torch.rand(n, 6, generator=g)
        kernel = torch.rand(ks ** 3, 6, 5, generator=g) - 0.5
        out_key = C.CoordinateMapKey(4)
        out = C.ConvolutionTransposeForwardCPU(feats, kernel, [ks] * 3, [st] * 3, [1] * 3, CUBE, EMPTY, True, MODE,
                                               in_key, out_key, mgr)
        gy = torch.rand(out.shape, generator=g)
        gi, gw = C.ConvolutionTransposeBackwardCPU(feats, gy, kernel, [ks] * 3, [st] * 3, [1] * 3, CUBE, EMPTY, MODE,
                                                   in_key, out_key, mgr)
        data.update({f"{name}/coords": mgr.get_coordinates(in_key).numpy(), f"{name}/feats": feats.numpy(),
                     f"{name}/kernel": kernel.numpy(), f"{name}/out_coords": mgr.get_coordinates(out_key).numpy(),
                     f"{name}/out": out.numpy(), f"{name}/grad_out": gy.numpy(), f"{name}/grad_in": gi.numpy(),
                     f"{name}/grad_kernel": gw.numpy(),
                     f"{name}/out_tensor_stride": np.array(out_key.get_tensor_stride(), np.int32)})
    # ---- expanding (non-transposed) convolution: k = 3, stride 2 -----------------------------------------
    coords = cloud(300, 10, 5)
    mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 4)
    in_key, _ = mgr.insert_and_map(coords, [1] * 3, "")
    feats = torch.rand(mgr.size(in_key), 4, generator=g)
    kernel = torch.rand(27, 4, 3, generator=g) - 0.5
    out_key = C.CoordinateMapKey(4)
    out = C.ConvolutionForwardCPU(feats, kernel, [3] * 3, [2] * 3, [1] * 3, CUBE, EMPTY, True, MODE, in_key, out_key, mgr)
    data.update({"expand/coords": mgr.get_coordinates(in_key).numpy(), "expand/feats": feats.numpy(),
                 "expand/kernel": kernel.numpy(), "expand/out_coords": mgr.get_coordinates(out_key).numpy(),
                 "expand/out": out.numpy()})
    # ---- pruning ---------------------------------------------------------------------------------------------
    coords = cloud(500, 14, 7)
    mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 4)
    in_key, _ = mgr.insert_and_map(coords, [1] * 3, "")
    n = mgr.size(in_key)
    feats = torch.rand(n, 5, generator=g)
    keep = torch.rand(n, generator=g) < 0.4
    out_key = C.CoordinateMapKey(4)
    out = C.PruningForwardCPU(feats, keep, in_key, out_key, mgr)
    gy = torch.rand(out.shape, generator=g)
    gi = C.PruningBackwardCPU(gy, in_key, out_key, mgr)
    data.update({"prune/coords": mgr.get_coordinates(in_key).numpy(), "prune/feats": feats.numpy(),
                 "prune/keep": keep.numpy(), "prune/out_coords": mgr.get_coordinates(out_key).numpy(),
                 "prune/out": out.numpy(), "prune/grad_out": gy.numpy(), "prune/grad_in": gi.numpy()})
    # ---- union -------------------------------------------------------------------------------------------------
    a, b = cloud(300, 8, 9), cloud(300, 8, 10)
    mgr = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, 4)
    ka, _ = mgr.insert_and_map(a, [1] * 3, "")
    kb, _ = mgr.insert_and_map(b, [1] * 3, "b")
    fa, fb = torch.rand(mgr.size(ka), 3, generator=g), torch.rand(mgr.size(kb), 3, generator=g)
    ku = C.CoordinateMapKey(4)
    maps = mgr.union_map([ka, kb], ku)
    n_out = mgr.size(ku)
    outf = torch.zeros(n_out, 3)
    for f, m in zip((fa, fb), maps):
        outf[m[1].long()] += f[m[0].long()]
    data.update({"union/a": mgr.get_coordinates(ka).numpy(), "union/b": mgr.get_coordinates(kb).numpy(),
                 "union/fa": fa.numpy(), "union/fb": fb.numpy(), "union/out_coords": mgr.get_coordinates(ku).numpy(),
                 "union/out": outf.numpy()})
    path = os.path.join(HERE, "generative_3d.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), {k: v.shape for k, v in data.items() if k.endswith("out_coords")})
