"""Deterministic network weights derived from parameter names (shared by make_golden_minkunet.py and
tests/test_gpu_minkunet.py)."""
import torch


def seeded_parameters(named_parameters, rename=lambda n: n):
    """Deterministic weights derived from the parameter's name (the reference's names, which examples/minkunet.py
    of this repository shares; so the fixture does not have to store 46 MB of weights): uniform in +-sqrt(3 / fan_in) for kernels, 1 +- 0.1 / +- 0.1 for batch
    norm weight / bias."""
    import zlib
    with torch.no_grad():
        for name, p in named_parameters:
            g = torch.Generator().manual_seed(zlib.crc32(rename(name).encode()))
            r = torch.rand(p.shape, generator=g) - 0.5
            if p.dim() == 3:      # [K, Cin, Cout]
                p.copy_(r * 2 * (3.0 / (p.shape[0] * p.shape[1])) ** 0.5)
            elif p.dim() == 2:    # 1x1 kernels stored as [Cin, Cout], or the [1, Cout] bias
                p.copy_(r * 2 * (3.0 / max(p.shape[0], 1)) ** 0.5 if p.shape[0] > 1 else r * 0.2)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.2 * r)
            else:
                p.copy_(0.2 * r)


def config3_inputs():
    """Inputs of the full-size config-3 fixture (BASELINE configs[2]: MinkUNet34C, 200k-voxel scene): coordinates of
    SURVEY.md 8d's plane-union scene, features, the loss weights of loss = sum(out * w) and the 2048 sample rows,
    all from fixed seeds (CPU generators: identical in the generating script and in the test)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "examples"))
    import minkunet as ours
    coords = ours.synthetic_scene(200000)
    g = torch.Generator().manual_seed(11)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    w = torch.rand(coords.shape[0], 20, generator=g) - 0.5
    rows = torch.randperm(coords.shape[0], generator=g)[:2048]
    return coords, feats, w, rows


def grad_slice(g):
    """the part of a parameter gradient the config-3 fixture stores: everything up to 64k elements, else the
    leading [k, :32, :32] corner of every kernel offset"""
    if g.numel() <= 65536:
        return g
    return g[:, :32, :32].contiguous()
