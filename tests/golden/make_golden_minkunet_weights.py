"""Deterministic network weights derived from parameter names (shared by make_golden_minkunet.py and
tests/test_gpu_minkunet.py)."""
import torch


def seeded_parameters(named_parameters, rename=lambda n: n):
    """Deterministic weights derived from the parameter's name in THIS repository's naming (so the fixture does
    not have to store 46 MB of weights): uniform in +-sqrt(3 / fan_in) for kernels, 1 +- 0.1 / +- 0.1 for batch
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
