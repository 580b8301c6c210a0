"""Host-side checks of the example helpers that bench.py's config-3 step uses."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_composed_cross_entropy_equals_torch():
    import minkunet as MU
    g = torch.Generator().manual_seed(0)
    z = (torch.randn(1000, 20, generator=g) * 3).requires_grad_(True)
    y = torch.randint(0, 20, (1000,), generator=g)
    a = MU.cross_entropy(z, y)
    (ga,) = torch.autograd.grad(a, z)
    b = torch.nn.CrossEntropyLoss()(z, y)
    (gb,) = torch.autograd.grad(b, z)
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-8)
    # bf16 logits are promoted to fp32 like `.F.float()` in the training step
    zb = z.detach().bfloat16().requires_grad_(True)
    c = MU.cross_entropy(zb, y)
    d = torch.nn.CrossEntropyLoss()(zb.float(), y)
    assert torch.allclose(c, d, rtol=1e-6, atol=1e-6)
