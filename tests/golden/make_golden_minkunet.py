"""Golden fixtures for the whole-network rows: the REFERENCE's own MinkUNet (examples/minkunet.py on the
reference's Python package + its CPU operators compiled unmodified into oracle/_ref/_C.so) run forward and
backward on a synthetic scene.  examples/minkunet.py of this repository uses the reference's module names, so
parameters are addressed by the same names on both sides; the weights are derived from those names
(make_golden_minkunet_weights.py) and never stored.

  minkunet14_3k.npz      MinkUNet14, 3k-voxel scene: full output, input gradient, five parameter gradients
  minkunet34c_200k.npz   MinkUNet34C on the 200k-voxel scene of BASELINE configs[2]: the loss, a
                         2048-row sample of the output and of the input gradient, nine parameter gradients
                         (sliced where large) — the full-size config-3 check

Every quantity is stored twice: from the reference run in float32 (its production arithmetic) and in float64
(`...64`, the arbiter SURVEY.md 8c names), plus `noise/<name>` = max |float32 run - float64 run|, the reference's OWN
fp32 error.  The loss sum(out * w) has random-sign weights, so gradients are sums of ~10^5 cancelling terms and their
fp32 error is far above 1e-4 of their size in ANY implementation; the tests therefore bound |ours - float64 truth|
by a small multiple of the reference's own fp32 noise instead of a fixed fraction.

Run in the authoring container from any directory (needs /root/reference):
    python tests/golden/make_golden_minkunet.py [14] [34c]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref  # noqa: E402

RME = ref.import_reference_package()
from examples.minkunet import MinkUNet14, MinkUNet34C  # noqa: E402

from make_golden_minkunet_weights import config3_inputs, grad_slice, seeded_parameters  # noqa: E402

GRADS_14 = ("conv0p1s1.kernel", "final.kernel", "block1.0.conv1.kernel", "bn0.bn.weight", "convtr7p2s2.kernel")
GRADS_34C = ("conv0p1s1.kernel", "final.kernel", "final.bias", "block1.0.conv1.kernel", "bn0.bn.weight",
             "convtr7p2s2.kernel", "block4.5.conv2.kernel", "block6.0.conv1.kernel", "bntr4.bn.bias")


def _run(net_cls, n_out, coords, feats, w, dtype):
    """one forward + backward of the reference network in `dtype` -> (out, grad_feats, {name: grad}, net)"""
    net = net_cls(3, n_out, D=3)
    seeded_parameters(net.named_parameters())
    net = net.to(dtype).train()
    fin = feats.to(dtype).clone().requires_grad_(True)
    y = net(RME.SparseTensor(fin, coords))
    loss = (y.F * w.to(dtype)).sum()
    loss.backward()
    return y, loss, fin.grad, {n: p.grad for n, p in net.named_parameters()}, net


def _noise(a32, a64):
    """the reference's own fp32 error against its float64 run: what "equal to the reference" can mean at best"""
    return np.float64((a32.double() - a64).abs().max().item())


def small():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import minkunet as ours  # examples/minkunet.py of this repository (scene generator only)
    coords = ours.synthetic_scene(3000, grid=48, seed=3)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    w = torch.rand(coords.shape[0], 5, generator=g) - 0.5
    y, loss, gin, grads, net = _run(MinkUNet14, 5, coords, feats, w, torch.float32)
    y64, loss64, gin64, grads64, _ = _run(MinkUNet14, 5, coords, feats, w, torch.float64)
    assert torch.equal(y.C, y64.C)
    data = {"coords": coords.numpy(), "feats": feats.numpy(), "loss_weight": w.numpy(), "out": y.F.detach().numpy(),
            "out_coords": y.C.numpy(), "grad_feats": gin.numpy(),
            "out64": y64.F.detach().numpy(), "grad_feats64": gin64.numpy(),
            "noise/out": _noise(y.F.detach(), y64.F.detach()), "noise/grad_feats": _noise(gin, gin64)}
    data["param_names"] = np.array([n for n, _ in net.named_parameters()])
    for name in GRADS_14:
        data["grad/" + name] = grads[name].numpy()
        data["grad64/" + name] = grads64[name].numpy()
        data["noise/" + name] = _noise(grads[name], grads64[name])
    np.savez_compressed(os.path.join(HERE, "minkunet14_3k.npz"), **data)
    print("saved minkunet14_3k", y.F.shape, "params", sum(p.numel() for p in net.parameters()),
          {k: float(v) for k, v in data.items() if k.startswith("noise/")})


def config3():
    coords, feats, w, rows = config3_inputs()
    y, loss, gin, grads, net = _run(MinkUNet34C, 20, coords, feats, w, torch.float32)
    assert torch.equal(y.C, coords), "the stride-1 output map is the input map, rows in input order"
    y64, loss64, gin64, grads64, _ = _run(MinkUNet34C, 20, coords, feats, w, torch.float64)
    data = {"n": np.int64(coords.shape[0]), "loss": np.float64(loss.item()), "loss64": np.float64(loss64.item()),
            "rows": rows.numpy(),
            "out_rows": y.F.detach()[rows].numpy(), "out_rows64": y64.F.detach()[rows].numpy(),
            "out_absmax": np.float32(y.F.detach().abs().max().item()),
            "out_sum": np.float64(y64.F.detach().sum().item()),
            "noise/out": _noise(y.F.detach(), y64.F.detach()),
            "grad_feats_rows": gin[rows].numpy(), "grad_feats_rows64": gin64[rows].numpy(),
            "grad_feats_absmax": np.float32(gin.abs().max().item()), "noise/grad_feats": _noise(gin, gin64)}
    data["param_names"] = np.array([n for n, _ in net.named_parameters()])
    for name in GRADS_34C:
        data["grad/" + name] = grad_slice(grads[name]).numpy()
        data["grad64/" + name] = grad_slice(grads64[name]).numpy()
        data["gradmax/" + name] = np.float32(grads[name].abs().max().item())
        data["noise/" + name] = _noise(grads[name], grads64[name])
    np.savez_compressed(os.path.join(HERE, "minkunet34c_200k.npz"), **data)
    print("saved minkunet34c_200k loss", loss.item(), loss64.item(), "params", sum(p.numel() for p in net.parameters()),
          {k: float(v) for k, v in data.items() if k.startswith("noise/")})


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["14", "34c"]
    if "14" in which:
        small()
    if "34c" in which:
        config3()
