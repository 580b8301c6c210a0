"""Golden fixtures for the whole-network rows: the REFERENCE's own MinkUNet (examples/minkunet.py on the
reference's Python package + its CPU operators compiled unmodified into oracle/_ref/_C.so) run forward and
backward on a synthetic scene.  examples/minkunet.py of this repository uses the reference's module names, so
parameters are addressed by the same names on both sides; the weights are derived from those names
(make_golden_minkunet_weights.py) and never stored.

  minkunet14_3k.npz      MinkUNet14, 3k-voxel scene: full output, input gradient, five parameter gradients
  minkunet34c_200k.npz   MinkUNet34C on the 200k-voxel scene of BASELINE configs[2]: the loss, a
                         2048-row sample of the output and of the input gradient, six parameter gradients
                         (sliced where large) — the full-size config-3 check

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


def small():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import minkunet as ours  # examples/minkunet.py of this repository (scene generator only)
    coords = ours.synthetic_scene(3000, grid=48, seed=3)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    net = MinkUNet14(3, 5, D=3)
    seeded_parameters(net.named_parameters())
    net.train()
    fin = feats.clone().requires_grad_(True)
    y = net(RME.SparseTensor(fin, coords))
    w = torch.rand(y.F.shape, generator=g) - 0.5
    (y.F * w).sum().backward()
    data = {"coords": coords.numpy(), "feats": feats.numpy(), "loss_weight": w.numpy(), "out": y.F.detach().numpy(),
            "out_coords": y.C.numpy(), "grad_feats": fin.grad.numpy()}
    data["param_names"] = np.array([n for n, _ in net.named_parameters()])
    for name, p in net.named_parameters():
        if name in GRADS_14:
            data["grad/" + name] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "minkunet14_3k.npz"), **data)
    print("saved minkunet14_3k", y.F.shape, "params", sum(p.numel() for p in net.parameters()))


def config3():
    coords, feats, w, rows = config3_inputs()
    net = MinkUNet34C(3, 20, D=3)
    seeded_parameters(net.named_parameters())
    net.train()
    fin = feats.clone().requires_grad_(True)
    y = net(RME.SparseTensor(fin, coords))
    assert torch.equal(y.C, coords), "the stride-1 output map is the input map, rows in input order"
    loss = (y.F * w).sum()
    loss.backward()
    data = {"n": np.int64(coords.shape[0]), "loss": np.float64(loss.item()), "rows": rows.numpy(),
            "out_rows": y.F.detach()[rows].numpy(), "out_absmax": np.float32(y.F.detach().abs().max().item()),
            "out_sum": np.float64(y.F.detach().double().sum().item()),
            "grad_feats_rows": fin.grad[rows].numpy(), "grad_feats_absmax": np.float32(fin.grad.abs().max().item())}
    data["param_names"] = np.array([n for n, _ in net.named_parameters()])
    for name, p in net.named_parameters():
        if name in GRADS_34C:
            data["grad/" + name] = grad_slice(p.grad).numpy()
            data["gradmax/" + name] = np.float32(p.grad.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "minkunet34c_200k.npz"), **data)
    print("saved minkunet34c_200k loss", loss.item(), "params", sum(p.numel() for p in net.parameters()))


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["14", "34c"]
    if "14" in which:
        small()
    if "34c" in which:
        config3()
