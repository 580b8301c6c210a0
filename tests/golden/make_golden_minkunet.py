"""Golden fixture for the whole-network row: the REFERENCE's own MinkUNet14 (examples/minkunet.py on the
reference's Python package + its CPU operators compiled unmodified into oracle/_ref/_C.so) run forward and
backward on a small synthetic scene; weights, input, output and a few gradients are saved under OUR module
names, so tests/test_gpu_minkunet.py can load them into examples/minkunet.py and compare.

Run in the authoring container from any directory (needs /root/reference):
    python tests/golden/make_golden_minkunet.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref  # noqa: E402

C = ref.load()
pkg = types.ModuleType("MinkowskiEngineBackend")
pkg._C = C
pkg.__path__ = []
sys.modules["MinkowskiEngineBackend"] = pkg
sys.modules["MinkowskiEngineBackend._C"] = C
sys.modules["open3d"] = types.ModuleType("open3d")          # examples/resnet.py imports it at module scope
sys.path.insert(0, os.environ.get("ME_REFERENCE_ROOT", "/root/reference"))
work = tempfile.mkdtemp()
os.chdir(work)
open("1.ply", "w").close()                                   # keeps examples/resnet.py from downloading
import MinkowskiEngine as RME  # noqa: E402
from examples.minkunet import MinkUNet14  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "examples"))


def our_name(ref_name):
    """reference parameter / buffer name -> name in examples/minkunet.py of this repository"""
    head, rest = ref_name.split(".", 1)
    table = {"conv0p1s1": "conv0p1s1", "bn0": "bn0", "final": "final"}
    for lvl, (c, b, blk) in enumerate([("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"),
                                       ("conv3p4s2", "bn3", "block3"), ("conv4p8s2", "bn4", "block4")]):
        table.update({c: f"down.{lvl}", b: f"down_bn.{lvl}", blk: f"enc.{lvl}"})
    for lvl, (c, b, blk) in enumerate([("convtr4p16s2", "bntr4", "block5"), ("convtr5p8s2", "bntr5", "block6"),
                                       ("convtr6p4s2", "bntr6", "block7"), ("convtr7p2s2", "bntr7", "block8")]):
        table.update({c: f"up.{lvl}", b: f"up_bn.{lvl}", blk: f"dec.{lvl}"})
    return table[head] + "." + rest


from make_golden_minkunet_weights import seeded_parameters  # noqa: E402


if __name__ == "__main__":
    torch.manual_seed(0)
    import minkunet as ours  # examples/minkunet.py of this repository (scene generator only)
    coords = ours.synthetic_scene(3000, grid=48, seed=3)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    net = MinkUNet14(3, 5, D=3)
    seeded_parameters(net.named_parameters(), our_name)
    net.train()
    fin = feats.clone().requires_grad_(True)
    x = RME.SparseTensor(fin, coords)
    y = net(x)
    w = torch.rand(y.F.shape, generator=g) - 0.5
    (y.F * w).sum().backward()
    data = {"coords": coords.numpy(), "feats": feats.numpy(), "loss_weight": w.numpy(), "out": y.F.detach().numpy(),
            "out_coords": y.C.numpy(), "grad_feats": fin.grad.numpy()}
    data["param_names"] = np.array([our_name(n) for n, _ in net.named_parameters()])
    for name, p in net.named_parameters():
        if name in ("conv0p1s1.kernel", "final.kernel", "block1.0.conv1.kernel", "bn0.bn.weight", "convtr7p2s2.kernel"):
            data["grad/" + our_name(name)] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "minkunet14_3k.npz"), **data)
    print("saved", y.F.shape, "params", sum(p.numel() for p in net.parameters()))
