"""Checkpoint compatibility of examples/minkunet.py with the reference's MinkUNet family
(/root/reference/examples/minkunet.py:35-245; SURVEY.md 5: "keep parameter names/shapes so reference checkpoints
load"): the parameter names recorded by the reference-generated fixtures, and — where the reference tree and its
compiled CPU extension are present — the full state_dict (names, shapes, buffers) of the reference's own
network, loaded strictly into ours."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


@pytest.mark.parametrize("cls,fixture", [("MinkUNet14", "minkunet14_3k.npz"), ("MinkUNet34C", "minkunet34c_200k.npz")])
def test_parameter_names_match_the_reference_fixture(cls, fixture):
    import minkunet
    path = os.path.join(GOLDEN_DIR, fixture)
    if not os.path.exists(path):
        pytest.skip(f"{fixture} not generated")
    z = np.load(path)
    net = getattr(minkunet, cls)(3, 5 if cls == "MinkUNet14" else 20, D=3)
    assert [n for n, _ in net.named_parameters()] == z["param_names"].tolist()


_DUMP = r"""
import json, sys
sys.path.insert(0, {root!r})
from oracle import ref
ref.import_reference_package()
from examples.minkunet import {cls}
net = {cls}(3, 20, D=3)
print("STATE" + json.dumps({{k: list(v.shape) for k, v in net.state_dict().items()}}))
"""


@pytest.mark.skipif(not ref.package_available(), reason="needs /root/reference and oracle/_ref/_C.so")
@pytest.mark.parametrize("cls", ["MinkUNet14", "MinkUNet18", "MinkUNet34C", "MinkUNet50"])
def test_reference_state_dict_loads_strictly(cls):
    import torch
    import minkunet
    out = subprocess.run([sys.executable, "-c", _DUMP.format(root=ROOT, cls=cls)], capture_output=True, text=True,
                         timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("STATE")]
    assert line, out.stderr[-2000:]
    ref_shapes = json.loads(line[0][5:])
    net = getattr(minkunet, cls)(3, 20, D=3)
    ours = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert list(ours) == list(ref_shapes), "state_dict keys (or their order) differ from the reference network"
    assert ours == ref_shapes
    # a state dict with the reference's names and shapes loads strictly
    sd = {k: torch.zeros(s) if "num_batches_tracked" not in k else torch.zeros((), dtype=torch.long)
          for k, s in ref_shapes.items()}
    net.load_state_dict(sd, strict=True)
