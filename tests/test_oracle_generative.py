"""SURVEY 8f rank 4 on the CPU: the oracle restatements of stride_region / prune / union against the fixture
produced by the reference's CPU operators (tests/golden/make_golden_generative.py).  The reference orders new
maps by hash-table iteration, so coordinate SETS are compared and rows are matched by coordinate."""
import os

import numpy as np

from oracle import me_oracle as O
from helpers import GOLDEN_DIR, row_mapping


def _z():
    return np.load(os.path.join(GOLDEN_DIR, "generative_3d.npz"))


def test_stride_region_matches_reference():
    z = _z()
    for name, ks, ts_out in (("gen_k2s2", 2, 1), ("gen_k3s1", 3, 1)):
        assert z[f"{name}/out_tensor_stride"].tolist() == [ts_out] * 3
        got = O.stride_region(z[f"{name}/coords"], O.make_region(3, ks, 1, ts_out))
        m = row_mapping(got, z[f"{name}/out_coords"])           # same coordinate set (asserted inside)
        assert len(got) == len(z[f"{name}/out_coords"]) and len(np.unique(m)) == len(m)
        # features: transposed kernel map = forward map from the new map onto the input map, roles swapped
        _, km = O.kernel_map(got, z[f"{name}/coords"], O.make_region(3, ks, 1, ts_out))
        kmt = {k: v[::-1].copy() for k, v in km.items()}
        out = O.conv_forward(z[f"{name}/feats"], z[f"{name}/kernel"], kmt, len(got))
        assert np.abs(out - z[f"{name}/out"][m]).max() < 1e-5
    got = O.stride_region(z["expand/coords"], O.make_region(3, 3, 1, 1), out_tensor_stride=2)
    m = row_mapping(got, z["expand/out_coords"])
    _, km = O.kernel_map(z["expand/coords"], got, O.make_region(3, 3, 1, 1))
    out = O.conv_forward(z["expand/feats"], z["expand/kernel"], km, len(got))
    assert np.abs(out - z["expand/out"][m]).max() < 1e-5


def test_prune_and_union_match_reference():
    z = _z()
    c, f = O.prune(z["prune/coords"], z["prune/feats"], z["prune/keep"])
    m = row_mapping(c, z["prune/out_coords"])
    assert np.array_equal(f, z["prune/out"][m])
    gi = np.zeros_like(z["prune/feats"])
    gi[z["prune/keep"]] = z["prune/grad_out"][m]
    assert np.array_equal(gi, z["prune/grad_in"])
    c, f = O.union([z["union/a"], z["union/b"]], [z["union/fa"], z["union/fb"]])
    m = row_mapping(c, z["union/out_coords"])
    assert np.abs(f - z["union/out"][m]).max() < 1e-6
