"""Oracle vs the reference itself, live: oracle/_ref/_C.so (the reference's CPU extension compiled
unmodified by oracle/build_ref.py).  Skipped when the shared object is absent."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from oracle import ref
from helpers import make_cloud, rel_err

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/_C.so not built")


@pytest.mark.parametrize("n,extent,D,ks,cin,cout", [
    (4000, 24, 3, 3, 8, 16),      # dense-ish 3-D
    (3000, 60, 3, 3, 16, 8),      # sparse 3-D
    (2000, 10, 4, 3, 4, 4),       # 4-D, 81 offsets
    (1500, 40, 2, 5, 3, 5),       # 2-D k=5
])
def test_kernel_map_and_conv_match_reference(n, extent, D, ks, cin, cout):
    coords = make_cloud(n, extent, D, seed=n, negative=True)
    rc = ref.RefConv(coords, ks)
    co = coords.numpy()
    _, km = O.kernel_map(co, co, O.make_region(D, ks))
    O.assert_same_kernel_map(rc.kernel_map(), km)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(len(co), cin, generator=g)
    w = torch.rand(ks ** D, cin, cout, generator=g) - 0.5
    y = rc.forward(x, w)
    assert rel_err(O.conv_forward(x.numpy(), w.numpy(), km, len(co)), y.numpy()) < 2e-6
    gy = torch.rand(y.shape, generator=g)
    gi, gw = rc.backward(x, gy, w)
    gi2, gw2 = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), km)
    assert rel_err(gi2, gi.numpy()) < 2e-6 and rel_err(gw2, gw.numpy()) < 2e-6


def test_insert_and_map_matches_reference():
    coords = make_cloud(5000, 12, 3, seed=3, batch=2, dup=700, negative=True)
    rc = ref.RefConv(coords, 3)
    um, inv = O.insert_and_map(coords.numpy())
    assert np.array_equal(um, rc.unique_map.numpy()) and np.array_equal(inv, rc.inverse_map.numpy())
    assert np.array_equal(coords.numpy()[um], rc.in_coordinates().numpy())
