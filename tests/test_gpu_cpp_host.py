"""The C ABI without Python in the loop: examples/cpp/conv_layer (plain C++ + hipMalloc, linked against libme_amd.so)
runs a whole convolution layer — insert, kernel map, plans, forward, input gradient, weight gradient — on inputs
written by this test, and its outputs are compared with the oracle per element."""
import os
import subprocess

import numpy as np
import pytest

from oracle import me_oracle as O
from helpers import assert_close, make_cloud

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "cpp", "conv_layer")


@pytest.mark.parametrize("n,extent,D,cin,cout,ks", [(4000, 14, 3, 64, 128, 3),    # bf16x6 split kernels
                                                     (3000, 14, 3, 16, 32, 3),     # fp32-MFMA kernels
                                                     (1500, 8, 4, 32, 64, 3)])     # 4-D, K = 81
def test_cpp_host_layer_matches_the_oracle(tmp_path, n, extent, D, cin, cout, ks):
    if not os.path.exists(BIN):
        from minkowskiengine_amd import build as me_build
        me_build.build_cpp_example()
    coords = make_cloud(n, extent, D, seed=n + cin, batch=1).numpy().astype(np.int32)
    n = coords.shape[0]
    rng = np.random.default_rng(n)
    feats = rng.random((n, cin), dtype=np.float32)
    kernel = (rng.random((ks ** D, cin, cout), dtype=np.float32) - 0.5).astype(np.float32)
    gout = rng.random((n, cout), dtype=np.float32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([n, D + 1, cin, cout, ks], dtype=np.int64).tofile(f)
        coords.tofile(f)
        feats.tofile(f)
        kernel.tofile(f)
        gout.tofile(f)
    env = dict(os.environ)
    for k in list(env):                      # the binary must not need anything Python-side
        if k.startswith("PYTHON"):
            del env[k]
    r = subprocess.run([BIN, str(fin), str(fout)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    print(r.stdout.strip())
    with open(fout, "rb") as f:
        n_unique, n_pairs, split = np.fromfile(f, dtype=np.int64, count=3)
        out = np.fromfile(f, dtype=np.float32, count=n * cout).reshape(n, cout)
        gin = np.fromfile(f, dtype=np.float32, count=n * cin).reshape(n, cin)
        gw = np.fromfile(f, dtype=np.float32, count=kernel.size).reshape(kernel.shape)
    assert n_unique == n
    assert bool(split) == (cin * cout >= 8192)
    _, km = O.kernel_map(coords, coords, O.make_region(D, ks))
    assert n_pairs == sum(io.shape[1] for io in km.values())
    assert_close(out, O.conv_forward(feats, kernel, km, n))
    want_gin, want_gw = O.conv_backward(feats, gout, kernel, km)
    assert_close(gin, want_gin)
    assert_close(gw, want_gw)
