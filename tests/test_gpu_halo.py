"""GPU parity of the output-stationary bf16 convolution on an LDS-staged source halo (csrc/conv_halo.hip, round 5).

Same oracle and tolerance as tests/test_gpu_bf16.py: the fp32 reference algorithm (oracle/me_oracle.py, pinned to the
compiled reference) applied to the bf16-rounded operands; element-wise |err| <= 2^-8 |ref| + 1e-3 max|ref| for bf16
outputs.  The halo kernel is FORCED here (me_debug_set_halo(1, ...)) for every shape it is instantiated for, on both
hosts, in both tile heights and with / without the empty-group skip; the halo plan itself (integer work) is compared
bit for bit with a numpy restatement."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import assert_close, make_cloud
from test_gpu_bf16 import _run_layer, assert_bf16_close, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture
def halo():
    """set(mode, tile_rows, kc, skip): forces the halo kernel; policy restored afterwards"""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        # (round 6, VERDICT r5 item 7: the shipped library holds the two wave shapes the auto policy selects — they are
        # oracle-checked by test_halo_auto_policy_selects_the_measured_shapes —; the matrix of forced shapes / tile heights /
        # channel chunks below needs the tuning build: scripts/build_debug.sh)
        pytest.skip("forced halo shapes are only in a -DME_DEBUG_VARIANTS build (scripts/build_debug.sh)")
    try:
        yield lib.me_debug_set_halo
    finally:
        lib.me_debug_set_halo(-1, 0, 0, 1)


def numpy_halo_plan(tbl, col_order, n_tgt, tile_rows, s_cap, src_pos=None):
    """the arrays of me_halo_plan_build, restated: per tile the distinct source rows sorted by position (src_pos: row ->
    position; None: by row), local slots, group masks"""
    volume = tbl.shape[0]
    tiles = -(-n_tgt // tile_rows)
    cnt = np.zeros(tiles, np.int32)
    rows = np.full((tiles, s_cap), -1, np.int32)
    lidx = np.zeros((tiles, volume, tile_rows), np.uint16)
    kmask = np.zeros((tiles, volume), np.uint32)
    for t in range(tiles):
        p = np.arange(t * tile_rows, min(n_tgt, (t + 1) * tile_rows))
        cols = col_order[p] if col_order is not None else p
        v = tbl[:, cols]                                    # [volume, rows_here]
        u = np.unique(v[v >= 0])
        if src_pos is not None:
            u = u[np.argsort(src_pos[u], kind="stable")]
        cnt[t] = len(u)
        rows[t, :min(len(u), s_cap)] = u[:s_cap]
        rank = np.empty(int(u.max()) + 1 if len(u) else 1, np.int64)
        rank[u] = np.arange(len(u))
        slot = rank[np.maximum(v, 0)] if len(u) else np.zeros_like(v)
        li = np.where(v >= 0, np.where(slot < s_cap, slot + 1, 0xffff), 0)
        lidx[t, :, :len(p)] = li
        for g in range(tile_rows // 16):
            anyp = (v[:, g * 16:(g + 1) * 16] >= 0).any(1)
            kmask[t] |= (anyp.astype(np.uint32) << g)
    return cnt, rows, lidx, kmask


@pytest.mark.parametrize("n,extent,tile_rows,s_cap", [(3000, 14, 128, 383), (3000, 14, 64, 255), (2000, 40, 128, 383),
                                                       (777, 9, 64, 40), (130, 6, 128, 383)])
def test_halo_plan_matches_numpy(device, n, extent, tile_rows, s_cap):
    """bit-exact: halo sizes, halo rows (ascending), local slots (0 absent / 1 + slot / 0xffff beyond s_cap), group masks;
    (777, 9, 64, 40) forces overflowing halos"""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    coords = make_cloud(n, extent, 3, seed=n, batch=2, negative=True).to(device)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    for target in ("out", "in"):
        tbl, native = km.table_pos(target)
        n_tgt = km.n_out if target == "out" else km.n_in
        col_order = None if native is not None else km._flat_order(target, "spatial")
        tiles = -(-n_tgt // tile_rows)
        cnt = torch.empty(tiles, dtype=torch.int32, device=device)
        rows = torch.full((tiles * s_cap,), -1, dtype=torch.int32, device=device)
        lidx = torch.empty(tiles * km.volume * tile_rows, dtype=torch.int16, device=device)
        kmask = torch.empty(tiles * km.volume, dtype=torch.int32, device=device)
        # both slot orders: by row (no position arrays) and by the source map's Z-order
        smap = km.in_map if target == "out" else km.out_map
        for spos, sord in ((None, None), (smap.zorder_inv(), smap.zorder())):
            _check_plan(lib, _lib, km, tbl, col_order, spos, sord, n_tgt, tile_rows, s_cap, device, target)


def _check_plan(lib, _lib, km, tbl, col_order, spos, sord, n_tgt, tile_rows, s_cap, device, target):
    if True:
        tiles = -(-n_tgt // tile_rows)
        cnt = torch.empty(tiles, dtype=torch.int32, device=device)
        rows = torch.full((tiles * s_cap,), -1, dtype=torch.int32, device=device)
        lidx = torch.empty(tiles * km.volume * tile_rows, dtype=torch.int16, device=device)
        kmask = torch.empty(tiles * km.volume, dtype=torch.int32, device=device)
        _lib.check(lib.me_halo_plan_build(tbl.data_ptr(), col_order.data_ptr() if col_order is not None else None,
                                          spos.data_ptr() if spos is not None else None,
                                          sord.data_ptr() if sord is not None else None, n_tgt,
                                          km.volume, tile_rows, s_cap, cnt.data_ptr(), rows.data_ptr(), lidx.data_ptr(),
                                          kmask.data_ptr(), None))
        torch.cuda.synchronize()
        r_cnt, r_rows, r_lidx, r_kmask = numpy_halo_plan(tbl.cpu().numpy()[:, :n_tgt], None if col_order is None else
                                                          col_order.cpu().numpy(), n_tgt, tile_rows, s_cap,
                                                          None if spos is None else spos.cpu().numpy())
        assert np.array_equal(cnt.cpu().numpy(), r_cnt), target
        assert np.array_equal(rows.cpu().numpy().reshape(tiles, s_cap), r_rows), target
        assert np.array_equal(lidx.cpu().numpy().view(np.uint16).reshape(r_lidx.shape), r_lidx), target
        assert np.array_equal(kmask.cpu().numpy().view(np.uint32).reshape(r_kmask.shape), r_kmask), target
        if s_cap == 40:
            assert (r_cnt > s_cap).any(), "the case is meant to overflow"


HALO_CASES = [
    # n, extent, cin, cout, ks, dil
    (3000, 14, 64, 128, 3, 1),      # config-2 channel shape
    (3000, 40, 64, 128, 3, 1),      # sparse map: most groups skipped
    (3000, 14, 32, 32, 3, 1),       # one column wave, four row waves
    (2500, 14, 32, 96, 3, 1),       # three column waves
    (2500, 14, 96, 32, 3, 1),       # 96-channel chunk (padded stage rows)
    (2500, 14, 96, 96, 3, 1),
    (2000, 12, 192, 128, 3, 1),     # three 64-channel passes over the offsets
    (1500, 10, 256, 256, 3, 1),     # two column slabs, four passes
    (2500, 14, 64, 64, 3, 1),       # 2 x 2 waves
    (2500, 14, 128, 64, 3, 1),
    (2500, 30, 64, 64, 3, 2),       # dilated: large halos
    (2500, 14, 32, 64, 2, 1),       # even kernel (K = 8), same map
    (40, 4, 64, 128, 3, 1),         # a single, partial tile
]


@pytest.mark.parametrize("tile_rows,skip", [(128, 1), (64, 1), (128, 0)], ids=["t128", "t64", "t128-noskip"])
@pytest.mark.parametrize("n,extent,cin,cout,ks,dil", HALO_CASES)
def test_halo_conv_forward_backward_vs_oracle(device, host_layer, halo, n, extent, cin, cout, ks, dil, tile_rows, skip):
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    halo(1, tile_rows, 0, skip)
    assert lib.me_conv_halo_use_bf16(n, ks ** 3, 10 * n, cin, cout) == 1
    coords = make_cloud(n, extent, 3, seed=n + cin, batch=2 if n > 100 else 1, negative=True)
    conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, 1, dil)
    in_c, out_c = coords.numpy(), y.C.cpu().numpy()
    _, km = O.kernel_map(in_c, out_c, O.make_region(3, ks, dil, 1))
    w = conv.kernel.detach().float().cpu().numpy()
    assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, len(out_c)), "forward")
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), gw)


@pytest.mark.parametrize("kc", [32, 128])
def test_halo_conv_channel_chunks_agree(device, halo, kc):
    """the channels staged per pass (32 / 64 / 128) regroup the walk — chunk-major: all offsets of a chunk, then the next
    chunk — so the fp32 sums are re-associated, not changed: the results agree to one bf16 rounding"""
    coords = make_cloud(3000, 14, 3, seed=5)
    halo(1, 128, 64, 1)
    ref = _run_layer(device, coords, 128, 128, 3, seed=7)
    halo(1, 128, kc, 1)
    got = _run_layer(device, coords, 128, 128, 3, seed=7)
    # (two results that are each within one bf16 ulp of the exact sums: up to two ulps apart)
    for what, a, b in (("forward", got[2].F, ref[2].F), ("grad_in", got[1].F.grad, ref[1].F.grad)):
        a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
        tol = 2.0 ** -7 * np.abs(b) + 2e-3 * max(1.0, np.abs(b).max())
        assert not (np.abs(a - b) > tol).any(), (what, float(np.abs(a - b).max()))


def test_halo_conv_overflowing_halo_takes_the_direct_path(device, halo):
    """a map whose tiles touch more source rows than the LDS image holds (dilation 3 on a dense cloud, row-ordered
    target tiles): the kernel's direct-gather path — same sums in the same order, so the same bits as the staged path
    of a geometry that fits"""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(6000, 20, 3, seed=2)
    halo(1, 64, 0, 1)
    conv, x, y, feats, gy = _run_layer(device, coords, 64, 64, 3, 1, 3)
    _, km = O.kernel_map(coords.numpy(), y.C.cpu().numpy(), O.make_region(3, 3, 3, 1))
    w = conv.kernel.detach().float().cpu().numpy()
    assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, len(coords)), "forward")
    gi, _ = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    kmap = x.coordinate_manager._manager
    counts = [v[2] for k, v in _halo_plans(kmap) if v is not None]
    assert counts and any(int((c > 319).sum()) > 0 for c in counts), "the case is meant to overflow the 319-slot image"


def _halo_plans(manager):
    out = []
    kms = getattr(manager, "_kernel_maps", None)
    if kms is None:
        return out
    for km in kms.values():
        for name, v in km._store.items():
            if isinstance(name, str) and "halo" in name:
                out.append((name, v))
    return out


def test_halo_conv_is_bitwise_reproducible_and_host_independent(device, halo):
    import minkowskiengine_amd as ME
    coords = make_cloud(4000, 14, 3, seed=9)
    halo(1, 128, 0, 1)
    prev = ME.get_host()
    res = {}
    try:
        for h in ("python", "native"):
            ME.set_host(h)
            r1 = _run_layer(device, coords, 64, 128, 3)
            r2 = _run_layer(device, coords, 64, 128, 3)
            assert torch.equal(r1[2].F, r2[2].F) and torch.equal(r1[1].F.grad, r2[1].F.grad)
            res[h] = r1
    finally:
        ME.set_host(prev)
    assert torch.equal(res["python"][2].F, res["native"][2].F)
    assert torch.equal(res["python"][1].F.grad, res["native"][1].F.grad)


def test_halo_conv_batch_norm_partials(device, halo):
    """the tile statistics the halo kernel leaves behind give the batch norm the mean / variance of the stored matrix"""
    import minkowskiengine_amd as ME
    halo(1, 128, 0, 1)
    coords = make_cloud(5000, 20, 3, seed=4)
    g = torch.Generator().manual_seed(0)
    feats = bf16_round(torch.rand(coords.shape[0], 64, generator=g) - 0.4)
    for host in ("python", "native"):
        prev = ME.get_host()
        ME.set_host(host)
        try:
            conv = ME.MinkowskiConvolution(64, 96, kernel_size=3, dimension=3).to(device)
            bn = ME.MinkowskiBatchNorm(96).to(device)
            conv.train(), bn.train()
            x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
            y = conv(x)
            z = bn(y)
            yf = y.F.float()
            mean, var = yf.mean(0), yf.var(0, unbiased=False)
            ref = (yf - mean) / torch.sqrt(var + bn.bn.eps) * bn.bn.weight.float() + bn.bn.bias.float()
            err = (z.F.float() - ref).abs().max().item()
            assert err <= 2.0 ** -7 * ref.abs().max().item() + 1e-2, (host, err)
            assert torch.allclose(bn.bn.running_mean.float(), 0.1 * mean, atol=1e-3, rtol=1e-2), host
        finally:
            ME.set_host(prev)


@pytest.mark.parametrize("cin,cout", [(192, 128), (32, 64)])
def test_halo_auto_policy_selects_the_measured_shapes(device, host_layer, cin, cout):
    """ME_AMD_HALO=auto (the default): the shapes the policy lists — 192 -> 128 forward, 256 -> 384 and 64 -> 32 as input
    gradients (here: the backward pass of 32 -> 64) — run on the halo kernel when the map is large and dense enough, every
    other launch of the layer on the tile-plan kernels; results against the oracle as everywhere"""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    assert lib.me_debug_halo_mode() == -1
    assert lib.me_conv_halo_use_bf16(80000, 27, 700000, 192, 128) == 1
    assert lib.me_conv_halo_use_bf16(80000, 27, 700000, 64, 32) == 1 and lib.me_conv_halo_use_bf16(21000, 27, 220000, 256, 384) == 1
    assert lib.me_conv_halo_use_bf16(80000, 27, 700000, 128, 128) == 0      # not listed
    assert lib.me_conv_halo_use_bf16(80000, 27, 200000, 192, 128) == 0      # too sparse
    assert lib.me_conv_halo_use_bf16(5000, 27, 60000, 192, 128) == 0        # too small
    assert lib.me_conv_halo_use_bf16(80000, 8, 80000, 192, 128) == 0        # not a 3^3 kernel
    coords = make_cloud(12000, 24, 3, seed=21)
    conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, 3)
    # the FIRST launch on a kernel-map side runs on the tile-plan kernel: a halo plan costs more than one launch saves
    # (me_conv_halo_min_uses() == 2 under the policy) — a scene that is used once never builds it
    assert lib.me_conv_halo_min_uses() == 2
    assert not [n for n, v in _halo_plans(x.coordinate_manager._manager) if v is not None]
    first = (y.F.detach().clone(), x.F.grad.detach().clone())
    # ... the second launch on the same maps (a reused scene) builds it: exactly one side of the layer takes the halo kernel
    x.F.grad = None
    conv.kernel.grad = None
    y = conv(x)
    y.F.backward(gy.to(device).to(torch.bfloat16))
    plans = [(n, v) for n, v in _halo_plans(x.coordinate_manager._manager) if v is not None]
    if host_layer == "python":
        assert len(plans) == 1, [n for n, _ in plans]
        assert ("halo_out" in plans[0][0]) == (cin == 192)
    # the two schedules agree to bf16 rounding (another summation order)
    for what, a, b in (("forward", y.F, first[0]), ("grad_in", x.F.grad, first[1])):
        a, b = a.detach().double().cpu().numpy(), b.double().cpu().numpy()
        tol = 2.0 ** -7 * np.abs(b) + 2e-3 * max(1.0, np.abs(b).max())
        assert not (np.abs(a - b) > tol).any(), (what, float(np.abs(a - b).max()))
    _, km = O.kernel_map(coords.numpy(), y.C.cpu().numpy(), O.make_region(3, 3, 1, 1))
    w = conv.kernel.detach().float().cpu().numpy()
    assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, len(coords)), "forward")
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), gw)
