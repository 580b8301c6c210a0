"""Pin the oracle (oracle/me_oracle.{c,py}) against the reference's OWN golden vectors
(/root/reference/tests/cpp/kernel_region_cpu_test.py:12-116, coordinate_map_cpu_test.py:12-125,
tests/python/coordinate_manager.py:183-200, tests/python/convolution.py:226-245) — re-typed here —
and against the fixtures generated from the compiled reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import me_oracle as O
from helpers import golden_cases, golden_kmap, rel_err, row_mapping


def test_region_odd():  # kernel_region_cpu_test.py:22-43
    reg = O.region_coordinates(np.array([[0, 1, -1], [0, 2, 1]]), O.make_region(2, [3, 3]))
    assert len(reg) == 2 * 9
    assert reg[:9].tolist() == [[0, 0, -2], [0, 1, -2], [0, 2, -2], [0, 0, -1], [0, 1, -1], [0, 2, -1],
                                [0, 0, 0], [0, 1, 0], [0, 2, 0]]


def test_region_even():  # kernel_region_cpu_test.py:45-61
    reg = O.region_coordinates(np.array([[0, 1, -1], [0, 2, 1]]), O.make_region(2, [3, 2]))
    assert len(reg) == 2 * 6
    assert reg[:6].tolist() == [[0, 0, -1], [0, 1, -1], [0, 2, -1], [0, 0, 0], [0, 1, 0], [0, 2, 0]]


def test_region_even3():  # kernel_region_cpu_test.py:63-86
    reg = O.region_coordinates(np.array([[0, 1, -1, 3], [0, 2, 1, -2]]), O.make_region(3, [3, 2, 2]))
    assert len(reg) == 2 * 12
    assert reg[:12].tolist() == [[0, 0, -1, 3], [0, 1, -1, 3], [0, 2, -1, 3], [0, 0, 0, 3], [0, 1, 0, 3],
                                 [0, 2, 0, 3], [0, 0, -1, 4], [0, 1, -1, 4], [0, 2, -1, 4], [0, 0, 0, 4],
                                 [0, 1, 0, 4], [0, 2, 0, 4]]


def test_kernel_map_volume1():  # kernel_region_cpu_test.py:88-98
    _, km = O.kernel_map(np.array([[0, 1, -1], [0, 2, 1]]), np.array([[0, 1, -1], [0, 2, 1], [1, 2, 1]]),
                         O.make_region(2, [1, 1]))
    assert km[0][0].tolist() == [0, 1] and km[0][1].tolist() == [0, 1]


def test_kernel_map_two_points():  # kernel_region_cpu_test.py:100-116
    _, km = O.kernel_map(np.array([[0, 1, -1], [0, 2, 1]]), np.array([[0, 1, 0], [0, 1, 2], [1, 2, 1]]),
                         O.make_region(2, [3, 3]))
    assert km[1][0].tolist() == [0] and km[1][1].tolist() == [0]
    assert km[2][0].tolist() == [1] and km[2][1].tolist() == [1]


def test_insert_dedup():  # coordinate_map_cpu_test.py:12-27
    coords = np.array([[0, 1], [1, 2], [2, 3], [2, 3]], np.int32)
    um, inv = O.insert_and_map(coords)
    assert len(um) == 3
    assert np.array_equal(coords, coords[um][inv])


def test_find():  # coordinate_map_cpu_test.py:47-65
    rows = O.find(np.array([[0, 1], [1, 2], [2, 3], [2, 3]]), np.array([[-1, 1], [1, 2], [2, 3], [2, 3], [0, 0]]))
    valid = np.nonzero(rows >= 0)[0]
    assert valid.tolist() == [1, 2, 3]
    # the reference map holds the 3 unique rows (value = unique index); row ids of the raw list differ
    um, inv = O.insert_and_map(np.array([[0, 1], [1, 2], [2, 3], [2, 3]], np.int32))
    assert [int(inv[r]) for r in rows[valid]] == [1, 2, 2]


def test_stride_sizes():  # coordinate_map_cpu_test.py:67-125
    c, _ = O.stride_map(np.array([[0, 1], [0, 2], [0, 3], [0, 3]]), [2])
    assert len(c) == 2
    coords = np.array([[0, 1, 1], [0, 2, 1], [0, 1, 0], [1, 0, 3], [1, 0, 2]])
    assert len(O.stride_map(coords, [1, 1])[0]) == 5
    assert len(O.stride_map(coords, [2, 1])[0]) == 5
    assert len(O.stride_map(coords, [4, 4])[0]) == 2
    assert len(O.stride_map(np.array([[0, -1], [0, -2], [0, 1], [0, 0]]), [2])[0]) == 2


def test_negative_coords_floor():  # tests/python/coordinate_manager.py:183-200
    coords = np.array([[0, -3], [0, -2], [0, -1], [0, 0], [0, 1], [0, 2], [0, 3]])
    c, _ = O.stride_map(coords, [2])
    assert sorted(c[:, 1].tolist()) == [-4, -2, 0, 2]


def test_analytic_1d_conv():  # tests/python/convolution.py:226-245 (value: compiled reference)
    z = np.load(golden_cases("ref_analytic1d.npz")[0])
    _, km = O.kernel_map(z["coords"], z["coords"], O.make_region(1, 2))
    out = O.conv_forward(z["feats"], z["kernel"], km, 3)
    assert out.tolist() == [[2, 2], [2, 3], [3, 3]]
    assert np.array_equal(out, z["out"])


def test_reference_fixture_26_pairs():  # data_loader fixture, conv k=3 s=2
    z = np.load(golden_cases("ref_fixture2d_k3s2.npz")[0])
    assert int(z["kmap_n"].sum()) == 26 and len(z["out_coords"]) == 10


@pytest.mark.parametrize("path", golden_cases())
def test_oracle_matches_reference_fixture(path):
    """Oracle vs the reference's outputs stored in tests/golden: unique/inverse maps bit-exact,
    output coordinate SET identical, kernel-map pair sets identical after relabelling the
    (implementation-ordered) output rows, features within fp32 round-off of the reference's MKL."""
    z = np.load(path)
    D = z["coords"].shape[1] - 1
    um, inv = O.insert_and_map(z["coords"])
    assert np.array_equal(um, z["unique_map"]) and np.array_equal(inv, z["inverse_map"])
    in_coords = z["coords"][um]
    assert np.array_equal(in_coords, z["in_coords"])
    ks, st, dl = z["kernel_size"].tolist(), z["stride"].tolist(), z["dilation"].tolist()
    if any(s > 1 for s in st):
        out_coords, _ = O.stride_map(in_coords, z["out_tensor_stride"].tolist())
    else:
        out_coords = in_coords
    m = row_mapping(out_coords, z["out_coords"])          # our out row -> reference out row
    _, km = O.kernel_map(in_coords, out_coords, O.make_region(D, ks, dl, 1))
    km_ref_rows = {k: np.stack((v[0].astype(np.int64), m[v[1]])) for k, v in km.items()}
    O.assert_same_kernel_map(km_ref_rows, golden_kmap(z))
    out = O.conv_forward(z["feats"], z["kernel"], km, len(out_coords))
    assert rel_err(out, z["out"][m]) < 2e-6
    gi, gk = O.conv_backward(z["feats"], z["grad_out"][m], z["kernel"], km)
    assert rel_err(gi, z["grad_in"]) < 2e-6 and rel_err(gk, z["grad_kernel"]) < 2e-6
    if "up" in z.files:  # transposed conv back to the fine map = swapped forward map
        km_t = {k: np.stack((v[1], v[0])) for k, v in km.items()}
        up = O.conv_forward(z["out"][m], z["kernel_t"], km_t, len(in_coords))
        assert rel_err(up, z["up"]) < 2e-6
        tgi, tgk = O.conv_backward(z["out"][m], z["up_grad_out"], z["kernel_t"], km_t)
        assert rel_err(tgi, z["up_grad_in"][m]) < 2e-6 and rel_err(tgk, z["up_grad_kernel"]) < 2e-6
