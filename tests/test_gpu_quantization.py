"""Input pipeline on the device: sparse_quantize / quantize_label / voxel-averaged
SparseTensor features against the fixture produced by the reference's own `sparse_quantize` and against
the oracle.  Index maps and labels bit-exact; averaged features within 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import GOLDEN_DIR

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]   # both host layers


def test_sparse_quantize_matches_reference_fixture(device):
    import minkowskiengine_amd as ME
    z = np.load(os.path.join(GOLDEN_DIR, "quantize_3d_6k.npz"))
    q = float(z["quantization_size"])
    pts, feats = torch.from_numpy(z["points"]), torch.from_numpy(z["feats"])
    # torch on the CPU in -> torch on the CPU out, hashed on the GPU
    c, f, idx, inv = ME.utils.sparse_quantize(pts, feats, quantization_size=q, return_index=True, return_inverse=True)
    assert not c.is_cuda and c.dtype == torch.int32
    assert np.array_equal(c.numpy(), z["q_coords"]) and np.array_equal(f.numpy(), z["q_feats"])
    assert np.array_equal(idx.numpy(), z["q_index"]) and np.array_equal(inv.numpy(), z["q_inverse"])
    # numpy in -> numpy out; GPU tensors in -> GPU tensors out
    cn = ME.utils.sparse_quantize(z["points"], quantization_size=q)
    assert isinstance(cn, np.ndarray) and np.array_equal(cn, z["q_coords"])
    cg, ig = ME.utils.sparse_quantize(pts.to(device), quantization_size=[q, q, q], return_index=True)
    assert cg.is_cuda and np.array_equal(cg.cpu().numpy(), z["q_coords"]) and np.array_equal(ig.cpu().numpy(), z["q_index"])
    um = ME.utils.sparse_quantize(pts, quantization_size=q, return_index=True, return_maps_only=True)
    assert np.array_equal(um.numpy(), z["q_index"])
    # labels (fixture ordered so that the reference's misplaced ignore write cannot trigger)
    c2, l2, idx2, inv2 = ME.utils.sparse_quantize(torch.from_numpy(z["lab_points"]),
                                                  labels=torch.from_numpy(z["lab_labels"]), ignore_label=-100,
                                                  quantization_size=q, return_index=True, return_inverse=True)
    assert np.array_equal(c2.numpy(), z["lab_coords"]) and np.array_equal(l2.numpy(), z["lab_colabels"])
    assert np.array_equal(idx2.numpy(), z["lab_index"]) and np.array_equal(inv2.numpy(), z["lab_inverse"])
    # labels in arbitrary order: the oracle (intended semantics)
    vox = np.floor(z["points"] / np.float32(q)).astype(np.int32)
    o_um, o_inv, o_col = O.quantize_label(vox, z["labels"], -100)
    um3, inv3, col3 = ME.utils.quantize_label(vox, z["labels"], -100)
    assert np.array_equal(um3, o_um) and np.array_equal(inv3, o_inv) and np.array_equal(col3, o_col)


def test_voxel_average_features(device):
    import minkowskiengine_amd as ME
    z = np.load(os.path.join(GOLDEN_DIR, "quantize_3d_6k.npz"))
    q = float(z["quantization_size"])
    pts, feats = torch.from_numpy(z["points"]), torch.from_numpy(z["feats"])
    bc = ME.utils.batched_coordinates([torch.floor(pts / q)]).to(device)
    st = ME.SparseTensor(feats.to(device), bc, quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    assert np.array_equal(st.C.cpu().numpy(), z["avg_coords"])
    assert np.abs(st.F.cpu().numpy() - z["avg_feats"]).max() < 1e-6
    st2 = ME.SparseTensor(feats.to(device), bc, quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    assert torch.equal(st.F, st2.F)                                          # fixed summation order
    ss = ME.SparseTensor(feats.to(device), bc, quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_SUM)
    ref = O.segment_mean(z["feats"], z["q_inverse"], len(z["q_index"]), average=False)
    assert np.abs(ss.F.cpu().numpy() - ref).max() < 1e-5
    # odd channel count (scalar path) and a large cloud with many duplicates per voxel
    g = torch.Generator().manual_seed(0)
    n = 300000
    c = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.randint(0, 24, (n, 3), generator=g).int()], 1)
    f = torch.rand(n, 3, generator=g)
    big = ME.SparseTensor(f.to(device), c.to(device), quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    um, inv = O.insert_and_map(c.numpy())
    refm = O.segment_mean(f.numpy(), inv, len(um))
    assert big.F.shape[0] == len(um) and np.abs(big.F.cpu().numpy() - refm).max() < 1e-5
