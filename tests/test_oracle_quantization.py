"""Input-pipeline row on the CPU: the oracle restatement of quantize_label / voxel averaging against the
compiled reference (oracle/_ref) and against the fixture produced by the reference's own Python
`sparse_quantize` (tests/golden/make_golden_quantize.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from oracle import ref
from helpers import GOLDEN_DIR


def _fixture():
    return np.load(os.path.join(GOLDEN_DIR, "quantize_3d_6k.npz"))


def test_oracle_quantize_matches_reference_fixture():
    z = _fixture()
    q = float(z["quantization_size"])
    vox = np.floor(z["points"] / np.float32(q)).astype(np.int32)
    um, inv = O.insert_and_map(vox)
    um2, inv2, col = O.quantize_label(vox, z["labels"], -100)
    assert np.array_equal(um, um2) and np.array_equal(inv, inv2)
    assert np.array_equal(um2, z["q_index"]) and np.array_equal(inv2, z["q_inverse"])
    assert np.array_equal(vox[um2], z["q_coords"]) and np.array_equal(z["feats"][um2], z["q_feats"])
    # labels: fixture ordered so that the reference's misplaced ignore write cannot trigger
    vox_l = np.floor(z["lab_points"] / np.float32(q)).astype(np.int32)
    um3, inv3, col3 = O.quantize_label(vox_l, z["lab_labels"], -100)
    assert np.array_equal(um3, z["lab_index"]) and np.array_equal(inv3, z["lab_inverse"])
    assert np.array_equal(col3, z["lab_colabels"])
    assert (col3 == -100).sum() > 50 and (col3 != -100).sum() > 1000
    # voxel-averaged features of the reference SparseTensor (rows in first-occurrence order)
    mean = O.segment_mean(z["feats"], inv2, len(um2))
    bc = np.concatenate([np.zeros((len(um2), 1), np.int32), vox[um2]], 1)
    assert np.array_equal(bc, z["avg_coords"])
    assert np.abs(mean - z["avg_feats"]).max() < 1e-5


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/_C.so not built")
def test_oracle_quantize_label_vs_compiled_reference():
    C = ref.load()
    g = torch.Generator().manual_seed(3)
    uniq = torch.unique(torch.randint(-6, 6, (900, 3), generator=g), dim=0)
    uniq = uniq[torch.randperm(uniq.shape[0], generator=g)].int()
    dups = uniq[torch.randint(0, uniq.shape[0], (1500,), generator=g)]
    coords = torch.cat([uniq, dups]).contiguous()             # first occurrences first (see the fixture script)
    labels = torch.randint(0, 3, (coords.shape[0],), generator=g).int()
    um, inv, col = C.quantize_label_th(coords, labels, -1)
    o_um, o_inv, o_col = O.quantize_label(coords.numpy(), labels.numpy(), -1)
    assert np.array_equal(um.numpy(), o_um) and np.array_equal(inv.numpy(), o_inv)
    assert np.array_equal(col.numpy(), o_col)
    # the divergence, pinned: with duplicates BEFORE a voxel's row the reference marks another voxel
    coords2 = torch.tensor([[0, 0, 0], [0, 0, 0], [1, 0, 0], [1, 0, 0]], dtype=torch.int32)
    labels2 = torch.tensor([5, 5, 1, 2], dtype=torch.int32)
    _, _, col_ref = C.quantize_label_th(coords2, labels2, -1)
    _, _, col_orc = O.quantize_label(coords2.numpy(), labels2.numpy(), -1)
    assert col_orc.tolist() == [5, -1]                        # voxel 1 has labels {1, 2}
    assert col_ref.tolist() == [-1, 1]                        # reference: colabels[inverse_mapping[1]] = colabels[0]
