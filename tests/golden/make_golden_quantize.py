"""Golden fixture for the input-pipeline row: the REFERENCE's own MinkowskiEngine.utils.sparse_quantize
(Python package on its CPU extension compiled unmodified into oracle/_ref/_C.so) and
SparseTensor(quantization_mode=UNWEIGHTED_AVERAGE) on a small synthetic point cloud.

Run in the authoring container (needs /root/reference):  python tests/golden/make_golden_quantize.py
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
from oracle import ref  # noqa: E402

C = ref.load()
pkg = types.ModuleType("MinkowskiEngineBackend")
pkg._C = C
pkg.__path__ = []
sys.modules["MinkowskiEngineBackend"] = pkg
sys.modules["MinkowskiEngineBackend._C"] = C
sys.path.insert(0, os.environ.get("ME_REFERENCE_ROOT", "/root/reference"))
os.chdir(tempfile.mkdtemp())
import MinkowskiEngine as RME  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator().manual_seed(7)
    n = 6000
    pts = (torch.rand(n, 3, generator=g) - 0.5) * 2.0                  # points in [-1, 1)^3 (negative coordinates)
    feats = torch.rand(n, 5, generator=g)
    # labels: voxels in the half space x < 0 are pure (label from the voxel index), the others mixed
    q = 0.08
    vox = torch.floor(pts / q).int()
    labels = torch.where(pts[:, 0] < 0, (vox[:, 1] % 7 + 7) % 7, torch.randint(0, 7, (n,), generator=g).int()).int()
    # --- no labels: coordinates, features, index and inverse maps
    c, f, idx, inv = RME.utils.sparse_quantize(pts, feats, quantization_size=q, return_index=True,
                                               return_inverse=True)
    # --- labels.  The reference's quantize_label (src/quantization.cpp:189) writes the ignore label to
    # colabels[inverse_mapping[u]] instead of colabels[u]; that is only the same voxel when the first
    # n_unique input rows are all distinct, so the label fixture orders the points "first occurrences first".
    order = torch.cat([idx, torch.tensor(sorted(set(range(n)) - set(idx.tolist())), dtype=torch.long)])
    pts2, labels2 = pts[order], labels[order]
    c2, l2, idx2, inv2 = RME.utils.sparse_quantize(pts2, labels=labels2, ignore_label=-100, quantization_size=q,
                                                   return_index=True, return_inverse=True)
    assert torch.equal(idx2, torch.arange(idx2.numel()))
    # --- voxel-averaged features through the reference SparseTensor
    bc = RME.utils.batched_coordinates([torch.floor(pts / q)])
    st = RME.SparseTensor(feats, bc, quantization_mode=RME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    out = os.path.join(HERE, "quantize_3d_6k.npz")
    np.savez_compressed(out, points=pts.numpy(), feats=feats.numpy(), labels=labels.numpy(), quantization_size=q,
                        q_coords=c.numpy(), q_feats=f.numpy(), q_index=idx.numpy(), q_inverse=inv.numpy(),
                        lab_points=pts2.numpy(), lab_labels=labels2.numpy(), lab_coords=c2.numpy(),
                        lab_colabels=l2.numpy(), lab_index=idx2.numpy(), lab_inverse=inv2.numpy(),
                        avg_coords=st.C.numpy(), avg_feats=st.F.numpy())
    print(out, os.path.getsize(out), "voxels", c.shape[0], "ignored", int((l2 == -100).sum()))
