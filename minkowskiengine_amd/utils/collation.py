"""Input construction helpers (reference: MinkowskiEngine/utils/collation.py:30-188)."""
import numpy as np
import torch


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """List of [N_i, D] coordinates -> [sum N_i, D + 1] with the batch index prepended
    (collation.py:30-93)."""
    assert isinstance(coords, (list, tuple)), "coords must be a sequence of arrays or tensors"
    D = int(coords[0].shape[1])
    n_total = sum(int(c.shape[0]) for c in coords)
    out = torch.zeros((n_total, D + 1), dtype=dtype, device=device)
    s = 0
    for b, c in enumerate(coords):
        if isinstance(c, np.ndarray):
            c = torch.from_numpy(c)
        assert c.shape[1] == D, "all coordinate sets must have the same dimension"
        if c.dtype in (torch.float32, torch.float64):
            c = torch.floor(c)
        n = int(c.shape[0])
        out[s:s + n, 1:] = c.to(dtype=dtype, device=out.device)
        out[s:s + n, 0] = b
        s += n
    return out


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """(coords list, feats list[, labels list]) -> batched tensors (collation.py:96-188)."""
    bcoords = batched_coordinates(coords, dtype=dtype, device=device)
    tofeat = [torch.from_numpy(f) if isinstance(f, np.ndarray) else f for f in feats]
    bfeats = torch.cat(tofeat, 0)
    if device is not None:
        bfeats = bfeats.to(device)
    if labels is None:
        return bcoords, bfeats
    tolab = [torch.from_numpy(l) if isinstance(l, np.ndarray) else l for l in labels]
    blabels = torch.cat(tolab, 0)
    if device is not None:
        blabels = blabels.to(device)
    return bcoords, bfeats, blabels


def batch_sparse_collate(data, dtype=torch.int32, device=None):
    """collate_fn for torch.utils.data.DataLoader: list of (coords, feats[, labels]) tuples
    (collation.py:191-206)."""
    return sparse_collate(*list(zip(*data)), dtype=dtype, device=device)


class SparseCollation:
    """Callable collate_fn that caps the batch at `limit_numpoints` points and optionally returns the lists
    un-concatenated (collation.py:209-288)."""

    def __init__(self, limit_numpoints=-1, dtype=torch.int32, device=None):
        self.limit_numpoints = limit_numpoints
        self.dtype = dtype
        self.device = device

    def __call__(self, list_data):
        coords, feats, labels = list(zip(*list_data))
        keep_c, keep_f, keep_l = [], [], []
        total = 0
        for batch_id, _ in enumerate(coords):
            n = coords[batch_id].shape[0]
            total += n
            if 0 < self.limit_numpoints < total:
                import logging
                logging.warning("Cannot fit %d points into %d points limit. Truncating batch size at %d out of %d.",
                                sum(len(c) for c in coords), self.limit_numpoints, batch_id, len(coords))
                break
            keep_c.append(coords[batch_id])
            keep_f.append(feats[batch_id])
            keep_l.append(labels[batch_id])
        return sparse_collate(keep_c, keep_f, keep_l, dtype=self.dtype, device=self.device)
