"""Shared input generators and comparison helpers for the tests (SURVEY.md §8d inputs)."""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_cloud(n, extent, D=3, seed=0, batch=1, dup=0, negative=False):
    """Unique, unsorted int32 voxels drawn uniformly from [0, extent)^D (optionally centred on 0),
    batch index prepended; `dup` extra duplicated rows shuffled in."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for b in range(batch):
        lo = -(extent // 2) if negative else 0
        pts = torch.randint(lo, lo + extent, (int(1.6 * n) + 16, D), generator=g)
        pts = torch.unique(pts, dim=0)
        pts = pts[torch.randperm(pts.shape[0], generator=g)][:n]
        parts.append(torch.cat([torch.full((pts.shape[0], 1), b, dtype=torch.long), pts], 1))
    c = torch.cat(parts, 0)
    if dup:
        c = torch.cat([c, c[torch.randint(0, c.shape[0], (dup,), generator=g)]], 0)
        c = c[torch.randperm(c.shape[0], generator=g)]
    return c.int().contiguous()


def golden_cases(pattern="ref_*d_k*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, pattern)))


def golden_kmap(z):
    """npz -> {k: int32 [2, n_k]}"""
    out, s = {}, 0
    for k, n in zip(z["kmap_k"].tolist(), z["kmap_n"].tolist()):
        out[int(k)] = z["kmap_pairs"][:, s:s + n]
        s += n
    return out


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def row_mapping(coords_a, coords_b):
    """m with coords_b[m[i]] == coords_a[i] (same coordinate sets, different row order)."""
    from oracle import me_oracle as O
    ca, cb = np.asarray(coords_a), np.asarray(coords_b)
    assert ca.shape == cb.shape, f"coordinate sets differ in size: {ca.shape} vs {cb.shape}"
    ra, rb = O.coordinate_rank(ca), O.coordinate_rank(cb)
    assert np.array_equal(ca[ra], cb[rb]), "coordinate sets differ"
    m = np.empty(len(ra), np.int64)
    m[ra] = rb
    return m
