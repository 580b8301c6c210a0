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
    """max |a - b| normalised by the largest reference magnitude (a coarse, GLOBAL figure: used only for
    oracle-vs-reference sanity checks; the GPU parity tests use assert_close, which is per element)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def close_excess(a, b, atol=1e-4, rtol=1e-4):
    """max over elements of |a - b| / (atol + rtol * |b|): <= 1 means every element is within tolerance."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"shape mismatch {a.shape} vs {b.shape}"
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) / (atol + rtol * np.abs(b))).max())


def assert_close(a, b, atol=1e-4, rtol=1e-4, what=""):
    """PER-ELEMENT parity check |a - b| <= atol + rtol * |b| — the fp32 bar of BASELINE.json's north_star
    (1e-4 absolute + relative), not a max-normalised figure."""
    if hasattr(a, "detach"):
        a = a.detach().float().cpu().numpy()
    if hasattr(b, "detach"):
        b = b.detach().float().cpu().numpy()
    ex = close_excess(a, b, atol, rtol)
    if not ex <= 1.0:
        a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
        i = np.unravel_index(np.argmax(np.abs(a64 - b64) / (atol + rtol * np.abs(b64))), a64.shape)
        raise AssertionError(f"{what} element {i}: got {a64[i]!r} want {b64[i]!r} "
                             f"(|diff| {abs(a64[i] - b64[i]):.3e} = {ex:.2f} x tolerance {atol}+{rtol}|b|)")


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
