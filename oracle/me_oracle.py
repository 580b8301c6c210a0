"""TEST INFRASTRUCTURE ONLY — numpy/C oracle of the reference algorithms on the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package minkowskiengine_amd never does (tests/test_no_oracle_in_product.py enforces it).

Integer work (coordinate dedup, striding, kernel offsets, kernel maps) is restated in plain C
(oracle/me_oracle.c, compiled with gcc by build()); feature arithmetic is restated with numpy
matmul following the reference loops (src/convolution_kernel.hpp:33-144).  Pinning: see the header
of me_oracle.c.  Floating-point values are pinned against the compiled reference itself
(oracle/_ref) — the reference's own tests hold no value-level vectors for the BLAS boundary
(SURVEY.md §8c).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(HERE, "me_oracle.c")
_SO = os.path.join(HERE, "_build", "libme_oracle.so")
ORC_MAX_DIM = 7
HYPER_CUBE, HYPER_CROSS = 0, 1


class OrcRegion(ctypes.Structure):
    _fields_ = [("ncol", ctypes.c_int32), ("region_type", ctypes.c_int32),
                ("kernel_size", ctypes.c_int32 * ORC_MAX_DIM), ("dilation", ctypes.c_int32 * ORC_MAX_DIM),
                ("tensor_stride", ctypes.c_int32 * ORC_MAX_DIM)]


def build(force=False):
    """gcc -O2 -shared me_oracle.c -> oracle/_build/libme_oracle.so"""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", _SRC, "-o", _SO, "-lm"])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        rgp = ctypes.POINTER(OrcRegion)
        lib.orc_insert_and_map.restype = ctypes.c_int64
        lib.orc_insert_and_map.argtypes = [i32p, ctypes.c_int64, ctypes.c_int32, i64p, i64p]
        lib.orc_stride_coordinates.restype = None
        lib.orc_stride_coordinates.argtypes = [i32p, ctypes.c_int64, ctypes.c_int32, i32p, i32p]
        lib.orc_region_volume.restype = ctypes.c_int64
        lib.orc_region_volume.argtypes = [rgp]
        lib.orc_region_coordinates.restype = None
        lib.orc_region_coordinates.argtypes = [rgp, i32p, ctypes.c_int64, i32p]
        lib.orc_kernel_map.restype = ctypes.c_int64
        lib.orc_kernel_map.argtypes = [i32p, ctypes.c_int64, i32p, ctypes.c_int64, rgp, i32p, i64p]
        lib.orc_find.restype = ctypes.c_int
        lib.orc_find.argtypes = [i32p, ctypes.c_int64, ctypes.c_int32, i32p, ctypes.c_int64, i32p]
        _lib = lib
    return _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _as_list(v, d):
    if np.isscalar(v):
        return [int(v)] * d
    v = [int(x) for x in v]
    assert len(v) == d
    return v


def make_region(D, kernel_size, dilation=1, tensor_stride=1, region_type=HYPER_CUBE):
    rg = OrcRegion()
    rg.ncol = D + 1
    rg.region_type = int(region_type)
    ks, dl, ts = _as_list(kernel_size, D), _as_list(dilation, D), _as_list(tensor_stride, D)
    for d in range(ORC_MAX_DIM):
        rg.kernel_size[d] = ks[d] if d < D else 1
        rg.dilation[d] = dl[d] if d < D else 1
        rg.tensor_stride[d] = ts[d] if d < D else 1
    return rg


# ---- coordinate maps --------------------------------------------------------------------------
def insert_and_map(coords):
    """-> (unique_map int64 [n_unique], inverse_map int64 [n]); first occurrence wins
    (src/coordinate_map_cpu.hpp:353-380)."""
    coords = _i32(coords)
    n, ncol = coords.shape
    um = np.zeros(max(n, 1), np.int64)
    inv = np.zeros(max(n, 1), np.int64)
    nu = _load().orc_insert_and_map(coords, n, ncol, um, inv)
    assert nu >= 0
    return um[:nu].copy(), inv[:n].copy()


def stride_coordinates(coords, out_tensor_stride):
    """floor((float)c / ts) * ts per spatial axis (src/coordinate_map.hpp:58-66); not deduplicated."""
    coords = _i32(coords)
    n, ncol = coords.shape
    ts = _i32(_as_list(out_tensor_stride, ncol - 1))
    out = np.zeros_like(coords)
    _load().orc_stride_coordinates(coords, n, ncol, ts, out)
    return out


def stride_map(coords, out_tensor_stride):
    """CoordinateMapCPU::stride (src/coordinate_map_cpu.hpp:418-437) -> unique strided coordinates in
    first-occurrence order of the INPUT rows (the reference iterates its hash table, so its row order
    is arbitrary: compare after relabelling by coordinate) and the in-row -> out-row map."""
    s = stride_coordinates(coords, out_tensor_stride)
    um, inv = insert_and_map(s)
    return s[um], inv


def region_volume(region):
    return int(_load().orc_region_volume(ctypes.byref(region)))


def region_coordinates(coords, region):
    """[n * volume, D+1]: every neighbour coordinate of every point, offsets fastest
    (src/kernel_region.hpp:198-247)."""
    coords = _i32(coords)
    n = coords.shape[0]
    out = np.zeros((n * region_volume(region), coords.shape[1]), np.int32)
    _load().orc_region_coordinates(ctypes.byref(region), coords, n, out)
    return out


def kernel_map(in_coords, out_coords, region):
    """-> (nbr int32 [volume, n_out], {k: int32 [2, n_k]} non-empty offsets only), pairs sorted by
    output row (src/coordinate_map_cpu.hpp:569-670; dict format
    src/coordinate_map_manager.cpp:1358-1387)."""
    in_coords, out_coords = _i32(in_coords), _i32(out_coords)
    vol = region_volume(region)
    n_out = out_coords.shape[0]
    nbr = np.full((vol, max(n_out, 1)), -1, np.int32)
    counts = np.zeros(vol, np.int64)
    total = _load().orc_kernel_map(in_coords, in_coords.shape[0], out_coords, n_out, ctypes.byref(region), nbr,
                                   counts)
    assert total >= 0
    nbr = nbr[:, :n_out]
    maps = {}
    for k in range(vol):
        outs = np.nonzero(nbr[k] >= 0)[0].astype(np.int32)
        if outs.size:
            maps[k] = np.stack((nbr[k, outs], outs))
    return nbr, maps


def find(map_coords, queries):
    map_coords, queries = _i32(map_coords), _i32(queries)
    rows = np.zeros(max(queries.shape[0], 1), np.int32)
    rc = _load().orc_find(map_coords, map_coords.shape[0], map_coords.shape[1], queries, queries.shape[0], rows)
    assert rc == 0
    return rows[:queries.shape[0]]


# ---- feature arithmetic (src/convolution_kernel.hpp:33-144) ------------------------------------
def conv_forward(in_feat, kernel, kmap, n_out, dtype=np.float64):
    """out[out_k] += in[in_k] @ W_k for every non-empty offset k (convolution_kernel.hpp:50-78)."""
    in_feat = np.asarray(in_feat, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    out = np.zeros((n_out, kernel.shape[2]), dtype)
    for k, io in kmap.items():
        buf = in_feat[io[0]] @ kernel[k]           # gather + gemm
        np.add.at(out, io[1], buf)                 # scatter-add (rows unique within k)
    return out


def conv_backward(in_feat, grad_out, kernel, kmap, dtype=np.float64):
    """-> (grad_in, grad_kernel) (convolution_kernel.hpp:98-142)."""
    in_feat = np.asarray(in_feat, dtype=dtype)
    grad_out = np.asarray(grad_out, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    grad_in = np.zeros_like(in_feat)
    grad_kernel = np.zeros_like(kernel)
    for k, io in kmap.items():
        g = grad_out[io[1]]
        np.add.at(grad_in, io[0], g @ kernel[k].T)
        grad_kernel[k] += in_feat[io[0]].T @ g
    return grad_in, grad_kernel


# ---- pooling / broadcast (numpy restatement; float32 by default to follow the reference's arithmetic) --------
def pool_forward(in_feat, kmap, n_out, mode, dtype=np.float32):
    """mode "sum" | "avg" | "max" -> (out, num_nonzero | max_index).
    NonzeroAvgPoolingForwardKernelCPU (pooling_avg_kernel.hpp:41-108) / MaxPoolingForwardKernelCPU
    (pooling_max_kernel.hpp:36-96): offsets in ascending k, rows in list order."""
    x = np.asarray(in_feat, dtype=dtype)
    c = x.shape[1]
    if mode == "max":
        out = np.full((n_out, c), -np.finfo(dtype).max, dtype)
        idx = np.full((n_out, c), -1, np.int32)
        for k in sorted(kmap):
            for i, o in zip(*[np.asarray(a) for a in kmap[k]]):
                better = out[o] < x[i]
                out[o] = np.where(better, x[i], out[o])
                idx[o] = np.where(better, i * c + np.arange(c), idx[o])
        return out, idx
    out = np.zeros((n_out, c), dtype)
    cnt = np.zeros(n_out, dtype)
    for k in sorted(kmap):
        for i, o in zip(*[np.asarray(a) for a in kmap[k]]):
            out[o] += x[i]
            cnt[o] += 1
    if mode == "avg":
        nz = cnt > 0
        out[nz] = out[nz] / cnt[nz, None]
    return out, cnt


def pool_backward(grad_out, kmap, n_in, mode, aux, dtype=np.float32):
    """aux = num_nonzero (avg) / max_index (max).  pooling_avg_kernel.hpp:110-150,
    pooling_max_kernel.hpp:98-117."""
    g = np.asarray(grad_out, dtype=dtype)
    c = g.shape[1]
    grad_in = np.zeros((n_in, c), dtype)
    if mode == "max":
        flat = grad_in.reshape(-1)
        m = np.asarray(aux).reshape(-1)
        np.add.at(flat, m[m >= 0], g.reshape(-1)[m >= 0])
        return grad_in
    for k in sorted(kmap):
        for i, o in zip(*[np.asarray(a) for a in kmap[k]]):
            if mode == "avg":
                if aux[o] > 0:
                    grad_in[i] += g[o] / dtype(aux[o])
            else:
                grad_in[i] += g[o]
    return grad_in


def global_pool_forward(in_feat, batch_rows, n_batch, mode, dtype=np.float32):
    """Rows of every batch index reduced in row order (global_pooling_cpu.cpp:43-238)."""
    x = np.asarray(in_feat, dtype=dtype)
    rows = np.asarray(batch_rows)
    c = x.shape[1]
    if mode == "max":
        out = np.full((n_batch, c), -np.finfo(dtype).max, dtype)
        idx = np.full((n_batch, c), -1, np.int32)
        for i, b in enumerate(rows):
            better = out[b] < x[i]
            out[b] = np.where(better, x[i], out[b])
            idx[b] = np.where(better, i * c + np.arange(c), idx[b])
        return out, idx
    out = np.zeros((n_batch, c), np.float64)
    np.add.at(out, rows, x.astype(np.float64))
    cnt = np.bincount(rows, minlength=n_batch).astype(dtype)
    if mode == "avg":
        out = out / np.maximum(cnt, 1)[:, None]
    return out.astype(dtype), cnt


def broadcast_forward(in_feat, glob, batch_rows, multiply):
    x, g = np.asarray(in_feat), np.asarray(glob)[np.asarray(batch_rows)]
    return x * g if multiply else x + g


# ---- canonicalisers (how "bit-exact index maps" is defined) ----------------------
def pairs_by_offset(kmap):
    """{k: [2, n_k] array-like} -> {k: sorted int64 [n_k, 2] of (in, out)}; empty offsets dropped."""
    out = {}
    for k, io in kmap.items():
        a = np.asarray(io if not hasattr(io, "cpu") else io.cpu().numpy()).astype(np.int64)
        if a.size == 0:
            continue
        p = a.T
        out[int(k)] = p[np.lexsort((p[:, 1], p[:, 0]))]
    return out


def assert_same_kernel_map(a, b):
    pa, pb = pairs_by_offset(a), pairs_by_offset(b)
    assert sorted(pa) == sorted(pb), f"non-empty offsets differ: {sorted(pa)} vs {sorted(pb)}"
    for k in pa:
        assert pa[k].shape == pb[k].shape and np.array_equal(pa[k], pb[k]), f"pair set of offset {k} differs"


def coordinate_rank(coords):
    """Row permutation that sorts coordinates lexicographically (for relabelling maps whose row order
    is implementation-defined)."""
    c = np.asarray(coords)
    return np.lexsort(tuple(c[:, i] for i in range(c.shape[1] - 1, -1, -1)))


def relabel_kernel_map(kmap, in_coords_a, in_coords_b, out_coords_a, out_coords_b):
    """Express kernel map `kmap` (rows of coordinate lists *_a) in the row numbering of lists *_b,
    which hold the same coordinate sets in a different order."""
    def mapping(ca, cb):
        ra, rb = coordinate_rank(ca), coordinate_rank(cb)
        assert np.array_equal(np.asarray(ca)[ra], np.asarray(cb)[rb]), "coordinate sets differ"
        m = np.empty(len(ra), np.int64)
        m[ra] = rb
        return m
    mi, mo = mapping(in_coords_a, in_coords_b), mapping(out_coords_a, out_coords_b)
    out = {}
    for k, io in kmap.items():
        a = np.asarray(io if not hasattr(io, "cpu") else io.cpu().numpy()).astype(np.int64)
        out[k] = np.stack((mi[a[0]], mo[a[1]]))
    return out


# ---- voxelisation (input pipeline) ------------------------------------------------------------------
def quantize_label(coords, labels, ignore_label):
    """-> (unique_map, inverse_map, colabels): first-occurrence dedup; a voxel keeps its first point's label
    unless a later point of the voxel disagrees -> ignore_label (src/quantization.cpp:140-196, with the
    ignore label written to colabels[u]; the reference's line 189 indexes colabels[inverse_mapping[u]])."""
    coords = _i32(coords)
    labels = np.asarray(labels, np.int32)
    unique_map, inverse_map = [], np.empty(len(coords), np.int64)
    colabels, seen = [], {}
    for row in range(len(coords)):
        key = coords[row].tobytes()
        u = seen.get(key)
        if u is None:
            seen[key] = len(unique_map)
            inverse_map[row] = len(unique_map)
            unique_map.append(row)
            colabels.append(int(labels[row]))
        else:
            if colabels[u] != labels[row] and colabels[u] != ignore_label:
                colabels[u] = ignore_label
            inverse_map[row] = u
    return np.asarray(unique_map, np.int64), inverse_map, np.asarray(colabels, np.int32)


def segment_mean(features, inverse_map, n_unique, average=True):
    """Voxel features of duplicate coordinates: sum / mean of the rows of each voxel in input order
    (MinkowskiSparseTensor.py:317-341: spmm with a [n_unique, N] COO matrix of ones, then / row count)."""
    f = np.asarray(features, np.float64)
    out = np.zeros((n_unique, f.shape[1]), np.float64)
    cnt = np.zeros(n_unique, np.float64)
    for row, u in enumerate(np.asarray(inverse_map)):
        out[u] += f[row]
        cnt[u] += 1
    return out / cnt[:, None] if average else out


# ---- generative maps, pruning, union ------------------------------------------------
def stride_region(coords, region, out_tensor_stride=None):
    """CoordinateMapCPU::stride_region (src/coordinate_map_cpu.hpp:446-487): the coordinates of the kernel region
    around every input coordinate, deduplicated in first-occurrence order (input row, then offset).
    `out_tensor_stride` given = the non-transposed branch (:470-484): only coordinates aligned to it are kept."""
    cand = region_coordinates(coords, region)
    if out_tensor_stride is not None:
        ts = np.asarray(_as_list(out_tensor_stride, cand.shape[1] - 1), np.int64)
        cand = cand[(cand[:, 1:] % ts == 0).all(1)]
    um, _ = insert_and_map(cand)
    return cand[um]


def prune(coords, feats, keep):
    """CoordinateMapCPU::prune + PruningForwardKernelCPU (src/coordinate_map_cpu.hpp:519-536,
    src/pruning_cpu.cpp:40-105): kept rows in row order."""
    keep = np.asarray(keep, bool)
    return _i32(coords)[keep], np.asarray(feats)[keep]


def union(coord_sets, feat_sets):
    """union_map + MinkowskiUnionFunction.forward (MinkowskiUnion.py:41-60): union coordinates in first-occurrence
    order over the concatenated inputs, features of coinciding coordinates added."""
    allc = np.concatenate([_i32(c) for c in coord_sets], 0)
    allf = np.concatenate([np.asarray(f, np.float64) for f in feat_sets], 0)
    um, inv = insert_and_map(allc)
    out = np.zeros((len(um), allf.shape[1]), np.float64)
    np.add.at(out, inv, allf)
    return allc[um], out
