"""Voxelisation on the device (reference: MinkowskiEngine/utils/quantization.py:136-333 and
src/quantization.cpp).  The reference hashes on the host (and requires CPU tensors when labels are given);
here the coordinates go through the same HIP hash map as every other coordinate map
(me_coords_insert_and_map), labels and duplicate features are reduced by two small kernels, and results
come back in the caller's container (numpy in -> numpy out, torch in -> torch on the input's device)."""
from collections.abc import Sequence

import numpy as np
import torch

from .. import _lib
from .. import host as _host


def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("minkowskiengine_amd voxelises on the GPU (MI355X); no GPU is visible")
    if device is None or (isinstance(device, str) and device == "cpu") or \
            (isinstance(device, torch.device) and device.type == "cpu"):
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _to_dev(a, dev):
    return (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(dev)


def unique_coordinate_map(coordinates, tensor_stride=1):
    """(unique_map, inverse_map) of integer coordinates: coordinates[unique_map] are the unique rows in
    first-occurrence order, coordinates[unique_map][inverse_map] == coordinates (quantization.py:336-363)."""
    assert coordinates.ndim == 2, "Coordinates must be a matrix"
    assert isinstance(coordinates, torch.Tensor)
    src = coordinates.device
    c = coordinates.to(_device()).int().contiguous()
    mgr = _host.backend().CoordinateMapManagerGPU_c10()      # the operator module in charge (native | Python twin)
    D = c.shape[1] - 1
    ts = [int(tensor_stride)] * D if np.isscalar(tensor_stride) else [int(t) for t in tensor_stride]
    _, (um, inv) = mgr.insert_and_map(c, ts, "")
    return um.to(src), inv.to(src)


def quantize(coords):
    """unique / inverse index maps of int32 coordinates (quantization.py:66-103)."""
    is_np = isinstance(coords, np.ndarray)
    assert is_np or isinstance(coords, torch.Tensor), "Invalid coords type"
    if is_np:
        assert coords.dtype == np.int32, f"Invalid coords type {coords.dtype} != np.int32"
    um, inv = unique_coordinate_map(_to_dev(coords, "cpu" if is_np else coords.device).int())
    return (um.cpu().numpy(), inv.cpu().numpy()) if is_np else (um, inv)


def quantize_label(coords, labels, ignore_label):
    """(unique_map, inverse_map, colabels): a voxel keeps the label its points agree on, otherwise
    `ignore_label` (src/quantization.cpp:140-196; see include/me_amd.h for the one deliberate difference)."""
    is_np = isinstance(coords, np.ndarray)
    dev = _device()
    c = _to_dev(coords, dev).int().contiguous()
    lab = _to_dev(labels, dev).int().contiguous()
    assert c.shape[0] == lab.shape[0], "Coords nrows must be equal to label size."
    um, inv = unique_coordinate_map(c)
    col = torch.empty(um.numel(), dtype=torch.int32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.me_coords_quantize_labels(um.data_ptr(), um.numel(), inv.data_ptr(), lab.data_ptr(),
                                                 c.shape[0], int(ignore_label), col.data_ptr(),
                                                 torch.cuda.current_stream(dev).cuda_stream))
    if is_np:
        return um.cpu().numpy(), inv.cpu().numpy(), col.cpu().numpy()
    src = coords.device
    return um.to(src), inv.to(src), col.to(src)


def segment_reduce(features, inverse_map, n_unique, average=True):
    """Sum / mean of the feature rows of every voxel in input-row order (deterministic); the device-side
    twin of the reference's coo_spmm voxel averaging (MinkowskiSparseTensor.py:317-341)."""
    dev = features.device
    assert features.is_cuda, "features must be on the GPU"
    f32 = features.float().contiguous()
    inv = inverse_map.to(dev).long()
    perm = torch.argsort(inv, stable=True)
    seg = torch.zeros(n_unique + 1, dtype=torch.int64, device=dev)
    seg[1:] = torch.cumsum(torch.bincount(inv, minlength=n_unique), 0)
    out = torch.empty((n_unique, f32.shape[1]), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.me_segment_sum_f32(f32.data_ptr(), f32.shape[1], perm.data_ptr(), seg.data_ptr(), n_unique,
                                          1 if average else 0, out.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream))
    return out.to(features.dtype)


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device=None):
    """Voxelise points (quantization.py:136-333): floor(coordinates / quantization_size), one row per occupied
    voxel (the first point of each voxel), optional voxel labels.  Same arguments and return conventions as
    the reference; `device` is accepted for compatibility — the hashing always runs on the GPU."""
    assert isinstance(coordinates, (np.ndarray, torch.Tensor)), "Coords must be either np.array or torch.Tensor."
    is_np = isinstance(coordinates, np.ndarray)
    use_label, use_feat = labels is not None, features is not None
    assert coordinates.ndim == 2, \
        "The coordinates must be a 2D matrix. The shape of the input is " + str(coordinates.shape)
    if return_inverse:
        assert return_index, "return_reverse must be set with return_index"
    if use_feat:
        assert features.ndim == 2
        assert coordinates.shape[0] == features.shape[0]
    if use_label:
        assert coordinates.shape[0] == len(labels)
    dimension = coordinates.shape[1]
    dev = _device(device)
    src = torch.device("cpu") if is_np else coordinates.device
    c = _to_dev(coordinates, dev)
    if quantization_size is not None:
        if isinstance(quantization_size, (Sequence, np.ndarray, torch.Tensor)):
            assert len(quantization_size) == dimension, "Quantization size and coordinates size mismatch."
            q = torch.tensor([float(i) for i in quantization_size], device=dev)
            c = torch.floor(c / q)
        elif np.isscalar(quantization_size):
            c = torch.floor(c) if quantization_size == 1 else torch.floor(c / quantization_size)
        else:
            raise ValueError("Not supported type for quantization_size.")
    elif c.is_floating_point():
        c = torch.floor(c)
    discrete = c.int().contiguous()

    back = (lambda t: t.cpu().numpy()) if is_np else (lambda t: t.to(src))
    if use_label:
        um, inv, col = quantize_label(discrete, _to_dev(labels, dev), ignore_label)
    else:
        um, inv = unique_coordinate_map(discrete)
        if return_maps_only:
            return (back(um), back(inv)) if return_inverse else back(um)
    ret = [back(discrete[um])]
    if use_feat:
        ret.append(features[um.cpu().numpy()] if is_np else features[um.to(features.device)])
    if use_label:
        ret.append(back(col))
    if return_index:
        ret.append(back(um))
    if return_inverse:
        ret.append(back(inv))
    return ret[0] if len(ret) == 1 else tuple(ret)
