"""Argument helpers and operator dispatch shared by the Python API (same role as the reference's
MinkowskiEngine/MinkowskiCommon.py)."""
from collections.abc import Sequence

import numpy as np
import torch

from . import host as _host


def convert_to_int_list(arg, dimension):
    """scalar | sequence | ndarray | tensor -> list[int] of length `dimension`
    (MinkowskiCommon.py:40-56)."""
    if isinstance(arg, (Sequence, np.ndarray, torch.Tensor)) and not isinstance(arg, str):
        out = [int(v) for v in arg]
        assert len(out) == dimension, f"expected {dimension} values, got {out}"
        return out
    if np.isscalar(arg):
        return [int(arg)] * dimension
    raise ValueError("Input must be a scalar or a sequence")


def get_postfix(tensor):
    return "GPU" if tensor.is_cuda else "CPU"


def get_minkowski_function(name, variable, owner=None):
    """Resolve `<Op>{GPU,CPU}` in the backend by device, as MinkowskiCommon.py:110-120 does.
    Only the GPU (MI355X) entries exist.  `owner` (a coordinate map key or manager of the call): the operator comes
    from the host layer that made it — objects made under one host keep working after set_host()."""
    fn_name = name + get_postfix(variable)
    fn = getattr(_host.backend_of(owner) if owner is not None else _host.backend(), fn_name, None)
    if fn is None:
        raise ValueError(
            f"Function {fn_name} not available: minkowskiengine_amd implements the MI355X (GPU) path only; "
            "move the tensors to the GPU (`.cuda()`)." if not variable.is_cuda else
            f"Function {fn_name} not available.")
    return fn
