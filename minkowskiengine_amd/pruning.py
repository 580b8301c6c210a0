"""MinkowskiPruning (reference: MinkowskiEngine/MinkowskiPruning.py:38-133, src/pruning_cpu.cpp)."""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import host as _host
from .host import CoordinateMapKey  # noqa: F401
from .common import get_minkowski_function
from .sparse_tensor import SparseTensor


class MinkowskiPruningFunction(Function):
    @staticmethod
    def forward(ctx, in_feat, mask, in_coords_key, out_coords_key=None, coords_manager=None):
        ctx.in_coords_key = in_coords_key
        ctx.out_coords_key = out_coords_key
        ctx.coords_manager = coords_manager
        in_feat = in_feat.contiguous()
        fw_fn = get_minkowski_function("PruningForward", in_feat, in_coords_key)
        return fw_fn(in_feat, mask, ctx.in_coords_key, ctx.out_coords_key, ctx.coords_manager._manager)

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        bw_fn = get_minkowski_function("PruningBackward", grad_out_feat, ctx.in_coords_key)
        grad_in_feat = bw_fn(grad_out_feat, ctx.in_coords_key, ctx.out_coords_key, ctx.coords_manager._manager)
        return grad_in_feat, None, None, None, None


class MinkowskiPruning(Module):
    """Remove the coordinates (and features) where `mask` is False."""

    def __init__(self):
        super().__init__()
        self.pruning = MinkowskiPruningFunction

    def forward(self, input, mask):
        assert isinstance(input, SparseTensor)
        assert isinstance(mask, torch.Tensor) and mask.dtype in (torch.bool, torch.uint8), \
            "mask must be a boolean tensor"
        out_coords_key = _host.key_like(input.coordinate_map_key)
        output = self.pruning.apply(input.F, mask.to(input.F.device), input.coordinate_map_key, out_coords_key,
                                    input._manager)
        return SparseTensor(output, coordinate_map_key=out_coords_key, coordinate_manager=input._manager)

    def __repr__(self):
        return self.__class__.__name__ + "()"
