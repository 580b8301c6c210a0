"""Broadcast modules (reference: MinkowskiEngine/MinkowskiBroadcast.py:40-253): combine every row of a
sparse tensor with the row of its batch index in a globally pooled tensor."""
import torch
from torch.autograd import Function
from torch.nn import Module

from .backend import BroadcastMode
from .common import get_minkowski_function
from .sparse_tensor import SparseTensor


class MinkowskiBroadcastFunction(Function):
    @staticmethod
    def forward(ctx, input_features, input_features_global, operation_type, in_coords_key, glob_coords_key,
                coords_manager):
        assert isinstance(operation_type, BroadcastMode)
        input_features = input_features.contiguous()
        input_features_global = input_features_global.contiguous()
        ctx.saved_vars = (input_features, input_features_global, operation_type, in_coords_key, glob_coords_key,
                          coords_manager)
        fw_fn = get_minkowski_function("BroadcastForward", input_features, in_coords_key)
        return fw_fn(input_features, input_features_global, operation_type, in_coords_key, glob_coords_key,
                     coords_manager._manager)

    @staticmethod
    def backward(ctx, grad_out_feat):
        if not grad_out_feat.is_contiguous():
            grad_out_feat = grad_out_feat.contiguous()
        input_features, input_features_global, operation_type, in_key, glob_key, coords_manager = ctx.saved_vars
        bw_fn = get_minkowski_function("BroadcastBackward", grad_out_feat, in_key)
        grad_in_feat, grad_in_feat_glob = bw_fn(input_features, input_features_global, grad_out_feat,
                                                operation_type, in_key, glob_key, coords_manager._manager)
        return grad_in_feat, grad_in_feat_glob, None, None, None, None


class MinkowskiBroadcastBase(Module):
    def __init__(self, operation_type):
        super().__init__()
        assert isinstance(operation_type, BroadcastMode)
        self.operation_type = operation_type
        self.broadcast = MinkowskiBroadcastFunction

    def forward(self, input, input_glob):
        assert isinstance(input, SparseTensor)
        output = self.broadcast.apply(input.F, input_glob.F, self.operation_type, input.coordinate_map_key,
                                      input_glob.coordinate_map_key, input.coordinate_manager)
        return SparseTensor(output, coordinate_map_key=input.coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)

    def __repr__(self):
        return self.__class__.__name__


class MinkowskiBroadcastAddition(MinkowskiBroadcastBase):
    """y_i = x_i + g_batch(i)  (MinkowskiBroadcast.py:130-152)"""

    def __init__(self):
        super().__init__(BroadcastMode.ELEMENTWISE_ADDITON)


class MinkowskiBroadcastMultiplication(MinkowskiBroadcastBase):
    """y_i = x_i * g_batch(i)  (MinkowskiBroadcast.py:155-177)"""

    def __init__(self):
        super().__init__(BroadcastMode.ELEMENTWISE_MULTIPLICATION)


class MinkowskiBroadcast(Module):
    """Copy the global row of every batch index to the rows of `input` (MinkowskiBroadcast.py:180-213)."""

    def forward(self, input, input_glob):
        assert isinstance(input, SparseTensor)
        assert isinstance(input_glob, SparseTensor)
        zeros = torch.zeros(input.F.shape[0], input_glob.F.shape[1], dtype=input_glob.F.dtype,
                            device=input_glob.F.device)
        out = MinkowskiBroadcastFunction.apply(zeros, input_glob.F, BroadcastMode.ELEMENTWISE_ADDITON,
                                               input.coordinate_map_key, input_glob.coordinate_map_key,
                                               input.coordinate_manager)
        return SparseTensor(out, coordinate_map_key=input.coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)

    def __repr__(self):
        return self.__class__.__name__


class MinkowskiBroadcastConcatenation(MinkowskiBroadcast):
    """Concatenate the broadcast global row to the features (MinkowskiBroadcast.py:216-253)."""

    def forward(self, input, input_glob):
        broadcast = super().forward(input, input_glob)
        return SparseTensor(torch.cat((input.F, broadcast.F), dim=1), coordinate_map_key=input.coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)
