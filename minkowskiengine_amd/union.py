"""MinkowskiUnion (reference: MinkowskiEngine/MinkowskiUnion.py:33-135): union of the coordinates of several
sparse tensors, features of coinciding voxels added."""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import host as _host
from .host import CoordinateMapKey  # noqa: F401
from .sparse_tensor import SparseTensor


class MinkowskiUnionFunction(Function):
    @staticmethod
    def forward(ctx, in_coords_keys, out_coords_key, coordinate_manager, *in_feats):
        assert isinstance(in_feats, (list, tuple)), "Input must be a collection of Tensors"
        assert len(in_feats) > 1, "input must be a set with at least 2 Tensors"
        assert len(in_feats) == len(in_coords_keys), "The input features and keys must have the same length"
        union_maps = coordinate_manager.union_map(in_coords_keys, out_coords_key)
        n_out = coordinate_manager.size(out_coords_key)
        # rows of one voxel are added in input order by the segment-sum kernel (deterministic; the reference adds
        # with index_add, MinkowskiUnion.py:57-58)
        from .utils.quantization import segment_reduce
        allf = torch.cat([f.contiguous() for f in in_feats], 0)
        inverse = torch.cat([m[1] for m in union_maps], 0)
        out_feat = segment_reduce(allf, inverse, n_out, average=False)
        ctx.keys = (in_coords_keys, coordinate_manager)
        ctx.save_for_backward(*union_maps)
        return out_feat

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        union_maps = ctx.saved_tensors
        # every input row has exactly one union row: its gradient is that row of grad_out
        grad_in_feats = [grad_out_feat.index_select(0, m[1]) for m in union_maps]
        return (None, None, None, *grad_in_feats)


class MinkowskiUnion(Module):
    """Union of all input sparse tensors; overlapping features are added."""

    def __init__(self):
        super().__init__()
        self.union = MinkowskiUnionFunction

    def forward(self, *inputs):
        assert isinstance(inputs, (list, tuple)), "The input must be a list or tuple"
        for s in inputs:
            assert isinstance(s, SparseTensor), "Inputs must be sparse tensors."
        assert len(inputs) > 1, "input must be a set with at least 2 SparseTensors"
        ref_key = inputs[0].coordinate_map_key
        ref_manager = inputs[0].coordinate_manager
        for s in inputs:
            assert ref_manager == s.coordinate_manager, \
                "Invalid coordinate manager. All inputs must have the same coordinate manager."
        in_keys = [s.coordinate_map_key for s in inputs]
        out_key = _host.key_like(ref_key)
        output = self.union.apply(in_keys, out_key, ref_manager, *[s.F for s in inputs])
        return SparseTensor(output, coordinate_map_key=out_key, coordinate_manager=ref_manager)

    def __repr__(self):
        return self.__class__.__name__ + "()"
