"""Feature-only layers that MinkUNet-style networks need around the convolutions.  In the reference
these contain no native code: they apply torch modules to `.F` and re-wrap
(MinkowskiNormalization.py:51-140, MinkowskiNonlinearity.py, MinkowskiOps.py:141-158)."""
import torch
import torch.nn as nn

from .sparse_tensor import SparseTensor


def _rewrap(x, feats):
    return SparseTensor(feats, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x._manager)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, input):
        return _rewrap(input, self.bn(input.F))

    def __repr__(self):
        b = self.bn
        return (f"{self.__class__.__name__}({b.num_features}, eps={b.eps}, momentum={b.momentum}, "
                f"affine={b.affine}, track_running_stats={b.track_running_stats})")


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        nn.Module.__init__(self)
        self.bn = nn.SyncBatchNorm(num_features, eps=eps, momentum=momentum, affine=affine,
                                   track_running_stats=track_running_stats, process_group=process_group)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """MinkowskiNormalization.py:117-140"""
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            out = cls(module.bn.num_features, module.bn.eps, module.bn.momentum, module.bn.affine,
                      module.bn.track_running_stats, process_group)
            if module.bn.affine:
                with torch.no_grad():
                    out.bn.weight.copy_(module.bn.weight)
                    out.bn.bias.copy_(module.bn.bias)
            if module.bn.track_running_stats:
                out.bn.running_mean = module.bn.running_mean
                out.bn.running_var = module.bn.running_var
                out.bn.num_batches_tracked = module.bn.num_batches_tracked
        for name, child in module.named_children():
            out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        return out


class _Elementwise(nn.Module):
    MODULE = None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.module = self.MODULE(*args, **kwargs)

    def forward(self, input):
        return _rewrap(input, self.module(input.F))

    def __repr__(self):
        return self.__class__.__name__ + "()"


class MinkowskiReLU(_Elementwise):
    MODULE = nn.ReLU


class MinkowskiLeakyReLU(_Elementwise):
    MODULE = nn.LeakyReLU


class MinkowskiELU(_Elementwise):
    MODULE = nn.ELU


class MinkowskiSigmoid(_Elementwise):
    MODULE = nn.Sigmoid


class MinkowskiTanh(_Elementwise):
    MODULE = nn.Tanh


class MinkowskiDropout(_Elementwise):
    MODULE = nn.Dropout


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, input):
        return _rewrap(input, self.linear(input.F))


def cat(*sparse_tensors):
    """Concatenate the features of tensors that share one coordinate map (MinkowskiOps.py:141-158)."""
    if len(sparse_tensors) == 1 and isinstance(sparse_tensors[0], (list, tuple)):
        sparse_tensors = tuple(sparse_tensors[0])
    first = sparse_tensors[0]
    for s in sparse_tensors:
        assert isinstance(s, SparseTensor), "Inputs must be sparse tensors."
        assert s._manager is first._manager, "coordinate managers must match"
        assert s.coordinate_map_key == first.coordinate_map_key, "cat needs a shared coordinate map"
    return _rewrap(first, torch.cat([s.F for s in sparse_tensors], dim=1))
