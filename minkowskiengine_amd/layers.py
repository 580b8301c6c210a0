"""Feature-only layers that MinkUNet-style networks need around the convolutions.  In the reference
these contain no native code: they apply torch modules to `.F` and re-wrap
(MinkowskiNormalization.py:51-140, MinkowskiNonlinearity.py, MinkowskiOps.py:141-158)."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import backend as MEB
from . import host as _host
from .sparse_tensor import SparseTensor

_TORCH_BN = os.environ.get("ME_AMD_TORCH_BN", "0") != "0"   # 1: torch's batch-norm kernels (A/B timing)
_FUSE_RESIDUAL = os.environ.get("ME_AMD_FUSE_RESIDUAL", "1") != "0"   # bn + residual add + relu in one kernel


def _rewrap(x, feats):
    return SparseTensor(feats, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x._manager)


class _BatchNormTrainFunction(Function):
    """Training-mode batch norm of a feature matrix on the HIP kernels of csrc/norm.hip (statistics in fp32,
    fixed summation order).  weight / bias may be None."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu=False, num_batches_tracked=None):
        x = x.contiguous()
        w32 = weight.float() if weight is not None else None
        b32 = bias.float() if bias is not None else None
        mean, rstd = MEB.bn_stats(x, eps, momentum, running_mean, running_var, num_batches_tracked)
        y = MEB.bn_apply(x, mean, rstd, w32, b32, relu)
        ctx.save_for_backward(x, mean, rstd, w32, b32)
        ctx.param_dtype = weight.dtype if weight is not None else None
        ctx.has_bias = bias is not None
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w32, b32 = ctx.saved_tensors
        dx, gg, gb = MEB.bn_backward(x, dy, mean, rstd, w32, b32, ctx.relu)
        gw = gg.to(ctx.param_dtype) if ctx.param_dtype is not None else None
        gbias = gb.to(ctx.param_dtype) if ctx.has_bias else None
        return dx, gw, gbias, None, None, None, None, None, None


class _BatchNormResidualFunction(Function):
    """y = [relu] (batch_norm(x) + skip) on the fused kernels of csrc/norm.hip (me_bn_apply_residual /
    me_bn_backward_residual): the tail of a ResNet block in one pass per direction."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, running_mean, running_var, momentum, eps, relu, num_batches_tracked):
        x = x.contiguous()
        skip = skip.contiguous()
        w32 = weight.float() if weight is not None else None
        b32 = bias.float() if bias is not None else None
        mean, rstd = MEB.bn_stats(x, eps, momentum, running_mean, running_var, num_batches_tracked)
        y = MEB.bn_apply_residual(x, skip, mean, rstd, w32, b32, relu)
        ctx.save_for_backward(x, y if relu else None, mean, rstd, w32, b32)
        ctx.param_dtype = weight.dtype if weight is not None else None
        ctx.has_bias = bias is not None
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, w32, b32 = ctx.saved_tensors
        dx, dskip, gg, gb = MEB.bn_backward_residual(x, dy, y, mean, rstd, w32, b32, ctx.relu,
                                                     need_dskip=ctx.needs_input_grad[1])
        gw = gg.to(ctx.param_dtype) if ctx.param_dtype is not None else None
        gbias = gb.to(ctx.param_dtype) if ctx.has_bias else None
        return dx, dskip, gw, gbias, None, None, None, None, None, None


class MinkowskiBatchNorm(nn.Module):
    """torch.nn.BatchNorm1d semantics on the feature matrix (MinkowskiNormalization.py:35-82).  The parameters
    and running statistics live in `self.bn` (same state-dict names as the reference); on the GPU the arithmetic
    runs on this package's kernels."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        # Not in the reference: set by a model whose next layer is ALWAYS a MinkowskiReLU.  The rectification is
        # then done by the batch-norm kernels (forward clamp, backward mask) and the output tensor is marked so
        # that the following MinkowskiReLU passes it through; the results are those of the two separate layers.
        self.fuse_relu = False

    def _native(self, f):
        bn = self.bn
        return (not _TORCH_BN and type(bn) is nn.BatchNorm1d and f.is_cuda and f.dim() == 2 and f.shape[0] > 1
                and f.dtype in (torch.float32, torch.bfloat16) and bn.momentum is not None
                and f.shape[1] <= 2048)

    def forward(self, input):
        f = input.F
        bn = self.bn
        if not self._native(f):
            return _rewrap(input, bn(f))
        use_batch_stats = bn.training or not bn.track_running_stats
        if use_batch_stats:
            rm = bn.running_mean if (bn.training and bn.track_running_stats) else None
            rv = bn.running_var if (bn.training and bn.track_running_stats) else None
            # (num_batches_tracked is incremented by the statistics kernel: 62 one-element torch kernels per
            # MinkUNet34C pass otherwise)
            nbt = bn.num_batches_tracked if rm is not None else None
            if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
                nbt.add_(1)
                nbt = None
            if getattr(input._manager, "_native", False):      # C++ autograd function of the native host layer
                y = _host.native_module().batch_norm_train(f, None, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps,
                                                           bool(self.fuse_relu), nbt)
            else:
                y = _BatchNormTrainFunction.apply(f, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, self.fuse_relu,
                                                  nbt)
            out = _rewrap(input, y)
            out._rectified = self.fuse_relu
            return out
        else:
            # evaluation: an affine map per channel, differentiable through torch (cheap: two fused passes)
            # (the kernels take float32 vectors: after `model.bfloat16()` the running statistics are bf16)
            rmean = bn.running_mean.float().contiguous()
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps).contiguous()
            a = rstd * bn.weight.float() if bn.weight is not None else rstd
            b = (bn.bias.float() if bn.bias is not None else 0.0) - rmean * a
            y = (f * a.to(f.dtype) + b.to(f.dtype)) if f.requires_grad else \
                MEB.bn_apply(f.contiguous(), rmean, rstd,
                             bn.weight.float().contiguous() if bn.weight is not None else None,
                             bn.bias.float().contiguous() if bn.bias is not None else None)
        return _rewrap(input, y)

    def forward_residual(self, input, skip, relu=True):
        """[relu] (self(input) + skip) — the tail of a ResNet block (reference: modules/resnet_block.py:62-75 applies
        norm, `out += residual`, relu as three operators).  Training mode on the GPU: one fused kernel per direction,
        bit-identical to the separate operators; anything else falls back to them."""
        assert input.coordinate_map_key == skip.coordinate_map_key, "residual add needs a shared coordinate map"
        f, bn = input.F, self.bn
        fused = (_FUSE_RESIDUAL and self._native(f) and (bn.training or not bn.track_running_stats)
                 and skip.F.dtype == f.dtype and skip.F.shape == f.shape)
        if not fused:
            out = self.forward(input)
            y = out.F + skip.F
            return _rewrap(input, torch.relu(y) if relu else y)
        rm = bn.running_mean if (bn.training and bn.track_running_stats) else None
        rv = bn.running_var if (bn.training and bn.track_running_stats) else None
        nbt = bn.num_batches_tracked if rm is not None else None
        if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
            nbt.add_(1)
            nbt = None
        if getattr(input._manager, "_native", False):
            y = _host.native_module().batch_norm_train(f, skip.F, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps,
                                                       bool(relu), nbt)
        else:
            y = _BatchNormResidualFunction.apply(f, skip.F, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, bool(relu),
                                                 nbt)
        return _rewrap(input, y)

    def __repr__(self):
        b = self.bn
        return (f"{self.__class__.__name__}({b.num_features}, eps={b.eps}, momentum={b.momentum}, "
                f"affine={b.affine}, track_running_stats={b.track_running_stats})")


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        nn.Module.__init__(self)
        self.fuse_relu = False   # torch's SyncBatchNorm does the arithmetic: the following MinkowskiReLU rectifies
        self.bn = nn.SyncBatchNorm(num_features, eps=eps, momentum=momentum, affine=affine,
                                   track_running_stats=track_running_stats, process_group=process_group)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """MinkowskiNormalization.py:138-191.  Conscious divergence: the reference recurses into the children of a
        MinkowskiBatchNorm it has just converted and its `add_module("bn", <the old nn.BatchNorm1d>)` puts the
        UNSYNCHRONISED torch module back (MinkowskiNormalization.py:186-189), so its "sync" batch norm normalises with
        per-rank statistics; here a converted module keeps its nn.SyncBatchNorm
        (tests/test_gpu_distributed.py compares against one process on the batched scenes)."""
        out = module
        if isinstance(module, MinkowskiSyncBatchNorm):
            return module
        if isinstance(module, MinkowskiBatchNorm):
            out = cls(module.bn.num_features, module.bn.eps, module.bn.momentum, module.bn.affine,
                      module.bn.track_running_stats, process_group)
            if module.bn.affine:          # the same Parameter objects, as the reference (:176-179)
                out.bn.weight = module.bn.weight
                out.bn.bias = module.bn.bias
            if module.bn.track_running_stats:
                out.bn.running_mean = module.bn.running_mean
                out.bn.running_var = module.bn.running_var
                out.bn.num_batches_tracked = module.bn.num_batches_tracked
            return out
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

    def forward(self, input):
        # the output of a MinkowskiBatchNorm with fuse_relu is already rectified
        if getattr(input, "_rectified", False):
            return input
        return _rewrap(input, self.module(input.F))


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
