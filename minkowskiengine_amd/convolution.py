"""MinkowskiConvolution / MinkowskiConvolutionTranspose modules and their autograd Functions
(reference: MinkowskiEngine/MinkowskiConvolution.py:42-634).  The Functions resolve
`Convolution{Forward,Backward}GPU` in the backend by name, exactly as the reference does through
get_minkowski_function (MinkowskiCommon.py:110-120)."""
import math
import os

import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from . import host as _host
from .backend import ConvolutionMode, CoordinateMapKey, RegionType  # noqa: F401
from .common import get_minkowski_function
from .kernel_generator import KernelGenerator
from .sparse_tensor import SparseTensor, _get_coordinate_map_key


_MM_AS_CONV = os.environ.get("ME_AMD_MM_AS_CONV", "1") != "0"   # 1x1 convolutions on the conv kernels
# bf16 features: channel counts that are not multiples of 8 (a 3-channel stem, a 20-class head) are zero-padded to the
# next multiple of 8 around the operator — 16-byte row pieces for the gathers and the bf16 weight-gradient kernel instead
# of the scalar / fp32-MFMA fall-backs (MinkUNet34C stem 3 -> 32, K = 125: forward 121 us, weight gradient 138 us at
# 1.3 TFLOP/s before, profiles/r03_layers_minkunet34c_bf16_before.log); results are those of the unpadded operator
# (padded inputs and weights are zero, padded outputs are dropped), gradients flow through torch's pad / slice
_PAD_CHANNELS = os.environ.get("ME_AMD_PAD_CHANNELS", "1") != "0"


def _pad_channels(feats, kernel, cin, cout):
    """-> (feats, kernel, cout_padded | None) with channel counts rounded up to multiples of 8 (bf16 features only)"""
    if not (_PAD_CHANNELS and feats.is_cuda and feats.dtype == torch.bfloat16 and (cin % 8 or cout % 8)):
        return feats, kernel, None
    pi, po = (-cin) % 8, (-cout) % 8
    if pi:
        feats = torch.nn.functional.pad(feats, (0, pi))
    kernel = torch.nn.functional.pad(kernel, (0, po, 0, pi))
    return feats, kernel, (cout + po if po else None)


def _conv_apply(conv_fn, is_transpose, feats, kernel, kernel_generator, convolution_mode, in_key, out_key, manager,
                training=False):
    """One convolution with autograd: the native host layer's C++ autograd function when `manager` lives there (the
    backward pass then never enters Python), else the reference-style Python Function."""
    # (a training-mode batch norm may follow: bf16 forward launches then leave their tiles' statistics behind)
    want_stats = bool(training) and feats.dtype == torch.bfloat16
    if getattr(manager, "_native", False):
        _host.native_module().conv_bn_stats_hint(want_stats)
        return _host.native_module().conv_autograd(
            feats, kernel, kernel_generator.kernel_size, kernel_generator.kernel_stride, kernel_generator.kernel_dilation,
            int(kernel_generator.region_type), bool(kernel_generator.expand_coordinates), in_key, out_key,
            manager._manager, bool(is_transpose))
    from . import backend as _backend
    _backend.conv_bn_stats_hint(want_stats)
    return conv_fn.apply(feats, kernel, kernel_generator, convolution_mode, in_key, out_key, manager)


class MinkowskiConvolutionFunction(Function):
    @staticmethod
    def forward(ctx, input_features, kernel_weights, kernel_generator, convolution_mode, in_coordinate_map_key,
                out_coordinate_map_key=None, coordinate_manager=None):
        if out_coordinate_map_key is None:
            out_coordinate_map_key = _host.key_like(in_coordinate_map_key)
        input_features = input_features.contiguous()
        ctx.input_features = input_features
        ctx.kernel_weights = kernel_weights
        ctx.misc = (kernel_generator, convolution_mode, in_coordinate_map_key, out_coordinate_map_key,
                    coordinate_manager)
        fw_fn = get_minkowski_function("ConvolutionForward", input_features, in_coordinate_map_key)
        return fw_fn(input_features, kernel_weights, kernel_generator.kernel_size, kernel_generator.kernel_stride,
                     kernel_generator.kernel_dilation, kernel_generator.region_type,
                     kernel_generator.region_offsets, kernel_generator.expand_coordinates, convolution_mode,
                     in_coordinate_map_key, out_coordinate_map_key, coordinate_manager._manager)

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        kernel_generator, convolution_mode, in_key, out_key, coordinate_manager = ctx.misc
        bw_fn = get_minkowski_function("ConvolutionBackward", grad_out_feat, in_key)
        # (not in the reference: the input gradient is skipped when autograd does not ask for it — the first layer
        # of a network, whose input features are data)
        grad_in_feat, grad_kernel = bw_fn(ctx.input_features, grad_out_feat, ctx.kernel_weights,
                                          kernel_generator.kernel_size, kernel_generator.kernel_stride,
                                          kernel_generator.kernel_dilation, kernel_generator.region_type,
                                          kernel_generator.region_offsets, convolution_mode, in_key, out_key,
                                          coordinate_manager._manager, need_grad_in=ctx.needs_input_grad[0])
        return grad_in_feat, grad_kernel, None, None, None, None, None


class MinkowskiConvolutionTransposeFunction(Function):
    @staticmethod
    def forward(ctx, input_features, kernel_weights, kernel_generator, convolution_mode, in_coordinate_map_key,
                out_coordinate_map_key=None, coordinate_manager=None):
        if out_coordinate_map_key is None:
            out_coordinate_map_key = _host.key_like(in_coordinate_map_key)
        input_features = input_features.contiguous()
        ctx.input_features = input_features
        ctx.kernel_weights = kernel_weights
        ctx.misc = (kernel_generator, convolution_mode, in_coordinate_map_key, out_coordinate_map_key,
                    coordinate_manager)
        fw_fn = get_minkowski_function("ConvolutionTransposeForward", input_features, in_coordinate_map_key)
        return fw_fn(input_features, kernel_weights, kernel_generator.kernel_size, kernel_generator.kernel_stride,
                     kernel_generator.kernel_dilation, kernel_generator.region_type,
                     kernel_generator.region_offsets, kernel_generator.expand_coordinates, convolution_mode,
                     in_coordinate_map_key, out_coordinate_map_key, coordinate_manager._manager)

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        kernel_generator, convolution_mode, in_key, out_key, coordinate_manager = ctx.misc
        bw_fn = get_minkowski_function("ConvolutionTransposeBackward", grad_out_feat, in_key)
        grad_in_feat, grad_kernel = bw_fn(ctx.input_features, grad_out_feat, ctx.kernel_weights,
                                          kernel_generator.kernel_size, kernel_generator.kernel_stride,
                                          kernel_generator.kernel_dilation, kernel_generator.region_type,
                                          kernel_generator.region_offsets, convolution_mode, in_key, out_key,
                                          coordinate_manager._manager, need_grad_in=ctx.needs_input_grad[0])
        return grad_in_feat, grad_kernel, None, None, None, None, None


class MinkowskiModuleBase(Module):
    pass


class MinkowskiConvolutionBase(MinkowskiModuleBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, is_transpose=False, expand_coordinates=False,
                 convolution_mode=ConvolutionMode.DEFAULT, dimension=-1):
        super().__init__()
        assert dimension > 0, f"Invalid dimension. Please provide a valid dimension argument. dimension={dimension}"
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               expand_coordinates=expand_coordinates, dimension=dimension)
        else:
            kernel_generator.expand_coordinates = expand_coordinates
        self.is_transpose = is_transpose
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_generator = kernel_generator
        self.dimension = dimension
        # kernel volume 1 and all strides 1 -> a plain matrix product (MinkowskiConvolution.py:264-270)
        self.use_mm = (kernel_generator.kernel_volume == 1 and kernel_generator.requires_strided_coordinates)
        if self.use_mm:
            kernel_shape = (in_channels, out_channels)
        else:
            kernel_shape = (kernel_generator.kernel_volume, in_channels, out_channels)
        self.kernel = Parameter(torch.empty(*kernel_shape, dtype=torch.float32))
        self.bias = Parameter(torch.empty(1, out_channels, dtype=torch.float32)) if bias else None
        self.convolution_mode = convolution_mode
        self.conv = MinkowskiConvolutionTransposeFunction if is_transpose else MinkowskiConvolutionFunction

    def forward(self, input, coordinates=None):
        assert isinstance(input, SparseTensor)
        assert input.D == self.dimension
        if self.use_mm:
            # kernel volume 1, stride 1: out = F @ W on the same coordinate map (MinkowskiConvolution.py:304-308).
            out_coordinate_map_key = input.coordinate_map_key
            if _MM_AS_CONV and input.F.is_cuda and input.F.shape[0] > 0:
                # run it as a one-offset convolution on this package's kernels (no vendor BLAS on the path; rocBLAS /
                # hipBLASLt also pick slow kernels for the skinny products of a segmentation head: 200k x 96 @ 96 x 20
                # took 325 us).  A transposed 1x1 stride-1 layer is the same product on the same map, so it takes
                # the non-transposed operator too.
                feats, kernel, cpad = _pad_channels(input.F, self.kernel.unsqueeze(0), self.in_channels,
                                                    self.out_channels)
                outfeat = _conv_apply(MinkowskiConvolutionFunction, False, feats, kernel, self.kernel_generator,
                                      self.convolution_mode, input.coordinate_map_key, out_coordinate_map_key,
                                      input._manager, self.training)
                if cpad is not None:
                    outfeat = outfeat[:, :self.out_channels].contiguous()
            else:
                # bf16 features with fp32 master weights: the product runs in the feature dtype
                outfeat = input.F.mm(self.kernel if self.kernel.dtype == input.F.dtype
                                     else self.kernel.to(input.F.dtype))
        else:
            # (the reference passes expand_coordinates positionally into the tensor_stride slot,
            # MinkowskiConvolution.py:311-313; passed by keyword here)
            out_coordinate_map_key = _get_coordinate_map_key(
                input, coordinates, expand_coordinates=self.kernel_generator.expand_coordinates)
            feats, kernel, cpad = _pad_channels(input.F, self.kernel, self.in_channels, self.out_channels)
            outfeat = _conv_apply(self.conv, self.is_transpose, feats, kernel, self.kernel_generator,
                                  self.convolution_mode, input.coordinate_map_key, out_coordinate_map_key,
                                  input._manager, self.training)
            if cpad is not None:
                outfeat = outfeat[:, :self.out_channels].contiguous()
        if self.bias is not None:
            outfeat = outfeat + (self.bias if self.bias.dtype == outfeat.dtype else self.bias.to(outfeat.dtype))
        return SparseTensor(outfeat, coordinate_map_key=out_coordinate_map_key, coordinate_manager=input._manager)

    def reset_parameters(self, is_transpose=False):
        with torch.no_grad():
            n = (self.out_channels if is_transpose else self.in_channels) * self.kernel_generator.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)      # (in place on the parameter itself: bumps its version counter,
            if self.bias is not None:              #  which validates the packed weight images; `.data` would not)
                self.bias.uniform_(-stdv, stdv)

    def __repr__(self):
        s = f"(in={self.in_channels}, out={self.out_channels}, "
        if self.kernel_generator.region_type in [RegionType.CUSTOM]:
            s += f"region_type={self.kernel_generator.region_type}, kernel_volume={self.kernel_generator.kernel_volume}, "
        else:
            s += f"kernel_size={self.kernel_generator.kernel_size}, "
        s += f"stride={self.kernel_generator.kernel_stride}, dilation={self.kernel_generator.kernel_dilation})"
        return self.__class__.__name__ + s


class MinkowskiConvolution(MinkowskiConvolutionBase):
    """out_u = sum_k W_k^T x_{u + offset(k)} over the existing input voxels
    (MinkowskiConvolution.py:360-451)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=ConvolutionMode.DEFAULT,
                 dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=False, expand_coordinates=expand_coordinates,
                         convolution_mode=convolution_mode, dimension=dimension)
        self.reset_parameters()


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    """Transposed (up-sampling) convolution (MinkowskiConvolution.py:454-560)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=ConvolutionMode.DEFAULT,
                 dimension=None):
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               dimension=dimension)
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=True, expand_coordinates=expand_coordinates,
                         convolution_mode=convolution_mode, dimension=dimension)
        self.reset_parameters(True)


class MinkowskiGenerativeConvolutionTranspose(MinkowskiConvolutionBase):
    """Transposed convolution that GENERATES its output coordinates: every kernel offset around every input
    voxel becomes an output voxel (MinkowskiConvolution.py:539-634; coordinate generation
    src/coordinate_map_cpu.hpp:446-487)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, convolution_mode=ConvolutionMode.DEFAULT, dimension=None):
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               expand_coordinates=True, dimension=dimension)
        else:
            kernel_generator.expand_coordinates = True
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=True, expand_coordinates=True, convolution_mode=convolution_mode,
                         dimension=dimension)
        self.reset_parameters(True)
