"""Local / global pooling modules and their autograd Functions (reference:
MinkowskiEngine/MinkowskiPooling.py:40-780).  The Functions resolve `LocalPooling{Forward,Backward}GPU`,
`LocalPoolingTranspose*GPU` and `GlobalPooling*GPU` in the backend by name, as the reference does."""
import torch
from torch.autograd import Function
from torch.nn import Module

from .backend import PoolingMode
from . import host as _host
from .host import CoordinateMapKey  # noqa: F401
from .common import get_minkowski_function
from .kernel_generator import KernelGenerator
from .sparse_tensor import SparseTensor, _get_coordinate_map_key


class MinkowskiLocalPoolingFunction(Function):
    @staticmethod
    def forward(ctx, input_features, pooling_mode, kernel_generator, in_coordinate_map_key,
                out_coordinate_map_key=None, coordinate_manager=None):
        if out_coordinate_map_key is None:
            out_coordinate_map_key = _host.key_like(in_coordinate_map_key)
        input_features = input_features.contiguous()
        ctx.input_features = input_features
        ctx.misc = (pooling_mode, kernel_generator, in_coordinate_map_key, out_coordinate_map_key,
                    coordinate_manager)
        fw_fn = get_minkowski_function("LocalPoolingForward", input_features, in_coordinate_map_key)
        out_feat, num_nonzero = fw_fn(input_features, kernel_generator.kernel_size, kernel_generator.kernel_stride,
                                      kernel_generator.kernel_dilation, kernel_generator.region_type,
                                      kernel_generator.region_offsets, pooling_mode, in_coordinate_map_key,
                                      out_coordinate_map_key, coordinate_manager._manager)
        ctx.num_nonzero = num_nonzero
        return out_feat

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        pooling_mode, kernel_generator, in_key, out_key, coordinate_manager = ctx.misc
        bw_fn = get_minkowski_function("LocalPoolingBackward", grad_out_feat, in_key)
        grad_in_feat = bw_fn(ctx.input_features, grad_out_feat, ctx.num_nonzero, kernel_generator.kernel_size,
                             kernel_generator.kernel_stride, kernel_generator.kernel_dilation,
                             kernel_generator.region_type, kernel_generator.region_offsets, pooling_mode, in_key,
                             out_key, coordinate_manager._manager)
        return grad_in_feat, None, None, None, None, None


class MinkowskiLocalPoolingTransposeFunction(Function):
    @staticmethod
    def forward(ctx, input_features, pooling_mode, kernel_generator, in_coordinate_map_key,
                out_coordinate_map_key=None, coordinate_manager=None):
        if out_coordinate_map_key is None:
            out_coordinate_map_key = _host.key_like(in_coordinate_map_key)
        input_features = input_features.contiguous()
        ctx.input_features = input_features
        ctx.misc = (pooling_mode, kernel_generator, in_coordinate_map_key, out_coordinate_map_key,
                    coordinate_manager)
        fw_fn = get_minkowski_function("LocalPoolingTransposeForward", input_features, in_coordinate_map_key)
        out_feat, num_nonzero = fw_fn(input_features, kernel_generator.kernel_size, kernel_generator.kernel_stride,
                                      kernel_generator.kernel_dilation, kernel_generator.region_type,
                                      kernel_generator.region_offsets, kernel_generator.expand_coordinates,
                                      pooling_mode, in_coordinate_map_key, out_coordinate_map_key,
                                      coordinate_manager._manager)
        ctx.num_nonzero = num_nonzero
        return out_feat

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        pooling_mode, kernel_generator, in_key, out_key, coordinate_manager = ctx.misc
        bw_fn = get_minkowski_function("LocalPoolingTransposeBackward", grad_out_feat, in_key)
        grad_in_feat = bw_fn(ctx.input_features, grad_out_feat, ctx.num_nonzero, kernel_generator.kernel_size,
                             kernel_generator.kernel_stride, kernel_generator.kernel_dilation,
                             kernel_generator.region_type, kernel_generator.region_offsets, pooling_mode, in_key,
                             out_key, coordinate_manager._manager)
        return grad_in_feat, None, None, None, None, None


class MinkowskiPoolingBase(Module):
    """MinkowskiPooling.py:110-205"""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, is_transpose=False,
                 pooling_mode=PoolingMode.LOCAL_AVG_POOLING, dimension=-1):
        super().__init__()
        assert dimension is not None and dimension > 0, \
            f"Invalid dimension. Please provide a valid dimension argument. dimension={dimension}"
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               dimension=dimension)
        self.is_transpose = is_transpose
        self.kernel_generator = kernel_generator
        self.pooling_mode = pooling_mode
        self.dimension = dimension
        self.pooling = MinkowskiLocalPoolingTransposeFunction if is_transpose else MinkowskiLocalPoolingFunction

    def forward(self, input, coordinates=None):
        assert isinstance(input, SparseTensor)
        assert input.D == self.dimension
        out_coordinate_map_key = _get_coordinate_map_key(input, coordinates)
        outfeat = self.pooling.apply(input.F, self.pooling_mode, self.kernel_generator, input.coordinate_map_key,
                                     out_coordinate_map_key, input._manager)
        return SparseTensor(outfeat, coordinate_map_key=out_coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)

    def __repr__(self):
        kg = self.kernel_generator
        return (self.__class__.__name__ +
                f"(kernel_size={kg.kernel_size}, stride={kg.kernel_stride}, dilation={kg.kernel_dilation})")


class MinkowskiAvgPooling(MinkowskiPoolingBase):
    """Average of the input features inside the kernel, divided by the number of present neighbours
    (MinkowskiPooling.py:208-296)."""

    def __init__(self, kernel_size=-1, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__(kernel_size, stride, dilation, kernel_generator, False, PoolingMode.LOCAL_AVG_POOLING,
                         dimension)


class MinkowskiSumPooling(MinkowskiPoolingBase):
    """MinkowskiPooling.py:299-379"""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__(kernel_size, stride, dilation, kernel_generator, False, PoolingMode.LOCAL_SUM_POOLING,
                         dimension)


class MinkowskiMaxPooling(MinkowskiPoolingBase):
    """MinkowskiPooling.py:382-452"""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__(kernel_size, stride, dilation, kernel_generator, False, PoolingMode.LOCAL_MAX_POOLING,
                         dimension)


class MinkowskiPoolingTranspose(MinkowskiPoolingBase):
    """Unpooling: sums over the transposed kernel map (MinkowskiPooling.py:530-622)."""

    def __init__(self, kernel_size, stride, dilation=1, kernel_generator=None, expand_coordinates=False,
                 dimension=None):
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               expand_coordinates=expand_coordinates, dimension=dimension)
        super().__init__(kernel_size, stride, dilation, kernel_generator, True, PoolingMode.LOCAL_AVG_POOLING,
                         dimension)

    def forward(self, input, coordinates=None):
        assert isinstance(input, SparseTensor)
        assert input.D == self.dimension
        # (by keyword: positionally the flag would land in the tensor_stride slot of
        # MinkowskiSparseTensor.py:754-759 and insert a map with tensor stride [0, ...], the origin map's key)
        out_coordinate_map_key = _get_coordinate_map_key(
            input, coordinates, expand_coordinates=self.kernel_generator.expand_coordinates)
        outfeat = self.pooling.apply(input.F, self.pooling_mode, self.kernel_generator, input.coordinate_map_key,
                                     out_coordinate_map_key, input._manager)
        return SparseTensor(outfeat, coordinate_map_key=out_coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)


class MinkowskiGlobalPoolingFunction(Function):
    @staticmethod
    def forward(ctx, input_features, pooling_mode, in_coordinate_map_key, out_coordinate_map_key=None,
                coordinate_manager=None):
        if out_coordinate_map_key is None:
            out_coordinate_map_key = _host.key_like(in_coordinate_map_key)
        input_features = input_features.contiguous()
        ctx.input_features = input_features
        ctx.misc = (pooling_mode, in_coordinate_map_key, out_coordinate_map_key, coordinate_manager)
        fw_fn = get_minkowski_function("GlobalPoolingForward", input_features, in_coordinate_map_key)
        out_feat, num_nonzero = fw_fn(input_features, pooling_mode, in_coordinate_map_key, out_coordinate_map_key,
                                      coordinate_manager._manager)
        ctx.num_nonzero = num_nonzero
        return out_feat

    @staticmethod
    def backward(ctx, grad_out_feat):
        grad_out_feat = grad_out_feat.contiguous()
        pooling_mode, in_key, out_key, coordinate_manager = ctx.misc
        bw_fn = get_minkowski_function("GlobalPoolingBackward", grad_out_feat, in_key)
        grad_in_feat = bw_fn(ctx.input_features, grad_out_feat, ctx.num_nonzero, pooling_mode, in_key, out_key,
                             coordinate_manager._manager)
        return grad_in_feat, None, None, None, None


class MinkowskiGlobalPooling(Module):
    """Pool all features of each batch index into one row (MinkowskiPooling.py:660-716)."""

    def __init__(self, mode=PoolingMode.GLOBAL_AVG_POOLING_PYTORCH_INDEX):
        super().__init__()
        assert isinstance(mode, PoolingMode), f"Mode must be an instance of PoolingMode. mode={mode}"
        self.pooling_mode = mode
        self.pooling = MinkowskiGlobalPoolingFunction

    def forward(self, input, coordinates=None):
        assert isinstance(input, SparseTensor)
        out_coordinate_map_key = _get_coordinate_map_key(input, coordinates)
        output = self.pooling.apply(input.F, self.pooling_mode, input.coordinate_map_key, out_coordinate_map_key,
                                    input._manager)
        return SparseTensor(output, coordinate_map_key=out_coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)

    def __repr__(self):
        return self.__class__.__name__ + f"(mode={str(self.pooling_mode)})"


class MinkowskiGlobalSumPooling(MinkowskiGlobalPooling):
    def __init__(self, mode=PoolingMode.GLOBAL_SUM_POOLING_PYTORCH_INDEX):
        super().__init__(mode=mode)


class MinkowskiGlobalAvgPooling(MinkowskiGlobalPooling):
    def __init__(self, mode=PoolingMode.GLOBAL_AVG_POOLING_PYTORCH_INDEX):
        super().__init__(mode=mode)


class MinkowskiGlobalMaxPooling(MinkowskiGlobalPooling):
    def __init__(self, mode=PoolingMode.GLOBAL_MAX_POOLING_PYTORCH_INDEX):
        super().__init__(mode=mode)
