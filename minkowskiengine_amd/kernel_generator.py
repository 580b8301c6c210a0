"""KernelGenerator: kernel shape bookkeeping of a layer (reference:
MinkowskiEngine/MinkowskiKernelGenerator.py:245-345)."""
from functools import reduce

import torch

from .backend import RegionType
from .common import convert_to_int_list


def get_kernel_volume(region_type, kernel_size, region_offset, axis_types, dimension):
    """MinkowskiKernelGenerator.py:36-101 / src/kernel_region.hpp:250-270."""
    if region_type == RegionType.HYPER_CUBE:
        assert all(k > 0 for k in kernel_size), "kernel_size must be positive"
        return int(reduce(lambda a, b: a * b, kernel_size, 1))
    if region_type == RegionType.HYPER_CROSS:
        assert all(k > 0 for k in kernel_size), "kernel_size must be positive"
        assert all(k % 2 == 1 for k in kernel_size), "kernel_size must be odd for region_type HYPER_CROSS"
        return int(sum(k - 1 for k in kernel_size) + 1)
    if region_type == RegionType.CUSTOM:
        assert region_offset is not None and region_offset.numel() > 0, \
            "region_offset must be non empty when region_type is CUSTOM"
        assert region_offset.size(1) == dimension
        return int(region_offset.size(0))
    raise NotImplementedError()


class KernelGenerator:
    __slots__ = ("cache", "kernel_size", "kernel_stride", "kernel_dilation", "region_type", "region_offsets",
                 "axis_types", "dimension", "kernel_volume", "requires_strided_coordinates",
                 "expand_coordinates")

    def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False,
                 region_type=RegionType.HYPER_CUBE, region_offsets=None, expand_coordinates=False,
                 axis_types=None, dimension=-1):
        assert dimension > 0
        assert isinstance(region_type, RegionType)
        self.cache = {}
        self.kernel_size = convert_to_int_list(kernel_size, dimension)
        self.kernel_stride = convert_to_int_list(stride, dimension)
        self.kernel_dilation = convert_to_int_list(dilation, dimension)
        self.region_type = region_type
        self.region_offsets = region_offsets if region_offsets is not None else torch.IntTensor()
        self.axis_types = axis_types
        self.dimension = dimension
        self.kernel_volume = get_kernel_volume(region_type, self.kernel_size, region_offsets, axis_types,
                                               dimension)
        # NB the reference's name is a misnomer: it is True when ALL strides are 1
        # (MinkowskiKernelGenerator.py:307-309); kept for drop-in behaviour.
        self.requires_strided_coordinates = all(s == 1 for s in self.kernel_stride)
        self.expand_coordinates = expand_coordinates

    def __repr__(self):
        return (f"{self.__class__.__name__}(kernel_size={self.kernel_size}, kernel_stride={self.kernel_stride}, "
                f"kernel_dilation={self.kernel_dilation}, region_type={self.region_type}, "
                f"expand_coordinates={self.expand_coordinates}, dimension={self.dimension})")
