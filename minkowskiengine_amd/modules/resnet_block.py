"""Residual blocks of the MinkUNet / ResNet families (reference: MinkowskiEngine/modules/resnet_block.py:
BasicBlock :32-77, Bottleneck :80-135).  Two (three) k=3 (k=1,3,1) sparse convolutions with batch norm, ReLU
and an additive skip; `downsample` adapts the skip when stride or width change."""
import torch.nn as nn

from ..convolution import MinkowskiConvolution
from ..layers import MinkowskiBatchNorm, MinkowskiReLU
from ..sparse_tensor import SparseTensor


def _add(a, b):
    """Feature-wise sum of two sparse tensors on the same coordinate map."""
    assert a.coordinate_map_key == b.coordinate_map_key, "residual add needs a shared coordinate map"
    return SparseTensor(a.F + b.F, coordinate_map_key=a.coordinate_map_key, coordinate_manager=a.coordinate_manager)


def _norm_add_relu(norm, x, skip, relu):
    """relu(norm(x) + skip): one fused kernel per direction when `norm` is this package's batch norm in training mode
    (layers.MinkowskiBatchNorm.forward_residual), the three separate operators otherwise."""
    if type(norm) is MinkowskiBatchNorm and type(relu) is MinkowskiReLU:
        return norm.forward_residual(x, skip, relu=True)
    return relu(_add(norm(x), skip))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.norm1.fuse_relu = True   # forward() rectifies norm1's output directly
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        skip = x if self.downsample is None else self.downsample(x)
        return _norm_add_relu(self.norm2, self.conv2(out), skip, self.relu)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.norm1.fuse_relu = self.norm2.fuse_relu = True   # forward() rectifies their outputs directly
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.relu(self.norm2(self.conv2(out)))
        skip = x if self.downsample is None else self.downsample(x)
        return _norm_add_relu(self.norm3, self.conv3(out), skip, self.relu)
