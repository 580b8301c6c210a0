"""Squeeze-and-excitation residual blocks (reference: MinkowskiEngine/modules/senet_block.py: SELayer :33-50,
SEBasicBlock :53-90, SEBottleneck :93-137): the residual branch is rescaled per channel by a gate computed from the
scene's globally pooled features — global pooling, two linear layers, a sigmoid, broadcast multiplication.  The one
module-level consumer of the global pooling + broadcast operators next to a convolution stack.

The reference's blocks forward `D=D` to BasicBlock / Bottleneck, whose argument is called `dimension`
(resnet_block.py:35-43) — they cannot be constructed there as written; both spellings are accepted here."""
import torch.nn as nn

from ..broadcast import MinkowskiBroadcastMultiplication
from ..layers import MinkowskiLinear, MinkowskiReLU, MinkowskiSigmoid
from ..pooling import MinkowskiGlobalPooling
from .resnet_block import BasicBlock, Bottleneck, _add


class SELayer(nn.Module):
    def __init__(self, channel, reduction=16, D=-1):
        super().__init__()
        self.fc = nn.Sequential(MinkowskiLinear(channel, channel // reduction), MinkowskiReLU(inplace=True),
                                MinkowskiLinear(channel // reduction, channel), MinkowskiSigmoid())
        self.pooling = MinkowskiGlobalPooling()
        self.broadcast_mul = MinkowskiBroadcastMultiplication()

    def forward(self, x):
        return self.broadcast_mul(x, self.fc(self.pooling(x)))


class SEBasicBlock(BasicBlock):
    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, reduction=16, D=-1, dimension=-1):
        D = D if D > 0 else dimension
        super().__init__(inplanes, planes, stride=stride, dilation=dilation, downsample=downsample, dimension=D)
        self.se = SELayer(planes, reduction=reduction, D=D)

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.se(self.norm2(self.conv2(out)))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(_add(out, skip))


class SEBottleneck(Bottleneck):
    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, D=3, reduction=16, dimension=-1):
        D = D if D > 0 else dimension
        super().__init__(inplanes, planes, stride=stride, dilation=dilation, downsample=downsample, dimension=D)
        self.se = SELayer(planes * self.expansion, reduction=reduction, D=D)

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.relu(self.norm2(self.conv2(out)))
        out = self.se(self.norm3(self.conv3(out)))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(_add(out, skip))
