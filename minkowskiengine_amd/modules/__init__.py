from .resnet_block import BasicBlock, Bottleneck  # noqa: F401
from .senet_block import SEBasicBlock, SEBottleneck, SELayer  # noqa: F401
