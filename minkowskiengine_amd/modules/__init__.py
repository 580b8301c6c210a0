from .resnet_block import BasicBlock, Bottleneck  # noqa: F401
