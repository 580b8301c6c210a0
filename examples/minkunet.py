"""MinkUNet family on the MI355X engine (reference: examples/minkunet.py:35-245, examples/resnet.py:38-135).

A 4-level sparse U-Net: stem k=5 conv, four (k=2 s=2 down-conv + residual stage) encoder levels, four
(k=2 s=2 transposed conv + skip concatenation + residual stage) decoder levels, and a k=1 classifier.
The architecture table (PLANES / LAYERS / BLOCK) is the reference's; the code is written for this package.

    python examples/minkunet.py            # one forward/backward of MinkUNet34C on a synthetic scene (GPU)
"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minkowskiengine_amd as ME  # noqa: E402
from minkowskiengine_amd.modules.resnet_block import BasicBlock, Bottleneck  # noqa: E402


class MinkUNetBase(nn.Module):
    BLOCK = None
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
    INIT_DIM = 32

    def __init__(self, in_channels, out_channels, D=3):
        super().__init__()
        assert self.BLOCK is not None
        self.D = D
        P, L, E = self.PLANES, self.LAYERS, self.BLOCK.expansion
        conv = lambda i, o, k, s=1: ME.MinkowskiConvolution(i, o, kernel_size=k, stride=s, dimension=D)
        up = lambda i, o: ME.MinkowskiConvolutionTranspose(i, o, kernel_size=2, stride=2, dimension=D)

        # Module names are the reference's (examples/minkunet.py:50-123: conv{i}p{stride}s2 / bn{i} / block{i},
        # convtr{i}p{stride}s2 / bntr{i}), so a reference state_dict loads unchanged; `p` = tensor stride of the
        # layer's input.
        self.inplanes = self.INIT_DIM
        self.conv0p1s1, self.bn0 = conv(in_channels, self.inplanes, 5), ME.MinkowskiBatchNorm(self.inplanes)
        fused = [self.bn0]
        # encoder: tensor stride 1 -> 2 -> 4 -> 8 -> 16
        self._down, self._up = [], []
        for level in range(4):
            i, ts = level + 1, 2 ** level
            names = (f"conv{i}p{ts}s2", f"bn{i}", f"block{i}")
            setattr(self, names[0], conv(self.inplanes, self.inplanes, 2, 2))
            setattr(self, names[1], ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, names[2], self._make_layer(P[level], L[level]))
            self._down.append(names)
            fused.append(getattr(self, names[1]))
        # decoder: transposed conv back up, concatenate the encoder feature of that stride, residual stage
        skip_planes = [P[2] * E, P[1] * E, P[0] * E, self.INIT_DIM]
        for level in range(4):
            i, ts = level + 4, 2 ** (4 - level)
            names = (f"convtr{i}p{ts}s2", f"bntr{i}", f"block{i + 1}")
            setattr(self, names[0], up(self.inplanes, P[4 + level]))
            setattr(self, names[1], ME.MinkowskiBatchNorm(P[4 + level]))
            self.inplanes = P[4 + level] + skip_planes[level]
            setattr(self, names[2], self._make_layer(P[4 + level], L[4 + level]))
            self._up.append(names)
            fused.append(getattr(self, names[1]))
        self.final = ME.MinkowskiConvolution(P[7] * E, out_channels, kernel_size=1, bias=True, dimension=D)
        self.relu = ME.MinkowskiReLU(inplace=True)
        # every one of these batch norms feeds self.relu directly in forward(): let the batch-norm kernels rectify
        for bn in fused:
            bn.fuse_relu = True
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, (ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose)):
                nn.init.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")   # examples/resnet.py:80-82
            elif isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        E = self.BLOCK.expansion
        downsample = None
        if stride != 1 or self.inplanes != planes * E:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, planes * E, kernel_size=1, stride=stride, dimension=self.D),
                ME.MinkowskiBatchNorm(planes * E))
        layers = [self.BLOCK(self.inplanes, planes, stride=stride, downsample=downsample, dimension=self.D)]
        self.inplanes = planes * E
        layers += [self.BLOCK(self.inplanes, planes, dimension=self.D) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        out = self.relu(self.bn0(self.conv0p1s1(x)))
        skips = [out]                                   # tensor stride 1
        for conv, bn, block in self._down:
            out = self.relu(getattr(self, bn)(getattr(self, conv)(out)))
            out = getattr(self, block)(out)
            skips.append(out)                           # strides 2, 4, 8, 16
        for level, (conv, bn, block) in enumerate(self._up):
            out = self.relu(getattr(self, bn)(getattr(self, conv)(out)))
            out = ME.cat(out, skips[3 - level])
            out = getattr(self, block)(out)
        return self.final(out)


class MinkUNet14(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class MinkUNet18(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class MinkUNet34(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class MinkUNet50(MinkUNetBase):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class MinkUNet14A(MinkUNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet18A(MinkUNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet34A(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 64)


class MinkUNet34B(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 32)


class MinkUNet34C(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


def cross_entropy(logits, labels):
    """Mean cross-entropy of [n, classes] logits, the value and gradient of torch.nn.CrossEntropyLoss() — composed of
    logsumexp and gather because torch's nll_loss kernels reduce a 200k-row loss in a single workgroup on ROCm (230 us
    forward + 150 us backward per step in the round-1 profile of the config-3 step; this composition takes ~60 us)."""
    z = logits.float()
    return (torch.logsumexp(z, 1) - z.gather(1, labels.view(-1, 1)).squeeze(1)).mean()


def synthetic_scene(n=200000, grid=400, seed=0, batch_index=0):
    """voxels on a union of axis-aligned planes in a grid^3 volume (9 planes x 30k draws,
    unique, first n) — a deterministic, network-free stand-in for an indoor scan."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for p in range(9):
        axis = p % 3
        pts = torch.randint(0, grid, (30000 * max(1, n // 200000 + (n % 200000 > 0)), 3), generator=g)
        pts[:, axis] = int(torch.randint(grid // 8, grid - grid // 8, (1,), generator=g))
        parts.append(pts)
    pts = torch.unique(torch.cat(parts), dim=0)
    pts = pts[torch.randperm(pts.shape[0], generator=g)][:n]
    return torch.cat([torch.full((pts.shape[0], 1), batch_index, dtype=torch.long), pts], 1).int().contiguous()


if __name__ == "__main__":
    import time
    assert torch.cuda.is_available(), "needs a GPU"
    dev = torch.device("cuda:0")
    n = int(os.environ.get("POINTS", "200000"))
    coords = synthetic_scene(n)
    feats = torch.rand(coords.shape[0], 3)
    net = MinkUNet34C(3, 20, D=3).to(dev)
    print("parameters:", sum(p.numel() for p in net.parameters()))
    x = ME.SparseTensor(feats.to(dev), coords.to(dev))
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = net(x)
        y.F.sum().backward()
        torch.cuda.synchronize()
        print(f"iteration {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms, voxels {coords.shape[0]}, "
              f"out {tuple(y.F.shape)}")
