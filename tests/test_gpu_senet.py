"""Network-level test of global pooling + broadcast multiplication through their one module-level
consumer, the squeeze-and-excitation residual block (modules/senet_block.py; reference MinkowskiEngine/modules/
senet_block.py:33-137): a small SE-ResNet on a batch of scenes of different sizes, forward and backward, against the
same network whose SE gates are computed with plain torch index arithmetic per batch index."""
import pytest
import torch

from helpers import make_cloud

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]


def _manual_se(se, t_feats, batch_index, n_batch):
    """sigmoid(fc(mean over the rows of each batch)) broadcast back over the rows: torch ops only"""
    c = t_feats.shape[1]
    sums = torch.zeros(n_batch, c, dtype=t_feats.dtype, device=t_feats.device).index_add_(0, batch_index, t_feats)
    cnt = torch.bincount(batch_index, minlength=n_batch).to(t_feats.dtype).unsqueeze(1)
    pooled = sums / cnt
    l1, l2 = se.fc[0].linear, se.fc[2].linear
    gate = torch.sigmoid(torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(pooled, l1.weight, l1.bias)),
                                                    l2.weight, l2.bias))
    return t_feats * gate[batch_index]


@pytest.mark.parametrize("block", ["basic", "bottleneck"])
def test_se_blocks_match_a_torch_composition_of_their_gates(device, block):
    import minkowskiengine_amd as ME
    from minkowskiengine_amd.modules import SEBasicBlock, SEBottleneck
    torch.manual_seed(3)
    coords = make_cloud(1500, 12, 3, seed=5, batch=3)          # three scenes in one batch
    coords = coords[torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(1))[:3800]].contiguous()
    feats = torch.rand(coords.shape[0], 8, generator=torch.Generator().manual_seed(2))
    stem = ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3).to(device)
    if block == "basic":
        blk = SEBasicBlock(32, 32, reduction=4, D=3).to(device)
    else:
        down = torch.nn.Sequential(ME.MinkowskiConvolution(32, 64, kernel_size=1, dimension=3),
                                   ME.MinkowskiBatchNorm(64)).to(device)
        blk = SEBottleneck(32, 16, downsample=down, reduction=4, D=3).to(device)
    head = torch.nn.Sequential(ME.MinkowskiGlobalPooling(), ME.MinkowskiLinear(blk.se.fc[2].linear.out_features, 5)).to(device)
    params = list(stem.parameters()) + list(blk.parameters()) + list(head.parameters())
    target = torch.rand(3, 5, generator=torch.Generator().manual_seed(4)).to(device)

    def run(manual):
        for p in params:
            p.grad = None
        x = ME.SparseTensor(feats.to(device), coords.to(device))
        h = stem(x)
        bidx = h.C[:, 0].long()
        if not manual:
            out = blk(h)
        else:
            # the block's own layers, the SE layer replaced by the torch composition
            o = blk.relu(blk.norm1(blk.conv1(h)))
            if block == "basic":
                t = blk.norm2(blk.conv2(o))
            else:
                t = blk.norm3(blk.conv3(blk.relu(blk.norm2(blk.conv2(o)))))
            gated = ME.SparseTensor(_manual_se(blk.se, t.F, bidx, 3), coordinate_map_key=t.coordinate_map_key,
                                    coordinate_manager=t.coordinate_manager)
            skip = h if blk.downsample is None else blk.downsample(h)
            out = blk.relu(ME.SparseTensor(gated.F + skip.F, coordinate_map_key=t.coordinate_map_key,
                                           coordinate_manager=t.coordinate_manager))
        y = head(out)
        assert y.F.shape == (3, 5)
        ((y.F - target) ** 2).sum().backward()
        return out.F.detach().clone(), y.F.detach().clone(), [p.grad.detach().clone() for p in params]

    f_mod, y_mod, g_mod = run(False)
    f_man, y_man, g_man = run(True)
    assert torch.allclose(f_mod, f_man, rtol=1e-4, atol=1e-5), float((f_mod - f_man).abs().max())
    assert torch.allclose(y_mod, y_man, rtol=1e-4, atol=1e-5)
    for a, b, p in zip(g_mod, g_man, params):
        assert torch.allclose(a, b, rtol=2e-3, atol=1e-5 + 1e-4 * float(b.abs().max())), (tuple(p.shape), float((a - b).abs().max()))
