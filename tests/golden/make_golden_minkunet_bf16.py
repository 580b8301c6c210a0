"""bf16-EMULATING oracle for the whole-network bf16 rows.

The reference has no reduced-precision path (AT_DISPATCH_FLOATING_TYPES, src/convolution_gpu.cu:137-155), so "the
reference in bf16" does not exist.  What exists is the reference's fp32 network (examples/minkunet.py on its Python
package + its CPU operators, oracle/_ref/_C.so) — and the list of places where THIS implementation stores a tensor in
bf16.  This generator runs the reference network in fp32 and rounds (RNE, float32 -> bfloat16 -> float32) at exactly
those storage points, forward AND backward:

  * input features; the weights of every convolution (this implementation rounds the fp32 master weights when it packs
    them: me_conv_pack_weights_bf16); batch-norm weight / bias stay fp32;
  * the output of every convolution / transposed convolution (fp32 accumulation, one rounding: csrc/conv_bf16.hip), with
    a bias added in bf16 behind it (convolution.py: outfeat + bias);
  * the output of every batch norm (csrc/norm.hip k_bn_apply: fp32 arithmetic on the stored bf16 input, one rounding);
    ReLU and concatenation are exact on rounded values;
  * the residual sum `out += residual` (k_bn_apply<SKIP>: the normalised value is rounded, the sum is rounded again);
  * every upstream gradient where it is stored: the gradient arriving at a module's output (the accumulated sum of its
    consumers' contributions, rounded) and the input gradient a convolution / batch norm hands back (rounded BEFORE it
    is accumulated with other branches — a bf16 tensor on the GPU).  Parameter gradients stay fp32 (fp32 accumulation
    of exact products; csrc/conv_bf16.hip k_wgrad_bf16, norm.hip).

Products of bf16 values are exact in fp32, so what remains between this emulation and the HIP path is fp32 summation
order and the roundings that order flips.  Those flips do NOT stay small: a flipped rounding moves ~10^3 downstream sums
by a bf16 ulp of one addend, each of which flips with probability ~1 / sqrt(fan-in) — the set of flipped roundings grows
by an order of magnitude per layer, and two runs that differ by 1e-7 relative in one weight tensor end up with
INDEPENDENT rounding noise.  Measured here (MinkUNet14, 20k voxels): emulation vs the fp32 network 0.31 median relative
L2 over the parameter gradients; emulation vs the same emulation with every weight perturbed by 1e-7 relative: 0.29.  The
logits and the loss are well conditioned (1e-2 of the range, 5e-5), the gradients of a bf16-stored network are not — in
ANY implementation.  So the fixture stores, next to the emulated run, a second emulated run with the weights perturbed by
2^-20 relative, and `noise/<name>` = the relative L2 distance of the two per parameter tensor: the emulation's own
sensitivity.  tests/test_gpu_minkunet.py holds the HIP path to a small multiple of THAT (the way the fp32 fixtures bound
the fp32 path by the reference's own fp32 noise), holds loss and logits tightly, and separately proves that the bf16
gradient noise is UNBIASED (the mean of 16 decorrelated bf16 runs converges to the fp32 gradient): an operator wired to
the wrong tensor, a dropped term or a mis-scaled branch fails the second test, whatever the noise.

  minkunet14_bf16_20k.npz    MinkUNet14, 20k-voxel plane scene, cross-entropy on 5 classes: logits, loss, ALL parameter
                             gradients (sampled: every tensor's first 4096 elements of a fixed stride)
  minkunet34c_bf16_200k.npz  MinkUNet34C on BASELINE configs[2]'s 200k-voxel scene, cross-entropy on 20 classes: the same

Run in the authoring container (needs /root/reference):  python tests/golden/make_golden_minkunet_bf16.py [14] [34c]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref  # noqa: E402

RME = ref.import_reference_package()
from examples.minkunet import MinkUNet14, MinkUNet34C  # noqa: E402

from make_golden_minkunet_weights import config3_inputs, seeded_parameters  # noqa: E402

SAMPLE = 4096


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class RoundBoth(torch.autograd.Function):
    """y = bf16(x) forward, grad_x = bf16(grad_y) backward: a tensor that is STORED in bf16 in both directions"""

    @staticmethod
    def forward(ctx, x):
        return bf16(x)

    @staticmethod
    def backward(ctx, g):
        return bf16(g)


def rewrap(st, feats):
    return RME.SparseTensor(feats, coordinate_map_key=st.coordinate_map_key, coordinate_manager=st.coordinate_manager)


def instrument(net):
    """hooks that put the storage roundings around every convolution / batch norm; `+=` of sparse tensors rounds"""
    convs = (RME.MinkowskiConvolution, RME.MinkowskiConvolutionTranspose)
    handles = []

    def pre(module, args):
        x = args[0]
        return (rewrap(x, RoundBoth.apply(x.F)),) + tuple(args[1:])     # (identity forward: x.F is already rounded)

    def post(module, args, out):
        return rewrap(out, RoundBoth.apply(out.F))

    def conv_pre(module, args):
        module._bias_saved = module.bias
        module.bias = None                                                  # the bias is added in bf16 behind the rounding
        return pre(module, args)

    def conv_post(module, args, out):
        b = module._bias_saved
        module.bias = b
        f = RoundBoth.apply(out.F)
        if b is not None:
            f = RoundBoth.apply(f + RoundBoth.apply(b))
        return rewrap(out, f)

    for m in net.modules():
        if isinstance(m, convs):
            handles.append(m.register_forward_pre_hook(conv_pre))
            handles.append(m.register_forward_hook(conv_post))
        elif isinstance(m, RME.MinkowskiBatchNorm):
            handles.append(m.register_forward_pre_hook(pre))
            handles.append(m.register_forward_hook(post))
    return handles


def _iadd(self, other):
    self._is_same_key(other)
    self._F = RoundBoth.apply(self._F + other.F)
    return self


def sample(t):
    """a fixed-stride sample of at most SAMPLE elements (the tests index the same way)"""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // SAMPLE)
    return f[::step][:SAMPLE].numpy().astype(np.float32)


def emulated(net_cls, n_classes, coords, feats, labels, perturb=0.0):
    """one forward + backward of the instrumented reference network -> (net, logits, loss); perturb: every convolution
    weight times (1 + perturb * N(0, 1)) BEFORE the rounding to bf16 (the noise-floor run)"""
    net = net_cls(3, n_classes, D=3)
    seeded_parameters(net.named_parameters())
    gp = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (RME.MinkowskiConvolution, RME.MinkowskiConvolutionTranspose)):
                if perturb:
                    m.kernel.mul_(1 + perturb * torch.randn(m.kernel.shape, generator=gp))
                m.kernel.copy_(bf16(m.kernel))
    instrument(net)
    saved = RME.SparseTensor.__iadd__
    RME.SparseTensor.__iadd__ = _iadd
    try:
        x = RME.SparseTensor(bf16(feats), coords)
        y = net(x)
        logits = y.F                                        # stored bf16 values, read as float32 by the loss
        loss = (torch.logsumexp(logits, 1) - logits.gather(1, labels.view(-1, 1)).squeeze(1)).mean()
        loss.backward()
    finally:
        RME.SparseTensor.__iadd__ = saved
    return net, logits, loss


def run(net_cls, n_classes, coords, feats, labels, tag):
    net, logits, loss = emulated(net_cls, n_classes, coords, feats, labels)
    net2, logits2, loss2 = emulated(net_cls, n_classes, coords, feats, labels, perturb=2.0 ** -20)
    data = {"loss": np.float64(loss.item()), "labels": labels.numpy().astype(np.int64), "sample": np.int64(SAMPLE),
            "logits_sample": sample(logits), "logits_absmax": np.float64(logits.abs().max().item()),
            "noise/loss": np.float64(abs(loss2.item() - loss.item())),
            "noise/logits": np.float64((logits2 - logits).abs().max().item())}
    names, noise = [], []
    other = dict(net2.named_parameters())
    for n, p in net.named_parameters():
        names.append(n)
        a, b = sample(p.grad).astype(np.float64), sample(other[n].grad).astype(np.float64)
        data["grad/" + n] = a.astype(np.float32)
        data["noise/" + n] = np.float64(np.linalg.norm(b - a) / max(np.linalg.norm(a), 1e-30))
        noise.append(float(data["noise/" + n]))
    data["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **data)
    noise.sort()
    print("saved", tag, "loss", loss.item(), "parameters", len(names), "own noise (rel L2, 2^-20 weight perturbation): median",
          noise[len(noise) // 2], "worst", noise[-1], "logits", float(data["noise/logits"]), flush=True)


def small():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import minkunet as ours   # (scene generator only)
    coords = ours.synthetic_scene(20000, grid=96, seed=5)
    g = torch.Generator().manual_seed(2)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    labels = torch.randint(0, 5, (coords.shape[0],), generator=g)
    run(MinkUNet14, 5, coords, feats, labels, "minkunet14_bf16_20k")


def config3():
    coords, feats, _, _ = config3_inputs()
    labels = torch.randint(0, 20, (coords.shape[0],), generator=torch.Generator().manual_seed(11))
    run(MinkUNet34C, 20, coords, feats, labels, "minkunet34c_bf16_200k")


if __name__ == "__main__":
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    which = sys.argv[1:] or ["14", "34c"]
    if "14" in which:
        small()
    if "34c" in which:
        config3()
