"""Where the host time of one convolution layer goes (perf_counter around pieces, 2000 repetitions, no GPU sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
coords = make_scene(20000, 40, 0).to(dev)
x = ME.SparseTensor(torch.rand(20000, 64, device=dev).bfloat16(), coords, requires_grad=True)
conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(dev)
g = torch.ones(20000, 128, device=dev, dtype=torch.bfloat16)
y = conv(x); y.F.backward(g)
mgr = x.coordinate_manager._manager
km = mgr._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3]*3, [1]*3, [1]*3, ME.RegionType.HYPER_CUBE, None, False, False)
lib = _lib.load()
def T(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t
F = x.F.detach(); W = conv.kernel.detach()
print("conv(x) forward (module call)          %6.1f us" % T(lambda: conv(x)))
with torch.no_grad():
    print("conv(x) forward, no_grad               %6.1f us" % T(lambda: conv(x)))
print("  _conv_forward (pack + target)        %6.1f us" % T(lambda: MEB._conv_forward(F, W, km)))
print("  _conv_backward (dgrad + wgrad)       %6.1f us" % T(lambda: MEB._conv_backward(F, g, W, km)))
print("  ConvolutionForwardGPU                %6.1f us" % T(lambda: MEB.ConvolutionForwardGPU(F, W, [3]*3, [1]*3, [1]*3, ME.RegionType.HYPER_CUBE, None, False, ME.ConvolutionMode.DEFAULT, x.coordinate_map_key, ME.CoordinateMapKey(4), mgr)))
print("  SparseTensor(F, key, manager)        %6.1f us" % T(lambda: ME.SparseTensor(F, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)))
print("  torch.empty x2                       %6.1f us" % T(lambda: (torch.empty((20000, 128), dtype=torch.bfloat16, device=dev), torch.empty(221184, dtype=torch.bfloat16, device=dev))))
print("  _stream                              %6.1f us" % T(lambda: MEB._stream(dev)))
print("  mgr._kernel_map lookup               %6.1f us" % T(lambda: mgr._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3]*3, [1]*3, [1]*3, ME.RegionType.HYPER_CUBE, None, False, False)))
print("  mgr.stride(key, [1,1,1])             %6.1f us" % T(lambda: mgr.stride(x.coordinate_map_key, [1, 1, 1])))
def fb():
    conv.kernel.grad = None; x.F.grad = None
    conv(x).F.backward(g)
print("forward + backward                     %6.1f us" % T(fb, 1000))
