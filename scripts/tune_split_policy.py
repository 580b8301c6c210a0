"""fp32 forward / dgrad: the bf16x6 split kernel (k_conv_tile_f32x3) against the fp32-MFMA kernel (k_conv_tile_f32) per
(map density, channel shape): the MinkUNet scene at strides 1..16, the config-2 scene and its sparse variant, the
config-5 scene.  Prints microseconds and the statistic the dispatch rule uses (pairs per (tile, offset) item)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
from examples.minkunet import synthetic_scene

dev = torch.device("cuda:0")
lib = _lib.load()


def time_it(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(tag, km, n_in, n_out, shapes):
    for cin, cout in shapes:
        x = torch.rand(n_in, cin, device=dev)
        gy = torch.rand(n_out, cout, device=dev)
        w = torch.rand(km.volume, cin, cout, device=dev) - 0.5
        res = {}
        for split in (True, False):
            MEB._F32_SPLIT = split
            res[split] = (time_it(lambda: MEB._conv_forward(x, w, km, "mfma")),
                          time_it(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, name="d", transposed=True)))
        T = MEB.plan_config(n_out, km.volume, km.n_pairs, cin, cout, False, True)[0]
        tiles = -(-n_out // T)
        per_item = (km.n_pairs - min(n_in, n_out)) / max(1, (km.volume - 1) * tiles)
        print(f"{tag:26s} {cin:3d}->{cout:3d} n {n_out:6d} K {km.volume:3d} pairs/n {km.n_pairs / n_out:5.2f} T {T:3d} "
              f"pairs/item {per_item:6.1f} | fwd x3 {res[True][0]:7.1f} mfma {res[False][0]:7.1f} ({res[False][0] / res[True][0]:4.2f}x)"
              f" | dgrad x3 {res[True][1]:7.1f} mfma {res[False][1]:7.1f} ({res[False][1] / res[True][1]:4.2f}x)", flush=True)


SHAPES = [(32, 32), (64, 64), (96, 96), (128, 96), (128, 128), (256, 256)]
# MinkUNet scene, maps at strides 1 .. 16
coords = synthetic_scene(200000).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
for level in range(5):
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    n = km.n_out
    shapes = SHAPES if level < 2 else SHAPES[:2] + SHAPES[4:]
    run(f"unet scene stride {2 ** level}", km, n, n, shapes)
    if level < 4:
        nk = mgr.stride(key, [2, 2, 2])
        kd = mgr._kernel_map(key, nk, [2] * 3, [2] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        run(f"  down k2s2 from {2 ** level}", kd, kd.n_in, kd.n_out, [(32, 32), (128, 128)])
        key = nk
for extent in (70, 215):
    c = make_scene(100000, extent, 0).to(dev)
    m = MEB.CoordinateMapManagerGPU_c10()
    k, _ = m.insert_and_map(c, [1, 1, 1], "")
    km = m._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    run(f"config 2 scene extent {extent}", km, 100000, 100000, [(64, 128), (128, 64), (32, 64)])
c = make_scene(400000, (100, 100, 100, 8), 0, D=4).to(dev)
m = MEB.CoordinateMapManagerGPU_c10()
k, _ = m.insert_and_map(c, [1, 1, 1, 1], "")
km = m._kernel_map(k, k, [3] * 4, [1] * 4, [1] * 4, MEB.RegionType.HYPER_CUBE, None, False, False)
run("config 5 scene (4-D)", km, 400000, 400000, [(32, 64), (64, 32)])
