"""Forward / dgrad of the config-2 layer under a debug variant against the default kernels: bit-identity + timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
variants = [int(v) for v in os.environ.get("VARIANTS", "0,31,30").split(",")]
extent = int(os.environ.get("EXTENT", "70"))     # 70: the dense headline scene (P ~ 8.4 N); 215: the sparse one (P ~ 1.27 N)
for cin, cout in ((64, 128), (128, 64), (128, 128), (64, 64)):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, cin, device=dev) - 0.5
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    outs, times = {}, {}
    gy = torch.rand(100000, cout, device=dev) - 0.5
    # (every variant is timed in several rounds, the order reversed between rounds: clocks drift over a session)
    for rnd in range(4):
        for v in (variants if rnd % 2 == 0 else variants[::-1]):
            _lib.check(lib.me_debug_set_conv_variant(v))
            outs[v] = (MEB._conv_forward(x, w, km, "mfma").clone(),
                       MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True).clone())
            MEB.KERNEL_TIMER = MEB.KernelTimer()
            for _ in range(20):
                MEB._conv_forward(x, w, km, "mfma")
                MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True)
            torch.cuda.synchronize()
            sm = MEB.KERNEL_TIMER.summary()
            t = (sm["conv_forward"][1] * 1e3, sm["conv_dgrad"][1] * 1e3)
            times[v] = t if v not in times or rnd == 0 else (min(times[v][0], t[0]), min(times[v][1], t[1]))
            MEB.KERNEL_TIMER = None
    lib.me_debug_set_conv_variant(0)
    print(f"{cin}->{cout}: " + ", ".join(f"variant {v}: fwd {times[v][0]:.1f} dgrad {times[v][1]:.1f} us" for v in variants) +
          "; bit-identical to variant %d: %s" % (variants[0], all(torch.equal(outs[variants[0]][i], outs[v][i]) for v in variants for i in (0, 1))))
