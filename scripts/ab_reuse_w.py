"""A/B inside one process (TUNING build, scripts/build_debug.sh: the default build reads the variable once): weight
registers reused for the next batch of the same offset (ME_AMD_X3_REUSE_W) against a reload per batch — split fp32 tile kernel, forward / dgrad of the config-2 scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
for cin, cout in ((64, 128), (128, 64), (128, 128)):
    coords = make_scene(100000, 70, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, cin, device=dev) - 0.5
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    gy = torch.rand(100000, cout, device=dev) - 0.5
    outs, times = {}, {}
    for rnd in range(6):
        for v in (("1", "0") if rnd % 2 == 0 else ("0", "1")):
            os.environ["ME_AMD_X3_REUSE_W"] = v
            outs[v] = (MEB._conv_forward(x, w, km, "mfma").clone(),
                       MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True).clone())
            MEB.KERNEL_TIMER = MEB.KernelTimer()
            for _ in range(20):
                MEB._conv_forward(x, w, km, "mfma")
                MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True)
            torch.cuda.synchronize()
            sm = MEB.KERNEL_TIMER.summary()
            t = (sm["conv_forward"][1] * 1e3, sm["conv_dgrad"][1] * 1e3)
            times.setdefault(v, []).append(t)
            MEB.KERNEL_TIMER = None
    def med(v, i):
        a = sorted(t[i] for t in times[v])
        return a[len(a) // 2]
    print(f"{cin}->{cout}: reload fwd {med('0', 0):.1f} dgrad {med('0', 1):.1f} us; reuse fwd {med('1', 0):.1f} dgrad {med('1', 1):.1f} us "
          f"(median of 6 rounds of 20); bit-identical: {all(torch.equal(outs['0'][i], outs['1'][i]) for i in (0, 1))}")
