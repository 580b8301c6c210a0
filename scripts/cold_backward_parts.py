import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["ME_AMD_HOST"] = "python"
import torch
torch.zeros(1, device="cuda").item()
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB
from bench import make_scene
dev = torch.device("cuda:0")
coords = make_scene(100000, 70, 0).to(dev)
feats = torch.rand(100000, 64, device=dev)
w = torch.rand(27, 64, 128, device=dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t):.2f} ms", flush=True); return r
y = T("forward 1", lambda: MEB._conv_forward(feats, w, km, "mfma"))
g = torch.ones_like(y)
T("dgrad 1 (plan + launch)", lambda: MEB._conv_target(g, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))
T("dgrad 2", lambda: MEB._conv_target(g, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))
T("wgrad 1", lambda: MEB._conv_backward(feats, g, w, km, "mfma", need_grad_in=False))
T("wgrad 2", lambda: MEB._conv_backward(feats, g, w, km, "mfma", need_grad_in=False))
T("wgrad 3", lambda: MEB._conv_backward(feats, g, w, km, "mfma", need_grad_in=False))
x = feats.clone().requires_grad_(True)
import torch.autograd
T("autograd backward of a trivial graph 1", lambda: (x * 2).sum().backward())
T("autograd backward of a trivial graph 2", lambda: (x * 2).sum().backward())
