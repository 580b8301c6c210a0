"""Which tile plans does a MinkUNet34C bf16 step build? (python host: kernel map stores)"""
import os, sys
os.environ["ME_AMD_HOST"] = "python"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
coords = MU.synthetic_scene(200000, seed=0).to(dev)
x = ME.SparseTensor(torch.rand(coords.shape[0], 3, device=dev).bfloat16(), coords)
y = net(x)
y.F.float().sum().backward()
torch.cuda.synchronize()
mgr = x.coordinate_manager._manager
n = 0
for key, km in mgr._kernel_maps.items():
    ts = key[0][0]
    for name, v in km._store.items():
        if isinstance(name, str) and name.startswith("plan"):
            n += 1
            print("ts", ts, "ks", key[2], "st", key[3], "tr" if key[6] else "  ", "K", km.volume, "n_in", km.n_in, "n_out", km.n_out, name)
print("plans:", n)
