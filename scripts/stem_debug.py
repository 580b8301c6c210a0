import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import minkowskiengine_amd as ME
from minkowskiengine_amd import _lib
from helpers import make_cloud
lib = _lib.load()
dev = torch.device("cuda:0")
for n, cin, cout, ks in ((100, 3, 32, 3), (3000, 3, 32, 3), (6000, 3, 32, 5)):
    coords = make_cloud(n, 14, 3, seed=1).to(dev)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, dimension=3).to(dev)
    x = ME.SparseTensor(torch.rand(coords.shape[0], cin, device=dev).to(torch.bfloat16), coords)
    outs = []
    for mode in (0, 1):
        lib.me_debug_set_stem(mode, 2)
        with torch.no_grad():
            y = conv(x)
        torch.cuda.synchronize()
        outs.append(y.F.float())
        print(n, cin, cout, ks, "mode", mode, "ok", flush=True)
    print("max diff", (outs[0] - outs[1]).abs().max().item(), "max", outs[0].abs().max().item(), flush=True)
