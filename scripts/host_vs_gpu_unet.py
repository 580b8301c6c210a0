"""Is a MinkUNet34C step host- or GPU-bound?  Enqueue time (no sync) vs wall time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
for dts in ("bf16", "f32"):
    dt = torch.bfloat16 if dts == "bf16" else torch.float32
    coords = MU.synthetic_scene(200000, seed=0).to(dev)
    x = ME.SparseTensor(torch.rand(coords.shape[0], 3).to(dev).to(dt), coords)
    net = MU.MinkUNet34C(3, 20, D=3).to(dev)
    labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    def step():
        opt.zero_grad(set_to_none=True)
        crit(net(x).F.float(), labels).backward()
        opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{dts}: enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
