import torch, time, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/examples')
import minkunet as MU
dev=torch.device('cuda:0')
net=MU.MinkUNet34C(3,20,D=3).to(dev)
for p in net.parameters(): p.grad=torch.randn_like(p)
for kw in ({}, {"foreach":True}, {"fused":True}):
    try:
        opt=torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, **kw)
        for _ in range(3): opt.step()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): opt.step()
        torch.cuda.synchronize(); print(kw, f"{(time.perf_counter()-t0)/50*1e3:.3f} ms/step")
    except Exception as e: print(kw, "failed", e)
