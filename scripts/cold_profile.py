"""MinkUNet34C steps with every coordinate map / kernel map / tile plan rebuilt (MODE=cold, a new scene per
iteration as in training) or cached (MODE=warm); run under rocprofv3 --kernel-trace --stats to split the cold
overhead into map-kernel GPU time and host-synchronisation bubbles.  Prints wall ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "cold")
steps = int(os.environ.get("STEPS", "10"))
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = MU.synthetic_scene(200000, seed=0).to(dev)
feats = torch.rand(coords.shape[0], 3).to(dev).to(dt)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
ME.set_map_prefetch(os.environ.get("PREFETCH", "0") != "0")   # replay the previous scene's map requests at creation
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
crit = MU.cross_entropy
def step(x):
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x).F.float(), labels)
    loss.backward()
    opt.step()
x = ME.SparseTensor(feats, coords)
if mode == "pipelined":
    # the next scene's maps are built on a side stream while the GPU runs the current backward pass: the build's host
    # read-backs wait for the side stream only, its small kernels share the GPU with the backward kernels
    ME.set_map_prefetch(True)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()

    def next_scene():
        with torch.cuda.stream(side):
            t = ME.SparseTensor(feats, coords)          # insert + replay of the previous scene's build requests
        t.coordinate_manager.record_stream(main)        # built on `side`, used (and later freed) under `main`
        ev = torch.cuda.Event()
        ev.record(side)
        return t, ev

    def pstep(pending):
        t, ev = pending
        main.wait_event(ev)
        opt.zero_grad(set_to_none=True)
        loss = crit(net(t).F.float(), labels)
        loss.backward()
        nxt = next_scene()
        opt.step()
        return nxt

    pending = next_scene()
    for _ in range(3):
        pending = pstep(pending)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        pending = pstep(pending)
    torch.cuda.synchronize()
    print(f"MODE={mode} prefetch={ME.map_prefetch_enabled()} dtype={dt} wall {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step over {steps} steps")
    sys.exit(0)
for _ in range(3):
    step(x if mode == "warm" else ME.SparseTensor(feats, coords))
if os.environ.get("PROFILE", "0") != "0":      # host-side picture of the cold path (cProfile inflates everything ~2x)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step(x if mode == "warm" else ME.SparseTensor(feats, coords))
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(30)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step(x if mode == "warm" else ME.SparseTensor(feats, coords))
torch.cuda.synchronize()
print(f"MODE={mode} prefetch={ME.map_prefetch_enabled()} dtype={dt} wall {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step over {steps} steps")
