"""Microbenchmark of the row-wise kernel on a K = 1 identity map (scripts/gpu_r06.sh rowwise_bench)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minkowskiengine_amd as ME          # noqa: E402
from minkowskiengine_amd import _lib      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n, cin, cout = [int(v) for v in os.environ.get("SHAPE", "200000,128,96").split(",")]
x = torch.rand(n, cin, device=dev).bfloat16()
w = (torch.rand(1, cin, cout, device=dev) - 0.5)
rows = torch.arange(n, dtype=torch.int32, device=dev)
koffs = torch.tensor([0, n], dtype=torch.int64, device=dev)
elems = lib.me_conv_packed_weight_elems_bf16(1, cin, cout)
packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
_lib.check(lib.me_conv_pack_weights_bf16(w.data_ptr(), 1, 1, cin, cout, 0, packed.data_ptr(), None))
out = torch.empty(n, cout, dtype=torch.bfloat16, device=dev)


def run():
    _lib.check(lib.me_conv_rowwise_bf16(x.data_ptr(), n, cin, packed.data_ptr(), 1, cout, rows.data_ptr(), rows.data_ptr(),
                                        koffs.data_ptr(), n, out.data_ptr(), n, None))


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


t = timeit(run)
gb = (n * cin * 2 + n * cout * 2) / 1e9
print(f"abl={os.environ.get('ME_RW_ABL','0')} G={os.environ.get('ME_AMD_RW_G','-')} rowwise {n}x{cin}->{cout}: {t:.1f} us  {gb / (t * 1e-6):.0f} GB/s")
if os.environ.get("BASE"):
    wb = w[0].bfloat16()
    print(f"  torch mm: {timeit(lambda: torch.mm(x, wb)):.1f} us;  copy of the same bytes: "
          f"{timeit(lambda: out.copy_(x[:, :cout])):.1f} us")
    ref = torch.mm(x.float(), wb.float())
    run()
    torch.cuda.synchronize()
    print("  max err", float((out.float() - ref).abs().max()))
