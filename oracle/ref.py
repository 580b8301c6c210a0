"""TEST INFRASTRUCTURE ONLY — loader for the reference's own CPU extension (oracle/_ref/_C.so,
built unmodified from /root/reference by oracle/build_ref.py).  The shared object travels to the GPU
box; /root/reference itself does not: bench.py and smoke() use only the compiled operators, and the one test that
needs the reference's Python package on a GPU box finds the copy staged by oracle/stage_ref_package.py."""
import importlib.machinery
import importlib.util
import os

import torch  # noqa: F401  (libtorch must be loaded before the extension)

from . import build_ref

_mod = None


def available():
    return os.path.exists(build_ref.ref_so_path())


def load():
    """-> the pybind module (MinkowskiEngineBackend._C of the reference, CPU_ONLY build)."""
    global _mod
    if _mod is None:
        so = build_ref.ref_so_path()
        if not os.path.exists(so):
            so = build_ref.build()
        loader = importlib.machinery.ExtensionFileLoader("_C", so)
        spec = importlib.util.spec_from_file_location("_C", so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        _mod = mod
    return _mod


class RefConv:
    """One reference CPU convolution layer on fixed coordinates, through the reference's operators
    ConvolutionForwardCPU / ConvolutionBackwardCPU (src/convolution_cpu.cpp:42-203) or their
    transposed twins, with the reference's CoordinateMapManagerCPU."""

    def __init__(self, coords, kernel_size, stride=1, dilation=1, D=None, transpose_from=None, num_threads=None):
        C = load()
        coords = coords.contiguous().int().cpu()
        self.C = C
        self.D = coords.shape[1] - 1 if D is None else D
        D = self.D
        aslist = lambda v: [int(v)] * D if isinstance(v, int) else [int(x) for x in v]
        self.kernel_size, self.stride, self.dilation = aslist(kernel_size), aslist(stride), aslist(dilation)
        nthreads = num_threads if num_threads is not None else min(os.cpu_count() or 1, 20)
        self.manager = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, nthreads)
        self.in_key, (self.unique_map, self.inverse_map) = self.manager.insert_and_map(coords, [1] * D, "")
        # same rule as ConvolutionForwardCPU (src/convolution_cpu.cpp:78-108): out map = stride(in map)
        self.out_key = self.manager.stride(self.in_key, self.stride, "")
        self.empty_offset = torch.IntTensor()

    def forward(self, feats, kernel):
        C = self.C
        return C.ConvolutionForwardCPU(feats, kernel, self.kernel_size, self.stride, self.dilation,
                                       C.RegionType.HYPER_CUBE, self.empty_offset, False, C.ConvolutionMode.DEFAULT,
                                       self.in_key, self.out_key, self.manager)

    def backward(self, feats, grad_out, kernel):
        C = self.C
        return C.ConvolutionBackwardCPU(feats, grad_out, kernel, self.kernel_size, self.stride, self.dilation,
                                        C.RegionType.HYPER_CUBE, self.empty_offset, C.ConvolutionMode.DEFAULT,
                                        self.in_key, self.out_key, self.manager)

    def kernel_map(self):
        C = self.C
        return self.manager.kernel_map(self.in_key, self.out_key, self.kernel_size, self.stride, self.dilation,
                                       C.RegionType.HYPER_CUBE, self.empty_offset, False, False)

    def in_coordinates(self):
        return self.manager.get_coordinates(self.in_key)

    def out_coordinates(self):
        return self.manager.get_coordinates(self.out_key)


class RefPool(RefConv):
    """Reference CPU pooling / broadcast operators on fixed coordinates (src/local_pooling_cpu.cpp,
    src/global_pooling_cpu.cpp, src/broadcast_cpu.cpp)."""

    def _mode(self, name):
        PM = self.C.PoolingMode
        return {"sum": PM.LOCAL_SUM_POOLING, "avg": PM.LOCAL_AVG_POOLING, "max": PM.LOCAL_MAX_POOLING,
                "gsum": PM.GLOBAL_SUM_POOLING_KERNEL, "gavg": PM.GLOBAL_AVG_POOLING_KERNEL,
                "gmax": PM.GLOBAL_MAX_POOLING_KERNEL}[name]

    def pool_forward(self, feats, mode):
        C = self.C
        return C.LocalPoolingForwardCPU(feats, self.kernel_size, self.stride, self.dilation, C.RegionType.HYPER_CUBE,
                                        self.empty_offset, self._mode(mode), self.in_key, self.out_key, self.manager)

    def pool_backward(self, feats, grad_out, aux, mode):
        C = self.C
        return C.LocalPoolingBackwardCPU(feats, grad_out, aux, self.kernel_size, self.stride, self.dilation,
                                         C.RegionType.HYPER_CUBE, self.empty_offset, self._mode(mode), self.in_key,
                                         self.out_key, self.manager)

    def pool_kernel_map(self):
        C = self.C
        return self.manager.kernel_map(self.in_key, self.out_key, self.kernel_size, self.stride, self.dilation,
                                       C.RegionType.HYPER_CUBE, self.empty_offset, False, True)

    def global_forward(self, feats, mode):
        C = self.C
        self.glob_key = C.CoordinateMapKey(self.D + 1)
        return C.GlobalPoolingForwardCPU(feats, self._mode("g" + mode), self.in_key, self.glob_key, self.manager)

    def global_backward(self, feats, grad_out, aux, mode):
        C = self.C
        return C.GlobalPoolingBackwardCPU(feats, grad_out, aux, self._mode("g" + mode), self.in_key, self.glob_key,
                                          self.manager)

    def glob_coordinates(self):
        return self.manager.get_coordinates(self.glob_key)

    def broadcast_forward(self, feats, glob, multiply):
        C = self.C
        op = C.BroadcastMode.ELEMENTWISE_MULTIPLICATION if multiply else C.BroadcastMode.ELEMENTWISE_ADDITON
        return C.BroadcastForwardCPU(feats, glob, op, self.in_key, self.glob_key, self.manager)

    def broadcast_backward(self, feats, glob, grad_out, multiply):
        C = self.C
        op = C.BroadcastMode.ELEMENTWISE_MULTIPLICATION if multiply else C.BroadcastMode.ELEMENTWISE_ADDITON
        return C.BroadcastBackwardCPU(feats, glob, grad_out, op, self.in_key, self.glob_key, self.manager)


def reference_root():
    """Where the reference's Python package is: $ME_REFERENCE_ROOT, /root/reference (the authoring container), or the
    copy staged in oracle/_ref/reference_tree by oracle/stage_ref_package.py (the GPU box: git-ignored scratch that
    travels with the snapshot, like the compiled reference next to it)."""
    env = os.environ.get("ME_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/MinkowskiEngine"):
        return "/root/reference"
    staged = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_tree")
    return staged if os.path.isdir(os.path.join(staged, "MinkowskiEngine")) else "/root/reference"


def package_available():
    """the reference's Python package can only be imported where /root/reference exists (the authoring container)"""
    return available() and os.path.isdir(os.path.join(reference_root(), "MinkowskiEngine"))


def import_reference_package(backend=None):
    """Import the REFERENCE's own Python package (/root/reference/MinkowskiEngine, v0.5.4) on top of `backend`
    as its `MinkowskiEngineBackend._C` — by default the reference's own CPU operators (oracle/_ref/_C.so); the
    drop-in tests pass minkowskiengine_amd.backend instead.  Returns the imported `MinkowskiEngine` module.
    Side effects (this is why callers run it in a subprocess): sys.modules gets MinkowskiEngineBackend, a stub
    `open3d` (examples/resnet.py imports it at module scope), sys.path gets the reference root, and the working
    directory moves to a scratch directory holding an empty 1.ply (examples/resnet.py would download it)."""
    import sys
    import tempfile
    import types
    C = load() if backend is None else backend
    pkg = types.ModuleType("MinkowskiEngineBackend")
    pkg._C = C
    pkg.__path__ = []
    sys.modules["MinkowskiEngineBackend"] = pkg
    sys.modules["MinkowskiEngineBackend._C"] = C
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    if reference_root() not in sys.path:
        sys.path.insert(0, reference_root())
    work = tempfile.mkdtemp()
    os.chdir(work)
    open("1.ply", "w").close()
    import MinkowskiEngine as RME
    return RME


class RefConvStack:
    """TEST / BENCH INFRASTRUCTURE: the convolution layers of a network (a list of layer specs recorded from the
    MI355X run) replayed on the reference's own CPU operators on the same coordinates — the `cpu_baseline` of
    bench.py's MinkUNet workload where the reference's Python package (and so its network class) is absent.
    A spec is (is_transpose, kernel_size, stride, c_in, c_out, in_tensor_stride); k = 1, s = 1 layers are the
    reference's `use_mm` matmuls (MinkowskiConvolution.py:264-270, 304-308).  Batch norm / ReLU / concatenation are
    NOT replayed (they are torch ops in the reference): the figure is a lower bound of the reference's step time."""

    def __init__(self, coords, layers, num_threads=None, seed=0):
        C = load()
        self.C = C
        coords = coords.contiguous().int().cpu()
        D = coords.shape[1] - 1
        self.D = D
        nthreads = num_threads if num_threads is not None else min(os.cpu_count() or 1, 20)
        self.manager = C.CoordinateMapManagerCPU(C.MinkowskiAlgorithm.DEFAULT, nthreads)
        key, _ = self.manager.insert_and_map(coords, [1] * D, "")
        self.keys = {tuple([1] * D): key}
        self.empty_offset = torch.IntTensor()
        g = torch.Generator().manual_seed(seed)
        self.layers, self._feat, self._ones = [], {}, {}
        for (tr, ks, st, cin, cout, ts) in layers:
            ks, st, ts = [int(v) for v in ks], [int(v) for v in st], tuple(int(v) for v in ts)
            in_key = self.keys[ts]
            out_ts = tuple(t // s for t, s in zip(ts, st)) if tr else tuple(t * s for t, s in zip(ts, st))
            if out_ts not in self.keys:
                assert not tr, "a transposed convolution needs the finer map to exist"
                self.keys[out_ts] = self.manager.stride(in_key, st, "")
            out_key = self.keys[out_ts]
            n_in, n_out = self.manager.size(in_key), self.manager.size(out_key)
            volume = 1
            for k in ks:
                volume *= k
            mm = volume == 1 and all(s == 1 for s in st)
            w = (torch.rand((cin, cout) if mm else (volume, cin, cout), generator=g) - 0.5) * 0.1
            if (n_in, cin) not in self._feat:
                self._feat[(n_in, cin)] = torch.rand(n_in, cin, generator=g)
            if (n_out, cout) not in self._ones:
                self._ones[(n_out, cout)] = torch.ones(n_out, cout)
            self.layers.append((tr, mm, ks, st, in_key, out_key, w, (n_in, cin), (n_out, cout)))

    def run(self):
        """one forward + backward of every layer"""
        C = self.C
        dl = [1] * self.D
        for tr, mm, ks, st, in_key, out_key, w, fk, ok in self.layers:
            x, gy = self._feat[fk], self._ones[ok]
            if mm:
                x.mm(w)
                gy.mm(w.t())
                x.t().mm(gy)
                continue
            fwd = C.ConvolutionTransposeForwardCPU if tr else C.ConvolutionForwardCPU
            bwd = C.ConvolutionTransposeBackwardCPU if tr else C.ConvolutionBackwardCPU
            fwd(x, w, ks, st, dl, C.RegionType.HYPER_CUBE, self.empty_offset, False, C.ConvolutionMode.DEFAULT, in_key,
                out_key, self.manager)
            bwd(x, gy, w, ks, st, dl, C.RegionType.HYPER_CUBE, self.empty_offset, C.ConvolutionMode.DEFAULT, in_key,
                out_key, self.manager)
