"""TEST INFRASTRUCTURE ONLY — loader for the reference's own CPU extension (oracle/_ref/_C.so,
built unmodified from /root/reference by oracle/build_ref.py).  The shared object travels to the GPU
box; /root/reference itself does not, so only the compiled operators are used here (never the
reference's Python package)."""
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
