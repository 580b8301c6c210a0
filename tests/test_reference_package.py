"""Drop-in proof at the reference's own boundary: `MinkowskiEngineBackend._C` (the native extension the reference's
setup.py:312 builds) is provided by this repository, and the REFERENCE's unmodified Python package
(/root/reference/MinkowskiEngine/*.py) imports and runs on top of it.

* CPU (authoring container, where /root/reference exists): the reference package imports over our `_C`; its
  CoordinateManager / MinkowskiConvolution / pooling / broadcast modules construct; every operator the reference's
  autograd Functions resolve with get_minkowski_function(name, cuda tensor) for the hot path exists in our module
  and accepts exactly the positional arguments of the reference's call site (parsed from the reference's sources).
* GPU (`-m gpu`): the reference package's own layers run on the HIP kernels and agree with this repository's layers
  and the oracle.  /root/reference does not exist on a GPU box; the package is found in oracle/_ref/reference_tree,
  the git-ignored scratch copy that __graft_entry__.build() stages next to the compiled reference
  (oracle/stage_ref_package.py) and that travels with the snapshot.
Everything touching the reference tree runs in a subprocess: importing it rewires sys.path / sys.modules."""
import ast
import inspect
import os
import subprocess
import sys

import pytest

from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(ref.reference_root(), "MinkowskiEngine")),
                               reason="needs the reference package (/root/reference or oracle/_ref/reference_tree)")

HOT_PATH_FILES = ["MinkowskiConvolution.py", "MinkowskiPooling.py", "MinkowskiBroadcast.py", "MinkowskiPruning.py"]
# resolved by name but outside SURVEY.md 8 (no GPU kernel here): nothing in the four files above


def test_backend_alias_is_the_operator_module():
    import MinkowskiEngineBackend._C as C
    from minkowskiengine_amd import backend
    assert C is backend
    for name in ("CoordinateMapKey", "CoordinateMapManagerGPU_c10", "CoordinateMapManagerGPU_default", "RegionType",
                 "PoolingMode", "BroadcastMode", "ConvolutionMode", "MinkowskiAlgorithm", "CoordinateMapType",
                 "GPUMemoryAllocatorType", "is_cuda_available", "cuda_version", "cudart_version",
                 "get_gpu_memory_info"):
        assert hasattr(C, name), name


def _reference_call_sites():
    """[(file, operator base name, positional arg count)] for every `get_minkowski_function("X", t)` whose result is
    called in the reference's hot-path modules"""
    sites = []
    for fn in HOT_PATH_FILES:
        tree = ast.parse(open(os.path.join(ref.reference_root(), "MinkowskiEngine", fn)).read())
        for func in ast.walk(tree):
            if not isinstance(func, ast.FunctionDef):
                continue
            bound = {}
            for node in ast.walk(func):
                if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and \
                        getattr(node.value.func, "id", "") == "get_minkowski_function":
                    bound[node.targets[0].id] = node.value.args[0].value
            for node in ast.walk(func):
                if isinstance(node, ast.Call) and getattr(node.func, "id", None) in bound:
                    assert not node.keywords
                    sites.append((fn, bound[node.func.id], len(node.args)))
    return sites


@needs_ref
def test_every_hot_path_operator_of_the_reference_resolves_with_its_arity():
    from minkowskiengine_amd import backend
    sites = _reference_call_sites()
    names = {s[1] for s in sites}
    assert {"ConvolutionForward", "ConvolutionBackward", "ConvolutionTransposeForward", "ConvolutionTransposeBackward",
            "LocalPoolingForward", "LocalPoolingBackward", "LocalPoolingTransposeForward",
            "LocalPoolingTransposeBackward", "GlobalPoolingForward", "GlobalPoolingBackward", "BroadcastForward",
            "BroadcastBackward", "PruningForward", "PruningBackward"} <= names
    for fn, base, nargs in sites:
        op = getattr(backend, base + "GPU", None)
        assert op is not None, f"{fn}: {base}GPU missing"
        params = [p for p in inspect.signature(op).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        required = [p for p in params if p.default is p.empty]
        assert len(required) <= nargs <= len(params), f"{fn}: {base}GPU takes {len(params)} arguments, call has {nargs}"


_IMPORT = r"""
import sys
sys.path.insert(0, {root!r})
import MinkowskiEngineBackend._C as C
from oracle import ref
ME = ref.import_reference_package(backend=C)
assert ME.__file__.startswith({refroot!r}), ME.__file__
assert ME.is_cuda_available() is True
import torch
mgr = ME.CoordinateManager(D=3)
assert type(mgr._manager).__name__ == "CoordinateMapManagerGPU_c10"
conv = ME.MinkowskiConvolution(4, 8, kernel_size=3, dimension=3)
convt = ME.MinkowskiConvolutionTranspose(8, 4, kernel_size=2, stride=2, dimension=3)
pool = ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3)
gpool, bcast = ME.MinkowskiGlobalAvgPooling(), ME.MinkowskiBroadcastMultiplication()
bn = ME.MinkowskiBatchNorm(8)
key = ME.CoordinateMapKey([1, 1, 1], "")
assert key.get_key() == ([1, 1, 1], "") and key.is_key_set()
print("IMPORT_OK", ME.__version__)
"""


@needs_ref
def test_reference_package_imports_and_constructs_on_our_backend():
    out = subprocess.run([sys.executable, "-c", _IMPORT.format(root=ROOT, refroot=ref.reference_root())],
                         capture_output=True, text=True, timeout=600)
    assert "IMPORT_OK" in out.stdout, out.stderr[-3000:]


_GPU = r"""
import sys
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import torch
import minkowskiengine_amd as OURS
import MinkowskiEngineBackend._C as C
from oracle import me_oracle as O
from oracle import ref
from helpers import make_cloud, assert_close
ME = ref.import_reference_package(backend=C)          # the reference's Python, our kernels
dev = torch.device("cuda:0")
coords = make_cloud(4000, 16, 3, seed=1, batch=2)
g = torch.Generator().manual_seed(0)
feats = torch.rand(coords.shape[0], 16, generator=g)
w = torch.rand(27, 16, 32, generator=g) - 0.5
x = ME.SparseTensor(feats.to(dev), coords.to(dev), requires_grad=True)
conv = ME.MinkowskiConvolution(16, 32, kernel_size=3, dimension=3).to(dev)
with torch.no_grad():
    conv.kernel.copy_(w)
y = conv(x)
co = coords.numpy()
_, km = O.kernel_map(co, co, O.make_region(3, 3))
assert_close(y.F, O.conv_forward(feats.numpy(), w.numpy(), km, len(co)), what="reference-package conv forward")
gy = torch.rand(y.F.shape, generator=g)
y.F.backward(gy.to(dev))
gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w.numpy(), km)
assert_close(x.F.grad, gi, what="grad_in")
assert_close(conv.kernel.grad, gw, what="grad_kernel")
# strided conv -> transposed conv back, max pooling, global pooling + broadcast: reference layers vs our layers
down = ME.MinkowskiConvolution(16, 16, kernel_size=2, stride=2, dimension=3).to(dev)
up = ME.MinkowskiConvolutionTranspose(16, 8, kernel_size=2, stride=2, dimension=3).to(dev)
pool = ME.MinkowskiMaxPooling(kernel_size=3, stride=2, dimension=3)
gp, bm = ME.MinkowskiGlobalAvgPooling(), ME.MinkowskiBroadcastMultiplication()
x2 = ME.SparseTensor(feats.to(dev), coords.to(dev))
z = up(down(x2)); p = pool(x2); b = bm(x2, gp(x2))
o_down = OURS.MinkowskiConvolution(16, 16, kernel_size=2, stride=2, dimension=3).to(dev)
o_up = OURS.MinkowskiConvolutionTranspose(16, 8, kernel_size=2, stride=2, dimension=3).to(dev)
with torch.no_grad():
    o_down.kernel.copy_(down.kernel); o_up.kernel.copy_(up.kernel)
ox = OURS.SparseTensor(feats.to(dev), coords.to(dev))
oz = o_up(o_down(ox)); op = OURS.MinkowskiMaxPooling(kernel_size=3, stride=2, dimension=3)(ox)
ob = OURS.MinkowskiBroadcastMultiplication()(ox, OURS.MinkowskiGlobalAvgPooling()(ox))
assert torch.equal(z.C, oz.C) and torch.equal(z.F, oz.F)
assert torch.equal(p.C, op.C) and torch.equal(p.F, op.F)
assert torch.equal(b.F, ob.F)
assert z.coordinate_map_key == x2.coordinate_map_key
print("GPU_OK")
"""


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("host", ["python", "native"])
def test_reference_package_runs_on_the_hip_kernels(host):
    """The reference's unmodified Python package over `MinkowskiEngineBackend._C` = the Python twin, and = the native
    C++ extension (the product default, what the reference's own `_C` is: a pybind11 module)."""
    env = dict(os.environ, ME_AMD_HOST=host)
    script = _GPU.format(root=ROOT) + f"\nassert OURS.get_host() == {host!r} and getattr(C, '_host', 'python') == {host!r}\nprint('HOST_OK')\n"
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900, env=env)
    assert "GPU_OK" in out.stdout and "HOST_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


_IMPORT_NATIVE = r"""
import sys
sys.path.insert(0, {root!r})
import MinkowskiEngineBackend._C as C
assert getattr(C, "_host", None) == "native", C
for name in ("CoordinateMapKey", "CoordinateMapManagerGPU_c10", "CoordinateMapManagerGPU_default", "RegionType",
             "PoolingMode", "BroadcastMode", "ConvolutionMode", "MinkowskiAlgorithm", "CoordinateMapType",
             "GPUMemoryAllocatorType", "is_cuda_available", "cuda_version", "cudart_version", "get_gpu_memory_info",
             "ConvolutionForwardGPU", "ConvolutionBackwardGPU", "ConvolutionTransposeForwardGPU",
             "ConvolutionTransposeBackwardGPU", "LocalPoolingForwardGPU", "LocalPoolingBackwardGPU",
             "LocalPoolingTransposeForwardGPU", "LocalPoolingTransposeBackwardGPU", "GlobalPoolingForwardGPU",
             "GlobalPoolingBackwardGPU", "BroadcastForwardGPU", "BroadcastBackwardGPU", "PruningForwardGPU",
             "PruningBackwardGPU"):
    assert hasattr(C, name), name
from oracle import ref
ME = ref.import_reference_package(backend=C)
mgr = ME.CoordinateManager(D=3)
assert type(mgr._manager).__name__ == "CoordinateMapManagerGPU_c10"
key = ME.CoordinateMapKey([1, 1, 1], "")
assert key.get_key() == ([1, 1, 1], "") and key.is_key_set()
conv = ME.MinkowskiConvolution(4, 8, kernel_size=3, dimension=3)
print("IMPORT_OK", ME.__version__)
"""


@needs_ref
def test_reference_package_imports_on_the_native_module():
    """CPU leg of the above: with ME_AMD_HOST=native `MinkowskiEngineBackend._C` IS the compiled extension, exports the
    reference's operator / class / enum names, and the reference's Python package imports and constructs on it."""
    env = dict(os.environ, ME_AMD_HOST="native")
    out = subprocess.run([sys.executable, "-c", _IMPORT_NATIVE.format(root=ROOT)], capture_output=True, text=True,
                         timeout=600, env=env)
    assert "IMPORT_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
