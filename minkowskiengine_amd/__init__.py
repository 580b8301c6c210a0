"""minkowskiengine_amd — MI355X (gfx950) sparse-tensor convolution engine that keeps the
MinkowskiEngine Python API for the hot path: SparseTensor / CoordinateManager /
MinkowskiConvolution[Transpose] over hand-written HIP kernels (csrc/, C ABI in include/me_amd.h).

    import minkowskiengine_amd as ME
    x = ME.SparseTensor(features.cuda(), coordinates.cuda())
    y = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).cuda()(x)
"""
__version__ = "0.1.0"

from . import backend as MinkowskiEngineBackend  # noqa: F401  (the `_C`-compatible operator module)
from .backend import (  # noqa: F401
    BroadcastMode, ConvolutionMode, CoordinateMapType, GPUMemoryAllocatorType,
    MinkowskiAlgorithm, PoolingMode, RegionType, cuda_version, cudart_version, get_gpu_memory_info,
    is_cuda_available)
from .host import (  # noqa: F401  (native C++ host layer | backend.py)
    CoordinateMapKey, get_host, invalidate_packed_weights, is_native, set_host)
from .common import convert_to_int_list, get_minkowski_function  # noqa: F401
from .convolution import (  # noqa: F401
    MinkowskiConvolution, MinkowskiConvolutionFunction, MinkowskiConvolutionTranspose,
    MinkowskiConvolutionTransposeFunction, MinkowskiGenerativeConvolutionTranspose)
from .pruning import MinkowskiPruning, MinkowskiPruningFunction  # noqa: F401
from .union import MinkowskiUnion, MinkowskiUnionFunction  # noqa: F401
from .coordinate_manager import (  # noqa: F401
    CoordinateManager, set_gpu_allocator, set_memory_manager_backend, set_map_prefetch, map_prefetch_enabled, map_prefetch_tag)
from .kernel_generator import KernelGenerator, get_kernel_volume  # noqa: F401
from .layers import (  # noqa: F401
    MinkowskiBatchNorm, MinkowskiDropout, MinkowskiELU, MinkowskiLeakyReLU, MinkowskiLinear, MinkowskiReLU,
    MinkowskiSigmoid, MinkowskiSyncBatchNorm, MinkowskiTanh, cat)
from .pooling import (  # noqa: F401
    MinkowskiAvgPooling, MinkowskiGlobalAvgPooling, MinkowskiGlobalMaxPooling, MinkowskiGlobalPooling,
    MinkowskiGlobalPoolingFunction, MinkowskiGlobalSumPooling, MinkowskiLocalPoolingFunction,
    MinkowskiLocalPoolingTransposeFunction, MinkowskiMaxPooling, MinkowskiPoolingTranspose, MinkowskiSumPooling)
from .broadcast import (  # noqa: F401
    MinkowskiBroadcast, MinkowskiBroadcastAddition, MinkowskiBroadcastConcatenation, MinkowskiBroadcastFunction,
    MinkowskiBroadcastMultiplication)
from .sparse_tensor import (  # noqa: F401
    SparseTensor, SparseTensorOperationMode, SparseTensorQuantizationMode, clear_global_coordinate_manager,
    global_coordinate_manager, set_global_coordinate_manager, set_sparse_tensor_operation_mode,
    sparse_tensor_operation_mode)
from . import utils  # noqa: F401


def _install_optimizer_hook():
    """The weight-image cache is validated by tensor version counters, which writes through `p.data` do not bump
    (Apex / DeepSpeed-style optimizers update that way).  Every torch.optim.Optimizer step therefore marks the images of
    ITS parameters stale: the first convolution after it repacks them in one launch — which a training step does anyway.
    ME_AMD_PACK_CACHE_HOOK=0 leaves the cache to the version counters alone."""
    import os
    if os.environ.get("ME_AMD_PACK_CACHE_HOOK", "1") == "0":
        return
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:  # pragma: no cover  (torch < 2.0)
        return
    def _after_step(opt, args, kwargs):
        # only the stepping optimizer's own parameters: the images of frozen / teacher networks stay valid
        invalidate_packed_weights([p for g in opt.param_groups for p in g["params"]])

    register_optimizer_step_post_hook(_after_step)


_install_optimizer_hook()
