"""TEST INFRASTRUCTURE ONLY.  Build the reference's own CPU_ONLY extension, unmodified, from the
sources where they lie under /root/reference, into oracle/_ref/_C.so.

Source set = the "cpu" list of /root/reference/setup.py:224-242 plus pybind/minkowski.cpp; the only
missing dependency (a CBLAS header) is supplied by oracle/shim/cblas.h.  Nothing is copied out of
/root/reference: the compiler reads the files in place and only the object files / shared object
land in oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.
Usage:  python oracle/build_ref.py [--force]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ME_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref")

CPU_SOURCES = [
    "math_functions_cpu.cpp", "coordinate_map_manager.cpp", "convolution_cpu.cpp",
    "convolution_transpose_cpu.cpp", "local_pooling_cpu.cpp", "local_pooling_transpose_cpu.cpp",
    "global_pooling_cpu.cpp", "broadcast_cpu.cpp", "pruning_cpu.cpp", "interpolation_cpu.cpp",
    "quantization.cpp", "direct_max_pool.cpp",
]


def ref_so_path():
    return os.path.join(OUT, "_C.so")


def build(force=False, verbose=False):
    so = ref_so_path()
    if os.path.exists(so) and not force:
        return so
    if not os.path.isdir(os.path.join(REF, "src")):
        raise FileNotFoundError(
            f"{REF}/src not present (GPU box?) and {so} was not prebuilt; cannot build oracle/_ref")
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("MAX_JOBS", str(os.cpu_count() or 8))
    from torch.utils.cpp_extension import load
    src = os.path.join(REF, "src")
    sources = [os.path.join(src, f) for f in CPU_SOURCES] + [os.path.join(REF, "pybind", "minkowski.cpp")]
    t = time.time()
    load(name="_C", sources=sources,
         extra_cflags=["-DCPU_ONLY", "-fopenmp", "-O3", "-w"],
         extra_include_paths=[src, os.path.join(src, "3rdparty"), os.path.join(HERE, "shim")],
         extra_ldflags=["-fopenmp"], build_directory=OUT, verbose=verbose, is_python_module=False)
    print(f"[oracle/_ref] built {so} in {time.time() - t:.1f}s", file=sys.stderr)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
