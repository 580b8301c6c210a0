"""Build recipe of the HIP library: hipcc --offload-arch=gfx950 csrc/*.hip -> libme_amd.so (in-tree,
next to this file, so it travels with the source snapshot)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ME_AMD_LIB_TAG=x: a second library libme_amd_x.so (own object cache) next to the default one — A/B builds with
# ME_AMD_EXTRA_HIPCC_FLAGS for the tuning scripts; _lib.py loads the tagged library when the variable is set
_TAG = os.environ.get("ME_AMD_LIB_TAG", "")
OUT = os.path.join(HERE, f"libme_amd_{_TAG}.so" if _TAG else "libme_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fvisibility=hidden: only what include/me_amd.h and csrc/me_amd_debug.h declare (under `#pragma GCC visibility
# push(default)`) leaves the library — no C++-mangled internals, no device stubs
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-Wno-unused-lambda-capture"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


OBJ_DIR = os.path.join(HERE, "csrc", "build_" + _TAG if _TAG else "build")   # per-file objects (git-ignored): only edited files are recompiled


def _headers():
    return glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "me_amd.h")]


def _compile_one(src, extra, verbose):
    obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
    stamp = obj + ".flags"
    flags = " ".join(CFLAGS + extra)
    fresh = (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flags and
             all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [src] + _headers()))
    if not fresh:
        cmd = [HIPCC] + CFLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(flags)
    return obj


def build(force=False, verbose=False, extra_flags=()):
    """Compile every csrc/*.hip to its own object (in parallel, cached per file) and link libme_amd.so.
    `extra_flags`: e.g. ("-DME_DEBUG_VARIANTS",) for the tuning / ablation instantiations (scripts/ only)."""
    extra = list(extra_flags) + os.environ.get("ME_AMD_EXTRA_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(f)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, extra, verbose), sources()))
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


# ---- native host layer: csrc_host/*.cpp -> _me_host.so (pybind11 + libtorch, host code only: g++) -------------------
HOST_SRC = os.path.join(HERE, "csrc_host")
HOST_OUT = os.path.join(HERE, "_me_host.so")
HOST_OBJ_DIR = os.path.join(HOST_SRC, "build")


def _host_flags():
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"]]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cflags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-DTORCH_EXTENSION_NAME=_me_host",
              "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}",
              "-Wall", "-Wno-unused-function", "-Wno-sign-compare"] + [f"-I{i}" for i in inc]
    ldflags = ["-shared", f"-L{tl}", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip",
               "-L/opt/rocm/lib", "-lamdhip64", f"-L{HERE}", "-lme_amd", f"-Wl,-rpath,{tl}", "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,-rpath,$ORIGIN"]
    return cflags, ldflags


def build_host(force=False, verbose=False):
    """The native operator module (MinkowskiEngineBackend._C for the hot path): pybind11 + libtorch host code over the
    C ABI of libme_amd.so; one object per csrc_host/*.cpp, compiled in parallel with g++ (no device code)."""
    build()
    cxx = os.environ.get("CXX", "g++")
    cflags, ldflags = _host_flags()
    os.makedirs(HOST_OBJ_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HOST_SRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(HOST_SRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "me_amd.h")]

    def one(src):
        obj = os.path.join(HOST_OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + hdrs):
            cmd = [cxx] + cflags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, srcs))
    if force or not os.path.exists(HOST_OUT) or any(os.path.getmtime(o) > os.path.getmtime(HOST_OUT) for o in objs) \
            or os.path.getmtime(OUT) > os.path.getmtime(HOST_OUT):
        subprocess.check_call([cxx] + objs + ldflags + ["-o", HOST_OUT])
    return HOST_OUT


CPP_EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "cpp", "conv_layer.cpp")
CPP_EXAMPLE = os.path.join(HERE, "..", "examples", "cpp", "conv_layer")


def build_cpp_example(force=False):
    """examples/cpp/conv_layer: a convolution layer driven from plain C++ through the C ABI (include/me_amd.h), linked
    against the in-tree libme_amd.so with an $ORIGIN-relative rpath so that it runs wherever the tree is copied."""
    build()
    src, out = os.path.normpath(CPP_EXAMPLE_SRC), os.path.normpath(CPP_EXAMPLE)
    if not os.path.exists(src):
        return None
    deps = [src, OUT, os.path.join(HERE, "..", "include", "me_amd.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(HERE, "..", "include"),
                           src, "-L", HERE, "-lme_amd", "-Wl,-rpath,$ORIGIN/../../minkowskiengine_amd", "-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--host" in sys.argv:
        print(build_host(force="--force" in sys.argv, verbose=True))
