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
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
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
