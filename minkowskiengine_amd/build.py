"""Build recipe of the HIP library: hipcc --offload-arch=gfx950 csrc/*.hip -> libme_amd.so (in-tree,
next to this file, so it travels with the source snapshot)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libme_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-Wno-unused-lambda-capture"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "me_amd.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    cmd = [HIPCC] + FLAGS + sources() + ["-o", OUT]
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
