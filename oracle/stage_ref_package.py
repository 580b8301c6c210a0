"""TEST INFRASTRUCTURE ONLY.  Stage the reference's own PYTHON package next to the compiled reference operators so
that the strongest drop-in test — the reference's unmodified `MinkowskiEngine/*.py` running on this repository's HIP
kernels (tests/test_reference_package.py) — can execute on a GPU box, where /root/reference does not exist.

The files are copied verbatim from /root/reference/MinkowskiEngine into oracle/_ref/reference_tree/ — the same
git-ignored scratch directory that holds the compiled reference (`oracle/_ref/_C.so`): it never enters the history,
travels to the GPU box with the snapshot, and nothing in the product imports it (oracle/ref.py::reference_root finds
it when /root/reference is absent).

Usage:  python oracle/stage_ref_package.py
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ME_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref", "reference_tree")


def staged_root():
    return OUT


def stage(force=False):
    src = os.path.join(REF, "MinkowskiEngine")
    if not os.path.isdir(src):
        raise FileNotFoundError(f"{src} not present: the reference package can only be staged where the reference is")
    dst = os.path.join(OUT, "MinkowskiEngine")
    if os.path.isdir(dst) and not force:
        return OUT
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    return OUT


if __name__ == "__main__":
    print(stage(force="--force" in sys.argv))
