"""Builds the native parts in-tree with hipcc for gfx950 (no GPU needed to compile).

  cpprob_amd/lib/libcpprob_hip.so   C ABI + HIP kernels (include/cpprob_hip.h)

`python -m cpprob_amd.build` rebuilds everything; `build_all()` is what __graft_entry__.build() calls.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcpprob_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
             "-fvisibility=hidden", "-DCPPROB_HIP_BUILD"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def lib_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "cpprob_hip.h"))
    return srcs


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _newer(LIB, lib_sources()):
        return LIB
    cmd = [HIPCC] + HIP_FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", LIB, os.path.join(CSRC, "cpprob_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_all(force=False, verbose=False):
    return [build_lib(force, verbose)]


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
