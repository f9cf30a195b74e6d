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
    srcs += [os.path.join(HERE, "include", "cpprob", "detail", f) for f in ("rng.hpp", "dist.hpp", "hd.hpp", "fastmath.hpp", "wave.hpp", "fixed_mass.hpp")]
    return srcs


def source_hash():
    """sha256 over the library's sources (names and contents): embedded in the binary as cpprob_hip_build_id()."""
    import hashlib
    h = hashlib.sha256()
    for s in lib_sources():
        h.update(os.path.relpath(s, ROOT).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _newer(LIB, lib_sources()):
        return LIB
    cmd = [HIPCC] + HIP_FLAGS + ['-DCPPROB_BUILD_ID="%s"' % source_hash(), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "include"), "-o", LIB,
                                 os.path.join(CSRC, "cpprob_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


INC = os.path.join(HERE, "include")
EXAMPLES = os.path.join(HERE, "examples")
MODELS_LIB = os.path.join(LIBDIR, "libcpprob_models.so")
MAIN_BIN = os.path.join(HERE, "bin", "cpprob_main")


def _tree(d, exts):
    out = []
    for dp, _, fs in os.walk(d):
        out += [os.path.join(dp, f) for f in fs if f.endswith(exts)]
    return out


def build_models(force=False, verbose=False):
    """Model translation unit: the model source compiled for host AND device by hipcc (C++17),
    registered with the engine (cpprob/gpu.hpp)."""
    srcs = _tree(INC, (".hpp", ".h")) + _tree(EXAMPLES, (".hpp", ".hip")) + [LIB]
    if not force and not _newer(MODELS_LIB, srcs):
        return MODELS_LIB
    cmd = [HIPCC, "-O2", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", INC,
           "-I", EXAMPLES, "-o", MODELS_LIB, os.path.join(EXAMPLES, "registered_models.hip"), "-L", LIBDIR, "-lcpprob_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return MODELS_LIB


def build_main(force=False, verbose=False):
    """Host driver: plain C++14 (g++), no HIP -- the reference's call sites compile as they are."""
    os.makedirs(os.path.dirname(MAIN_BIN), exist_ok=True)
    srcs = _tree(INC, (".hpp", ".h")) + _tree(EXAMPLES, (".hpp", ".cpp")) + [MODELS_LIB]
    if not force and not _newer(MAIN_BIN, srcs):
        return MAIN_BIN
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", INC, "-I", EXAMPLES,
           "-o", MAIN_BIN, os.path.join(EXAMPLES, "cpprob_main.cpp"), "-L", LIBDIR, "-lcpprob_models", "-lcpprob_hip",
           "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return MAIN_BIN


def build_all(force=False, verbose=False):
    return [build_lib(force, verbose), build_models(force, verbose), build_main(force, verbose)]


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
