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


# Step kernels built per step (cpprob/gpu.hpp: model_step_kernel_at): one source, one object per (model, part) -- the optimiser needs
# ~a minute per build of a 128-observe model, so the builds are dealt over objects that compile side by side.
STEPS_SRC = os.path.join(EXAMPLES, "registered_steps.hip")
STEPS_FLAGS = ["-mllvm", "-unroll-threshold=2000000", "-mllvm", "-inline-threshold=10000000", "-mllvm", "-amdgpu-inline-max-bb=1000000",
               "-mllvm", "-memdep-block-scan-limit=2000", "-mllvm", "-memdep-block-number-limit=2000"]
STEPS_UNITS = [(0, 1), (1, 2), (2, 7), (3, 7)]          # (CPPROB_STEPS_MODEL, parts): hmm<16>, linear_gaussian_1d<25>, hmm<128>, linear_gaussian_1d<100>


def build_models(force=False, verbose=False):
    """Model translation units: the model source compiled for host AND device by hipcc (C++17), registered with the engine
    (cpprob/gpu.hpp); the per-step kernels in objects of their own, compiled in parallel."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    srcs = _tree(INC, (".hpp", ".h")) + _tree(EXAMPLES, (".hpp", ".hip")) + [LIB, os.path.abspath(__file__)]
    if not force and not _newer(MODELS_LIB, srcs):
        return MODELS_LIB
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    common = [HIPCC, "-O2", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-c", "-I", os.path.join(ROOT, "include"), "-I", INC, "-I", EXAMPLES]
    jobs = [(os.path.join(objdir, "registered_models.o"), common + ["-o", os.path.join(objdir, "registered_models.o"), os.path.join(EXAMPLES, "registered_models.hip")])]
    if os.environ.get("CPPROB_NO_STEP_BUILDS", "") == "":
        for model, parts in STEPS_UNITS:
            for part in range(parts):
                o = os.path.join(objdir, "steps_%d_%d.o" % (model, part))
                jobs.append((o, common + STEPS_FLAGS + ["-DCPPROB_STEPS_MODEL=%d" % model, "-DCPPROB_STEPS_PART=%d" % part, "-DCPPROB_STEPS_PARTS=%d" % parts, "-o", o, STEPS_SRC]))
    t0 = time.time()

    def run(job):
        t1 = time.time()
        subprocess.check_call(job[1])
        return job[0], time.time() - t1
    # (the longest jobs first: the 128- and 100-observe builds)
    order = sorted(jobs, key=lambda j: 0 if ("steps_2_" in j[0] or "steps_3_" in j[0]) else 1)
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        done = list(ex.map(run, order))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", MODELS_LIB] + [j[0] for j in jobs] + ["-L", LIBDIR, "-lcpprob_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if verbose:
        print("model units: %d objects in %.0f s wall (%.0f s of compiler time; slowest %.0f s), %s = %.1f MB" %
              (len(jobs), time.time() - t0, sum(d[1] for d in done), max(d[1] for d in done), os.path.basename(MODELS_LIB), os.path.getsize(MODELS_LIB) / 1e6))
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
