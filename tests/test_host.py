"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/cpprob_hip.h
declares, fails loudly without a GPU, and the multi-GPU host logic (gloo, world size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import cpprob_amd
    return cpprob_amd.load_library().cpprob_hip_device_count() <= 0


def test_library_exports_every_declared_symbol():
    import cpprob_amd
    from cpprob_amd import build as B
    B.build_lib()
    hdr = open(os.path.join(ROOT, "include", "cpprob_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cpprob_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    assert sorted(cpprob_amd.capi.SYMBOLS) == declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", cpprob_amd.capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (cpprob_hip_\w+)", out))
    assert set(declared) <= exported
    L = cpprob_amd.load_library()
    assert L.cpprob_hip_abi_version() == 3
    # nothing but the C ABI is exported (no C++ symbols leak)
    assert not [l for l in out.splitlines() if " T " in l and "cpprob_hip_" not in l and "_init" not in l and "_fini" not in l]


def test_config_struct_layout_matches_header():
    import ctypes as C
    from cpprob_amd import capi
    assert C.sizeof(capi.Config) == 8 * 4 + 8 + 4 * 8
    assert capi.Config.flags.offset == 24 and capi.Config.ess_threshold.offset == 32 and capi.Config.seed.offset == 40
    assert C.sizeof(capi.Summary) == 4 * 8 + 6 * 4


def test_fails_loudly_without_gpu():
    import cpprob_amd
    if not _no_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(cpprob_amd.CpprobHipError) as e:
        cpprob_amd.Engine(0)
    assert "no HIP device" in str(e.value) and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cpprob_amd/ or include/ may reference it."""
    bad = []
    for base in ("cpprob_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle|liboracle|cpprob_oracle|#include\s+[\"<].*oracle", txt, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_shard_bounds_partition():
    from cpprob_amd import distributed as D
    for n, w in [(10, 3), (1000000, 8), (7, 8), (100000001, 8)]:
        seen = 0
        for r in range(w):
            lo, cnt = D.shard_bounds(n, w, r)
            assert lo == seen
            seen += cnt
        assert seen == n


def test_combine_islands_is_evidence_weighting():
    from cpprob_amd import distributed as D
    rng = np.random.default_rng(0)
    R, T = 4, 5
    log_z = rng.normal(size=R) - 30
    p = rng.dirichlet(np.ones(3), size=(R, T))
    out, lz, w, iess = D.combine_islands(log_z, p, True)
    wz = np.exp(log_z - log_z.max()); wz /= wz.sum()
    assert np.allclose(out, np.einsum("r,rtk->tk", wz, p)) and np.allclose(out.sum(1), 1)
    assert abs(lz - np.log(np.mean(np.exp(log_z)))) < 1e-12
    mv = np.stack([rng.normal(size=(R, T)), rng.random((R, T)) + 0.1], axis=2)
    out, _, _, _ = D.combine_islands(log_z, mv, False)
    mean = wz @ mv[:, :, 0]
    raw2 = wz @ (mv[:, :, 1] + mv[:, :, 0] ** 2)
    assert np.allclose(out[:, 0], mean) and np.allclose(out[:, 1], raw2 - mean ** 2)
    # equal islands: identity
    out, lz, w, iess = D.combine_islands(np.full(3, -5.0), np.tile(p[:1], (3, 1, 1)), True)
    assert np.allclose(out, p[0]) and abs(lz + 5.0) < 1e-12 and abs(iess - 3.0) < 1e-12


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from cpprob_amd import distributed as D
from oracle import oracle as O, exact as E
world, rank, local = D.init_process_group(device_is_gpu=False)
assert world == 2 and dist.get_backend() == "gloo"
# every rank runs its island with the CPU oracle standing in for the device (test only) and the
# product's host logic combines them
obs = np.load(os.path.join(%(root)r, "tests", "golden", "observations.npz"))["hmm16"]
n = 40000
lo, cnt = D.shard_bounds(2 * n, world, rank)
assert cnt == n and lo == rank * n
r = O.smc(O.MODEL_HMM3, obs, n, 100 + rank, O.RESAMPLE_SYSTEMATIC, 2.0)
st = O.smoothing(r["hist"], r["anc"], r["logw"])
vec = np.concatenate([[r["log_z"]], st.reshape(-1)])
allv = D.allgather_vector(vec)
assert allv.shape == (2, 1 + 48)
assert np.array_equal(allv[rank], vec)
out, lz, w, iess = D.combine_islands(allv[:, 0], allv[:, 1:].reshape(2, 16, 3), True)
t = torch.tensor([lz], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert abs(float(t) - lz) < 1e-15          # identical on both ranks
g = np.load(os.path.join(%(root)r, "tests", "golden", "observations.npz"))["hmm16_smooth"]
assert np.abs(out - g).max() < 0.03 and np.allclose(out.sum(1), 1)
assert iess > 1.9
# exchange scope: the all-to-all-v of lineage records through host memory (what TorchCollective uses under gloo)
begins = D.shard_begins(11, 2)
assert list(begins) == [0, 6, 11]
send = torch.arange(10, dtype=torch.int32) + 100 * rank
ins = [4, 6] if rank == 0 else [3, 7]
outs = [4, 3] if rank == 0 else [6, 7]
got = D.host_all_to_all(send, ins, outs)
want = torch.cat([torch.arange(0, 4), torch.arange(100, 103)]) if rank == 0 else torch.cat([torch.arange(4, 10), torch.arange(103, 110)])
assert torch.equal(got, want.to(torch.int32))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gloo_world2_island_combine(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    import socket
    for attempt in range(2):                                        # one retry if the probed port was taken in between
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


_GLOO_EXCHANGE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from cpprob_amd import distributed as D, capi
from oracle import oracle as O
world, rank, local = D.init_process_group(device_is_gpu=False)
assert dist.get_backend() == "gloo" and world == %(world)d
# DEFAULT scope of a sharded run (exchange: one joint population, exact global resampling).  Every process holds ONE shard of a
# population whose generations come from the single-process oracle run (so that a rank's sources carry the right weights and
# lineages); the oracle's SHARDED resampler stands in for the device on each rank, the ranks' offspring bounds and the plan of who
# sends how many lineages to whom are the product's host statements (D.offspring_bounds_fixed, D.exchange_counts), the lineages
# travel through D.host_all_to_all -- and what every rank ends up holding for its outputs must be, output for output, the lineage
# of the ancestor the single-process run drew.
z = np.load(os.path.join(%(root)r, "tests", "golden", "observations.npz"))
obs = z["lgssm100"][:12]
N, seed = %(n)d, 17
ref = O.smc_ref(O.MODEL_LINEAR_GAUSSIAN_1D, obs, N, seed, O.REF_STATEMENT_BOUND, O.RESAMPLE_SYSTEMATIC, 2.0)       # every step resamples
paths_all = [np.take_along_axis(ref["hist"][: t + 1], O.lineage(ref["anc"][: t + 1]), axis=1) for t in range(len(obs))]   # [t + 1][N]: lineages of generation t
sizes = %(sizes)r
begins = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
assert int(begins[-1]) == N
lo, hi = int(begins[rank]), int(begins[rank + 1])
bound = -0.5 * np.log(2 * np.pi)
L = capi.load_library()
n_moved = 0
for t in range(len(obs) - 1):
    # generation t's weights on this rank's sources (every step resamples: they start from zero), as integers
    lw = -0.5 * ((obs[t] - ref["hist"][t][lo:hi]) ** 2 + np.log(2 * np.pi))
    q = O.fix_weights(lw, bound)
    mine = np.array([int(q.astype(np.uint64).sum())], np.float64)                 # (exact below 2^53)
    masses = D.allgather_vector(mine)[:, 0]
    total, before = int(masses.sum()), int(masses[:rank].sum())
    u0 = L.cpprob_hip_systematic_offset(seed, t + 1)
    # the stand-in for the device: ancestors of EVERY output among this rank's sources (-1: another rank's)
    anc = O.resample_fixed_systematic(q, seed, t + 1, before=before, total=total, last_shard=(rank + 1 == world), j0=0, n_out=N, n_total_out=N)
    owned = np.flatnonzero(anc >= 0)
    o = D.offspring_bounds_fixed(masses, N, u0)
    assert len(owned) == 0 or (owned[0] == int(o[rank]) and owned[-1] + 1 == int(o[rank + 1]) and len(owned) == int(o[rank + 1] - o[rank]))    # the host's bounds = what the resampler drew
    first, sc, rc = D.exchange_counts(o, begins, rank)
    width = t + 1
    send = []
    for p in range(world):
        j = np.arange(first[p], first[p] + sc[p])
        assert np.all(anc[j] >= 0)
        send.append(paths_all[t][:, lo + anc[j]].T.reshape(-1))                   # records: the lineage x_0..x_t of each ancestor, in output order
    h_send = torch.from_numpy(np.concatenate(send) if send else np.zeros(0))
    h_recv = D.host_all_to_all(h_send, [int(c) * width for c in sc], [int(c) * width for c in rc]).numpy()
    n_moved += int(sc.sum())
    # assemble this rank's outputs: own sources where it owns the output, else the record of the rank that does, in source-rank order
    got = np.zeros((width, hi - lo))
    off = 0
    for src in range(world):
        if src == rank:
            j = np.arange(max(int(o[rank]), lo), min(int(o[rank + 1]), hi))
            got[:, j - lo] = paths_all[t][:, lo + anc[j]]
            continue
        olo = max(int(o[src]), lo)
        recs = h_recv[off: off + int(rc[src]) * width].reshape(int(rc[src]), width)
        got[:, olo - lo: olo - lo + int(rc[src])] = recs.T
        off += int(rc[src]) * width
    want = paths_all[t][:, ref["anc"][t + 1][lo:hi]]                                # the lineages the single-process run's ancestors carry
    assert np.array_equal(got, want), (rank, t)
tot = torch.tensor([n_moved]); dist.all_reduce(tot)
assert int(tot) > 0                                                              # (lineages did cross ranks)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world,sizes", [(2, [3000, 3000]), (3, [2500, 1700, 1801])])
def test_gloo_exchange_scope_moves_the_single_process_lineages(tmp_path, world, sizes):
    """The N > 1 path's DEFAULT scope across real processes, on CPU: plan, counts and ordering of the migration (cpprob_amd/distributed.py)
    against the single-process oracle run -- world 2, and world 3 with uneven shards."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_EXCHANGE_WORKER % {"root": ROOT, "world": world, "sizes": sizes, "n": sum(sizes)})
    import socket
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-2500:] + out.stderr[-2500:]
    assert out.stdout.count("ok") == world


_GLOO_STRATA_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from cpprob_amd import distributed as D
from oracle import oracle as O
world, rank, local = D.init_process_group(device_is_gpu=False)
assert dist.get_backend() == "gloo" and world == %(world)d
# MULTINOMIAL resampling (thesis Alg. 1, strata form) of ONE population over real processes, on CPU.  Every process holds one shard of
# a population whose generations come from the single-process oracle run; the oracle's SHARD form stands in for the device's search on
# each rank, the plan -- regular intervals, the strata the ranks' boundaries cut, every migrant's annex column -- is the product's
# host statement (D.StrataCutPlan = csrc/strata_cut.hpp), the lineages travel through D.host_all_to_all, and what every rank ends up
# holding for its outputs (own sources + annex columns, in the plan's order) must be the lineage of the ancestor the single-process run drew.
z = np.load(os.path.join(%(root)r, "tests", "golden", "observations.npz"))
obs = z["lgssm100"][:10]
N, seed = %(n)d, 19
ref = O.smc(O.MODEL_LINEAR_GAUSSIAN_1D, obs, N, seed, O.RESAMPLE_MULTINOMIAL, 2.0)       # every step resamples; fixed-point masses
paths_all = [np.take_along_axis(ref["hist"][: t + 1], O.lineage(ref["anc"][: t + 1]), axis=1) for t in range(len(obs))]
sizes = %(sizes)r
begins = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
lo, hi = int(begins[rank]), int(begins[rank + 1])
bound = -0.5 * np.log(2 * np.pi)
k = O.lib().orc_strata_levels(N); K = 1 << k
n_moved = 0
for t in range(len(obs) - 1):
    lw = -0.5 * ((obs[t] - ref["hist"][t][lo:hi]) ** 2 + np.log(2 * np.pi))
    q = O.fix_weights(lw, bound)
    masses = [int(m) for m in D.allgather_vector(np.array([float(int(q.astype(np.uint64).sum()))]))[:, 0]]      # (exact below 2^53)
    P = [0]
    for m in masses: P.append(P[-1] + m)
    total = P[-1]
    anc = O.resample_fixed_multinomial_strata_shard(q, seed, t + 1, P[rank], total, N)       # the stand-in for the device: -1 = another rank's
    tau = [int(x) for x in O.strata_thresholds_fixed(total, seed, t + 1, N)[0]]
    plan = D.StrataCutPlan(P, [(total * w) >> k for w in range(K + 1)], O.multinomial_strata(seed, t + 1, N), begins, lambda a, b: tau[a:b])
    mine = sorted(plan.sends(rank))
    assert all(anc[s] >= 0 for s, d in mine) and sum(anc >= 0) == len(mine) + sum(1 for s in range(lo, hi) if anc[s] >= 0)
    width = t + 1
    send, sc = [], []
    for p in range(world):
        ss = [s for s, d in mine if d == p]
        sc.append(len(ss))
        send.append(paths_all[t][:, lo + anc[ss]].T.reshape(-1) if ss else np.zeros(0))
    rc = [sum(1 for s, d in plan.sends(p) if d == rank) for p in range(world)]
    assert sum(rc) == plan.arrivals(rank)
    h_recv = D.host_all_to_all(torch.from_numpy(np.concatenate(send)), [c * width for c in sc], [c * width for c in rc]).numpy()
    n_moved += sum(sc)
    annex = np.full((width, plan.arrivals(rank)), np.nan)
    off = 0
    for p in range(world):
        ss = sorted(s for s, d in plan.sends(p) if d == rank)
        recs = h_recv[off: off + len(ss) * width].reshape(len(ss), width)
        for i, s in enumerate(ss):
            annex[:, plan.column(rank, s)] = recs[i]
        off += len(ss) * width
    assert not np.isnan(annex).any()                                              # every annex column was filled exactly ... (below) with the right lineage
    got = np.zeros((width, hi - lo))
    for s in range(lo, hi):
        got[:, s - lo] = paths_all[t][:, lo + anc[s]] if anc[s] >= 0 else annex[:, plan.column(rank, s)]
    want = paths_all[t][:, ref["anc"][t + 1][lo:hi]]
    assert np.array_equal(got, want), (rank, t)
tot = torch.tensor([n_moved]); dist.all_reduce(tot)
assert int(tot) > 0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,sizes", [(2, [3000, 3001]), (3, [2500, 1701, 1800])])
def test_gloo_multinomial_exchange_moves_the_single_process_lineages(tmp_path, world, sizes):
    """Multinomial resampling of one population across real processes, on CPU: the strata cut plan (cpprob_amd/distributed.py) against
    the single-process oracle run -- world 2, and world 3 with uneven shards that start at odd particles."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_STRATA_WORKER % {"root": ROOT, "world": world, "sizes": sizes, "n": sum(sizes)})
    import socket
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-2500:] + out.stderr[-2500:]
    assert out.stdout.count("ok") == world


def test_bench_cli_parses_and_graft_entry_builds():
    import importlib
    ge = importlib.import_module("__graft_entry__")
    assert callable(ge.build) and callable(ge.smoke)
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src


def test_reference_models_header_compiles_unchanged_for_device(tmp_path):
    """north_star: 'existing models in namespace models compile unchanged'.  Where the reference tree is
    present (this container, not the GPU box) its own include/models/models.hpp is compiled for gfx950
    against the compatibility headers, host + device, and registered."""
    ref = "/root/reference/include"
    if not os.path.exists(os.path.join(ref, "models", "models.hpp")):
        pytest.skip("reference tree not present")
    out = str(tmp_path / "libmodels_ref.so")
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DCPPROB_USE_REFERENCE_MODELS",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cpprob_amd", "include"), "-I", ref,
           "-o", out, os.path.join(ROOT, "cpprob_amd", "examples", "registered_models.hip"),
           "-L", os.path.join(ROOT, "cpprob_amd", "lib"), "-lcpprob_hip"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    syms = subprocess.check_output(["nm", "-DC", out]).decode()
    assert "models::hmm<16ul>" in syms and "models::gaussian_unknown_mean<double>" in syms
    # the device code object holds a kernel per registered model
    assert "model_kernel" in subprocess.check_output(["strings", out]).decode()


_ONE_MODEL_TU = r"""
#include "cpprob/gpu.hpp"
#pragma clang force_cuda_host_device begin
#include "target_models.hpp"
#pragma clang force_cuda_host_device end
CPPROB_REGISTER_MODEL(models::hmm<16>);
"""


def test_statements_are_inlined_into_the_model_kernels_at_every_optimisation_level(tmp_path):
    """The statements are part of the model's kernel (the launch's mode reaches them as compile-time facts, their counters live in its
    registers, the step's observe ends its wavefront): at -O3 hipcc used to leave cpprob::sample / observe as CALLS -- and that build
    faulted on the device.  The device code of a model translation unit holds no call at -O3 (nor at -O2)."""
    src = tmp_path / "one_model.hip"
    src.write_text(_ONE_MODEL_TU)
    for opt in ("-O3", "-O2"):
        out = str(tmp_path / ("one_model%s.s" % opt))
        cmd = ["/opt/rocm/bin/hipcc", opt, "-std=c++17", "--offload-arch=gfx950", "-fPIC", "--cuda-device-only", "-S",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cpprob_amd", "include"), "-I", os.path.join(ROOT, "cpprob_amd", "examples"),
               "-o", out, str(src)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        asm = open(out).read()
        assert "model_step_kernel" in asm and "model_kernel" in asm
        assert "s_swappc_b64" not in asm and "s_setpc_b64" not in asm, opt


_STEP_BUILD_TU = r"""
#include "cpprob/gpu.hpp"
#pragma clang force_cuda_host_device begin
#include "target_models.hpp"
#pragma clang force_cuda_host_device end
using C = cpprob::gpu::FunctionCaller<decltype(&models::linear_gaussian_1d<25>), &models::linear_gaussian_1d<25>>;
using T = C::observes_t;
template __global__ void cpprob::gpu::model_step_kernel_at<C, T, 12, 1, 1, true, false>(cpprob::gpu::ModelKernelArgs, const T*);    // the step 12 alone
template __global__ void cpprob::gpu::model_step_kernel_at<C, T, 16, 1, 1, false, false>(cpprob::gpu::ModelKernelArgs, const T*);   // the steps from 16 on
template __global__ void cpprob::gpu::model_step_kernel<C, T>(cpprob::gpu::ModelKernelArgs, const T*);                              // the run-time kernel
"""


def test_step_kernels_built_for_a_step_hold_no_dead_iteration(tmp_path):
    """The step ordinal as a compile-time fact (cpprob/gpu.hpp: model_step_kernel_at): with the model's loop unrolled and the statement
    counters forwarded through LDS, the iterations in front of the step fold away -- the device code of the build for step 12 of
    linear_gaussian_1d<25> draws ONE normal variate (one Philox block: the step's own sample) and ends in its observe; the build for
    the steps from 16 on keeps the 9 iterations that may be live; both far smaller than the unrolled model (25 draws)."""
    import re
    from cpprob_amd import build as B
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_stats
    src = tmp_path / "steps.hip"
    src.write_text(_STEP_BUILD_TU)
    out = str(tmp_path / "steps.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "--cuda-device-only", "-S"] + B.STEPS_FLAGS + \
          ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cpprob_amd", "include"), "-I", os.path.join(ROOT, "cpprob_amd", "examples"), "-o", out, str(src)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    asm = open(out).read()
    assert "s_swappc_b64" not in asm                                              # the model is part of the kernel whatever its size
    # split the listing by kernel and count the Philox blocks (its first multiplier is loaded once per block)
    parts = re.split(r"\n(?=_ZN6cpprob3gpu\w+:)", asm)
    draws, size = {}, {}
    st = isa_stats.stats(out)
    for part in parts:
        name = part.split(":", 1)[0]
        if not name.startswith("_ZN6cpprob3gpu"):
            continue
        key = "exact" if "Li12ELi1ELi1ELb1ELb0E" in name else ("from" if "Li16ELi1ELi1ELb0ELb0E" in name else "runtime")
        draws[key] = part.lower().count("0xd2511f53")
        size[key] = st[name]["n"]
    assert draws["exact"] == 1 and draws["from"] == 9 and draws["runtime"] >= 25, draws
    assert size["exact"] < size["from"] < size["runtime"] and size["exact"] < 4000, size


def test_host_driver_is_plain_cpp14_and_fails_loudly_without_gpu():
    from cpprob_amd import build as B
    B.build_all()
    if not _no_gpu():
        pytest.skip("a GPU is present")
    p = subprocess.run([B.MAIN_BIN, "--model", "gaussian_unknown_mean", "--sis", "--observes", "3 4"], capture_output=True, text=True)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr
    # no HIP symbols are referenced by the host driver itself
    und = subprocess.check_output(["nm", "-u", B.MAIN_BIN]).decode()
    assert "hipMalloc" not in und and "hipLaunch" not in und


_NDARRAY_PROG = r"""
#include <cassert>
#include <cmath>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include "cpprob/cpprob.hpp"
#include "cpprob/postprocess/stats_printer.hpp"
int main(int argc, char** argv)
{
    using cpprob::NDArray;
    // text form (reference ndarray.hpp:273-288): one element -> bare number, vector -> [..], higher rank -> [..] s[..]
    std::ostringstream o;
    o << NDArray<double>(2.5) << '|' << NDArray<double>(std::vector<double>{1.5, -2}) << '|' << NDArray<double>(std::vector<double>{1, 2, 3, 4}, std::vector<std::size_t>{2, 2});
    assert(o.str() == "2.5|[1.5 -2]|[1 2 3 4] s[2 2]");
    NDArray<double> a, b, c;
    std::istringstream i1("[1.5 -2] 7 [1 2 3 4] s[2 2])");
    i1 >> a >> b >> c;
    assert(a == NDArray<double>(std::vector<double>{1.5, -2}) && b == NDArray<double>(7.0) && c.shape().size() == 2 && c.values().size() == 4);
    char close; i1 >> close; assert(close == ')');
    assert((a * a - a) == NDArray<double>(std::vector<double>{0.75, 6}) && (a * 2.0)[1] == -4 && (NDArray<double>() += a) == a);
    // diagonal multivariate normal: logpdf is the sum of the components' (utils_multivariate_normal.hpp:22-33)
    cpprob::multivariate_normal_distribution<> prior{{1, 2}, {std::sqrt(5), std::sqrt(3)}};
    const NDArray<double> x(std::vector<double>{0.3, 2.9});
    const double want = cpprob::logpdf<boost::random::normal_distribution<double>>()(boost::random::normal_distribution<double>(1, std::sqrt(5)), 0.3) +
                        cpprob::logpdf<boost::random::normal_distribution<double>>()(boost::random::normal_distribution<double>(2, std::sqrt(3)), 2.9);
    assert(std::fabs(cpprob::logpdf<cpprob::multivariate_normal_distribution<>>()(prior, x) - want) < 1e-15);
    std::mt19937 g(1);
    const auto draw = prior(g);
    assert(draw.size() == 2 && draw.shape() == std::vector<std::size_t>{2} && prior.mean() == NDArray<double>(std::vector<double>{1, 2}) && std::fabs(prior.covariance()[1] - 3) < 1e-12);
    cpprob::multivariate_normal_distribution<> lik{draw.begin(), draw.end(), 0.5};
    assert(lik.distr().size() == 2 && lik.distr()[1].sigma() == 0.5);
    // StatsPrinter on a vector-valued .real file: elementwise mean / variance (stats_printer.hpp:44-58, empirical_distribution.hpp:52-81)
    const std::string base = argv[1];
    { std::ofstream f(base + ".ids"); f << "Mu\n"; }
    { std::ofstream f(base + ".real"); f << "([(0 [1.0e+00 2.0e+00])] 0.0e+00)\n([(0 [3.0e+00 6.0e+00])] 0.0e+00)\n"; }
    cpprob::StatsPrinter sp(base);
    const auto m = sp.real(0)[0].mean();
    const auto v = sp.real(0)[0].variance(m);
    assert(m == NDArray<double>(std::vector<double>{2, 4}) && v == NDArray<double>(std::vector<double>{1, 4}));
    std::cout << sp;
    return 0;
}
"""


def test_ndarray_multivariate_normal_and_stats_printer_host_side(tmp_path):
    """SURVEY 8(f) row 4, host side (plain C++14): NDArray text form and arithmetic, the diagonal multivariate normal and its
    logpdf, StatsPrinter's elementwise estimators on a vector-valued posterior file."""
    src = tmp_path / "nd.cpp"
    src.write_text(_NDARRAY_PROG)
    exe = str(tmp_path / "nd")
    cmd = ["g++", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cpprob_amd", "include"), str(src), "-o", exe,
           "-L", os.path.join(ROOT, "cpprob_amd", "lib"), "-lcpprob_hip", "-Wl,-rpath," + os.path.join(ROOT, "cpprob_amd", "lib")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    out = subprocess.run([exe, str(tmp_path / "post")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Mean: [2 4]" in out.stdout and "Variance: [1 4]" in out.stdout and "Mu:" in out.stdout


def _cpp_nested(x, elem):
    if isinstance(x, list):
        depth = lambda v: 1 + depth(v[0]) if isinstance(v, list) else 0
        t = elem
        for _ in range(depth(x)):
            t = "std::vector<%s>" % t
        return "%s{%s}" % (t, ", ".join(_cpp_nested(e, elem) for e in x))
    return repr(float(x)) + ("f" if elem == "float" else "") if elem != "int" else str(int(x))


def test_ndarray_constructors_against_the_reference_test_vectors(tmp_path):
    """The reference's own NDArray tests (tests/cpprob/ndarray.cpp), as data: nested ragged containers become zero-padded
    row-major boxes; element types convert.  tests/golden/ndarray_cases.json holds inputs and expected values / shapes."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ndarray_cases.json")))
    body = []
    k = 0
    for c in g["cases"]:
        for elem in (["int", "float", "double"] if c["type"] == "double" else ["int"]):
            k += 1
            body.append("  { auto in = %s; cpprob::NDArray<%s> a(in.begin(), in.end());\n    check(%d, a.values() == std::vector<%s>{%s} && a.shape() == std::vector<std::size_t>{%s}); }" % (
                _cpp_nested(c["input"], elem), c["type"], k, c["type"], ", ".join(str(v) for v in c["values"]), ", ".join(str(v) for v in c["shape"])))
    prog = """#include <cstdio>
#include <vector>
#include "cpprob/ndarray.hpp"
static int bad = 0;
static void check(int k, bool ok) { if (!ok) { std::printf("case %%d FAILED\\n", k); ++bad; } }
int main() {
  { cpprob::NDArray<double> e{}; cpprob::NDArray<int> ei{}; check(-1, e.values().empty() && e.shape().empty() && ei.values().empty() && ei.shape().empty()); }
  { cpprob::NDArray<double> a{2.0}; cpprob::NDArray<int> b{3}; cpprob::NDArray<double> c{2}; cpprob::NDArray<double> d{3.0f};
    check(-2, a.values() == std::vector<double>{2.0} && a.shape() == std::vector<std::size_t>{1} && b.values() == std::vector<int>{3} &&
              c.values() == std::vector<double>{2.0} && d.values() == std::vector<double>{3.0} && d.shape() == std::vector<std::size_t>{1}); }
%s
  std::printf("%%d cases, %%d failed\\n", %d, bad);
  return bad;
}
""" % ("\n".join(body), k)
    src = tmp_path / "ndc.cpp"
    src.write_text(prog)
    exe = str(tmp_path / "ndc")
    p = subprocess.run(["g++", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "cpprob_amd", "include"), str(src), "-o", exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "0 failed" in out.stdout, out.stdout


_DISCRETE_PROG = r"""
#include <array>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include <boost/random/discrete_distribution.hpp>
int main()
{
    const std::array<double, 3> w{{0.2, 0.2, 0.6}};
    boost::random::discrete_distribution<std::size_t> d{w.begin(), w.end()};
    const auto pr = d.probabilities();
    const double tot = (0.2 + 0.2) + 0.6;
    if (pr.size() != 3 || pr[0] != 0.2 / tot || pr[1] != 0.2 / tot || pr[2] != 0.6 / tot) return 1;       // normalised on demand, the engine's order of sums
    if (d.weights()[2] != 0.6 || d.min() != 0 || d.max() != 2) return 2;                                   // the weights as given
    const std::vector<double> big{3, 1};
    boost::random::discrete_distribution<int> e{big.begin(), big.end()};
    if (e.probabilities()[0] != 0.75 || e.probabilities()[1] != 0.25) return 3;
    boost::random::discrete_distribution<int> one;                                                         // boost's default: a single outcome
    if (one.max() != 0 || one.probabilities()[0] != 1.0) return 4;
    boost::random::discrete_distribution<int> il{1.0, 1.0, 2.0};
    std::mt19937 g(5);
    int cnt[3] = {0, 0, 0};
    for (int i = 0; i < 40000; ++i) cnt[il(g)]++;
    if (std::fabs(cnt[2] / 40000.0 - 0.5) > 0.02 || std::fabs(cnt[0] / 40000.0 - 0.25) > 0.02) return 5;
    std::puts("ok");
    return 0;
}
"""


def test_discrete_distribution_stand_in_normalises_on_demand(tmp_path):
    """cpprob_amd/include/boost/random/discrete_distribution.hpp (clean-room interface stand-in): construction copies the weights and
    nothing else (a distribution built in a replayed, dead iteration of a model's loop costs no division on the device),
    probabilities() normalises when asked -- the same numbers as normalising at construction -- and the host draw follows them."""
    src = tmp_path / "dd.cpp"
    src.write_text(_DISCRETE_PROG)
    exe = str(tmp_path / "dd")
    p = subprocess.run(["g++", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "cpprob_amd", "include"), str(src), "-o", exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stdout, out.stderr)
