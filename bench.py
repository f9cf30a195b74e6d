#!/usr/bin/env python3
"""Headline benchmark: particles/sec of the SIS/SMC hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 it is launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL); run directly with --gpus N > 1 it starts that launcher itself as a child.

A "step" is one complete inference run of the workload: BASELINE.json's metric is quoted on
"HMM T=16 SMC" = configs[2]: 3-state HMM (reference include/models/models.hpp:114-141), T = 16
observes, 10^6 particles per GPU, systematic resampling after every step, including the final
posterior read-out (smoothing marginals).  Observes are the committed synthetic vector
tests/golden/observations.npz (simulated from the model); particles are drawn on
device, so inputs are resident in HBM when the timed region starts.

value        = particles (complete T-step traces) per second over all GPUs
roofline     = dominant kernel (smc_step): algorithmic bytes per launch (56 B per
               particle-step for the HMM) / mean launch duration measured with HIP events on the
               engine's own stream in a separate profiled pass of the same K steps
cpu_baseline = the CPU oracle (oracle/cpprob_oracle.c, a port: the reference itself cannot be
               built here) on one host core, compute-only, on a bounded sample
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES = {"hmm": 56, "lgssm": 72, "gaussian_sis": 32}  # SURVEY 8(d) per particle(-step)
# What THIS data layout has to move through HBM per particle-step (every array read or written once, reads of
# neighbouring tiles and gathers of distinct ancestors served by the L2), by step form -- the bytes `frac_layout` prices a launch
# against; it cannot exceed 1 by construction, unlike the SURVEY 8(d) convention (`frac`), which prices arrays the layout does not have.
#   counts (headline)  : write state 1 + ancestor 4 + trace word 4; read source states 1 + ancestor's trace word 4
#   fixed, int8 states : write state 1 + ancestor 4 (resampling launches) + integer weight 4 + log-weight 8; read state 1 + weight 4 / log-weight 8
#   fixed, fp64 states : write value 8 + ancestor 4 + integer weight 4 + log-weight 8; read value 8 + weight 4 (or the carried log-weight 8)
LAYOUT_BYTES = {("hmm", 1): 14, ("hmm", 2): 22, ("hmm", 0): 56, ("lgssm", 2): 36, ("lgssm", 0): 72, ("gaussian_sis", 0): 16, ("gaussian_sis", 1): 16, ("gaussian_sis", 2): 16}
FORM_NAMES = {0: "floating-point", 1: "prefix counts (table weights, every-step schedule)", 2: "fixed-point masses"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--particles", type=int, default=1_000_000, help="particles per GPU")
    p.add_argument("--workload", default="hmm16_smc", choices=["hmm16_smc", "hmm128_smc_ess", "lgssm100_smc", "gaussian_sis"])
    p.add_argument("--scope", default="auto", choices=["auto", "global", "global-deferred", "island", "exchange"])
    p.add_argument("--resampler", default="systematic", choices=["systematic", "stratified", "multinomial", "multinomial_literal"],
                   help="one GPU: the resampler of the timed context (thesis Alg. 1 p.36 is multinomial; the headline metric is quoted on systematic)")
    p.add_argument("--seed", type=int, default=12345)
    p.add_argument("--flags", type=int, default=0, help="cpprob_hip_config::flags of the timed context (A/B forms, include/cpprob_hip.h); recorded in config")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=10_000_000, help="particles of the CPU-baseline sample")
    p.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (pipelined runs, gaussian SIS)")
    p.add_argument("--no-live-pmc", action="store_true", help="do not spawn the two rocprofv3 --pmc passes that measure roofline.traffic (e.g. when bench.py itself runs under a profiler)")
    p.add_argument("--in-flight", type=int, default=3, help="contexts in flight for the secondary pipelined measurement")
    p.add_argument("--loopback-ranks", type=int, default=0, help="N = 1 only: run the library's multi-GPU driver with this many ranks on the one GPU "
                   "(loopback transport: the whole exchange protocol, program order instead of collectives); --particles is then the WHOLE population")
    p.add_argument("--python-host", action="store_true", help="N > 1: drive the exchange scope from Python / torch.distributed instead of the library's C++ driver")
    return p.parse_args()


def relaunch_under_torchrun(args):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def workload_spec(name, golden):
    import numpy as np
    import cpprob_amd as cp
    z = np.load(golden)
    if name == "hmm16_smc":
        return dict(alg=cp.ALG_SMC, model=cp.MODEL_HMM3, obs=z["hmm16"], ess=2.0, bytes_key="hmm", exact=z["hmm16_smooth"], exact_logz=float(z["hmm16_logz"]),
                    desc="3-state HMM (models.hpp:114-141) SMC, T=16, systematic resampling every step")
    if name == "hmm128_smc_ess":
        return dict(alg=cp.ALG_SMC, model=cp.MODEL_HMM3, obs=z["hmm128"], ess=0.5, bytes_key="hmm", exact=z["hmm128_smooth"], exact_logz=float(z["hmm128_logz"]),
                    desc="3-state HMM SMC, T=128, resample when ESS < N/2")
    if name == "lgssm100_smc":
        return dict(alg=cp.ALG_SMC, model=cp.MODEL_LINEAR_GAUSSIAN_1D, obs=z["lgssm100"], ess=0.5, bytes_key="lgssm", exact_logz=float(z["lgssm100_logz"]),
                    exact=np.stack([z["lgssm100_smooth_mean"], z["lgssm100_smooth_var"]], 1),
                    desc="linear_gaussian_1d<100> (models.hpp:67-80) SMC, resample when ESS < N/2")
    return dict(alg=cp.ALG_SIS, model=cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, obs=np.array([3.0, 4.0]), ess=2.0, bytes_key="gaussian_sis",
                exact=np.array([[3.0833333333333335, 0.8333333333333334]]), desc="gaussian_unknown_mean (models.hpp:22-35) SIS, observes (3,4)")


def resampler_of(name):
    """(cpprob_hip resampler id, extra cpprob_hip_config::flags) of a --resampler name."""
    import cpprob_amd as cp
    return {"systematic": (cp.RESAMPLE_SYSTEMATIC, 0), "stratified": (cp.RESAMPLE_STRATIFIED, 0), "multinomial": (cp.RESAMPLE_MULTINOMIAL, 0),
            "multinomial_literal": (cp.RESAMPLE_MULTINOMIAL, cp.capi.FLAG_MULTINOMIAL_LITERAL)}[name]


def usable_cores():
    """Host cores this process may actually use: its affinity mask, cut by the cgroup's CPU quota where there is one."""
    import math
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:       # noqa
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        quota = q / float(f2.read().split()[0])
            break
        except Exception:   # noqa
            continue
    if quota is not None:
        n = max(1, min(n, int(math.ceil(quota))))
    return n, quota


def timed_runs(eng, steps, warmup, world, device, island, first_index=0, exchange=False, counters=None):
    """W untimed + exactly K timed runs bracketed by barrier + synchronize; returns max-over-ranks seconds."""
    import torch
    import torch.distributed as dist
    from cpprob_amd import distributed as D

    coll = D.TorchCollective(eng) if ((world > 1 and not island) or exchange) else None
    batch = D.IslandBatch(eng, min(max(steps, warmup, 1), 64)) if (world > 1 and island) else None
    bufs = None
    if coll is not None:
        bufs = (torch.zeros(4, dtype=torch.float64, device=device), torch.zeros(3 * world, dtype=torch.float64, device=device), None)
    jslots = None
    if coll is not None and not exchange:
        jdepth = min(max(steps, warmup, 1), 64)
        jslots = torch.zeros((jdepth, 4 + eng.T * eng.K), dtype=torch.float64, device=device)

    torch.cuda.synchronize()          # buffers above were filled on torch's stream; the engine works on its own

    def one(i):
        if exchange:
            st, _ = D.run_exchange(eng, coll, i, counters)   # per step: all-gather of 3 doubles + one all-to-all-v of migrating lineages
            return (st,)
        if world > 1 and island:
            batch.run(i % batch.depth, i)                 # no data-path collective; one async all-gather of summaries per run
            return None
        if world > 1:
            D.run_joint(eng, coll, i, bufs, slot=jslots[i % jslots.shape[0]])   # one RCCL all-gather of 3 doubles per rank per step; no host sync
            return None
        eng.run(i)
        return None

    for i in range(warmup):
        one(first_index + i)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = one(first_index + warmup + i)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if batch is not None:
        last = batch.results((first_index + warmup + steps - 1) % batch.depth)
    elif jslots is not None:
        st, js = D.joint_results(eng, jslots[(first_index + warmup + steps - 1) % jslots.shape[0]])
        last = (st, js["log_evidence"], 0.0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def guarded(fn, what):
    """Runs fn() on a helper thread with a deadline.  Creating an RCCL communicator and the first run over real links are
    collective and blocking: a transport that never completes cannot be recovered in-process, so the job ends with a message
    instead of holding the node until the launcher's own limit.  Exceptions of fn() are re-raised here."""
    import threading
    box = {}
    cur = None
    try:
        import torch
        cur = torch.cuda.current_device() if torch.cuda.is_available() else None
    except Exception:               # noqa
        cur = None
    def body():
        try:
            if cur is not None:
                import torch
                torch.cuda.set_device(cur)          # (the current device is a per-thread setting: a fresh thread starts on device 0)
            box["value"] = fn()
        except BaseException as e:        # noqa: handed to the caller
            box["error"] = e
    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(float(os.environ.get("CPPROB_BENCH_COLLECTIVE_TIMEOUT", "180")))
    if th.is_alive():
        sys.stderr.write("bench.py: %s did not complete within its deadline (rank %s); exiting\n" % (what, os.environ.get("RANK", "?")))
        sys.stderr.flush()
        os._exit(3)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def timed_group_runs(group, steps, warmup, world, device, first_index=0):
    """The same bracket for the library's own multi-GPU driver (cpprob_hip_group_run: a whole exchange-scope run per call,
    RCCL collectives on the context's stream, no host synchronisation inside)."""
    import torch
    import torch.distributed as dist
    # settle the transport before anything is timed: one run + results() (collective: every rank enters it), which repeats the run
    # with larger segments / annex if they proved too small -- a pipelined batch is never repeated, so it must not overflow
    def first():
        group.run(first_index)
        return group.results()[2]
    settle = guarded(first, "the first multi-GPU run (RCCL all-gather / direct stores or send / receive on the contexts' streams)") if world > 1 else first()
    for i in range(warmup):
        group.run(first_index + i)
    group.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        group.run(first_index + warmup + i)
    group.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    stats, summ, reruns = group.results()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # behind the timed region: ONE more run with HIP events between the launches of every step -- where a rank-step's time goes
    # (step + totals, all-gather, pack, barrier, commit; how long the mailbox waits spun) and which transports the group settled on
    breakdown = None
    try:
        group.profile(True)
        group.run(first_index + warmup + steps)
        group.results()
        breakdown = group.profile_read()
        group.profile(False)
        breakdown["note"] = ("HIP events around the launches of one extra run (us per rank-step, the slowest local rank; a loopback group: the sum over its "
                             "ranks, which share a stream); mailbox_wait = of all that, the time the waits spun on their peers' sequence numbers")
        breakdown["transport_note"] = group.note()
    except Exception as e:        # noqa: reported under the key
        breakdown = {"error": str(e)}
    return dt, stats, summ, {"settling": settle, "timed_batch": reruns - settle, "rank_step_breakdown_us": breakdown}


def make_rank_group(cp, world, rank, local):
    """This process's rank of a group spread over the launcher's processes: rank 0 draws the RCCL unique id, torch.distributed
    (already initialised by the launcher's environment) hands it to everybody."""
    import torch.distributed as dist
    box = [cp.Group.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return cp.Group([local], world=world, first_rank=rank, unique_id=box[0])


def make_external_group(cp, world, rank, local, device):
    """This process's rank of a group whose set-up exchanges and final reduction go through torch.distributed (the launcher's own
    process group: gloo or nccl); inside a run the ranks talk through their mapped mailboxes and direct stores
    (cpprob_hip_group_create_external)."""
    import torch
    import torch.distributed as dist
    on_host = dist.get_backend() == "gloo"

    def allgather(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        if on_host:
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            return b"".join(bytes(o.numpy().tobytes()) for o in outs)
        t = t.to(device)
        out = torch.empty(world * t.numel(), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(out, t)
        return bytes(out.cpu().numpy().tobytes())
    return cp.Group([local], world=world, first_rank=rank, allgather=allgather)


def _oracle_run(args):
    """One oracle run (worker of the all-cores baseline; top-level so that multiprocessing can pickle it)."""
    alg_is_sis, model, obs, n_sample, seed, ess = args
    from oracle import oracle as O
    if alg_is_sis:
        vals, lw = O.sis(model, obs, n_sample, seed)
        O.weighted_moments(vals[0], lw)
    else:
        r = O.smc(model, obs, n_sample, seed, O.RESAMPLE_SYSTEMATIC, ess)
        O.smoothing(r["hist"], r["anc"], r["logw"])
    return n_sample


def cpu_baseline_worker(workload, n_sample, seed):
    """Runs in a FRESH interpreter (no GPU state, so forking a process pool is safe): the oracle timed on the host
    (BASELINE.md section 2), bounded samples:
    value      = mode B: one core, compute-only, in memory (the reference is single-threaded with global state);
    all_cores  = mode C: independent replicas with distinct seeds on every host core (what processes could do)."""
    import multiprocessing as mp
    import cpprob_amd as cp
    from oracle import oracle as O
    O.lib()
    spec = workload_spec(workload, os.path.join(ROOT, "tests", "golden", "observations.npz"))
    is_sis = spec["alg"] == cp.ALG_SIS
    t0 = time.perf_counter()
    _oracle_run((is_sis, spec["model"], spec["obs"], n_sample, seed, spec["ess"]))
    dt = time.perf_counter() - t0
    out = {"value": n_sample / dt, "unit": "particles/s", "cores": 1, "kind": "port",
           "sample": "%d particles of the same workload (all T steps + read-out), oracle/cpprob_oracle.c -O2, in-memory (no file dumps), %.1f s"
                     % (n_sample, dt), "host_cores_advertised": os.cpu_count()}
    try:
        # mode C of BASELINE.md: every core this process may USE (affinity mask, cgroup quota), independent replicas.  If the
        # replicas do not scale (a lease that grants less than it shows), the pool is halved until they do: `cores` is the count
        # at which speedup / cores >= 0.7, not the count the box advertises.
        procs, quota = usable_cores()
        out["host_cores_usable"] = procs
        out["cgroup_cpu_quota"] = quota
        per = max(10000, n_sample // 16)
        tried = []
        while True:
            t0 = time.perf_counter()
            with mp.get_context("fork").Pool(procs) as pool:
                done = sum(pool.map(_oracle_run, [(is_sis, spec["model"], spec["obs"], per, seed + 1 + i, spec["ess"]) for i in range(procs)]))
            dt2 = time.perf_counter() - t0
            speedup = (done / dt2) / out["value"]
            tried.append({"cores": procs, "value": done / dt2, "parallel_speedup": speedup})
            if speedup / procs >= 0.7 or procs == 1 or len(tried) >= 6:
                break
            procs = max(1, min(procs // 2, int(speedup * 1.3) + 1))
        # (the baseline is the BEST absolute rate over the pool sizes tried -- a larger pool at lower efficiency may still be the faster
        #  machine; the efficiency of the pool that gave it rides along)
        best = max(tried, key=lambda r: r["value"])
        out["all_cores"] = {"value": best["value"], "cores": best["cores"], "parallel_speedup": best["parallel_speedup"], "efficiency": best["parallel_speedup"] / best["cores"],
                            "sample": "%d independent replicas of %d particles each (last pool: %.1f s)" % (best["cores"], per, dt2), "pool_sizes_tried": tried}
    except Exception as e:   # the baseline is a report, never a reason to fail the bench
        out["all_cores"] = {"error": str(e)}
    return out


def cpu_baseline(workload, n_sample, seed):
    """Child process: the CPU legs never share a process with the HIP runtime."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", workload, str(n_sample), str(seed)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        return {"error": p.stderr[-500:]}
    return json.loads(p.stdout.strip().splitlines()[-1])


def cpu_as_shipped(n_sample, seed, model=None, obs=None, address="Mu", label="gaussian_unknown_mean"):
    """Mode A of BASELINE.md for the SIS path: what the reference actually does per particle -- three append-mode
    file open/write/close (src/cpprob/state.cpp:193-202,262-267) and a progress line every 100 traces."""
    import tempfile
    import numpy as np
    from oracle import oracle as O
    d = tempfile.mkdtemp(prefix="cpprob_as_shipped_")
    t0 = time.perf_counter()
    O.sis_faithful(O.MODEL_GAUSSIAN_UNKNOWN_MEAN if model is None else model, np.array([3.0, 4.0]) if obs is None else obs, n_sample, seed,
                   os.path.join(d, "posterior"), address, progress=False)
    dt = time.perf_counter() - t0
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
    return {"value": n_sample / dt, "unit": "particles/s", "cores": 1, "sample": "%d particles, %s SIS with per-particle file dumps, %.1f s" % (n_sample, label, dt)}


def pmc_child(workload, n, resampler, seed, flags):
    """What the live PMC passes profile: the timed context's configuration, a few runs, nothing else (no floor run, no extras)."""
    import torch  # noqa: F401
    import cpprob_amd as cp
    spec = workload_spec(workload, os.path.join(ROOT, "tests", "golden", "observations.npz"))
    rid, rfl = resampler_of(resampler)
    eng = cp.Engine(0)
    eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=seed, resampler=rid, ess_threshold=spec["ess"], flags=flags | rfl)
    for i in range(4):
        eng.run(i)
    eng.sync()
    eng.close()


def live_pmc_traffic(workload, n, resampler, seed, flags, dom):
    """HBM bytes per launch of the dominant kernel, measured NOW: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: nothing
    but --pmc on the command line, as MI355X_MICROARCH.md's HBM section prescribes) of a child process that runs this configuration,
    corrected as that section says for gfx950 (FETCH_SIZE counts half the bytes of wide coalesced reads).  None when the profiler
    is not there or a pass fails: the line then falls back to the committed profile and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    sq = None
    # (a third pass of its own for the issue counters: wave cycles, cycles waiting on anything, cycles issuing vector instructions, instruction counts)
    for group in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES")):
        ctr = group[0]
        d = tempfile.mkdtemp(prefix="cpprob_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc"] + list(group) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child", workload, str(n), resampler, str(seed), str(flags)]
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not fs:
                if len(group) > 1:
                    sq = {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(group), p.returncode, p.stderr[-200:])}
                    continue
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctr, p.returncode, p.stderr[-200:]), None
            acc = {c: [] for c in group}
            for r in csv.DictReader(open(fs[0])):
                name = r.get("Kernel_Name", "")
                hit = ("smc_step" in name) if dom == "smc_step" else ("sis_" in name and "finish" not in name)
                if hit and r.get("Counter_Name") in acc:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if len(group) > 1:
                if all(acc[c] for c in group):
                    tot = {c: sum(acc[c]) for c in group}
                    sq = {"wait_frac": tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1.0), "valu_issue_frac": tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1.0),
                          "valu_insts_per_wave": tot["SQ_INSTS_VALU"] / max(tot["SQ_WAVES"], 1.0), "salu_insts_per_wave": tot["SQ_INSTS_SALU"] / max(tot["SQ_WAVES"], 1.0),
                          "launches": len(acc["SQ_WAVE_CYCLES"]),
                          "measured_in": "this run: one rocprofv3 --pmc pass of the issue counters (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES) over %d launches" % len(acc["SQ_WAVE_CYCLES"])}
                else:
                    sq = {"error": "no issue-counter samples of the %s kernel" % dom}
                continue
            if not acc[ctr]:
                return None, "no %s samples of the %s kernel" % (ctr, dom), None
            vals[ctr] = (sum(acc[ctr]) / len(acc[ctr]), len(acc[ctr]))
        except Exception as e:      # noqa: the line falls back and says so
            if len(group) > 1:
                sq = {"error": "rocprofv3 --pmc (issue counters): %s" % e}
                continue
            return None, "rocprofv3 --pmc %s: %s" % (ctr, e), None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = 2.0 * vals["FETCH_SIZE"][0] * 1024.0 + vals["WRITE_SIZE"][0] * 1024.0
    return traffic, ("measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE over %d launches, WRITE_SIZE over %d) of a child process on this GPU, "
                     "bytes = 2 x FETCH_SIZE KB x 1024 + WRITE_SIZE KB x 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reads half)" % (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1])), sq


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        print(json.dumps(cpu_baseline_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))))
        return
    if len(sys.argv) >= 7 and sys.argv[1] == "--pmc-child":
        pmc_child(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]))
        return
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        sys.exit(relaunch_under_torchrun(args))   # child process; nothing has touched the GPU yet
    # stdout carries ONE line, the JSON: libraries that print there (RCCL announces its version on the first communicator) are
    # sent to stderr for the whole run; the line itself goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import cpprob_amd as cp
    from cpprob_amd import distributed as D

    world, rank, local = D.init_process_group(device_is_gpu=True)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    eng = cp.Engine(local)

    spec = workload_spec(args.workload, os.path.join(ROOT, "tests", "golden", "observations.npz"))
    n = args.particles
    T = 1 if args.workload == "gaussian_sis" else len(spec["obs"])
    scope = args.scope
    smc = spec["alg"] == cp.ALG_SMC
    if scope == "auto":
        # SURVEY 8(e) / north_star: one joint population, resampled exactly as ONE GPU holding all particles would -- per step an
        # all-gather of the rank totals and the redistribution of the offspring whose ancestor sits on another GPU (exchange scope).
        # The migration-free scopes (island / global) are measured beside it as labelled secondary numbers.
        scope = "exchange" if (world > 1 and smc) else "global"
    island = scope in ("island", "global-deferred")
    exchange = scope == "exchange" and smc
    n_global = world * n
    # the library's own C++ driver issues the collectives (RCCL) unless several ranks share one GPU (test hook: RCCL refuses that)
    native = exchange and world > 1 and not D._host_collectives() and not args.python_host
    host = "python (torch.distributed)"
    group = None
    native_error = None
    if native:
        try:
            group = guarded(lambda: make_rank_group(cp, world, rank, local), "creating the RCCL communicator of the library's driver")
            host = "C++ (cpprob_hip_group_run: RCCL for set-up and the final reduction, mailboxes per step, no host synchronisation inside a run)"
        except Exception as e:      # reported, never silent: the next rung is the same driver on torch.distributed's collectives
            native_error = str(e)
    if exchange and world > 1 and group is None and not args.python_host:
        # the library's driver with the launcher's process group for the set-up exchanges and the final reduction (several ranks on
        # one GPU -- the gloo test hook -- come here directly: RCCL refuses duplicate devices)
        try:
            group = guarded(lambda: make_external_group(cp, world, rank, local, device), "creating the library's driver on torch.distributed's collectives")
            host = "C++ (cpprob_hip_group_run: torch.distributed %s for set-up and the final reduction, mailboxes per step)" % torch.distributed.get_backend()
        except Exception as e:
            native_error = (native_error + "; " if native_error else "") + str(e)
    moved = {}
    reruns = 0
    xtraffic = None
    rs_id, rs_flags = resampler_of(args.resampler)
    if args.resampler not in ("systematic", "stratified", "multinomial") and (world > 1 or args.loopback_ranks > 1):
        sys.stderr.write("bench.py: --resampler %s runs on one GPU (the exchange scope resamples systematically, stratified or multinomially in the strata form)\n" % args.resampler)
        sys.exit(2)
    if world == 1 and args.loopback_ranks > 1 and smc:
        group = cp.Group([local] * args.loopback_ranks)
        n_global, exchange, scope = n, True, "exchange"
        host = "C++ (cpprob_hip_group_run), %d loopback ranks on one GPU" % args.loopback_ranks
    preflight = None
    if group is not None and world > 1:
        # Before anything is timed over real links: the transport the group settles on, and a small exchange-scope run whose every
        # surviving trace is compared, bit for bit, with the same population run on THIS rank's GPU alone (the integer forms of the
        # step make that an exact test: any stale peer read, lost remote store or mis-ordered flag shows as a different trace).  On a
        # mismatch the conservative rungs are tried in turn and the line says which one produced the single-GPU answer.
        def run_preflight():
            n_pf = 200_000
            n_tot = n_pf * world
            eng.begin(spec["alg"], spec["model"], spec["obs"], n_tot, seed=args.seed + 7, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"])
            eng.run()
            ref_paths = eng.paths()[:, rank * n_pf:(rank + 1) * n_pf]
            ref_lz = eng.summary()["log_evidence"]
            rungs = [("default", 0), ("library collectives", cp.capi.GROUP_LIBRARY_COLLECTIVES), ("library collectives + shipped lineages", cp.capi.GROUP_LIBRARY_COLLECTIVES | cp.capi.GROUP_SHIP_LINEAGES),
                     ("send / receive", cp.capi.GROUP_LIBRARY_COLLECTIVES | cp.capi.GROUP_SENDRECV)]
            rep = {"particles_per_rank": n_pf, "rungs": []}
            for name, fl in rungs:
                group.transport(flags=fl)
                group.begin(spec["alg"], spec["model"], spec["obs"], n_tot, seed=args.seed + 7, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"], shard_sizes=[n_pf] * world)
                t0 = time.perf_counter()
                group.run(0)
                _, gs, rr = group.results()
                ms = (time.perf_counter() - t0) * 1e3
                e = group.context(0); e.n = n_pf; e.T = T
                same = bool(np.array_equal(e.paths(), ref_paths)) and gs["log_evidence"] == ref_lz
                flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device="cpu" if torch.distributed.get_backend() == "gloo" else device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = bool(flag.item() > 0.5)
                rep["rungs"].append({"transport": name, "traces_equal_single_gpu_run_on_every_rank": ok, "first_run_ms": ms, "reruns": rr, "note": group.note()})
                if ok:
                    rep["settled_on"] = name
                    break
            else:
                rep["settled_on"] = None
            try:
                import torch.cuda as tc
                rep["peer_access"] = [[bool(a == b or tc.can_device_access_peer(a, b)) for b in range(tc.device_count())] for a in range(tc.device_count())] if rank == 0 else None
            except Exception as e2:      # noqa
                rep["peer_access"] = str(e2)
            return rep
        try:
            preflight = guarded(run_preflight, "the multi-GPU preflight (a 2e5-particle-per-rank exchange-scope run against the single-GPU run)")
        except Exception as e:          # noqa: reported under the key, never a reason to lose the line
            preflight = {"error": str(e)}
        if isinstance(preflight, dict) and preflight.get("settled_on") not in (None, "default"):
            native_error = (native_error + "; " if native_error else "") + "preflight: the default transport did not reproduce the single-GPU traces; measuring '%s'" % preflight["settled_on"]
        if isinstance(preflight, dict) and "rungs" in preflight and preflight.get("settled_on") is None:
            # no rung reproduced the single-GPU traces: say so, and time the DEFAULT transport rather than whichever rung was tried last
            native_error = (native_error + "; " if native_error else "") + "preflight: NO transport reproduced the single-GPU traces; the default transport is measured and its posterior checked below"
            group.transport(flags=0)
    if group is not None:
        def begin_group():
            group.begin(spec["alg"], spec["model"], spec["obs"], n_global, seed=args.seed, resampler=rs_id, ess_threshold=spec["ess"])
        # (begin is collective over real links: it all-gathers hipIpc handles and proves the mailboxes by a round trip)
        guarded(begin_group, "cpprob_hip_group_begin (peer mappings, mailbox round trip)") if world > 1 else begin_group()
        dt, stats, summ, reruns = timed_group_runs(group, args.steps, args.warmup, world, device)
        xtraffic = group.traffic()
        # Over real links the default transport (direct stores into peers' memory, mailbox collectives, trace words across ranks) has
        # only ever run on one GPU: if the answer it produced is not the posterior, say so and measure the conservative path instead
        # (library collectives, lineages shipped, read-out by the walk) -- a wrong number is never the line.  Every rank takes the
        # same decision: the statistics are the all-reduced ones.
        if world > 1 and float(np.abs(stats - spec["exact"]).max()) > 0.05:
            sys.stderr.write("bench.py: the default multi-GPU transport produced a wrong posterior (max abs err %.3g); measuring the conservative transport instead\n"
                             % float(np.abs(stats - spec["exact"]).max()))
            native_error = (native_error + "; " if native_error else "") + "default transport produced a wrong posterior: conservative transport measured"
            group.transport(flags=cp.capi.GROUP_LIBRARY_COLLECTIVES | cp.capi.GROUP_SHIP_LINEAGES)
            def begin_conservative():
                group.begin(spec["alg"], spec["model"], spec["obs"], n_global, seed=args.seed, resampler=rs_id, ess_threshold=spec["ess"],
                            flags=cp.capi.FLAG_WALK_READOUT)
            guarded(begin_conservative, "cpprob_hip_group_begin (conservative transport)")
            dt, stats, summ, reruns = timed_group_runs(group, args.steps, args.warmup, world, device)
            xtraffic = group.traffic()
        last = (stats,)
    else:
        eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=rs_id, ess_threshold=spec["ess"],
                  particle_offset=rank * n, n_global=n_global, scope=cp.SCOPE_ISLAND if island else (cp.SCOPE_EXCHANGE if exchange else cp.SCOPE_GLOBAL), flags=args.flags | rs_flags)
        dt, last = timed_runs(eng, args.steps, args.warmup, world, device, island, exchange=exchange, counters=moved)
        summ = None
    value = n_global * args.steps / dt

    # posterior of the last timed run against the exact answer (validity of what was timed)
    if last is not None:
        stats = last[0]
    else:
        stats = eng.stats()
    err = float(np.abs(stats - spec["exact"]).max())
    if summ is None:
        summ = eng.summary()
    if last is not None and len(last) == 3:
        summ["log_evidence"] = last[1]          # evidence of the joint population (shards combined), not of this rank's shard
    if exchange and xtraffic is not None:
        tname = {0: "none", 1: "direct stores into the receiving rank's buffer (peer access / hipIpc), ordered by a 1-double all_gather", 2: "ncclSend/ncclRecv of fixed-capacity segments"}[xtraffic["transport"]]
        collective = "all_gather(24 bytes/rank) per step; migrating lineages: %s; all_reduce(T*K+3 doubles) per run" % tname
    elif exchange:
        collective = "all_gather(3 doubles/rank) + all_to_all_v of the migrating lineages per step (torch.distributed host); all_reduce(T*K+1 doubles) per run"
    else:
        collective = "none" if world == 1 else ("all_gather(4+T*K doubles/rank) once per run, asynchronous (no host synchronisation between runs)" if island else "all_gather(3 doubles/rank) per step + all_reduce(T*K doubles) per run")

    # profiled pass: same K steps with HIP events around every launch on the engine's stream
    # (per-shard kernels only: on several GPUs each rank profiles its own shard as an island)
    n_prof = n if args.loopback_ranks <= 1 else max(1, n // args.loopback_ranks)       # (one loopback rank's shard)
    if (world > 1 and not island) or exchange or group is not None:
        eng.begin(spec["alg"], spec["model"], spec["obs"], n_prof, seed=args.seed, resampler=rs_id, ess_threshold=spec["ess"],
                  particle_offset=rank * n_prof, n_global=world * n_prof, scope=cp.SCOPE_ISLAND, flags=rs_flags)
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    for i in range(args.steps):
        eng.run(10_000 + i)
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)
    dom = "sis" if spec["alg"] == cp.ALG_SIS else "smc_step"
    dom_ms, dom_calls = prof[dom]
    avg_s = dom_ms * 1e-3 / max(dom_calls, 1)
    n_res_prof = eng.summary()["n_resampled"]
    step_form = int(eng.summary().get("step_form", 0))
    # the floor of a launch of this chain: the same run at 4096 particles (the same launches with nothing in them)
    eng.begin(spec["alg"], spec["model"], spec["obs"], 4096, seed=args.seed, resampler=rs_id, ess_threshold=spec["ess"], flags=rs_flags)
    for i in range(3):
        eng.run(i)
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    for i in range(max(5, min(args.steps, 20))):
        eng.run(11_000 + i)
    floor = eng.profile_read(reset=True)
    eng.profile_enable(False)
    floor_us = floor[dom][0] * 1e3 / max(floor[dom][1], 1)
    bytes_per_unit = ALGO_BYTES[spec["bytes_key"]]
    light = {"hmm": 24, "lgssm": 32}.get(spec["bytes_key"], bytes_per_unit)      # a step that does not resample moves 2s + 2w
    if spec["alg"] == cp.ALG_SIS:
        bytes_per_unit = 16  # the sis kernel's own share of the 32 B: it writes value + logw; the read-out pass reads them back
        bytes_per_launch = bytes_per_unit * n_prof
    else:
        # launches of a run: step 0 and the steps behind a generation that was not resampled move the light figure, the others the full one
        full_launches = n_res_prof
        bytes_per_launch = n_prof * (full_launches * bytes_per_unit + (T - full_launches) * light) / T
    achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM bytes per launch and issue counters from the PMC passes: collected by separate rocprofv3 --pmc runs of this same command
    # (tools/profile_round.sh) and committed with their correction notes; bench.py itself cannot run the profiler
    traffic, traffic_src, valu_frac, wait_frac, sq_src, sq_live = None, None, None, None, None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if pmcs and spec["alg"] == cp.ALG_SMC and args.loopback_ranks <= 1:
        with open(pmcs[-1]) as f:
            pj = json.load(f)
        rec = pj.get("workloads", {}).get("%s@%d" % (args.workload, n_prof))
        if rec:
            traffic = rec["step_kernel"]["hbm_bytes_per_launch_corrected"]
            traffic_src = "committed profile profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on another box, gfx950-corrected): NOT measured in this run" % os.path.basename(pmcs[-1])
            valu_frac = rec["step_kernel"].get("valu_issue_frac")
            wait_frac = rec["step_kernel"].get("wait_frac")
            sq_src = "committed profile profiles/%s: NOT measured in this run" % os.path.basename(pmcs[-1])
    if rank == 0 and world == 1 and args.loopback_ranks <= 1 and not args.no_live_pmc:
        live, why, sq_live = live_pmc_traffic(args.workload, n_prof, args.resampler, args.seed, args.flags, dom)
        if sq_live and "wait_frac" in sq_live:
            valu_frac, wait_frac, sq_src = sq_live["valu_issue_frac"], sq_live["wait_frac"], sq_live["measured_in"]
        elif sq_live:
            sq_src = (sq_src or "no committed profile") + "; live issue-counter pass unavailable: " + sq_live.get("error", "?")
        if live is not None:
            traffic, traffic_src = live, why
        else:
            traffic_src = (traffic_src or "no committed profile for this workload and size") + "; live PMC passes unavailable: " + str(why)
    hbm_frac_measured = (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_s > 0) else None
    # what bounds the dominant kernel at THIS size: within 2x of an empty launch of the same chain it is the chain's latency
    # (kernel boundary, first round trip to memory, search, gather), whatever the byte convention says
    bound = "latency" if (avg_s * 1e6 <= 2.0 * floor_us) else "hbm"
    layout_bytes = LAYOUT_BYTES.get((spec["bytes_key"], step_form), bytes_per_unit)
    achieved_layout = layout_bytes * n_prof / avg_s / 1e9 if avg_s > 0 else 0.0
    roofline = {"bound": bound, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "step_form": FORM_NAMES.get(step_form, str(step_form)) if spec["alg"] == cp.ALG_SMC else None,
                "layout_bytes_per_unit": layout_bytes, "achieved_layout": achieved_layout, "frac_layout": achieved_layout / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_measured_in": traffic_src, "hbm_frac_measured": hbm_frac_measured, "valu_issue_frac": valu_frac, "wait_frac": wait_frac, "issue_counters_measured_in": sq_src,
                "valu_insts_per_wave": sq_live.get("valu_insts_per_wave") if sq_live else None, "salu_insts_per_wave": sq_live.get("salu_insts_per_wave") if sq_live else None,
                "launch_floor_us": floor_us, "floor_frac": floor_us / (avg_s * 1e6) if avg_s > 0 else None, "algorithmic_bytes_per_unit": bytes_per_unit, "algorithmic_bytes_per_unit_no_resampling": light if spec["alg"] == cp.ALG_SMC else None,
                "units_per_launch": n_prof, "avg_launch_us": avg_s * 1e6, "launches": int(dom_calls), "resampling_launches_per_run": n_res_prof if spec["alg"] == cp.ALG_SMC else None,
                "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]}}

    out = {
        "metric": "particles/sec (whole node) + achieved HBM GB/s, HMM T=16 SMC" if args.workload == "hmm16_smc" else "particles/sec (whole node), " + args.workload,
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 log-weights, 32-bit fixed-point masses" if (spec["alg"] == cp.ALG_SMC and step_form == 2) else "f64", "data": "synthetic",
        "config": {"workload": "%s, %d particles per GPU" % (spec["desc"].replace("systematic resampling", args.resampler.replace("_", " ") + " resampling"), n), "particles_per_gpu": n, "T": T,
                   "resampler": args.resampler, "ess_threshold": spec["ess"], "scope": scope, "n_global": n_global, "collective": collective,
                   "host": host if (world > 1 or group is not None) else "one context, one stream", "exchange_reruns": reruns, "flags": args.flags},
        "particle_steps_per_sec": value * T,
        "roofline": roofline,
        "posterior_max_abs_err_vs_exact": err, "log_evidence": summ["log_evidence"], "n_resampled": summ["n_resampled"],
    }

    if isinstance(reruns, dict) and "rank_step_breakdown_us" in reruns:
        bd = reruns.pop("rank_step_breakdown_us")
        if bd is not None:
            out["rank_step_breakdown_us"] = bd
            out["transport_note"] = bd.get("transport_note") if isinstance(bd, dict) else None
    if preflight is not None:
        out["preflight"] = preflight
    if native_error:
        out["config"]["native_driver_error"] = native_error
    if xtraffic is not None:
        # bytes of ONE run (the last one collected): what crossed the links for the migrating lineages, and the small collectives
        out["exchange_traffic_per_run"] = {"records": xtraffic["records"], "payload_bytes": xtraffic["payload_bytes"], "wire_bytes": xtraffic["wire_bytes"],
                                           "collective_bytes": xtraffic["collective_bytes"],
                                           "transport": {0: "none", 1: "direct", 2: "sendrecv"}[xtraffic["transport"]],
                                           "remote_lineages": bool(xtraffic["remote_lineages"]), "mailbox_collectives": bool(xtraffic["mailbox_collectives"])}

    def emit():
        if rank == 0:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if world > 1 and exchange and not args.no_extras:
        # secondary, labelled: the migration-free scopes on the same shards (every rank takes part).  "island": independent
        # populations combined once per run by their evidence -- no data-path collective; "global": one joint population, per-step
        # all-gather of 3 doubles per rank, resampling local to the shard (mass share carried) -- a different estimator from the
        # headline's exact global resampling ("offer it as an option, not the default").
        # They go through torch.distributed's collectives; the headline above is already measured, so nothing here may cost it:
        # an error is reported under the key, and a collective that never completes ends the job AFTER the line has been written.
        import threading
        sec = {}
        def secondary():
            torch.cuda.set_device(local)            # (per-thread setting)
            k2, w2 = max(2, min(args.steps, 20)), max(1, min(args.warmup, 3))
            for name, isl in (("island", True), ("global", False)):
                try:
                    eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"],
                              particle_offset=rank * n, n_global=n_global, scope=cp.SCOPE_ISLAND if isl else cp.SCOPE_GLOBAL)
                    sdt, slast = timed_runs(eng, k2, w2, world, device, isl, first_index=70_000)
                    sec[name] = {"particles_per_sec": n_global * k2 / sdt, "ms_per_run": sdt / k2 * 1e3, "runs": k2,
                                 "posterior_max_abs_err_vs_exact": float(np.abs(slast[0] - spec["exact"]).max()) if slast is not None else None,
                                 "collective": "all_gather(4+T*K doubles/rank) once per run" if isl else "all_gather(3 doubles/rank) per step, no migration"}
                except Exception as e:      # noqa: reported under the key
                    sec[name] = {"error": str(e)}
                    return
        th = threading.Thread(target=secondary, daemon=True)
        th.start()
        th.join(float(os.environ.get("CPPROB_BENCH_SECONDARY_TIMEOUT", "240")))
        out["secondary_scopes"] = dict(sec)
        if th.is_alive() or any("error" in v for v in sec.values()):
            # (a rank that failed or stalled leaves its peers inside a collective: no orderly shutdown is possible)
            out["secondary_scopes"]["incomplete"] = "timed out" if th.is_alive() else "error"
            emit()
            os._exit(0)

    if rank == 0 and world == 1 and not args.no_extras and smc and group is None:
        # secondary: the multi-GPU protocol at world = 1 -- the library's C++ driver with a single rank: per step the sharded step
        # kernel and the shard totals (a group of one has nobody to gather from or exchange with); then the same with EVERY call of
        # the multi-GPU path issued anyway (ncclAllGather per step, the ordering all-gather of the direct transport, ncclAllReduce)
        for key, flags in (("exchange_world1", 0), ("exchange_world1_all_collectives", cp.capi.GROUP_WORLD1_COLLECTIVES | cp.capi.GROUP_LIBRARY_COLLECTIVES),
                           ("exchange_world1_mailbox_collectives", cp.capi.GROUP_WORLD1_COLLECTIVES)):
            try:
                g1 = cp.Group([local])
                g1.transport(flags=flags)
                g1.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"])
                gdt, gstats, _, grr = timed_group_runs(g1, args.steps, args.warmup, 1, device, first_index=60_000)
                out[key] = {"ms_per_run": gdt / args.steps * 1e3, "particles_per_sec": n * args.steps / gdt, "reruns": grr,
                            "posterior_max_abs_err_vs_exact": float(np.abs(gstats - spec["exact"]).max()),
                            "note": "cpprob_hip_group_run, world = 1" + ("" if not flags else (", every RCCL collective of the multi-GPU path issued on the context's stream" if flags & cp.capi.GROUP_LIBRARY_COLLECTIVES
                                                                                       else ", the per-step collectives as stores into the rank's own mailbox (what ranks that can map each other's memory run by default)"))}
                g1.close()
            except Exception as e:
                out[key] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "hmm16_smc":
        # secondary: the SAME workload through the unchanged-model path -- models::hmm<16> as written against the CPProb statement API,
        # compiled for the device (CPPROB_REGISTER_MODEL) and called through cpprob::inference from the C++14 host (cpprob_main --generic):
        # trace replay, one launch of the model body per observe, device-side bookkeeping in between
        try:
            exe = os.path.join(ROOT, "cpprob_amd", "bin", "cpprob_main")
            obs_s = "[" + " ".join(repr(float(x)) for x in spec["obs"]) + "]"
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                cmd = [exe, "--model_folder", td, "--model", "hmm16", "--smc", "--observes", obs_s, "--n_samples", str(n), "--seed", str(args.seed),
                       "--ess_threshold", "2.0", "--generic", "--no_dump", "--json", "--repeat", "12"]
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            gj = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")][-1]
            gst = np.array([p_["p"] for p_ in gj["predicts"]])
            runs = [l for l in pr.stdout.splitlines() if l.startswith("run ")]
            first = runs[0].split()
            warm = sorted(float(l.split()[2]) for l in runs[1:])          # (every call but the first: each a run of its own seed)
            g_ms = warm[len(warm) // 2]
            out["generic_path"] = {"ms_per_run": g_ms, "ms_per_run_min": warm[0], "ms_per_run_max": warm[-1], "particles_per_sec": n / (g_ms * 1e-3), "replay_window": gj["replay_window"],
                                   "step_form": {0: "model launch + three bookkeeping launches", 1: "resampling inside the model's launch (dry-run bounds)", 2: "resampling inside the model's launch + exact-maximum passes",
                                                 3: "resampling inside the model's launch, four particles a lane"}[gj["step_form"]],
                                   "launches_per_step": gj["launches_per_step"],
                                   "ms_first_call": float(first[2]) + float(first[5]), "ms_setup_first_call": float(first[5]), "ms_setup_warm_call": gj["setup_seconds"] * 1e3,
                                   "posterior_max_abs_err_vs_exact": float(np.abs(gst - spec["exact"]).max()),
                                   "vs_fused_kernels": g_ms / (dt / args.steps * 1e3),
                                   "note": "cpprob_main --generic --repeat 12: ms_per_run = the median over the eleven warm calls of a call's device work (the read-out of every predict hit included), ms_first_call = "
                                           "the first cpprob::inference call whole (context, workspace, Markov pilot, run), ms_setup_warm_call = what a later call adds to its run"}
            # like for like: the SAME call and clock (cpprob::inference's Result::run_seconds through cpprob_main) on the built-in registration,
            # and the four-particles-a-lane step form (--step_form 3), measured and not the engine's choice
            def main_ms(extra):
                with tempfile.TemporaryDirectory() as td2:
                    cmd2 = [exe, "--model_folder", td2, "--model", "hmm16", "--smc", "--observes", obs_s, "--n_samples", str(n), "--seed", str(args.seed),
                            "--ess_threshold", "2.0", "--no_dump", "--json", "--repeat", "12"] + extra
                    pr2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=300)
                w2 = sorted(float(l.split()[2]) for l in pr2.stdout.splitlines() if l.startswith("run "))[:-1]
                return w2[len(w2) // 2]
            b_ms = main_ms([])
            out["generic_path"]["builtin_registration_same_call_ms"] = b_ms
            out["generic_path"]["vs_builtin_registration_same_call"] = g_ms / b_ms
            out["generic_path"]["four_particles_a_lane_ms"] = main_ms(["--generic", "--step_form", "3"])
            # the step kernels built per step (model_step_kernel_at) against the run-time kernel alone: the same call, the builds switched off
            out["generic_path"]["step_builds_used"] = gj.get("step_builds_used")
            out["generic_path"]["run_time_kernel_only_ms"] = main_ms(["--generic", "--no_step_builds"])
            # the same population as FOUR ranks of one joint population on this GPU (loopback: the pull migration and the per-step host all-gather at work)
            with tempfile.TemporaryDirectory() as td:
                cmd = [exe, "--model_folder", td, "--model", "hmm16", "--smc", "--observes", obs_s, "--n_samples", str(n), "--seed", str(args.seed),
                       "--ess_threshold", "2.0", "--generic", "--no_dump", "--json", "--repeat", "12", "--devices", "0,0,0,0"]
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            jj = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")][-1]
            jwarm = sorted(float(l.split()[2]) for l in pr.stdout.splitlines() if l.startswith("run "))[:-1]       # (the first call is the slowest)
            out["generic_path"]["joint_4_loopback_ranks"] = {"ms_per_run": jwarm[len(jwarm) // 2], "joint": jj["joint"], "log_evidence_equals_one_rank": abs(jj["log_evidence"] - gj["log_evidence"]) <= 1e-13 * abs(gj["log_evidence"]),
                                                             "note": "cpprob_main --generic --devices 0,0,0,0: one joint population, four ranks on this one GPU"}
        except Exception as e:
            out["generic_path"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "hmm16_smc" and args.resampler == "systematic":
        # secondary: the other two resamplers (thesis Alg. 1 p.36 is multinomial; remark p.36: systematic / stratified) on the headline
        # workload and on configs[3]'s per-GPU shape.  All three run on integers inside the step launch and equal the oracle's
        # ancestors bit for bit (tests/test_gpu_inference.py); multinomial in the strata form unless the literal form is named.
        zz = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))
        res = {}
        for wl, model, obs, nn, ess_w, exact in (("hmm16_smc@%d" % n, cp.MODEL_HMM3, zz["hmm16"], n, 2.0, zz["hmm16_smooth"]),
                                                 ("lgssm100_smc@1250000", cp.MODEL_LINEAR_GAUSSIAN_1D, zz["lgssm100"], 1_250_000, 0.5,
                                                  np.stack([zz["lgssm100_smooth_mean"], zz["lgssm100_smooth_var"]], 1))):
            row = {}
            for name in ("systematic", "stratified", "multinomial", "multinomial_literal"):
                try:
                    rid, rfl = resampler_of(name)
                    eng.begin(cp.ALG_SMC, model, obs, nn, seed=args.seed, resampler=rid, ess_threshold=ess_w, flags=rfl)
                    k2 = max(3, min(args.steps, 20))
                    rdt, _ = timed_runs(eng, k2, 2, 1, device, False, first_index=80_000)
                    eng.profile_enable(True); eng.profile_read(reset=True)
                    for i in range(k2):
                        eng.run(81_000 + i)
                    rp = eng.profile_read(reset=True)
                    eng.profile_enable(False)
                    sm = eng.summary()
                    # (the step launches' HIP-event time over the profiled runs / the launches those runs made: k2 runs x T steps)
                    row[name] = {"ms_per_run": rdt / k2 * 1e3, "step_us": rp["smc_step"][0] * 1e3 / (k2 * len(obs)),
                                 "resample_only_us_per_run": rp["resample"][0] * 1e3 / k2, "step_form": FORM_NAMES.get(int(sm.get("step_form", 0))),
                                 "n_resampled": sm["n_resampled"], "posterior_max_abs_err_vs_exact": float(np.abs(eng.stats() - exact).max())}
                except Exception as e:      # noqa: reported under the key
                    row[name] = {"error": str(e)}
            for name in ("stratified", "multinomial", "multinomial_literal"):
                if "ms_per_run" in row.get(name, {}) and "ms_per_run" in row.get("systematic", {}):
                    row[name]["vs_systematic"] = row[name]["ms_per_run"] / row["systematic"]["ms_per_run"]
            res[wl] = row
        res["note"] = ("ms per run of a whole inference (read-out included), same seed and sizes; step_us = HIP-event mean of the step launches; resample_only_us_per_run = the launches a "
                       "resampler adds in front of its steps (strata form of multinomial: every step's stratum counts in two launches at the run's start)")
        out["resamplers"] = res

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "hmm16_smc" and args.resampler == "systematic" and group is None:
        # secondary: the three resamplers as ONE joint population over eight loopback shards of this GPU (the whole exchange protocol on one
        # stream; exact: every run equals the one-GPU run's traces bit for bit, tests/test_gpu_group.py) -- multinomial (thesis Alg. 1) in
        # the strata form: regular intervals + the strata the ranks' boundaries cut (csrc/strata_cut.hpp)
        sh = {}
        for name in ("systematic", "stratified", "multinomial"):
            try:
                rid, rfl = resampler_of(name)
                g8 = cp.Group([local] * 8)
                g8.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=rid, ess_threshold=spec["ess"], flags=rfl)
                k2 = max(3, min(args.steps, 10))
                gdt, gstats, gsum, grr = timed_group_runs(g8, k2, 2, 1, device, first_index=90_000)
                tr8 = g8.traffic()
                sh[name] = {"ms_per_run": gdt / k2 * 1e3, "reruns": grr if not isinstance(grr, dict) else grr.get("reruns"), "records_per_run": tr8["records"],
                            "posterior_max_abs_err_vs_exact": float(np.abs(gstats - spec["exact"]).max())}
                g8.close()
            except Exception as e:      # noqa: reported under the key
                sh[name] = {"error": str(e)}
        sh["note"] = "cpprob_hip_group_run, 8 loopback ranks sharing this GPU and one stream (the ranks' launches serialise: a protocol figure, not a scaling one); %d particles in all" % n
        out["sharded_resamplers"] = sh

    if rank == 0 and world == 1 and not args.no_extras:
        # error bars: five run seeds of the headline configuration against the exact posterior
        eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"])
        errs, lzs = [], []
        for k in range(5):
            eng.run(50_000 + k)
            errs.append(float(np.abs(eng.stats() - spec["exact"]).max()))
            lzs.append(eng.summary()["log_evidence"])
        out["five_seeds"] = {"posterior_max_abs_err_vs_exact": {"mean": float(np.mean(errs)), "sd": float(np.std(errs, ddof=1)), "max": float(np.max(errs))},
                             "log_evidence": {"mean": float(np.mean(lzs)), "sd": float(np.std(lzs, ddof=1))},
                             "exact_log_evidence": spec.get("exact_logz")}

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "hmm16_smc":
        # secondary: the same particles without their history (keep_history = 0: two rows of values, no ancestors, no lineage
        # walk); predict hit t's statistics are then the FILTERING ones -- a different estimand from the headline's whole-trace
        # posterior, which is why this is not `value`
        zf = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))
        eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"], keep_history=False)
        fdt, _ = timed_runs(eng, args.steps, args.warmup, 1, device, False)
        out["filtering_only"] = {"ms_per_run": fdt / args.steps * 1e3, "particles_per_sec": n * args.steps / fdt,
                                 "filter_max_abs_err_vs_exact": float(np.abs(eng.stats() - zf["hmm16_filter"]).max()),
                                 "note": "keep_history = 0: particle store O(N) instead of O(N T); 16 step launches + one bookkeeping wavefront"}

    if rank == 0 and world == 1 and not args.no_extras and spec["alg"] == cp.ALG_SMC and not exchange:
        # secondary: the same runs with several contexts in flight.  One run is a dependent chain of ~19 launches whose
        # step kernels leave CUs idle at their tails; independent runs (replicates, other observation sets) on separate
        # contexts / HIP streams fill them.  Every run is unchanged -- same kernels, same 10^6 particles -- only overlapped.
        # `value` above stays the one-run-at-a-time figure that the roofline / rocprof numbers describe.
        R = args.in_flight
        engs = [eng] + [cp.Engine(local) for _ in range(R - 1)]
        eng.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"])
        for e in engs[1:]:
            e.begin(spec["alg"], spec["model"], spec["obs"], n, seed=args.seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=spec["ess"])
        total = args.steps * R
        for i in range(args.warmup * R):
            engs[i % R].run(30_000 + i)
        for e in engs:
            e.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(total):
            engs[i % R].run(40_000 + i)
        for e in engs:
            e.sync()
        pdt = time.perf_counter() - t0
        perr = max(float(np.abs(e.stats() - spec["exact"]).max()) for e in engs)
        out["pipelined"] = {"runs_in_flight": R, "runs": total, "particles_per_sec": n * total / pdt, "ms_per_run": pdt / total * 1e3,
                            "posterior_max_abs_err_vs_exact": perr,
                            "note": "independent runs on separate contexts/streams of one GPU; each run identical to the headline's"}
        for e in engs[1:]:
            e.close()

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "hmm16_smc":
        # secondary: BASELINE.json configs[1], gaussian_unknown_mean SIS at 10^7 particles
        g = workload_spec("gaussian_sis", os.path.join(ROOT, "tests", "golden", "observations.npz"))
        ng = 10_000_000
        eng.begin(g["alg"], g["model"], g["obs"], ng, seed=args.seed)
        gdt, _ = timed_runs(eng, args.steps, args.warmup, 1, device, False)
        gst = eng.stats()
        eng.profile_enable(True); eng.profile_read(reset=True)
        for i in range(args.steps):
            eng.run(20_000 + i)
        gp = eng.profile_read(reset=True)
        eng.profile_enable(False)
        sis_s = gp["sis"][0] * 1e-3 / max(gp["sis"][1], 1)
        sm_s = gp["smooth"][0] * 1e-3 / max(gp["smooth"][1], 1)      # 0 calls when the read-out rides the normalisation (fused SIS read-out)
        norm_s = gp["scan_partials"][0] * 1e-3 / max(gp["scan_partials"][1], 1)
        out["gaussian_sis_1e7"] = {"particles_per_sec": ng * args.steps / gdt, "ms_per_run": gdt / args.steps * 1e3,
                                   "posterior_mean_var": gst[0].tolist(), "analytic_mean_var": g["exact"][0].tolist(),
                                   "sis_kernel_us": sis_s * 1e6, "sis_kernel_GBs": 16 * ng / sis_s / 1e9 if sis_s else None,
                                   "readout_kernel_us": sm_s * 1e6, "readout_kernel_GBs": 16 * ng / sm_s / 1e9 if sm_s else None,
                                   "normalise_and_readout_us": norm_s * 1e6,
                                   "end_to_end_GBs_at_32B": 32 * ng * args.steps / gdt / 1e9}
        if not args.no_cpu_baseline:
            out["gaussian_sis_1e7"]["cpu_compute_only_1core"] = cpu_baseline("gaussian_sis", 4_000_000, args.seed)
            out["gaussian_sis_1e7"]["cpu_as_shipped_1core"] = cpu_as_shipped(200_000, args.seed)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample, args.seed)
        if "value" in out["cpu_baseline"]:
            out["gpu_over_cpu_1core"] = value / out["cpu_baseline"]["value"]
        if spec["model"] == cp.MODEL_HMM3:
            # mode A of BASELINE.md (as shipped: one core, three file appends per particle) for this workload's model run as SIS --
            # the only mode the reference has
            try:
                out["cpu_baseline"]["as_shipped_sis_1core"] = cpu_as_shipped(100_000, args.seed, model=cp.MODEL_HMM3, obs=spec["obs"], address="State",
                                                                             label="hmm<%d>" % T)
            except Exception as e:
                out["cpu_baseline"]["as_shipped_sis_1core"] = {"error": str(e)}

    # the secondary numbers as scalars where the driver's record keeps them whole (`roofline`): every number README quotes
    def _get(*ks):
        d = out
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    wl0 = "hmm16_smc@%d" % n
    out["roofline"].update({
        "generic_ms": _get("generic_path", "ms_per_run"), "generic_vs_fused": _get("generic_path", "vs_fused_kernels"),
        "generic_vs_builtin_same_call": _get("generic_path", "vs_builtin_registration_same_call"), "generic_step_builds_used": _get("generic_path", "step_builds_used"),
        "generic_no_step_builds_ms": _get("generic_path", "run_time_kernel_only_ms"),
        "stratified_vs_systematic": _get("resamplers", wl0, "stratified", "vs_systematic"), "multinomial_vs_systematic": _get("resamplers", wl0, "multinomial", "vs_systematic"),
        "lgssm100_1250000_systematic_ms": _get("resamplers", "lgssm100_smc@1250000", "systematic", "ms_per_run"),
        "lgssm100_1250000_multinomial_vs_systematic": _get("resamplers", "lgssm100_smc@1250000", "multinomial", "vs_systematic"),
        "gaussian_sis_particles_per_s": _get("gaussian_sis_1e7", "particles_per_sec"),
        "gaussian_sis_hbm_frac": (_get("gaussian_sis_1e7", "sis_kernel_GBs") / HBM_PEAK_GBS) if _get("gaussian_sis_1e7", "sis_kernel_GBs") else None,
        "pipelined_particles_per_s": _get("pipelined", "particles_per_sec"), "filtering_only_particles_per_s": _get("filtering_only", "particles_per_sec"),
        "multinomial_8_loopback_shards_ms": _get("sharded_resamplers", "multinomial", "ms_per_run"), "systematic_8_loopback_shards_ms": _get("sharded_resamplers", "systematic", "ms_per_run"),
    })
    emit()
    if group is not None:
        group.close()
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
