"""Multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process, single-threaded (src/cpprob/state.cpp:20-21); sharding is new.
Particles are independent between resampling points, so the population splits into
contiguous shards, rank r holding global ids [r*n, (r+1)*n); RNG counters use global ids.

Two scopes (include/cpprob_hip.h, cpprob_hip_config.resample_scope):

  ISLAND  every rank runs SMC on its own population of n particles (local ESS test, local
          resampling): no data-path collective at all.  The shards are combined ONCE at the
          end by importance-weighting each island with its evidence estimate Z_r:
              E[f] = sum_r Z_r E_r[f] / sum_r Z_r,     Z = mean_r Z_r
          (unbiased Z; consistent posterior; for iid shards of >= 1e5 particles the Z_r agree
          to ~1e-3 relative, so the efficiency loss against one joint population is negligible).
          Communication: one all-gather of (1 + T*K) doubles per rank per run.

  GLOBAL  one joint population of n_global particles: per step every rank all-gathers its
          (max, sum, sum-of-squares) of weights (3 doubles per rank -- latency-bound, RCCL over xGMI)
          so all ranks agree ON DEVICE on the joint normaliser, evidence, ESS and the resampling
          decision (cpprob_hip_smc_step_begin / _end).  Resampling is then local to each shard
          and the shard's particles carry its share of the mass (distributed resampling with
          non-proportional allocation): particles never migrate, nothing but the 3 doubles crosses
          xGMI, and there is no host synchronisation inside a run.  With one rank this is exactly
          the single-GPU algorithm.  At the end the un-normalised weighted sums are all-reduced.
          When the resampling schedule is static (every step) the other shards' totals are never
          needed inside a run -- the carried mass share of shard r is its own evidence estimate --
          so the joint algorithm is executed as ISLAND + one final combine (same estimator,
          tests/test_gpu_inference.py::..._static_schedule_equals_evidence_weighted_islands).

  EXCHANGE  GLOBAL plus particle migration: the sharded run draws exactly the ancestors one GPU holding all
          n_global particles would (systematic resampling).  The all-gathered rank totals give
          every rank the offspring interval [o_r, o_{r+1}) of every rank's sources, hence a deterministic
          plan of who sends how many lineages to whom; the lineages (trace x_0..x_t of each remote ancestor)
          travel in ONE all-to-all-v per resampling step over xGMI (point-to-point links: every pair of
          GPUs talks directly) and become annex columns of the receiving shard.  Expected volume is
          O(sqrt(n)) records per rank per step for well-mixed shards; worst case n.  One host
          synchronisation per step (the split sizes of the all-to-all must be known on the host).

Host logic in this file is pure numpy/torch and is covered by gloo world_size-2 tests on CPU.
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(device_is_gpu=True):
    """Initialises torch.distributed from the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # CPPROB_DIST_BACKEND=gloo + CPPROB_FORCE_DEVICE=k: test hook that lets several ranks share one GPU (RCCL refuses
        # duplicate devices); small collectives then travel through host memory
        backend = os.environ.get("CPPROB_DIST_BACKEND", "nccl" if device_is_gpu else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if "CPPROB_FORCE_DEVICE" in os.environ:
        local = int(os.environ["CPPROB_FORCE_DEVICE"])
    return world, rank, local


def _host_collectives():
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "gloo"


def shard_bounds(n_global, world, rank):
    """Contiguous shard of rank: particle i lives on rank floor(i * world / n_global)."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def logsumexp(a):
    a = np.asarray(a, np.float64)
    m = np.max(a)
    if not np.isfinite(m):
        return m
    return m + np.log(np.sum(np.exp(a - m)))


def combine_islands(log_z, stats, is_int):
    """Evidence-weighted combination of per-island posterior summaries.

    log_z : [R]        per-island log evidence estimates
    stats : [R, T, K]  per-island summaries: int -> probabilities; real -> (mean, variance)
    Returns (stats[T, K], log_evidence, island_weights[R], island_ess).
    """
    log_z = np.asarray(log_z, np.float64)
    stats = np.asarray(stats, np.float64)
    lse = logsumexp(log_z)
    w = np.exp(log_z - lse)                       # normalised island weights
    if is_int:
        out = np.tensordot(w, stats, axes=(0, 0))
    else:
        mean_r = stats[:, :, 0]
        raw2_r = stats[:, :, 1] + mean_r * mean_r   # variance(mean) = raw2 - mean^2 (empirical_distribution.hpp:78-81)
        mean = w @ mean_r
        raw2 = w @ raw2_r
        out = np.stack([mean, raw2 - mean * mean], axis=1)
    return out, float(lse - np.log(len(log_z))), w, float(1.0 / np.sum(w * w))


def allgather_vector(vec, device=None):
    """all-gather of a small float64 vector: returns [world, len(vec)] as numpy."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    v = np.ascontiguousarray(vec, np.float64)
    if world == 1:
        return v[None, :]
    t = torch.from_numpy(v)
    if device is not None and not _host_collectives():
        t = t.to(device)
    out = torch.empty((world, v.size), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t) if t.device.type == "cuda" else dist.all_gather(list(out.unbind(0)), t)
    return out.cpu().numpy()


def run_islands(engine, run_index=0, device=None):
    """One island-scope run on every rank + the final evidence-weighted combination.
    `engine` must have been begun with scope=SCOPE_ISLAND.  Returns (stats, log_evidence, island_ess)."""
    engine.run(run_index)
    s = engine.summary()
    st = engine.stats()
    vec = np.concatenate([[s["log_evidence"]], st.reshape(-1)])
    allv = allgather_vector(vec, device)
    out, lz, _, iess = combine_islands(allv[:, 0], allv[:, 1:].reshape(allv.shape[0], *st.shape), engine.is_int)
    return out, lz, iess


class IslandBatch:
    """Island-scope runs back to back with NO host synchronisation between them: every run leaves
    {log_evidence, ess, log_norm, max_logw, stats} in a device slot (cpprob_hip_infer_results_device) and the slot is
    all-gathered over the ranks by RCCL on the engine's stream; the host looks at a slot only when asked (results()).
    This is what keeps N GPUs at N times one GPU's rate: the only inter-GPU traffic is 8 * (4 + T*K) bytes per rank per
    run, and nobody waits for it."""

    def __init__(self, engine, depth):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.engine = torch, dist, engine
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        dev = torch.device("cuda", engine.device)
        self.width = 4 + engine.T * engine.K
        self.depth = int(depth)
        self.packed = torch.zeros((self.depth, self.width), dtype=torch.float64, device=dev)
        self.gathered = torch.zeros((self.depth, self.world, self.width), dtype=torch.float64, device=dev)
        self.stream = torch.cuda.ExternalStream(engine.stream_ptr, device=dev)
        torch.cuda.synchronize()                  # the fills above ran on torch's stream; the engine's stream is not ordered after it
        self.works = [None] * self.depth          # pending all-gathers (the engine's stream never waits for them)
        if self.world > 1 and not _host_collectives():
            # first use of a collective builds RCCL's communicator and channels (milliseconds): do it here, outside anybody's timing
            with torch.cuda.stream(self.stream):
                dist.all_gather_into_tensor(self.gathered[0].view(-1), self.packed[0])
            torch.cuda.synchronize()

    def run(self, slot, run_index):
        e, torch = self.engine, self.torch
        if self.works[slot] is not None:          # the slot's previous all-gather must have read packed[slot] before this run rewrites it:
            with torch.cuda.stream(self.stream):  # Work.wait() orders the CURRENT stream after the collective -- make that the engine's
                self.works[slot].wait()
            self.works[slot] = None
        e.run(run_index)
        e.results_device(self.packed[slot])
        with torch.cuda.stream(self.stream):
            if self.world == 1:
                self.gathered[slot, 0].copy_(self.packed[slot])
            elif _host_collectives():          # test hook (gloo): through host memory, synchronising
                mine = self.packed[slot].cpu()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(parts, mine)
                self.gathered[slot].copy_(torch.stack(parts).to(self.gathered.device))
            else:
                # async_op: RCCL's stream waits for the engine's (it sees the packed slot), but the engine's stream does NOT
                # wait for the collective -- the next run starts while the 8 * width bytes per rank are still travelling
                self.works[slot] = self.dist.all_gather_into_tensor(self.gathered[slot].view(-1), self.packed[slot], async_op=True)

    def results(self, slot):
        """Synchronises and combines the islands of one slot: (stats[T, K], log_evidence, island_ess)."""
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None
        self.engine.sync()
        self.torch.cuda.synchronize()
        g = self.gathered[slot].cpu().numpy()
        e = self.engine
        out, lz, _, iess = combine_islands(g[:, 0], g[:, 4:].reshape(self.world, e.T, e.K), e.is_int)
        return out, lz, iess


# ---- joint population (GLOBAL scope) ------------------------------------------------------------

class TorchCollective:
    """all-gather / all-reduce of small float64 device tensors on the ENGINE's stream (no host sync):
    torch.distributed orders the RCCL kernel after the work already queued on the current stream."""

    def __init__(self, engine):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.torch = torch
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.stream = torch.cuda.ExternalStream(engine.stream_ptr, device=torch.device("cuda", engine.device))
        if self.world > 1 and not _host_collectives():
            # build RCCL's communicator and channels now (first use costs milliseconds), outside anybody's timing
            dev = torch.device("cuda", engine.device)
            probe = torch.zeros(3, dtype=torch.float64, device=dev)
            out = torch.zeros(3 * self.world, dtype=torch.float64, device=dev)
            with torch.cuda.stream(self.stream):
                dist.all_gather_into_tensor(out, probe)
                dist.all_reduce(probe, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()

    def all_gather(self, local, out):
        if self.world == 1:
            with self.torch.cuda.stream(self.stream):
                out.copy_(local[: out.numel()])
            return
        with self.torch.cuda.stream(self.stream):
            if _host_collectives():      # test hook (gloo): through host memory
                mine = local[: out.numel() // self.world].cpu()
                parts = [self.torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(parts, mine)
                out.copy_(self.torch.cat(parts).to(out.device))
            else:
                self.dist.all_gather_into_tensor(out, local[: out.numel() // self.world].contiguous())

    def all_reduce_sum(self, t):
        if self.world == 1:
            return
        with self.torch.cuda.stream(self.stream):
            if _host_collectives():
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
                t.copy_(h.to(t.device))
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)


    def all_to_all_records(self, send, recv, send_counts, recv_counts, width):
        """all-to-all-v of lineage records: rank r gets send_counts[r] records of `width` elements from `send` (grouped by
        destination, rank order); `recv` is filled grouped by source."""
        ins = [int(c) * width for c in send_counts]
        outs = [int(c) * width for c in recv_counts]
        with self.torch.cuda.stream(self.stream):
            if _host_collectives():
                h_send = send[: sum(ins)].cpu()
                h_recv = host_all_to_all(h_send, ins, outs)
                recv[: sum(outs)].copy_(h_recv.to(recv.device))
            else:
                self.dist.all_to_all_single(recv[: sum(outs)], send[: sum(ins)], output_split_sizes=outs, input_split_sizes=ins)


def host_all_to_all(h_send, in_splits, out_splits):
    """all-to-all-v of a 1-D CPU tensor over the default (gloo) group, as point-to-point pairs."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    h_recv = torch.empty(sum(out_splits), dtype=h_send.dtype)
    send_parts = list(torch.split(h_send, in_splits))
    recv_parts = list(torch.split(h_recv, out_splits))
    recv_parts[rank].copy_(send_parts[rank])
    reqs = []
    for r in range(world):
        if r == rank:
            continue
        if in_splits[r]:
            reqs.append(dist.isend(send_parts[r].contiguous(), dst=r))
        if out_splits[r]:
            reqs.append(dist.irecv(recv_parts[r], src=r))
    for q in reqs:
        q.wait()
    return h_recv


def normalise_joint_stats(raw, log_norm, max_logw, is_int):
    """raw [T, K]: all-reduced un-normalised sums relative to exp(max_logw).  Returns StatsPrinter's numbers."""
    raw = np.asarray(raw, np.float64)
    W = np.exp(log_norm - max_logw)
    if is_int:
        return raw / W
    mean = raw[:, 0] / W
    return np.stack([mean, raw[:, 1] / W - mean * mean], axis=1)      # raw_moment(2) - mean^2, empirical_distribution.hpp:78-81


def run_joint(engine, collective, run_index=0, buffers=None, slot=None):
    """run_joint_once; a single-rank population whose fixed-point weights lost too many bits is repeated in the floating-point form
    (several locally resampling ranks run in that form anyway).  With `slot` nothing is read back and nothing is checked."""
    from . import capi
    try:
        return run_joint_once(engine, collective, run_index, buffers, slot)
    except capi.CpprobHipError as e:
        if getattr(e, "code", 0) != capi.EPRECISION:
            raise
    engine.rebegin(capi.FLAG_FLOATING_POINT_STEP)
    return run_joint_once(engine, collective, run_index, buffers, slot)


def run_joint_once(engine, collective, run_index=0, buffers=None, slot=None):
    """One run of a joint population sharded over `collective.world` ranks (engine begun with
    scope=SCOPE_GLOBAL, n_global = sum of shards).  Returns (stats[T, K], summary dict)."""
    import torch
    from . import capi
    dev = torch.device("cuda", engine.device)
    world, rank = collective.world, collective.rank
    if buffers is None:
        buffers = (torch.zeros(4, dtype=torch.float64, device=dev), torch.zeros(3 * world, dtype=torch.float64, device=dev),
                   torch.zeros(engine.T * engine.K, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()                  # fills on torch's stream before the engine's stream touches them
    local, allt, _ = buffers
    steps = [engine.T - 1] if engine.cfg.algorithm == capi.ALG_SIS else range(engine.T)
    for t in steps:
        engine.step_begin(t, local, run_index)
        collective.all_gather(local, allt)
        engine.step_end(t, allt, world, rank)
    engine.finish()
    if slot is not None:
        # no host synchronisation: {log_evidence, ess, log_norm, max_logw, raw sums...} stay on the device; the sums are
        # all-reduced in place on the engine's stream; joint_results() reads the slot later
        engine.results_device(slot)
        collective.all_reduce_sum(slot[4:])
        return None
    s = engine.summary()
    raw = torch.from_numpy(engine.stats()).to(dev)
    torch.cuda.current_stream().synchronize()     # the copy ran on torch's stream, the collective runs on the engine's
    collective.all_reduce_sum(raw)
    engine.sync()                                 # the reduction ran on the engine's stream; .cpu() below runs on torch's
    stats = normalise_joint_stats(raw.cpu().numpy(), s["log_norm"], s["max_logw"], engine.is_int)
    return stats, s


def joint_results(engine, slot):
    """Host read-out of a slot left by run_joint(..., slot=...): (stats[T, K], summary dict)."""
    import torch
    engine.sync()
    torch.cuda.synchronize()
    h = slot.cpu().numpy()
    s = {"log_evidence": float(h[0]), "ess_final": float(h[1]), "log_norm": float(h[2]), "max_logw": float(h[3])}
    stats = normalise_joint_stats(h[4:].reshape(engine.T, engine.K), s["log_norm"], s["max_logw"], engine.is_int)
    return stats, s


def offspring_bounds_fixed(masses, n_global, u0):
    """Offspring bounds of the ranks in the fixed-point form, as the device derives them from the all-gathered totals
    (cpprob_amd/csrc/step_fixed.hpp: plan_bounds_fixed): rank r's sources own the outputs [o[r], o[r + 1]),
    o[r] = ceil(fma(double(mass before rank r), N / double(total mass), -u0)), o[0] = 0, o[world] = N.  masses: one integer per rank."""
    import math
    from fractions import Fraction
    masses = [int(m) for m in masses]
    total = sum(masses)
    n = float(int(n_global))
    inv = n / float(total)
    o, before = [], 0
    for m in masses:
        # fma(double(before), inv, -u0) with ONE rounding: the exact rational value, correctly rounded (float(Fraction) is)
        fused = float(Fraction(float(before)) * Fraction(inv) - Fraction(float(u0)))
        o.append(max(0.0, float(math.ceil(fused))))
        before += m
    o[0] = 0.0
    o.append(n)
    return np.array(o, np.float64)


def exchange_counts(o, begins, rank):
    """The exchange plan of one resampling step from the ranks' offspring bounds o[0..world] and shard begins[0..world] (what
    cpprob_hip_exchange_plan computes on the device, cpprob_amd/csrc/exchange.hpp: exchange_plan_*): rank `rank`'s sources own the
    outputs [o[rank], o[rank + 1]); those inside another rank p's shard [begins[p], begins[p + 1]) are the lineages it SENDS to p
    (in output order), and it RECEIVES from q the outputs of its own shard that q's sources own.  Returns (send_first[world] -- the
    first such output per destination --, send_counts[world], recv_counts[world]); the diagonal is zero."""
    world = len(begins) - 1
    o = [int(x) for x in o]
    b = [int(x) for x in begins]
    def overlap(src, dst):
        lo, hi = max(o[src], b[dst]), min(o[src + 1], b[dst + 1])
        return lo, max(hi - lo, 0)
    send_first, send_counts, recv_counts = np.zeros(world, np.int64), np.zeros(world, np.int64), np.zeros(world, np.int64)
    for p in range(world):
        if p == rank:
            continue
        send_first[p], send_counts[p] = overlap(rank, p)
        recv_counts[p] = overlap(p, rank)[1]
    return send_first, send_counts, recv_counts


class StrataCutPlan:
    """The exchange plan of one MULTINOMIAL resampling step (strata form) over shards, as every rank's device derives it from the
    all-gathered totals (cpprob_amd/csrc/exchange.hpp: exchange_cut_kernel, strata_kept_before).

    The thresholds are generated stratum by stratum: stratum w holds the outputs [offs[w], offs[w + 1]) and its thresholds lie in
    [B[w], B[w + 1]).  Rank r's sources hold the mass range [P[r], P[r + 1]).  A stratum that lies inside one rank's range sends all
    its outputs to that rank's sources ("regular": one interval of outputs per rank, as under systematic resampling); only the
    <= world - 1 strata that a rank boundary CUTS have to look at their outputs one by one:
        lo[b] = min{w : B[w] >= P[b]},  hi[b] = max{w : B[w] <= P[b]}  (table form: B[w] < P[b] -- its thresholds may round UP to B[w + 1]),
        A[b] = offs[lo[b]],  Z[b] = offs[hi[b]]:    rank r's regular outputs = [A[r], Z[r + 1]),  boundary b's cut stratum = [Z[b], A[b])
    and for a cut stratum a table over its outputs: src(s) = max{r : P[r] <= tau_s} and cum(s) = #{s' < s in the stratum : src(s') =
    the rank whose shard holds s'} ("home" outputs).  An output that does not descend from its own shard's sources takes the next free
    annex column of its shard IN OUTPUT ORDER:   col(s) = (s - begin[d]) - kept_before(d, s),   kept_before(d, s) = #{s' in [begin[d], s) :
    src(s') = d} -- the regular part by interval arithmetic, the <= 2 cut strata around rank d by two table look-ups.

    thresholds(s_lo, s_hi) -> the thresholds of the outputs [s_lo, s_hi) (the device draws them: Philox, draw kResampleDrawBase2 + step)."""

    def __init__(self, P, B, offs, begins, thresholds, table_form=False):
        self.P = list(P); self.B = B; self.offs = [int(o) for o in offs]; self.begins = [int(b) for b in begins]
        self.world = len(self.P) - 1
        world, K, N = self.world, len(B) - 1, self.begins[-1]
        assert self.offs[K] == N and len(self.begins) == world + 1
        self.A, self.Z = [0] * (world + 1), [0] * (world + 1)
        self.A[world] = self.Z[world] = N
        self.tab = {}
        for b in range(1, world):
            lo = next(w for w in range(K + 1) if B[w] >= self.P[b])
            below = [w for w in range(K + 1) if (B[w] < self.P[b] if table_form else B[w] <= self.P[b])]
            hi = below[-1] if below else lo
            self.A[b], self.Z[b] = self.offs[lo], self.offs[hi]
            if self.A[b] > self.Z[b]:
                st, en = self.Z[b], self.A[b]
                tau = thresholds(st, en)
                src = [max(r for r in range(world) if self.P[r] <= t) for t in tau]
                dst = [self.shard_of(s) for s in range(st, en)]
                cum, acc = [], 0
                for a, d in zip(src, dst):
                    cum.append(acc); acc += 1 if a == d else 0
                cum.append(acc)
                self.tab[b] = (src, cum)

    def shard_of(self, s):
        return max(r for r in range(self.world) if self.begins[r] <= s)

    def cut_of(self, d):
        """the boundaries whose cut strata may hold outputs of rank d's sources: d and d + 1, once if they share a stratum"""
        bs = [b for b in (d, d + 1) if b in self.tab]
        if len(bs) == 2 and (self.Z[bs[0]], self.A[bs[0]]) == (self.Z[bs[1]], self.A[bs[1]]):
            bs = bs[:1]
        return bs

    def kept_before(self, d, s):
        sb = self.begins[d]
        kept = max(0, min(s, self.Z[d + 1]) - max(sb, self.A[d]))
        for b in self.cut_of(d):
            st, en = self.Z[b], self.A[b]
            x0, x1 = min(max(sb, st), en), min(max(s, st), en)
            if x1 > x0:
                kept += self.tab[b][1][x1 - st] - self.tab[b][1][x0 - st]
        return kept

    def column(self, d, s):
        """annex column (relative to the step's first) of output s on the rank d whose shard holds it; s must not descend from d"""
        return (s - self.begins[d]) - self.kept_before(d, s)

    def arrivals(self, d):
        return (self.begins[d + 1] - self.begins[d]) - self.kept_before(d, self.begins[d + 1])

    def sends(self, r):
        """[(output, destination rank)] of the outputs rank r's sources own in other ranks' shards: the regular interval cut by
        the shards, then the cut strata's table entries -- what the packing launch of rank r walks"""
        out = []
        for d in range(self.world):
            if d == r:
                continue
            lo, hi = max(self.A[r], self.begins[d]), min(self.Z[r + 1], self.begins[d + 1])
            out += [(s, d) for s in range(lo, hi)]
        for b in self.cut_of(r):
            st = self.Z[b]
            for i, a in enumerate(self.tab[b][0]):
                d = self.shard_of(st + i)
                if a == r and d != r:
                    out.append((st + i, d))
        return out


def shard_begins(n_global, world):
    return np.array([shard_bounds(n_global, world, r)[0] for r in range(world)] + [int(n_global)], np.uint64)


def run_exchange(engine, collective, run_index=0, counters=None):
    """One run of a joint population with EXACT global resampling (engine begun with scope=SCOPE_EXCHANGE on the shard
    shard_bounds(n_global, world, rank)).  Returns (stats[T, K], summary).  counters (dict) receives the number of lineage records
    this rank sent / received.  A generation whose fixed-point weights lost their bits is repaired in the run, in integers, as one
    GPU repairs it (cpprob_hip_smc_repair_begin / _end: every rank holds the same first offending generation -- the books come from the
    all-gathered totals -- so every rank enters the same collectives); CPPROB_HIP_FLAG_REPEAT_IN_FLOATING_POINT keeps the older
    behaviour, the whole run again in the floating-point form."""
    from . import capi
    try:
        return run_exchange_once(engine, collective, run_index, counters)
    except capi.CpprobHipError as e:
        if getattr(e, "code", 0) != capi.EPRECISION:
            raise
    engine.rebegin(capi.FLAG_FLOATING_POINT_STEP)
    return run_exchange_once(engine, collective, run_index, counters)


def run_exchange_once(engine, collective, run_index=0, counters=None):
    import torch
    from . import capi
    dev = torch.device("cuda", engine.device)
    world, rank = collective.world, collective.rank
    n_global = int(engine.cfg.n_global)
    begins = shard_begins(n_global, world)
    local = torch.zeros(4, dtype=torch.float64, device=dev)
    allt = torch.zeros(3 * world, dtype=torch.float64, device=dev)
    vdtype = torch.int32 if engine.is_int else torch.float64
    torch.cuda.synchronize()                      # fills on torch's stream before the engine's stream touches them
    bufs = {"send": None, "recv": None}
    moved = {"sent": 0, "recv": 0}

    def steps(t_from, resumed):
        """steps t_from .. T - 1; resumed: generation t_from exists already (its totals are in `local`: a repaired generation)"""
        for t in range(t_from, engine.T):
            if not (resumed and t == t_from):
                engine.step_begin(t, local, run_index)
            collective.all_gather(local, allt)
            engine.step_end(t, allt, world, rank)
            if t + 1 == engine.T:
                break
            _, sc, rc = engine.exchange_plan(t, world, rank, begins)
            width = t + 1
            ns, nr = int(sc.sum()), int(rc.sum())
            if bufs["send"] is None or bufs["send"].numel() < max(ns, 1) * engine.T:
                bufs["send"] = torch.empty(max(ns, 1) * engine.T, dtype=vdtype, device=dev)
            if bufs["recv"] is None or bufs["recv"].numel() < max(nr, 1) * engine.T:
                bufs["recv"] = torch.empty(max(nr, 1) * engine.T, dtype=vdtype, device=dev)
            engine.exchange_pack(t, bufs["send"])
            if world > 1:        # every rank takes part even with nothing to move: the peers' counts are not known here
                collective.all_to_all_records(bufs["send"], bufs["recv"], sc, rc, width)
            engine.exchange_commit(t, bufs["recv"])
            moved["sent"] += ns
            moved["recv"] += nr
        engine.finish()

    steps(0, False)
    repeat_whole = bool(int(engine.cfg.flags) & capi.FLAG_REPEAT_IN_FLOATING_POINT)
    last = -1
    for _ in range(engine.T + 1):
        g, _gap = engine.first_bad_generation()
        if g < 0 or g <= last or repeat_whole or not engine._begin_args.get("keep_history", True):
            break
        engine.repair_begin(g, local)
        collective.all_gather(local, allt)
        engine.repair_end(g, allt, world, rank, local)
        steps(g, True)
        last = g
    n_sent, n_recv = moved["sent"], moved["recv"]
    s = engine.summary()
    raw = torch.from_numpy(engine.stats()).to(dev)
    torch.cuda.current_stream().synchronize()     # the copy ran on torch's stream, the collective runs on the engine's
    collective.all_reduce_sum(raw)
    engine.sync()                                 # the reduction ran on the engine's stream; .cpu() below runs on torch's
    stats = normalise_joint_stats(raw.cpu().numpy(), s["log_norm"], s["max_logw"], engine.is_int)
    if counters is not None:
        counters["records_sent"] = n_sent
        counters["records_received"] = n_recv
    return stats, s
