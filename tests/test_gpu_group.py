"""The library's own multi-GPU driver (cpprob_amd/csrc/group.hpp) on one GPU: every rank of the group sits on cuda:0
("loopback": one stream, copies instead of collectives), so the WHOLE stream-ordered exchange protocol -- device-side
plan, fixed-capacity transport segments, annex commit, sharded prefix-count steps -- runs exactly as it would over RCCL,
and its results are compared with ONE context holding all particles."""
import os

import numpy as np
import pytest

import cpprob_amd as cp

pytestmark = pytest.mark.gpu


def _obs(golden_dir, key):
    return np.load(os.path.join(golden_dir, "observations.npz"))[key]


def _single(engine, alg, model, obs, n, seed, ess, run_index=0):
    engine.begin(alg, model, obs, n, seed=seed, ess_threshold=ess)
    engine.run(run_index)
    return engine.stats().copy(), engine.summary(), engine.paths(), engine.logw()


@pytest.mark.parametrize("shards", [[50000, 50000], [30000, 50001, 19999], [25000] * 8, [1000, 2000, 70000, 3000]])
def test_group_hmm_every_step_is_bit_identical_to_one_gpu(engine, golden_dir, shards):
    """Table-weight model, every-step schedule: integer prefix counts make the sharded run draw EXACTLY the ancestors one
    GPU would -- every trace, every weight, the evidence and the posterior, bit for bit, for any shard layout."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, "hmm16")
    n = int(sum(shards))
    ref_stats, ref_sum, ref_paths, ref_logw = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 31, 2.0)
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=31, ess_threshold=2.0, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), True) for r in range(len(shards))], axis=1)
    assert np.array_equal(paths, ref_paths)
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    g.close()


def test_group_long_traces_extract_lineages_through_skip_rows(engine, golden_dir):
    """T >= 24 in the exchange scope: a migrating particle's lineage is extracted in blocks of eight generations through the skip
    rows (csrc/exchange.hpp).  Count form, uneven shards, outlying observations that move mass between shards: every trace equals
    the one-GPU run's, bit for bit -- and the same with the skip rows switched off (CPPROB_HIP_FLAG_NO_SKIP_ROWS)."""
    obs = np.array(_obs(golden_dir, "hmm128")[:44])
    obs[[5, 23]] *= 4.0
    shards = [30000, 50001, 19999, 7]
    n = int(sum(shards))
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 13, 2.0)
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=13, ess_threshold=2.0, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), True) for r in range(len(shards))], axis=1)
    g.close()
    assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=13, ess_threshold=2.0, shard_sizes=shards, flags=cp.capi.FLAG_NO_SKIP_ROWS)
    g.run()
    stats2, s2, _ = g.results()
    paths2 = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), True) for r in range(len(shards))], axis=1)
    g.close()
    assert np.array_equal(paths2, ref_paths) and s2["log_evidence"] == s["log_evidence"] and np.array_equal(stats2, stats)


def _ctx_paths(g, r, n_r, T, is_int):
    e = g.context(r)
    e.n = n_r
    e.T = T
    return e.paths()


@pytest.mark.parametrize("ess", [2.0, 0.5])
def test_group_lgssm_matches_one_gpu(engine, golden_dir, ess):
    """Continuous weights, device-side decision when ESS-triggered: the fixed-point form (integer weights, exact 64-bit prefix
    masses) makes the sharded run draw EXACTLY the ancestors one GPU would -- every trace, the decisions, the evidence, bit for bit."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, "lgssm100")[:20]
    shards = [40000, 40001, 39999]
    n = sum(shards)
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, 5, ess)
    g = cp.Group([0] * 3)
    g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=5, ess_threshold=ess, shard_sizes=shards)
    g.run()
    stats, s, _ = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), False) for r in range(3)], axis=1)
    assert np.array_equal(paths, ref_paths)
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    g.close()


def test_group_hmm128_ess_triggered_matches_one_gpu(engine, golden_dir):
    import torch  # noqa: F401
    obs = _obs(golden_dir, "hmm128")
    shards = [60000, 60000]
    n = sum(shards)
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 9, 0.5)
    g = cp.Group([0, 0])
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=9, ess_threshold=0.5, shard_sizes=shards)
    g.run()
    stats, s, _ = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), True) for r in range(2)], axis=1)
    assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    assert s["n_resampled"] == ref_sum["n_resampled"] and 0 < s["n_resampled"] < 127
    g.close()


def test_group_repopulation_from_one_rank_enlarges_the_transport(engine, golden_dir):
    """An outlying first observation leaves the mass on a few particles of one shard: whole shards are repopulated from it --
    far more immigrants than the default segments and annex hold.  The overflow flag travels with the final all-reduce and the
    driver repeats the run with full-size segments; the answer is the one-GPU answer."""
    import torch  # noqa: F401
    obs = np.array(_obs(golden_dir, "lgssm100")[:12])
    obs[0] = 7.5
    shards = [40000, 40000, 40000]
    n = sum(shards)
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, 5, 0.5)
    g = cp.Group([0] * 3)
    g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=5, ess_threshold=0.5, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    assert reruns >= 1
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), False) for r in range(3)], axis=1)
    # (fixed-point form: the very traces; had the outlier cost the weights their bits, both runs would have been repeated in the
    #  floating-point form, whose sharded CDF is summed in another order -- a couple of boundary offspring may then differ)
    assert ref_sum["step_form"] == s["step_form"]
    differ = (paths != ref_paths).any(axis=0).sum()
    assert differ == 0 if s["step_form"] == cp.capi.FORM_FIXED else differ <= 2
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-4 if differ else 1e-11)
    g.close()


def test_group_sis_and_back_to_back_runs(engine, golden_dir):
    import torch  # noqa: F401
    n = 100003
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], n, seed=3)
    engine.run()
    ref = engine.stats().copy()
    g = cp.Group([0, 0, 0])
    g.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], n, seed=3)
    g.run()
    stats, s, _ = g.results()
    np.testing.assert_allclose(stats, ref, rtol=1e-12)
    # several runs in flight, one synchronisation: the last one's results; run indices decorrelate
    obs = _obs(golden_dir, "hmm16")
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, 90000, seed=1, ess_threshold=2.0)
    for i in range(5):
        g.run(i)
    a, sa, _ = g.results()
    g.run(4)
    b, sb, _ = g.results()
    assert np.array_equal(a, b) and sa["log_evidence"] == sb["log_evidence"]
    g.run(0)
    c, _, _ = g.results()
    assert not np.array_equal(a, c)
    g.close()


def test_group_over_rccl_with_one_rank(engine, golden_dir):
    """world = 1 over the real transport: the library's RCCL calls (communicator, all-gather, all-reduce on the context's
    stream) run on this GPU; bit-identical to the plain run."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, "hmm16")
    n = 200000
    ref_stats, ref_sum, _, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 4, 2.0)
    g = cp.Group([0])
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=2.0)
    g.run()
    stats, s, reruns = g.results()
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    assert s["log_evidence"] == ref_sum["log_evidence"] and reruns == 0
    g.close()


def test_config5_whole_population_over_eight_ranks(engine, golden_dir):
    """BASELINE.json configs[4] at FULL size -- hmm<128>, 10^8 particles, eight ranks, ESS-triggered resampling with ancestor
    redistribution -- with all eight ranks on this one GPU (loopback transport; 68 GB of its HBM): the whole stream-ordered protocol
    the 8-GPU run uses, against the exact posterior and against ONE context holding all 10^8 particles."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    obs, n = z["hmm128"], 100_000_000
    g = cp.Group([0] * 8)
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=7, ess_threshold=0.5)
    g.run()
    stats, s, reruns = g.results()
    tr = g.traffic()
    per = [g.context(r).exchange_traffic() for r in range(8)]
    g.close()
    # bytes on the links = the migrants' states and origin slots (remote lineages): records x (1 + 8) bytes
    assert tr["transport"] == cp.capi.TRANSPORT_DIRECT and tr["remote_lineages"] == 1
    assert tr["records"] == sum(p[1] for p in per) and tr["wire_bytes"] == tr["payload_bytes"] == tr["records"] * 9
    print("configs[4] traffic", tr)
    assert reruns <= 1 and 20 <= s["n_resampled"] <= 60
    assert np.abs(stats - z["hmm128_smooth"]).max() < 3e-3
    assert abs(s["log_evidence"] - float(z["hmm128_logz"])) < 5e-3
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=7, ess_threshold=0.5)
    engine.run()
    one, one_stats = engine.summary(), engine.stats()
    # fixed-point form: integer weights and exact 64-bit prefix masses -- the eight-rank run IS the one-GPU run, at 10^8 particles too
    assert one["n_resampled"] == s["n_resampled"] and one["log_evidence"] == s["log_evidence"]
    np.testing.assert_allclose(stats, one_stats, rtol=0, atol=1e-12)
    assert np.abs(one_stats - z["hmm128_smooth"]).max() < 3e-3
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm16"], 1000, seed=1)          # (hand the 68 GB back)


def test_config4_whole_population_over_eight_ranks(engine, golden_dir):
    """BASELINE.json configs[3] at full size -- linear_gaussian_1d<100>, 10^7 particles over eight ranks -- likewise on one GPU,
    against Kalman / RTS."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    obs, n = z["lgssm100"], 10_000_000
    g = cp.Group([0] * 8)
    g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=7, ess_threshold=0.5)
    g.run()
    stats, s, reruns = g.results()
    tr = g.traffic()
    per = [g.context(r).exchange_traffic() for r in range(8)]
    g.close()
    assert tr["transport"] == cp.capi.TRANSPORT_DIRECT and tr["remote_lineages"] == 1
    assert tr["records"] == sum(p[1] for p in per) and tr["wire_bytes"] == tr["payload_bytes"] == tr["records"] * 16
    print("configs[3] traffic", tr)
    assert reruns == 0                 # (the default annex holds a well-mixed run's immigrants: sqrt(N) per step, T steps)
    # (smoothing by ancestral lines degenerates towards t = 0: the filtering-quality end is tight, the far end Monte-Carlo-limited)
    assert np.abs(stats[-1, 0] - z["lgssm100_smooth_mean"][-1]) < 5e-3 and np.abs(stats[-1, 1] - z["lgssm100_smooth_var"][-1]) < 5e-3
    assert np.abs(stats[:, 0] - z["lgssm100_smooth_mean"]).max() < 5e-2
    assert abs(s["log_evidence"] - float(z["lgssm100_logz"])) < 2e-2



def _run_group(g, alg, model, obs, n, seed, ess, shards):
    g.begin(alg, model, obs, n, seed=seed, ess_threshold=ess, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    return stats, s, reruns, g.traffic()


@pytest.mark.parametrize("model,key,T,ess", [(cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 30, 0.5), (cp.MODEL_HMM3, "hmm128", 40, 0.5)])
def test_group_transports_agree_and_account_their_bytes(engine, golden_dir, model, key, T, ess):
    """Three ways to move a migrating particle, identical results (every trace of every shard, the evidence, the posterior):
    REMOTE LINEAGES (default where every rank can address every rank's store): the migrant takes its state and the slot it leaves
    along -- value + 8 bytes per record, stored straight into the receiving rank's annex column and origin table -- and its history stays where it is; the read-out walks into that rank's store;
    DIRECT + shipped lineages: the packing kernel stores the whole lineage (t + 1 values) into the receiving rank's buffer;
    SEND/RECV: fixed-capacity segments of lineages (the fall-back).  A step that does not resample moves nothing."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, key)[:T]
    shards = [30000, 50001, 19999, 40000]
    n = int(sum(shards))
    vsz = 1 if model == cp.MODEL_HMM3 else 8
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, model, obs, n, 21, ess)
    out = {}
    for name, flags in (("remote", 0), ("ship", cp.capi.GROUP_SHIP_LINEAGES), ("sendrecv", cp.capi.GROUP_SENDRECV)):
        g = cp.Group([0] * len(shards))
        g.transport(flags=flags)
        stats, s, reruns, tr = _run_group(g, cp.ALG_SMC, model, obs, n, 21, ess, shards)
        per_rank = [g.context(r).exchange_traffic() for r in range(len(shards))]
        _, resampled = g.context(0).step_trace()
        paths = np.concatenate([_ctx_paths(g, r, shards[r], T, model == cp.MODEL_HMM3) for r in range(len(shards))], axis=1)
        g.close()
        assert reruns == 0 and np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"], name
        np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
        out[name] = (stats, tr, per_rank)
        for p in per_rank:
            assert np.all(p[0][:T - 1][resampled[:T - 1] == 0] == 0) and p[0][T - 1] == 0        # a step that does not resample sends nothing
    records = sum(p[1] for p in out["remote"][2])
    lineage_bytes = sum(int((p[0] * (np.arange(T) + 1)).sum()) * vsz for p in out["ship"][2])
    tr = out["remote"][1]
    assert tr["transport"] == cp.capi.TRANSPORT_DIRECT and tr["remote_lineages"] == 1
    words = key == "hmm16"                        # (T <= 16 states of 2 bits on the every-step schedule: the particles carry their traces, 4 bytes more)
    assert tr["records"] == records > 0 and tr["payload_bytes"] == tr["wire_bytes"] == records * (vsz + 8 + (4 if words else 0))
    tr = out["ship"][1]
    assert tr["transport"] == cp.capi.TRANSPORT_DIRECT and tr["remote_lineages"] == 0
    assert tr["records"] == records and tr["payload_bytes"] == tr["wire_bytes"] == lineage_bytes
    tr = out["sendrecv"][1]
    cap = (int(8.0 * np.sqrt(n)) // 1024) * 1024 + 4096
    assert tr["transport"] == cp.capi.TRANSPORT_SENDRECV and tr["records"] == records and tr["payload_bytes"] == lineage_bytes
    assert tr["wire_bytes"] == (2 * len(shards) - 2) * min(cap, max(shards)) * vsz * (T - 1) * T // 2
    assert np.array_equal(out["ship"][0], out["sendrecv"][0])
    if words:       # (trace-word read-out: integer counts per rank against the walk's floating-point sums -- the same numbers to rounding)
        np.testing.assert_allclose(out["remote"][0], out["ship"][0], rtol=0, atol=1e-14)
    else:
        assert np.array_equal(out["remote"][0], out["ship"][0])
    assert out["ship"][1]["wire_bytes"] < out["sendrecv"][1]["wire_bytes"] // 10
    if T * vsz > 2 * (vsz + 12): assert out["remote"][1]["wire_bytes"] < out["ship"][1]["wire_bytes"]      # (short traces of bytes: the lineage is smaller than the origin word)


W1 = cp.capi.GROUP_WORLD1_COLLECTIVES
@pytest.mark.parametrize("flags", [W1 | cp.capi.GROUP_LIBRARY_COLLECTIVES, W1 | cp.capi.GROUP_LIBRARY_COLLECTIVES | cp.capi.GROUP_SENDRECV, W1, W1 | cp.capi.GROUP_SHIP_LINEAGES])
def test_group_of_one_rank_runs_every_rccl_call_of_the_multi_gpu_path(engine, golden_dir, flags):
    """world = 1 over the real library: with CPPROB_HIP_GROUP_WORLD1_COLLECTIVES the rank is its own peer, so the per-step
    collectives and the final ncclAllReduce all execute on this GPU.  LIBRARY_COLLECTIVES: ncclAllGather per step and the ordering
    all-gather of the direct transport -- or, with SENDRECV, the grouped ncclSend / ncclRecv of a whole segment to and from itself.
    Without it (the default wherever ranks can map each other's memory): the mailbox collectives -- the rank posts into and spins on
    its own mailbox.  Results are bit-identical to the plain run, count form and fixed-point form."""
    import torch  # noqa: F401
    for model, key, T, ess in ((cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12, 0.5)):
        obs = _obs(golden_dir, key)[:T]
        n = 150000
        ref_stats, ref_sum, _, _ = _single(engine, cp.ALG_SMC, model, obs, n, 4, ess)
        g = cp.Group([0])
        g.transport(flags=flags)
        g.begin(cp.ALG_SMC, model, obs, n, seed=4, ess_threshold=ess)
        for i in range(3):
            g.run(i)
        g.run(0)
        stats, s, reruns = g.results()
        tr = g.traffic()
        g.close()
        np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
        assert s["log_evidence"] == ref_sum["log_evidence"] and reruns == 0
        assert tr["transport"] == (cp.capi.TRANSPORT_SENDRECV if flags & cp.capi.GROUP_SENDRECV else cp.capi.TRANSPORT_DIRECT)
        assert tr["records"] == 0 and tr["collective_bytes"] > 0
        assert tr["mailbox_collectives"] == (0 if flags & cp.capi.GROUP_LIBRARY_COLLECTIVES else 1)
        if flags & cp.capi.GROUP_SENDRECV:
            assert tr["wire_bytes"] > 0          # the stale segment did travel (to itself)


@pytest.mark.parametrize("model,key,T,ess,extra", [(cp.MODEL_HMM3, "hmm16", 16, 2.0, 0), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 30, 0.5, 0),
                                                  (cp.MODEL_HMM3, "hmm128", 40, 0.5, cp.capi.GROUP_SHIP_LINEAGES)])
def test_group_mailbox_collectives_on_loopback_ranks(engine, golden_dir, model, key, T, ess, extra):
    """The per-step collectives as stores into the peers' mailboxes (csrc/device_collectives.hpp), on loopback ranks: every rank posts,
    then every rank finds what it waits for (one stream: program order delivered it) -- sequence numbers, parities, the ordering
    barrier of the direct stores and the proof round at begin all run; the answer is the one-GPU answer, run after run."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, key)[:T]
    shards = [30000, 50001, 19999, 40000, 2048]
    n = int(sum(shards))
    g = cp.Group([0] * len(shards))
    g.transport(flags=cp.capi.GROUP_MAILBOX_COLLECTIVES | extra)
    g.begin(cp.ALG_SMC, model, obs, n, seed=21, ess_threshold=ess, shard_sizes=shards)
    for run in (0, 1, 2, 1):
        g.run(run)
        stats, s, reruns = g.results()
        tr = g.traffic()
        ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, model, obs, n, 21, ess, run_index=run)
        paths = np.concatenate([_ctx_paths(g, r, shards[r], T, model == cp.MODEL_HMM3) for r in range(len(shards))], axis=1)
        assert reruns == 0 and tr["mailbox_collectives"] == 1 and tr["records"] > 0
        assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"]
        np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    g.close()


def test_group_can_fall_back_to_the_conservative_transport_after_a_run(engine, golden_dir):
    """What bench.py does when the default transport's answer is not the posterior on hardware it has never seen: the same group
    object, begun again with library collectives, shipped lineages and the walk read-out -- here on one GPU (RCCL world 1 with the rank
    as its own peer), where both transports give the one-GPU answer."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, "hmm16")
    n = 120000
    ref_stats, ref_sum, _, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 4, 2.0)
    g = cp.Group([0])
    g.transport(flags=cp.capi.GROUP_WORLD1_COLLECTIVES)
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=2.0)
    g.run()
    stats, s, _ = g.results()
    assert g.traffic()["mailbox_collectives"] == 1 and s["log_evidence"] == ref_sum["log_evidence"]
    g.transport(flags=cp.capi.GROUP_WORLD1_COLLECTIVES | cp.capi.GROUP_LIBRARY_COLLECTIVES | cp.capi.GROUP_SHIP_LINEAGES)
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=2.0, flags=cp.capi.FLAG_WALK_READOUT)
    g.run()
    stats2, s2, _ = g.results()
    tr = g.traffic()
    g.close()
    assert tr["mailbox_collectives"] == 0 and tr["remote_lineages"] == 0 and s2["log_evidence"] == ref_sum["log_evidence"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)
    np.testing.assert_allclose(stats2, ref_stats, rtol=0, atol=1e-13)


def test_group_world_limit_is_63_ranks(engine, golden_dir):
    """The plan lives on one wavefront whose lane r holds the bound o_r, r = 0 .. world: 63 ranks are served, 64 refused."""
    obs = _obs(golden_dir, "hmm16")[:6]
    with pytest.raises(cp.CpprobHipError):
        cp.Group([0] * 64)
    world = 63
    shards = [1500 + 7 * (r % 5) for r in range(world)]
    n = int(sum(shards))
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 17, 2.0)
    g = cp.Group([0] * world)
    g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=17, ess_threshold=2.0, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], len(obs), True) for r in range(world)], axis=1)
    g.close()
    assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)


def test_group_over_two_processes_maps_the_peer_buffers_through_hipipc(golden_dir):
    """One rank per PROCESS, both on this GPU, the caller's collectives (gloo through cpprob_hip_group_create_external): each
    process maps the other's receive buffer with hipIpcOpenMemHandle and its packing kernel stores the migrating lineages there --
    the transport of the one-process-per-GPU form (bench.py under torchrun), minus RCCL, which refuses two ranks on one device.
    Bit-identical to one context holding all particles (count form) / within boundary flips (floating-point form)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "ext_group_worker.py"), os.path.join(golden_dir, "observations.npz")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "EXT_GROUP_OK" in p.stdout


@pytest.mark.parametrize("model,key,T,ess,n", [(cp.MODEL_HMM3, "hmm16", 6, 2.0, 13_000_777), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 8, 0.5, 9_000_001),
                                               (cp.MODEL_HMM3, "hmm128", 8, 0.5, 16_700_000), (cp.MODEL_HMM3, "hmm16", 5, 2.0, 20_000_000)])
def test_three_level_hierarchy_against_sharded_runs(engine, golden_dir, model, key, T, ess, n):
    """Above 4096 tiles the hierarchy has three levels (a block's last arriver forwards its totals).  One context holding everything
    and eight loopback shards (two-level hierarchies of their own) must draw the very same traces: totals, prefix sums, probes and
    the descent all take part."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, key)[:T]
    world = 8
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, model, obs, n, 41, ess)
    engine.begin(cp.ALG_SMC, model, obs, 1024, seed=1)              # (give the big store back before the shards allocate theirs)
    g = cp.Group([0] * world)
    g.begin(cp.ALG_SMC, model, obs, n, seed=41, ess_threshold=ess)
    g.run()
    stats, s, reruns = g.results()
    base, rem = divmod(n, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    ok = True
    off = 0
    for r in range(world):
        p = _ctx_paths(g, r, sizes[r], T, model == cp.MODEL_HMM3)
        ok = ok and np.array_equal(p, ref_paths[:, off:off + sizes[r]])
        off += sizes[r]
    g.close()
    assert ok and s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-13)


@pytest.mark.parametrize("shards", [[50000, 50000], [30000, 50001, 19999, 40000, 2048]])
def test_group_shards_carry_trace_words_across_ranks(engine, golden_dir, shards):
    """Short discrete traces in the exchange scope (remote lineages): a migrant takes its 4-byte trace word along, every shard reads its
    posterior sums out of its own particles' words -- integer counts, no lineage walk into other ranks' stores -- and the ranks' sums
    meet in the run's final all-reduce.  Against CPPROB_HIP_FLAG_WALK_READOUT (the walk through the origin tables) and against ONE
    context: the same traces, the same evidence, the same statistics to rounding; 4 bytes more per migrant on the links."""
    import torch  # noqa: F401
    obs = _obs(golden_dir, "hmm16")
    n = int(sum(shards))
    ref_stats, ref_sum, ref_paths, _ = _single(engine, cp.ALG_SMC, cp.MODEL_HMM3, obs, n, 12, 2.0)
    out = {}
    for name, flags in (("words", 0), ("walk", cp.capi.FLAG_WALK_READOUT)):
        g = cp.Group([0] * len(shards))
        g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=12, ess_threshold=2.0, shard_sizes=shards, flags=flags)
        for run in (3, 0):
            g.run(run)
        stats, s, reruns = g.results()
        tr = g.traffic()
        paths = np.concatenate([_ctx_paths(g, r, shards[r], 16, True) for r in range(len(shards))], axis=1)
        g.close()
        assert reruns == 0 and np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"]
        np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-14)
        out[name] = (stats, tr)
    assert out["words"][1]["records"] == out["walk"][1]["records"] > 0
    assert out["words"][1]["wire_bytes"] == out["words"][1]["records"] * 13 and out["walk"][1]["wire_bytes"] == out["walk"][1]["records"] * 9
    np.testing.assert_allclose(out["words"][0], out["walk"][0], rtol=0, atol=1e-14)


@pytest.mark.parametrize("shards", [[50000, 50000], [30000, 50001, 19999], [25000] * 8, [1000, 2000, 70000, 3000, 999]])
@pytest.mark.parametrize("model,key,T,ess", [(cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.MODEL_HMM3, "hmm128", 40, 0.5), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 25, 0.5),
                                             (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12, 2.0)])
def test_group_stratified_resampling_is_bit_identical_to_one_gpu_and_the_oracle(engine, golden_dir, shards, model, key, T, ess):
    """Stratified resampling through the exchange scope: output j sits at j + u_j (the uniform of OUTPUT j: the same on whichever
    rank holds it), so the outputs a rank's sources own are ONE interval [A(mass before the rank), A(mass up to its end)) that every
    rank derives from the all-gathered totals -- the plan, the packing launch's ancestor search and the sharded step kernel run the
    stratified comb on integers (prefix counts for the every-step HMM, fixed-point masses otherwise) exactly as one GPU does: every
    surviving trace, the decisions and the evidence equal the single-context run's, which equals the oracle's, for 2 .. 8 uneven
    loopback shards."""
    import torch  # noqa: F401
    from oracle import oracle as O
    obs = _obs(golden_dir, key)[:T]
    n = int(sum(shards))
    engine.begin(cp.ALG_SMC, model, obs, n, seed=23, resampler=cp.RESAMPLE_STRATIFIED, ess_threshold=ess)
    engine.run()
    ref_stats, ref_sum, ref_paths, ref_anc = engine.stats().copy(), engine.summary(), engine.paths(), engine.ancestors()
    orc = O.smc(model, obs, n, 23, O.RESAMPLE_STRATIFIED, ess)
    assert np.array_equal(ref_anc, orc["anc"]) and ref_sum["n_resampled"] == int(orc["resampled"].sum())
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, model, obs, n, seed=23, resampler=cp.RESAMPLE_STRATIFIED, ess_threshold=ess, shard_sizes=shards)
    g.run()
    stats, s, _ = g.results()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], T, model == cp.MODEL_HMM3) for r in range(len(shards))], axis=1)
    g.close()
    assert s["step_form"] == ref_sum["step_form"] != cp.capi.FORM_FLOAT
    assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-12)


@pytest.mark.parametrize("shards", [[50000, 50000], [30000, 50001, 19999], [25000] * 8, [1000, 2000, 70000, 3000, 999]])
@pytest.mark.parametrize("model,key,T,ess", [(cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.MODEL_HMM3, "hmm128", 40, 0.5), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 25, 0.5),
                                             (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12, 2.0)])
def test_group_multinomial_resampling_is_bit_identical_to_one_gpu_and_the_oracle(engine, golden_dir, shards, model, key, T, ess):
    """Thesis Alg. 1's own resampler (multinomial, strata form) through the exchange scope: the thresholds come stratum by stratum, so a
    rank's sources own ONE interval of outputs plus their share of the <= world - 1 strata the ranks' boundaries cut -- those are
    classified output by output in a launch every rank runs on the all-gathered totals alone (exchange_cut_kernel), and a migrant's
    annex column is (s - shard begin) - kept_before(s) on both sides (csrc/strata_cut.hpp).  Both integer forms (prefix counts for
    the every-step HMM, fixed-point masses otherwise): every surviving trace, the decisions and the evidence equal the single-context
    run's, which equals the oracle's, for 2 .. 8 uneven loopback shards (odd shard begins included)."""
    import torch  # noqa: F401
    from oracle import oracle as O
    obs = _obs(golden_dir, key)[:T]
    n = int(sum(shards))
    engine.begin(cp.ALG_SMC, model, obs, n, seed=29, resampler=cp.RESAMPLE_MULTINOMIAL, ess_threshold=ess)
    engine.run()
    ref_stats, ref_sum, ref_paths, ref_anc = engine.stats().copy(), engine.summary(), engine.paths(), engine.ancestors()
    orc = O.smc(model, obs, n, 29, O.RESAMPLE_MULTINOMIAL, ess)
    assert np.array_equal(ref_anc, orc["anc"]) and ref_sum["n_resampled"] == int(orc["resampled"].sum())
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, model, obs, n, seed=29, resampler=cp.RESAMPLE_MULTINOMIAL, ess_threshold=ess, shard_sizes=shards)
    g.run()
    stats, s, reruns = g.results()
    tr = g.traffic()
    paths = np.concatenate([_ctx_paths(g, r, shards[r], T, model == cp.MODEL_HMM3) for r in range(len(shards))], axis=1)
    g.close()
    assert reruns == 0 and s["step_form"] == ref_sum["step_form"] != cp.capi.FORM_FLOAT
    assert tr["records"] > 0 and tr["remote_lineages"] == 1
    assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-12)


def test_group_multinomial_refuses_what_it_cannot_move(engine, golden_dir):
    """Multinomial migrants are not one interval per peer: the segment transports refuse them loudly (no silent change of estimator)."""
    obs = _obs(golden_dir, "hmm16")
    g = cp.Group([0, 0])
    g.transport(0, -1, cp.capi.GROUP_SHIP_LINEAGES)
    with pytest.raises(cp.CpprobHipError):
        g.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, 40000, seed=1, resampler=cp.RESAMPLE_MULTINOMIAL, ess_threshold=2.0)
    g.close()


@pytest.mark.parametrize("resampler", [cp.RESAMPLE_SYSTEMATIC, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL])
@pytest.mark.parametrize("shards,flags", [([2500, 2500], 0), ([1000, 2501, 1499], 0), ([700] * 6 + [800], cp.capi.GROUP_MAILBOX_COLLECTIVES)])
def test_group_run_that_loses_its_bits_is_repaired_in_integers_like_one_gpu(engine, golden_dir, resampler, shards, flags):
    """The group twin of test_fixed_point_run_that_loses_its_bits_is_repaired_from_the_offending_generation: an observation ~30 standard
    deviations from every particle costs a generation its bits.  The shards repair it as ONE context does -- log-weights of the first
    offending generation recomputed from every rank's store, the POPULATION's exact maximum all-gathered (24 bytes), masses against it,
    the books rewound, the steps behind it run again (cpprob_hip_smc_repair_begin / _end under cpprob_hip_group_results) -- so the
    sharded run still draws the one-GPU run's ancestors, which are the oracle's: no repeat of the run, no floating-point form.  Outlier
    in a middle, the last and the first generation; mailbox collectives too (their slots are numbered from the repaired generation)."""
    import torch  # noqa: F401
    from oracle import oracle as O
    n = int(sum(shards))
    for at, val in ((6, 40.0), (13, 35.0), (0, 9.0)):
        obs = np.array(_obs(golden_dir, "lgssm100")[:14])
        obs[at] = val
        engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, resampler=resampler, ess_threshold=0.5)
        engine.run()
        ref_stats, ref_sum, ref_paths, ref_anc = engine.stats().copy(), engine.summary(), engine.paths(), engine.ancestors()
        assert ref_sum["n_requantised"] >= 1
        orc = O.smc(cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, 8, resampler, 0.5)
        assert np.array_equal(ref_anc, orc["anc"])
        g = cp.Group([0] * len(shards))
        g.transport(flags=flags)
        g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, resampler=resampler, ess_threshold=0.5, shard_sizes=shards)
        g.run()
        stats, s, reruns = g.results()
        paths = np.concatenate([_ctx_paths(g, r, shards[r], 14, False) for r in range(len(shards))], axis=1)
        g.run(1)                                                # the group is as good as new behind a repair
        g.run(0)
        stats2, s2, reruns2 = g.results()
        g.close()
        # (a population that collapses onto one particle migrates wholesale: the transport may have had to grow -- a repeat with a larger
        #  annex, which is not the repeat in the floating-point form this test rules out: step_form stays the fixed-point one)
        assert reruns2 == reruns and s["step_form"] == cp.capi.FORM_FIXED and s["n_requantised"] == ref_sum["n_requantised"], (at, s, reruns, reruns2)
        assert np.array_equal(paths, ref_paths) and s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"], at
        np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-10)       # (sums over shards in another order, states up to 40: raw second moments of 1600)
        assert np.array_equal(stats2, stats) and s2 == s
