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


def _single(engine, alg, model, obs, n, seed, ess):
    engine.begin(alg, model, obs, n, seed=seed, ess_threshold=ess)
    engine.run()
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
    """Continuous weights (floating-point form of the step, device-side decision when ESS-triggered): the sharded run's
    CDF is evaluated in a different order, so at most a couple of boundary offspring may differ."""
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
    differ = (paths != ref_paths).any(axis=0).sum()
    assert differ <= 2
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-4 if differ else 1e-11)
    assert abs(s["log_evidence"] - ref_sum["log_evidence"]) < 1e-11 and s["n_resampled"] == ref_sum["n_resampled"]
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
    differ = (paths != ref_paths).any(axis=0).sum()
    assert differ <= 2
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-4 if differ else 1e-11)
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
    differ = (paths != ref_paths).any(axis=0).sum()
    assert differ <= 2
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
    g.close()
    assert reruns <= 1 and 20 <= s["n_resampled"] <= 60
    assert np.abs(stats - z["hmm128_smooth"]).max() < 3e-3
    assert abs(s["log_evidence"] - float(z["hmm128_logz"])) < 5e-3
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=7, ess_threshold=0.5)
    engine.run()
    one, one_stats = engine.summary(), engine.stats()
    # Floating-point form: the sharded CDF is summed in another order.  Up to ~3e7 particles the two runs are the same run; at this
    # size some offspring flips across a CDF boundary within the first resamplings, and a single flip shifts the systematic comb
    # against every later source (ancestors move to NEIGHBOURING slots, whose states are unrelated): from there on the two are
    # different, equally valid samples of the same posterior -- Monte-Carlo-close, not bit-close.  (The count form of the
    # every-step schedule is integer arithmetic and has no such sensitivity: test_group_hmm_every_step_is_bit_identical_to_one_gpu.)
    assert one["n_resampled"] == s["n_resampled"] and abs(one["log_evidence"] - s["log_evidence"]) < 2e-3
    np.testing.assert_allclose(stats, one_stats, rtol=0, atol=4e-3)
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
    g.close()
    assert reruns <= 2
    # (smoothing by ancestral lines degenerates towards t = 0: the filtering-quality end is tight, the far end Monte-Carlo-limited)
    assert np.abs(stats[-1, 0] - z["lgssm100_smooth_mean"][-1]) < 5e-3 and np.abs(stats[-1, 1] - z["lgssm100_smooth_var"][-1]) < 5e-3
    assert np.abs(stats[:, 0] - z["lgssm100_smooth_mean"]).max() < 5e-2
    assert abs(s["log_evidence"] - float(z["lgssm100_logz"])) < 2e-2

