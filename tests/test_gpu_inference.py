"""GPU parity of cpprob::inference (SIS and SMC) through the C ABI.

Against the oracle on the same seed (particle by particle), against the committed golden
observation vectors + exact posteriors, and through size-independent properties at full size.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from devmem import dcat, dtensor, dzeros, dzeros_like  # noqa: F401

import cpprob_amd as cp
from oracle import exact as E
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FP_TOL = 1e-11


def _obs(golden_dir, key):
    return np.load(os.path.join(golden_dir, "observations.npz"))[key]


@pytest.mark.parametrize("model,obs", [(cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0]), (cp.MODEL_GAUSSIAN_README, [3.0, 4.0]),
                                       (cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [8.0, 9.0])])
@pytest.mark.parametrize("n", [1, 1000, 10000, 4099, 8191])
def test_sis_gaussian_matches_oracle_per_particle(engine, model, obs, n):
    engine.begin(cp.ALG_SIS, model, obs, n, seed=2024)
    engine.run()
    vals, logw = O.sis(model, obs, n, 2024)
    np.testing.assert_allclose(engine.values(), vals, rtol=FP_TOL, atol=FP_TOL)
    np.testing.assert_allclose(engine.logw(), logw, rtol=FP_TOL, atol=FP_TOL)
    ref = O.weighted_moments(vals[0], logw)
    st = engine.stats()
    np.testing.assert_allclose(st[0], ref[:2], rtol=1e-9, atol=1e-10)
    s = engine.summary()
    assert abs(s["log_norm"] - ref[2]) < 1e-10
    assert abs(s["log_evidence"] - (ref[2] - np.log(n))) < 1e-10
    assert abs(s["ess_final"] - ref[3]) < 1e-6 * max(1.0, ref[3])
    assert s["n_predict"] == 1 and s["is_int"] == 0 and s["n_resampled"] == 0


@pytest.mark.parametrize("n", [1, 1000, 4099])
def test_sis_gaussian_2d_vector_statements_match_oracle_per_particle(engine, n):
    """reference models.hpp:38-49 through the C ABI: the two components of the vector-valued statements are rows of the
    particle store; the weight is the component sum of the vector observe."""
    obs = [3.0, 4.5]
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, obs, n, seed=2024)
    engine.run()
    vals, logw = O.sis(O.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, obs, n, 2024)
    np.testing.assert_allclose(engine.values(), vals, rtol=FP_TOL, atol=FP_TOL)
    np.testing.assert_allclose(engine.logw(), logw, rtol=FP_TOL, atol=FP_TOL)
    st = engine.stats()
    for d in range(2):
        ref = O.weighted_moments(vals[d], logw)
        np.testing.assert_allclose(st[d], ref[:2], rtol=1e-9, atol=1e-10)
    s = engine.summary()
    assert s["n_predict"] == 2 and s["n_resampled"] == 0
    engine.begin(cp.ALG_SMC, cp.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, obs, n, seed=2024)      # one observe statement: smc is sis
    engine.run()
    np.testing.assert_allclose(engine.values(), vals, rtol=FP_TOL, atol=FP_TOL)
    np.testing.assert_allclose(engine.logw(), logw, rtol=FP_TOL, atol=FP_TOL)
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, [1.0, 2.0, 3.0], 10)


def test_sis_config1_plumbing_10000_particles(engine, golden_dir):
    """BASELINE.json configs[0]: gaussian_unknown_mean, observes (3,4), 10^4 particles."""
    with open(os.path.join(golden_dir, "posteriors.json")) as f:
        post = json.load(f)["models_gaussian_obs_3_4"]
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], 10000, seed=1)
    engine.run()
    mean, var = engine.stats()[0]
    # MCSE at N=10^4, ESS/N = 0.344: sd(mean) ~ sqrt(0.833/3440) = 0.016
    assert abs(mean - post["mean"]) < 5 * 0.016
    assert abs(var - post["variance"]) < 5 * 0.025


def test_sis_gaussian_1e7_within_1e3_of_analytic(engine, golden_dir):
    """BASELINE.json configs[1] / north_star: posterior mean/variance within 1e-3 at 10^7 particles.
    MCSE(mean) = 3.4e-4, MCSE(var) = 4.3e-4; seed fixed."""
    with open(os.path.join(golden_dir, "posteriors.json")) as f:
        post = json.load(f)
    n = 10_000_000
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], n, seed=12345)
    engine.run()
    mean, var = engine.stats()[0]
    s = engine.summary()
    p = post["models_gaussian_obs_3_4"]
    assert abs(mean - p["mean"]) < 1e-3, mean
    assert abs(var - p["variance"]) < 1e-3, var
    assert abs(s["ess_final"] / n - 0.344) < 0.01
    assert abs(s["log_evidence"] - p["log_evidence"]) < 1e-3
    # README model, README.md:118
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_README, [3.0, 4.0], n, seed=12345)
    engine.run()
    mean, var = engine.stats()[0]
    assert abs(mean - 2.32353) < 1e-3 and abs(var - 1.05882) < 1.5e-3
    # thesis p.85: obs (8,9) -> N(7.25, 5/6); prior is far from the data, so ESS is lower
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [8.0, 9.0], n, seed=12345)
    engine.run()
    mean, var = engine.stats()[0]
    assert abs(mean - 7.25) < 1.5e-2 and abs(var - 5.0 / 6.0) < 2e-2   # ESS/N ~ 0.02 here: MCSE ~ 2.5e-3


@pytest.mark.parametrize("model,key,T", [(cp.MODEL_HMM3, "hmm16", 16), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12)])
def test_sis_state_space_matches_oracle(engine, golden_dir, model, key, T):
    obs = _obs(golden_dir, key)[:T]
    n = 20011
    engine.begin(cp.ALG_SIS, model, obs, n, seed=7)
    engine.run()
    vals, logw = O.sis(model, obs, n, 7)
    got = engine.values()
    if model == cp.MODEL_HMM3:
        assert np.array_equal(got, vals)                       # integer states: bit-exact
    else:
        np.testing.assert_allclose(got, vals, rtol=FP_TOL, atol=FP_TOL)
    np.testing.assert_allclose(engine.logw(), logw, rtol=1e-10, atol=1e-10)
    st = engine.stats()
    if model == cp.MODEL_HMM3:
        ref = np.array([O.weighted_hist(vals[t], logw, 3) for t in range(T)])
    else:
        ref = np.array([O.weighted_moments(vals[t], logw)[:2] for t in range(T)])
    np.testing.assert_allclose(st, ref, rtol=1e-8, atol=1e-10)
    # SIS: every trace is its own line
    assert np.array_equal(engine.paths(), got) if model == cp.MODEL_HMM3 else np.allclose(engine.paths(), got)


def _compare_smc(engine, model, obs, n, seed, resampler, ess):
    engine.begin(cp.ALG_SMC, model, obs, n, seed=seed, resampler=resampler, ess_threshold=ess)
    engine.run()
    ref = O.smc(model, obs, n, seed, resampler, ess)
    vals, anc, logw = engine.values(), engine.ancestors(), engine.logw()
    gess, gres = engine.step_trace()
    assert np.array_equal(gres, ref["resampled"])
    np.testing.assert_allclose(gess, ref["ess"], rtol=1e-6)
    T = len(obs)
    # every resampler runs on integers -- prefix counts (table weights, every-step schedule, systematic) or fixed-point masses
    # (everything else: systematic comb, stratified positions j + u_j, multinomial thresholds floor(u_j C_N)): the index work is
    # bit-exact against the oracle's statement of the same arithmetic
    assert np.array_equal(anc, ref["anc"]), np.mean(anc != ref["anc"])
    if model == cp.MODEL_HMM3:
        assert np.array_equal(vals, ref["hist"])
    else:
        np.testing.assert_allclose(vals, ref["hist"], rtol=0, atol=1e-9)
    s = engine.summary()
    assert abs(s["log_evidence"] - ref["log_z"]) < 1e-6
    assert s["n_resampled"] == int(ref["resampled"].sum())
    sm_ref = O.smoothing(ref["hist"], ref["anc"], ref["logw"])
    np.testing.assert_allclose(engine.stats(), sm_ref, atol=2e-3)
    # internal consistency, exact: device smoothing == oracle estimator applied to the DEVICE's own store
    if s["step_form"] == cp.capi.FORM_FIXED:
        # fixed-point form: the final weights ARE the integers q_i = rint(exp(lw_i - R) 2^32), R = summary()["max_logw"]
        sm_self = O.smoothing_linear(vals, anc, O.fix_weights(logw, s["max_logw"]).astype(np.float64))
        np.testing.assert_allclose(engine.stats(), sm_self, rtol=1e-11, atol=1e-13)
    else:
        sm_self = O.smoothing(vals, anc, logw)
        np.testing.assert_allclose(engine.stats(), sm_self, rtol=1e-9, atol=1e-11)
    # paths == lineage read-out of the device store (index work: bit-exact)
    path_idx = O.lineage(anc)
    expect = np.take_along_axis(vals, path_idx, axis=1)
    assert np.array_equal(engine.paths(), expect)
    return ref


@pytest.mark.parametrize("resampler", [cp.RESAMPLE_SYSTEMATIC, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL])
@pytest.mark.parametrize("ess", [2.0, 0.5])
def test_smc_hmm_matches_oracle(engine, golden_dir, resampler, ess):
    obs = _obs(golden_dir, "hmm16")
    _compare_smc(engine, cp.MODEL_HMM3, obs, 30000, 11, resampler, ess)


@pytest.mark.parametrize("n", [30_000, 300_000, 1_200_000, 2_000_000, 4_300_000])
def test_smc_hmm_every_step_is_bit_exact_against_the_oracle(engine, golden_dir, n):
    """Index work is bit-exact: table-weight models on an every-step schedule resample from INTEGER prefix counts
    (cpprob_amd/csrc/step_counts.hpp; oracle: orc_resample_table_systematic states the same arithmetic), so states and
    ancestors equal the oracle's at every size -- one level of the count hierarchy (<= 64 tiles), two (<= 4096) and three."""
    obs = _obs(golden_dir, "hmm16")
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=11, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0)
    engine.run()
    ref = O.smc(cp.MODEL_HMM3, obs, n, 11, cp.RESAMPLE_SYSTEMATIC, 2.0)
    vals, anc = engine.values(), engine.ancestors()
    assert np.array_equal(anc, ref["anc"])
    assert np.array_equal(vals, ref["hist"])
    gess, gres = engine.step_trace()
    assert np.array_equal(gres, ref["resampled"])
    np.testing.assert_allclose(gess, ref["ess"], rtol=1e-9)
    s = engine.summary()
    assert abs(s["log_evidence"] - ref["log_z"]) < 1e-9
    np.testing.assert_allclose(engine.stats(), O.smoothing(ref["hist"], ref["anc"], ref["logw"]), atol=1e-12)
    np.testing.assert_allclose(engine.logw(), ref["logw"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(engine.paths(), np.take_along_axis(vals, O.lineage(anc), axis=1))


def test_smc_hmm_every_step_floating_point_form_stays_within_its_flip_bound(engine, golden_dir):
    """The floating-point form of the same step (cpprob_hip_config::flags = CPPROB_HIP_FLAG_FLOATING_POINT_STEP; what stratified /
    multinomial resampling and non-exchange shards run) differs from the oracle only by CDF-boundary flips of the parallel
    summation order."""
    obs = _obs(golden_dir, "hmm16")
    n = 1200000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=11, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0, flags=cp.capi.FLAG_FLOATING_POINT_STEP)
    engine.run()
    ref = O.smc(cp.MODEL_HMM3, obs, n, 11, cp.RESAMPLE_SYSTEMATIC, 2.0)
    anc = engine.ancestors()
    T = len(obs)
    differs = [t for t in range(1, T) if not np.array_equal(anc[t], ref["anc"][t])]
    first = differs[0] if differs else T
    assert np.array_equal(engine.values()[:first], ref["hist"][:first])
    if first < T:
        d = anc[first].astype(np.int64) - ref["anc"][first].astype(np.int64)
        assert np.abs(d).max() == 1 and np.count_nonzero(d) <= 8, (np.abs(d).max(), np.count_nonzero(d))
    assert np.abs(engine.stats() - O.smoothing(ref["hist"], ref["anc"], ref["logw"])).max() < 5e-3


@pytest.mark.parametrize("resampler", [cp.RESAMPLE_SYSTEMATIC, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL])
@pytest.mark.parametrize("ess", [2.0, 0.5])
def test_smc_lgssm_matches_oracle(engine, golden_dir, resampler, ess):
    obs = _obs(golden_dir, "lgssm100")[:25]
    _compare_smc(engine, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, 20000, 5, resampler, ess)


@pytest.mark.parametrize("kind", ["stratified", "multinomial", "multinomial_literal"])
@pytest.mark.parametrize("model,key,T,ess,n", [(cp.MODEL_HMM3, "hmm16", 16, 2.0, 30_000), (cp.MODEL_HMM3, "hmm16", 16, 2.0, 1_200_000),
                                               (cp.MODEL_HMM3, "hmm16", 8, 0.5, 4_300_000), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12, 2.0, 1_250_000),
                                               (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 20, 0.5, 300_001)])
def test_smc_stratified_and_multinomial_are_bit_exact_against_the_oracle(engine, golden_dir, kind, model, key, T, ess, n):
    """Stratified and multinomial resampling (thesis Alg. 1 p.36 is multinomial) on integers -- the masses of the fixed-point form
    (cpprob/detail/fixed_mass.hpp: FixedCdf::first_stratified; csrc/step_fixed.hpp: multinomial_bin_kernel + binned_walk, and under
    CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL fixed_multinomial_ancestors) or the prefix counts of the table form (stratified, 3-state HMM,
    every-step schedule); oracle: orc_resample_fixed_stratified / _multinomial_binned / _multinomial, orc_resample_table_stratified.
    Ancestors, decisions and the HMM's states equal the oracle's at one, two and three levels of the mass hierarchy, on every-step
    and ESS-triggered schedules."""
    obs = _obs(golden_dir, key)[:T]
    resampler = cp.RESAMPLE_STRATIFIED if kind == "stratified" else cp.RESAMPLE_MULTINOMIAL
    engine.begin(cp.ALG_SMC, model, obs, n, seed=13, resampler=resampler, ess_threshold=ess, flags=cp.capi.FLAG_MULTINOMIAL_LITERAL if kind == "multinomial_literal" else 0)
    engine.run()
    s = engine.summary()
    # (the three-state HMM on an every-step schedule: stratified and strata-form multinomial resampling run on integer prefix COUNTS
    #  like systematic)
    counts = model == cp.MODEL_HMM3 and ess > 1.0 and kind in ("stratified", "multinomial")
    assert s["step_form"] == (cp.capi.FORM_COUNTS if counts else cp.capi.FORM_FIXED)
    ref = O.smc(model, obs, n, 13, O.RESAMPLE_MULTINOMIAL_LITERAL if kind == "multinomial_literal" else resampler, ess)
    anc, vals = engine.ancestors(), engine.values()
    gess, gres = engine.step_trace()
    assert np.array_equal(gres, ref["resampled"])
    assert np.array_equal(anc, ref["anc"]), [int(np.count_nonzero(anc[t] != ref["anc"][t])) for t in range(T)]
    if kind == "multinomial":                                   # strata form: an output's ancestor sits near it (thresholds sorted stratum by stratum)
        assert all(np.max(np.maximum.accumulate(anc[t]) - anc[t]) < 8192 for t in range(1, T) if gres[t - 1])
    if model == cp.MODEL_HMM3:
        assert np.array_equal(vals, ref["hist"])
    else:
        np.testing.assert_allclose(vals, ref["hist"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gess, ref["ess"], rtol=1e-9)
    assert abs(s["log_evidence"] - ref["log_z"]) < 1e-9
    if resampler == cp.RESAMPLE_STRATIFIED:
        assert all(np.all(np.diff(anc[t]) >= 0) for t in range(1, T))
    if counts:
        np.testing.assert_allclose(engine.stats(), O.smoothing(ref["hist"], ref["anc"], ref["logw"]), atol=1e-12)
    else:
        sm_self = O.smoothing_linear(vals, anc, O.fix_weights(engine.logw(), s["max_logw"]).astype(np.float64))
        np.testing.assert_allclose(engine.stats(), sm_self, rtol=1e-10, atol=1e-12)


def test_multinomial_strata_counts_beyond_16384_tiles(engine, golden_dir):
    """2 10^7 particles = 19 532 tiles = 2^15 strata: the bottom part of the strata tree runs nine levels per level-6 node (more nodes
    than a workgroup has threads: the thread-per-node form of strata_split), the step's stratum window and the three-level mass
    hierarchy at their largest tested size -- ancestors and states equal the oracle's."""
    obs = _obs(golden_dir, "hmm16")[:3]
    n = 20_000_000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=29, resampler=cp.RESAMPLE_MULTINOMIAL, ess_threshold=2.0)
    engine.run()
    ref = O.smc(cp.MODEL_HMM3, obs, n, 29, cp.RESAMPLE_MULTINOMIAL, 2.0)
    assert np.array_equal(engine.ancestors(), ref["anc"]) and np.array_equal(engine.values(), ref["hist"])
    assert abs(engine.summary()["log_evidence"] - ref["log_z"]) < 1e-9


@pytest.mark.parametrize("model,key,T", [(cp.MODEL_HMM3, "hmm128", 40), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 30)])
@pytest.mark.parametrize("resampler", [cp.RESAMPLE_SYSTEMATIC, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL])
def test_paired_step_launches_are_the_single_launch_bit_for_bit(engine, golden_dir, model, key, T, resampler):
    """CPPROB_HIP_FLAG_PAIRED_STEP_LAUNCH (an A/B form): on ESS-triggered schedules a step as two launches -- the carry form (no
    search, walk or gather in it) and the resampling form -- each of which ends at once when the step is the other's (csrc/step_fixed.hpp:
    smc_step_fixed_carry_body).  Against the single launch: the same decisions, ancestors, values, weights, evidence and posterior,
    and the oracle's."""
    obs = _obs(golden_dir, key)[:T]
    n = 70_001
    out = {}
    for name, flags in (("paired", cp.capi.FLAG_PAIRED_STEP_LAUNCH), ("single", 0)):
        engine.begin(cp.ALG_SMC, model, obs, n, seed=17, resampler=resampler, ess_threshold=0.5, flags=flags)
        engine.run()
        out[name] = (engine.ancestors(), engine.values(), engine.logw(), engine.summary(), engine.stats().copy(), engine.step_trace())
    a, b = out["paired"], out["single"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    assert a[3]["log_evidence"] == b[3]["log_evidence"] and a[3]["n_resampled"] == b[3]["n_resampled"] and 0 < a[3]["n_resampled"] < T - 1
    assert np.array_equal(a[5][0], b[5][0]) and np.array_equal(a[5][1], b[5][1])
    ref = O.smc(model, obs, n, 17, resampler, 0.5)
    assert np.array_equal(a[0], ref["anc"]) and np.array_equal(a[5][1], ref["resampled"])


@pytest.mark.parametrize("alg,model,key,T,ess", [(cp.ALG_SMC, cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 30, 0.5),
                                                 (cp.ALG_SIS, cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 12, 0.5)])
def test_results_in_one_call_equal_the_three_read_backs(engine, golden_dir, alg, model, key, T, ess):
    """cpprob_hip_infer_results (summary + statistics + step trace through pinned memory behind one stream synchronisation; a
    fixed-point run is settled on the same read-back) returns what cpprob_hip_infer_summary / _stats / _step_trace return."""
    obs = _obs(golden_dir, key)[:T]
    engine.begin(alg, model, obs, 50_001, seed=5, ess_threshold=ess)
    engine.run()
    s1, st1, ess1, res1 = engine.results()
    s2, st2, (ess2, res2) = engine.summary(), engine.stats(), engine.step_trace()
    assert s1 == s2 and np.array_equal(st1, st2) and np.array_equal(ess1, ess2) and np.array_equal(res1, res2)
    engine.run()                                                         # (the same run again starts its books over: SIS keeps them once, at the last observe)
    s3, st3, ess3, res3 = engine.results()
    assert s3 == s1 and np.array_equal(st3, st1) and np.array_equal(ess3, ess1) and np.array_equal(res3, res1)


@pytest.mark.parametrize("n", [1, 2, 1023, 1025, 4095, 4097, 12289])
def test_smc_tiny_and_ragged_populations(engine, golden_dir, n):
    obs = _obs(golden_dir, "hmm16")[:5]
    _compare_smc(engine, cp.MODEL_HMM3, obs, n, 3, cp.RESAMPLE_SYSTEMATIC, 2.0)          # (one and two particles included: ancestors, states, evidence)
    for rs in (cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL):                        # ... and the other two resamplers, both integer forms
        _compare_smc(engine, cp.MODEL_HMM3, obs, n, 3, rs, 2.0)
        _compare_smc(engine, cp.MODEL_LINEAR_GAUSSIAN_1D, _obs(golden_dir, "lgssm100")[:5], n, 3, rs, 0.5)
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=3)
    engine.run()
    st = engine.stats()
    np.testing.assert_allclose(st.sum(axis=1), 1.0, rtol=1e-12)


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("n", [777, 70_000])
def test_smc_short_traces_read_out_from_the_final_counts(engine, golden_dir, T, n):
    """T = 1 (the only step is the last one), 2 and 3: the count form's read-out takes weights and the normaliser from the final
    generation's counts; log-weights exist on request only (cpprob_hip_copy_logw), and asking for the traces first must not
    book the final generation twice."""
    obs = _obs(golden_dir, "hmm16")[:T]
    for rep in range(2):                                    # (the hierarchy's rotation carries over from run to run)
        engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=21 + rep, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0)
        engine.run()
        ref = O.smc(cp.MODEL_HMM3, obs, n, 21 + rep, cp.RESAMPLE_SYSTEMATIC, 2.0)
        paths = engine.paths()
        assert np.array_equal(engine.ancestors(), ref["anc"]) and np.array_equal(engine.values(), ref["hist"])
        assert np.array_equal(paths, np.take_along_axis(ref["hist"], O.lineage(ref["anc"]), axis=1))
        s = engine.summary()
        assert abs(s["log_evidence"] - ref["log_z"]) < 1e-10 and s["n_resampled"] == T - 1
        np.testing.assert_allclose(engine.logw(), ref["logw"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(engine.stats(), O.smoothing(ref["hist"], ref["anc"], ref["logw"]), atol=1e-12)
        engine.run(1)                                        # a second run on the same context, then the first again: same numbers
        engine.run(0)
        assert abs(engine.summary()["log_evidence"] - s["log_evidence"]) == 0.0


def _hmm_filter_from_states(hist, obs):
    """P(x_t = s | y_0..t) estimated by a generation that entered step t equally weighted: sum_i [x = s] e_s / sum_i e_x."""
    means = np.array([-1.0, 0.0, 1.0])
    out = np.zeros((len(obs), 3))
    for t, y in enumerate(obs):
        e = np.exp(-0.5 * (y - means) ** 2)
        cnt = np.bincount(hist[t], minlength=3).astype(np.float64)
        out[t] = cnt * e / (cnt * e).sum()
    return out


@pytest.mark.parametrize("n", [5_000, 70_000, 1_000_000])
def test_filtering_only_run_count_form(engine, golden_dir, n):
    """keep_history = 0: two rows of values, no ancestors; predict hit t's statistics are generation t's under its own weights.
    Count form (hmm, every-step schedule): the same particles as the history-keeping run, so the filtering marginals follow from
    the oracle's states exactly, the evidence is the same number, and the exact forward filter is within Monte-Carlo error."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    obs = z["hmm16"]
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=11, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0)
    engine.run()
    keep = engine.summary()
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=11, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0, keep_history=False)
    for rep in range(2):
        engine.run()
        s = engine.summary()
        assert s["log_evidence"] == keep["log_evidence"] and s["n_resampled"] == 15 and s["ess_final"] == keep["ess_final"]
        st = engine.stats()
        if n <= 70_000:
            ref = O.smc(cp.MODEL_HMM3, obs, n, 11, cp.RESAMPLE_SYSTEMATIC, 2.0)
            np.testing.assert_allclose(st, _hmm_filter_from_states(ref["hist"], obs), rtol=0, atol=1e-13)
            np.testing.assert_allclose(engine.logw(), ref["logw"], rtol=1e-12, atol=1e-12)
        assert np.abs(st - z["hmm16_filter"]).max() < 4.0 / np.sqrt(n)
    for getter in (engine.values, engine.ancestors, engine.paths):
        with pytest.raises(cp.CpprobHipError):
            getter()
    # the context goes back to keeping history without residue
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=11, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0)
    engine.run()
    assert engine.summary()["log_evidence"] == keep["log_evidence"] and engine.ancestors().shape == (16, n)


@pytest.mark.parametrize("model,key,ess,resampler", [(cp.MODEL_HMM3, "hmm16", 0.5, cp.RESAMPLE_SYSTEMATIC),
                                                       (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 0.5, cp.RESAMPLE_SYSTEMATIC),
                                                       (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 2.0, cp.RESAMPLE_STRATIFIED),
                                                       (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 2.0, cp.RESAMPLE_MULTINOMIAL)])
def test_filtering_only_run_weight_sums_form(engine, golden_dir, model, key, ess, resampler):
    """The same for continuous weights / ESS-triggered schedules: every step leaves its own weighted sums.  Same particles as the
    history-keeping run: the last predict hit's statistics agree (smoothing and filtering coincide there), the evidence is the
    same number, and the exact filter (forward algorithm / Kalman) is within Monte-Carlo error at every step."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    obs = z[key][:30]
    n = 300_000
    engine.begin(cp.ALG_SMC, model, obs, n, seed=5, resampler=resampler, ess_threshold=ess)
    engine.run()
    keep, keep_stats = engine.summary(), engine.stats().copy()
    engine.begin(cp.ALG_SMC, model, obs, n, seed=5, resampler=resampler, ess_threshold=ess, keep_history=False)
    engine.run()
    s, st = engine.summary(), engine.stats()
    assert s["log_evidence"] == keep["log_evidence"] and s["n_resampled"] == keep["n_resampled"]
    np.testing.assert_allclose(st[-1], keep_stats[-1], rtol=1e-9, atol=1e-12)
    # the oracle's filtering statistics of the same particles (every resampler runs on integer masses: the very same particles)
    ref = O.smc(model, obs, n, 5, resampler, ess)
    np.testing.assert_allclose(st, ref["filter"], rtol=0, atol=1e-9)
    if model == cp.MODEL_HMM3:
        assert np.abs(st - z["hmm16_filter"]).max() < 2e-2
    else:
        assert np.abs(st[:, 0] - z["lgssm100_filter_mean"][:30]).max() < 2e-2
        assert np.abs(st[:, 1] - z["lgssm100_filter_var"][:30]).max() < 2e-2
    with pytest.raises(cp.CpprobHipError):
        engine.paths()


@pytest.mark.parametrize("model,key,T,ess", [(cp.MODEL_HMM3, "hmm16", 16, 2.0), (cp.MODEL_HMM3, "hmm128", 48, 0.5), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 30, 0.5)])
def test_filtering_only_shards_of_a_joint_population_match_one_context(engine, golden_dir, model, key, T, ess):
    """keep_history = 0 for shards in the exchange scope: a migrating particle is its current state alone (records of one value,
    an annex that starts over every step), each rank holds two rows of its shard -- BASELINE configs[4] filtering-only fits 0.5 GB
    per GPU.  Four uneven shards on this GPU against ONE filtering-only context: the same evidence and decisions bit for bit, the
    same filtering statistics (prefix-count form: the very numbers; fixed-point form: the ranks' sums over the ranks' masses)."""
    obs = _obs(golden_dir, key)[:T]
    shards = [30000, 50001, 19999, 40000]
    n = int(sum(shards))
    engine.begin(cp.ALG_SMC, model, obs, n, seed=21, ess_threshold=ess, keep_history=False)
    engine.run()
    ref_stats, ref_sum = engine.stats().copy(), engine.summary()
    g = cp.Group([0] * len(shards))
    g.begin(cp.ALG_SMC, model, obs, n, seed=21, ess_threshold=ess, shard_sizes=shards, keep_history=False)
    g.run()
    stats, s, reruns = g.results()
    tr = g.traffic()
    e0 = g.context(0)
    e0.n = shards[0]
    with pytest.raises(cp.CpprobHipError):
        e0.paths()
    g.close()
    assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"] and reruns == 0
    if model == cp.MODEL_HMM3 and ess > 1.0:
        assert np.array_equal(stats, ref_stats)
    else:
        np.testing.assert_allclose(stats, ref_stats, rtol=1e-12, atol=1e-13)
    vsz = 1 if model == cp.MODEL_HMM3 else 8
    assert tr["records"] > 0 and tr["wire_bytes"] == tr["payload_bytes"] == tr["records"] * vsz        # one value per migrant
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SMC, model, obs, 1000, n_global=2000, scope=cp.SCOPE_GLOBAL, keep_history=False)      # a locally resampled shard keeps its history


@pytest.mark.parametrize("resampler", [cp.RESAMPLE_SYSTEMATIC, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_MULTINOMIAL])
def test_fixed_point_run_that_loses_its_bits_is_repaired_from_the_offending_generation(engine, golden_dir, resampler):
    """The fixed-point weights are taken against a reference known before the generation exists (the emission's density at its
    mode).  An observation many standard deviations from EVERY particle leaves the heaviest particle far below it -- 0.69 nats per
    lost bit, nothing left at 22 -- so every generation's gap is tracked and, past 6 nats, the OFFENDING generation is weighed again
    against its exact maximum (log-weights recomputed from the particle store, two order-free passes), the books are rewound to where
    they stood before it, and the steps behind it run again: the run stays in integers (no rerun of the whole run, no floating-point
    CDF), says so (n_requantised), and still equals the oracle -- which states the same rule -- ancestor for ancestor.  A run without
    such a generation repairs nothing.  CPPROB_HIP_FLAG_REPEAT_IN_FLOATING_POINT keeps the r03 / r04 behaviour: the whole run again
    in the floating-point form, bit for bit CPPROB_HIP_FLAG_FLOATING_POINT_STEP's numbers."""
    obs = np.array(_obs(golden_dir, "lgssm100")[:14])
    obs[6] = 40.0                                              # ~30 standard deviations from every particle
    n = 5000
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, resampler=resampler, ess_threshold=0.5)
    engine.run()
    st, s = engine.stats().copy(), engine.summary()
    assert s["step_form"] == cp.capi.FORM_FIXED and s["n_requantised"] >= 1
    ref = O.smc(cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, 8, resampler, 0.5)
    anc, vals = engine.ancestors(), engine.values()
    gess, gres = engine.step_trace()
    assert np.array_equal(gres, ref["resampled"]) and np.array_equal(anc, ref["anc"])
    np.testing.assert_allclose(vals, ref["hist"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gess, ref["ess"], rtol=1e-9)
    assert abs(s["log_evidence"] - ref["log_z"]) < 1e-9 and s["n_resampled"] == int(ref["resampled"].sum())
    sm_self = O.smoothing_linear(vals, anc, O.fix_weights(engine.logw(), s["max_logw"]).astype(np.float64))
    np.testing.assert_allclose(st, sm_self, rtol=1e-10, atol=1e-12)
    engine.run(1)                                               # the context is as good as new behind a repair
    engine.run(0)
    assert np.array_equal(engine.stats(), st) and engine.summary() == s
    # a last generation that trips (nothing behind it to run again) and a first one
    for at in (13, 0):
        o2 = np.array(_obs(golden_dir, "lgssm100")[:14])
        o2[at] = 35.0 if at else 9.0
        engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, o2, n, seed=8, resampler=resampler, ess_threshold=0.5)
        engine.run()
        s2 = engine.summary()
        r2 = O.smc(cp.MODEL_LINEAR_GAUSSIAN_1D, o2, n, 8, resampler, 0.5)
        assert s2["n_requantised"] >= 1 and np.array_equal(engine.ancestors(), r2["anc"]) and abs(s2["log_evidence"] - r2["log_z"]) < 1e-9, at
    if resampler != cp.RESAMPLE_SYSTEMATIC:
        return
    # no such generation: nothing repaired
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, _obs(golden_dir, "lgssm100")[:14], n, seed=8, ess_threshold=0.5)
    engine.run()
    assert engine.summary()["n_requantised"] == 0
    # the whole run again in the floating-point form, on request
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, ess_threshold=0.5, flags=cp.capi.FLAG_REPEAT_IN_FLOATING_POINT)
    engine.run()
    stf, sf = engine.stats().copy(), engine.summary()
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, ess_threshold=0.5, flags=cp.capi.FLAG_FLOATING_POINT_STEP)
    engine.run()
    s2 = engine.summary()
    assert np.array_equal(engine.stats(), stf) and sf["step_form"] == s2["step_form"] == cp.capi.FORM_FLOAT and s2 == sf
    np.testing.assert_allclose(stf, O.smoothing(engine.values(), engine.ancestors(), engine.logw()), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(stf, st, rtol=0, atol=5e-3)      # the two repairs agree as two arithmetic forms of one run do
    # the group driver (two loopback ranks) repairs in integers too -- the one-context run's evidence, bit for bit (tests/test_gpu_group.py:
    # test_group_run_that_loses_its_bits_...) -- and repeats in the floating-point form only when told to
    g = cp.Group([0, 0])
    g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, ess_threshold=0.5)
    g.run()
    gst, gs, reruns = g.results()
    assert gs["log_evidence"] == s["log_evidence"] and gs["n_requantised"] == s["n_requantised"] and gs["step_form"] == cp.capi.FORM_FIXED
    g.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=8, ess_threshold=0.5, flags=cp.capi.FLAG_REPEAT_IN_FLOATING_POINT)
    g.run()
    gst, gs, reruns = g.results()
    g.close()
    assert 1 <= reruns <= 4 and gs["step_form"] == cp.capi.FORM_FLOAT and abs(gs["log_evidence"] - sf["log_evidence"]) < 1e-9      # (the mass sits on a few particles: the transport may be enlarged too)
    np.testing.assert_allclose(gst, stf, rtol=0, atol=5e-3)


@pytest.mark.parametrize("workload", ["configs[3]: linear_gaussian_1d<100>, 10^7 particles", "configs[4] per-GPU shard: hmm<128>, 1.25 10^7 particles"])
def test_fixed_point_masses_at_full_size_arithmetic_bound_and_posterior(engine, golden_dir, workload):
    """The fixed-point form's arithmetic is narrower than the reference's `double log_w_` (include/cpprob/trace.hpp:59): a weight is
    an integer multiple of 2^-32 of exp(R_t).  What that costs, where BASELINE.json quotes the two multi-GPU configs:

    (1) THE ARITHMETIC, generation by generation (runs cut after 1, 7, 40 and all observes: the same particles as the full run's).
        Every weight is off by at most half a unit, so a generation's mass S (in units) is off by at most N / 2 and its logarithm --
        a factor of the evidence, the normaliser of every posterior sum -- by at most N / (2 S); measured against an fp64
        recomputation from the stored log-weights.  ESS: a weight's square is floor(floor(q / 2^8)^2 / 2^16), which under-counts
        a term by at most 2^9 / q + 2^32 / q^2: the integer ESS is never below the fp64 one and at most 1e-4 above it.
    (2) THE POSTERIOR.  Fixed-point against CPPROB_HIP_FLAG_FLOATING_POINT_STEP (fp64 weights, fp64 CDF) on the same seed is NOT a
        clean measure of (1): the floating-point form's CDF-boundary flips make the two runs different genealogies after a few
        steps, i.e. different Monte-Carlo realisations.  So their difference is compared with what it would have to exceed to mean
        anything -- the difference between two SEEDS of the fixed-point form -- and each form with the exact posterior (Kalman + RTS /
        forward-backward).  (north_star's 1e-3 is stated for the Gaussian SIS config at 10^7; the Monte-Carlo error of a T = 100 / 128
        smoothing trace at these sizes is several times that, in either arithmetic.)"""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    if workload.startswith("configs[3]"):
        model, obs, n, exact, tol = cp.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"], 10_000_000, np.stack([z["lgssm100_smooth_mean"], z["lgssm100_smooth_var"]], 1), 1.2e-2
    else:
        model, obs, n, exact, tol = cp.MODEL_HMM3, z["hmm128"], 12_500_000, z["hmm128_smooth"], 5e-3
    # (1)
    for Tt in (1, 7, 40, len(obs)):
        engine.begin(cp.ALG_SMC, model, obs[:Tt], n, seed=31, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=0.5)
        engine.run()
        s = engine.summary()
        assert s["step_form"] == cp.capi.FORM_FIXED
        lw = engine.logw()
        m = lw.max()
        w = np.exp(lw - m)
        lse, ess = m + np.log(w.sum()), w.sum() ** 2 / (w * w).sum()
        units = np.exp(s["log_norm"] - s["max_logw"]) * 2.0 ** 32                 # the generation's mass in units of 2^-32 exp(R)
        assert abs(s["log_norm"] - lse) <= n / (2.0 * units) + 1e-12, (Tt, s["log_norm"] - lse, n / (2.0 * units))
        assert 0.0 <= s["ess_final"] / ess - 1.0 <= 1e-4, (Tt, s["ess_final"], ess)
    # (2)
    out = {}
    for name, flags, seed in (("fixed", 0, 31), ("float", cp.capi.FLAG_FLOATING_POINT_STEP, 31), ("fixed, another seed", 0, 32)):
        engine.begin(cp.ALG_SMC, model, obs, n, seed=seed, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=0.5, flags=flags)
        engine.run()
        s = engine.summary()
        out[name] = (engine.stats().copy(), s["log_evidence"], s["step_form"], s["n_resampled"])
    assert out["fixed"][2] == cp.capi.FORM_FIXED and out["float"][2] == cp.capi.FORM_FLOAT and out["fixed"][3] == out["float"][3] > 0
    d_forms = float(np.abs(out["fixed"][0] - out["float"][0]).max())
    d_seeds = float(np.abs(out["fixed"][0] - out["fixed, another seed"][0]).max())
    z_forms, z_seeds = abs(out["fixed"][1] - out["float"][1]), abs(out["fixed"][1] - out["fixed, another seed"][1])
    assert d_forms <= 2.0 * d_seeds + 1e-3 and z_forms <= 2.0 * z_seeds + 2e-3, (d_forms, d_seeds, z_forms, z_seeds)
    for name in out:
        assert float(np.abs(out[name][0] - exact).max()) < tol, name
        assert float(np.abs(out[name][0] - exact)[-10:].max()) < tol / 3, name       # the last ten hits: hardly any path degeneracy
    print("%s: max |fixed - float| %.2e (two seeds of the fixed form: %.2e), log evidence %.2e (%.2e)" % (workload, d_forms, d_seeds, z_forms, z_seeds))


def test_smc_hmm16_config3_vs_forward_backward(engine, golden_dir):
    """BASELINE.json configs[2]: hmm<16>, systematic resampling every step, 10^6 particles."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    n = 1_000_000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm16"], n, seed=12345, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=2.0)
    engine.run()
    st = engine.stats()
    s = engine.summary()
    assert s["n_resampled"] == 15
    # filtering-quality at t = T-1, smoothing degrades towards t = 0 through path degeneracy
    assert np.abs(st[-1] - z["hmm16_smooth"][-1]).max() < 3e-3
    assert np.abs(st - z["hmm16_smooth"]).max() < 2e-2
    assert abs(s["log_evidence"] - float(z["hmm16_logz"])) < 5e-3
    np.testing.assert_allclose(st.sum(axis=1), 1.0, rtol=1e-12)
    anc = engine.ancestors()
    assert np.all(np.diff(anc[1:], axis=1) >= 0)      # systematic ancestors are sorted
    assert np.array_equal(anc[0], np.arange(n))


def test_smc_lgssm_vs_kalman(engine, golden_dir):
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    n = 1_000_000
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"], n, seed=12345, ess_threshold=0.5)
    engine.run()
    st = engine.stats()
    s = engine.summary()
    assert abs(st[-1, 0] - z["lgssm100_smooth_mean"][-1]) < 5e-3
    assert abs(st[-1, 1] - z["lgssm100_smooth_var"][-1]) < 5e-3
    assert abs(s["log_evidence"] - float(z["lgssm100_logz"])) < 2e-2
    # last 10 steps are barely degenerate
    assert np.abs(st[-10:, 0] - z["lgssm100_smooth_mean"][-10:]).max() < 3e-2


def test_smc_hmm128_config5_per_gpu_size_vs_forward_backward(engine, golden_dir):
    """BASELINE.json configs[4] at its per-GPU size: hmm<128>, FULL length T = 128, resampling when ESS < N/2, 1.25e7 particles
    (10^8 over 8 GPUs), against the exact forward-backward posterior and evidence."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    n = 12_500_000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm128"], n, seed=12345, resampler=cp.RESAMPLE_SYSTEMATIC, ess_threshold=0.5)
    engine.run()
    st, s = engine.stats(), engine.summary()
    ess, res = engine.step_trace()
    assert len(z["hmm128"]) == 128 and 0 < s["n_resampled"] < 127 and int(res.sum()) == s["n_resampled"]
    assert np.all(ess[res == 1] < 0.5 * n) and np.all(ess[:-1][res[:-1] == 0] >= 0.5 * n)      # the trigger, thesis p.37
    assert abs(s["log_evidence"] - float(z["hmm128_logz"])) < 5e-3
    np.testing.assert_allclose(st.sum(axis=1), 1.0, rtol=1e-12)
    assert np.abs(st[-16:] - z["hmm128_smooth"][-16:]).max() < 3e-3       # the recent past is barely degenerate
    assert np.abs(st - z["hmm128_smooth"]).max() < 5e-2                    # early marginals rest on few surviving lineages


@pytest.mark.parametrize("n", [1_250_000, 10_000_000])
def test_smc_lgssm_config4_sizes_vs_kalman(engine, golden_dir, n):
    """BASELINE.json configs[3]: linear_gaussian_1d<100>, T = 100, at its per-GPU size (1.25e6 = 10^7 / 8) and at the whole
    10^7 on one GPU, against the Kalman filter / RTS smoother."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"], n, seed=12345, ess_threshold=0.5)
    engine.run()
    st, s = engine.stats(), engine.summary()
    tol = 5e-3 if n < 5_000_000 else 2.5e-3
    assert abs(st[-1, 0] - z["lgssm100_smooth_mean"][-1]) < tol and abs(st[-1, 1] - z["lgssm100_smooth_var"][-1]) < tol
    assert abs(s["log_evidence"] - float(z["lgssm100_logz"])) < (5e-2 if n < 5_000_000 else 2e-2)        # sd of the estimator ~ 1.5e-2 at 1.25e6
    assert np.abs(st[-10:, 0] - z["lgssm100_smooth_mean"][-10:]).max() < 3e-2
    assert 0 < s["n_resampled"] < 99


def test_run_index_decorrelates_and_is_reproducible(engine, golden_dir):
    obs = _obs(golden_dir, "hmm16")
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, 50000, seed=1)
    engine.run(0); a = engine.stats().copy()
    engine.run(1); b = engine.stats().copy()
    engine.run(0); c = engine.stats().copy()
    assert np.array_equal(a, c)          # bitwise reproducible
    assert not np.array_equal(a, b)


def test_sharded_ids_reproduce_single_shard(engine):
    """Particle ids are global: shard [off, off+n) of an SIS population equals that slice."""
    obs = [3.0, 4.0]
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, obs, 10000, seed=5)
    engine.run()
    full = engine.values()[0].copy()
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, obs, 2500, seed=5, particle_offset=5000, n_global=10000, scope=cp.SCOPE_ISLAND)
    engine.run()
    assert np.array_equal(engine.values()[0], full[5000:7500])
    # a shard of a JOINT population must go through the step protocol
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, obs, 2500, seed=5, particle_offset=5000, n_global=10000)
    with pytest.raises(cp.CpprobHipError):
        engine.run()


def test_error_paths(engine):
    with pytest.raises(cp.CpprobHipError):
        engine.begin(0, cp.MODEL_HMM3, [0.0], 10)                       # StateType::compile: unsupported
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, [1.0, 2.0, 3.0], 10)
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SIS, 17, [1.0], 10)
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SIS, cp.MODEL_HMM3, [1.0], 0)
    engine.begin(cp.ALG_SIS, cp.MODEL_HMM3, [1.0], 10)
    with pytest.raises(cp.CpprobHipError):
        engine.stats()                                                  # no finished run
    engine.run()
    with pytest.raises(cp.CpprobHipError):
        engine.ancestors()                                              # SIS keeps none


# ---- joint population sharded over several contexts (virtual ranks on one GPU) -----------------------

class _InProcessCollective:
    """Test double for RCCL: the shards live in contexts of ONE process on one GPU; the all-gather is a
    concatenation.  (The product's TorchCollective is exercised with world = 1 below.)"""

    def __init__(self, engines, rank):
        self.engines, self.rank, self.world = engines, rank, len(engines)


def _run_joint_virtual(model, alg, obs, n_per, world, seed, ess):
    import torch
    from cpprob_amd import distributed as D
    engines = [cp.Engine(0) for _ in range(world)]
    for r, e in enumerate(engines):
        e.begin(alg, model, obs, n_per, seed=seed, ess_threshold=ess, particle_offset=r * n_per, n_global=world * n_per, scope=cp.SCOPE_GLOBAL)
    T, K = engines[0].T, engines[0].K
    locals_ = [dzeros(4, dtype=torch.float64) for _ in range(world)]
    allt = dzeros(3 * world, dtype=torch.float64)
    steps = [T - 1] if alg == cp.ALG_SIS else range(T)
    for t in steps:
        for r, e in enumerate(engines):
            e.step_begin(t, locals_[r])
        for e in engines:
            e.sync()
        allt.copy_(torch.cat([l[:3] for l in locals_]))
        torch.cuda.synchronize()
        for r, e in enumerate(engines):
            e.step_end(t, allt, world, r)
    raw = np.zeros((T, K))
    summ = None
    for e in engines:
        e.finish()
        raw += e.stats()
        s = e.summary()
        if summ is not None:
            assert s["log_evidence"] == summ["log_evidence"] and s["log_norm"] == summ["log_norm"]     # identical on every rank
        summ = s
    stats = D.normalise_joint_stats(raw, summ["log_norm"], summ["max_logw"], engines[0].is_int)
    traces = [e.step_trace() for e in engines]
    for e in engines:
        e.close()
    return stats, summ, traces


def _run_exchange_virtual(model, obs, n_pers, seed, ess, flags=0, repair=False):
    """EXCHANGE scope over virtual ranks: plan / pack / (in-process all-to-all) / commit.  Returns the joint stats, the
    summary, every shard's materialised traces [T, n_r] and the number of migrated lineage records per step."""
    import torch
    from cpprob_amd import distributed as D
    world = len(n_pers)
    begins = np.concatenate([[0], np.cumsum(n_pers)]).astype(np.uint64)
    engines = [cp.Engine(0) for _ in range(world)]
    for r, e in enumerate(engines):
        e.begin(cp.ALG_SMC, model, obs, n_pers[r], seed=seed, ess_threshold=ess, particle_offset=int(begins[r]), n_global=int(begins[-1]), scope=cp.SCOPE_EXCHANGE, flags=flags)
    T, K = engines[0].T, engines[0].K
    vdt = torch.int32 if engines[0].is_int else torch.float64
    locals_ = [dzeros(4, dtype=torch.float64) for _ in range(world)]
    allt = dzeros(3 * world, dtype=torch.float64)
    moved = []

    def steps(t_from, resumed):
        # steps t_from .. T - 1; resumed: generation t_from exists already (a repaired generation, its totals in locals_)
        for t in range(t_from, T):
            if not (resumed and t == t_from):
                for r, e in enumerate(engines):
                    e.step_begin(t, locals_[r])
            for e in engines:
                e.sync()
            allt.copy_(torch.cat([l[:3] for l in locals_]))
            torch.cuda.synchronize()
            for r, e in enumerate(engines):
                e.step_end(t, allt, world, r)
            if t + 1 == T:
                break
            plans = [e.exchange_plan(t, world, r, begins) for r, e in enumerate(engines)]
            decisions = {p[0] for p in plans}
            assert len(decisions) == 1                                              # same decision everywhere
            for a in range(world):
                for b in range(world):
                    assert plans[a][1][b] == plans[b][2][a]                         # what a sends to b is what b expects from a
            width = t + 1
            sends = []
            for r, e in enumerate(engines):
                buf = torch.empty(max(int(plans[r][1].sum()), 1) * width, dtype=vdt, device="cuda")
                e.exchange_pack(t, buf)
                e.sync()
                sends.append(buf)
            for dst, e in enumerate(engines):
                parts = []
                for src in range(world):
                    off = int(plans[src][1][:dst].sum()) * width
                    parts.append(sends[src][off: off + int(plans[src][1][dst]) * width])
                recv = torch.cat(parts) if parts else torch.empty(0, dtype=vdt, device="cuda")
                assert recv.numel() == int(plans[dst][2].sum()) * width
                torch.cuda.synchronize()
                e.exchange_commit(t, recv if recv.numel() else None)
                e.sync()
            moved.append(int(sum(p[1].sum() for p in plans)))
        for e in engines:
            e.finish()

    steps(0, False)
    last = -1
    for _ in range(T + 1 if repair else 0):
        # a generation that lost its fixed-point bits: every rank names the same one (the books come from the all-gathered totals)
        bad = [e.first_bad_generation()[0] for e in engines]
        assert len(set(bad)) == 1
        g = bad[0]
        if g < 0 or g <= last:
            break
        del moved[g:]
        for r, e in enumerate(engines):
            e.repair_begin(g, locals_[r])
        for e in engines:
            e.sync()
        allt.copy_(torch.cat([l[:3] for l in locals_]))
        torch.cuda.synchronize()
        for r, e in enumerate(engines):
            e.repair_end(g, allt, world, r, locals_[r])
        steps(g, True)
        last = g
    raw = np.zeros((T, K))
    summ = None
    for e in engines:
        raw += e.stats()
        s = e.summary()
        if summ is not None:
            assert s["log_evidence"] == summ["log_evidence"] and s["log_norm"] == summ["log_norm"]
        summ = s
    stats = D.normalise_joint_stats(raw, summ["log_norm"], summ["max_logw"], engines[0].is_int)
    paths = [e.paths() for e in engines]
    logw = [e.logw() for e in engines]
    for e in engines:
        e.close()
    return stats, summ, paths, logw, moved


@pytest.mark.parametrize("n_pers", [[50000, 50000], [30000, 50001, 19999], [1000, 2000, 70000, 3000]])
@pytest.mark.parametrize("ess", [2.0, 0.5])
def test_exchange_scope_draws_the_single_gpu_ancestors(engine, golden_dir, n_pers, ess):
    """SURVEY 8(e) default: exact global resampling over shards.  Every shard's traces must be the corresponding slice
    of the traces ONE context with all particles produces (same seed): same variates (global ids), same ancestors (one
    shared systematic offset, offspring ranges that meet without gap or overlap), lineages migrated intact."""
    obs = _obs(golden_dir, "hmm16")
    n = int(sum(n_pers))
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=31, ess_threshold=ess)
    engine.run()
    ref_paths, ref_stats, ref_sum, ref_logw = engine.paths(), engine.stats().copy(), engine.summary(), engine.logw()
    stats, s, paths, logw, moved = _run_exchange_virtual(cp.MODEL_HMM3, obs, n_pers, 31, ess)
    got = np.concatenate(paths, axis=1)
    # a 1-ulp difference between the two CDF evaluations can move one offspring across a boundary (~1e-3 per run)
    assert (got != ref_paths).any(axis=0).sum() <= 2
    np.testing.assert_allclose(np.concatenate(logw), ref_logw, rtol=0, atol=1e-12)
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-4 if (got != ref_paths).any() else 1e-12)
    assert abs(s["log_evidence"] - ref_sum["log_evidence"]) < 1e-12 and s["n_resampled"] == ref_sum["n_resampled"]
    assert sum(moved) > 0                                                       # particles did migrate


def test_exchange_scope_survives_extreme_imbalance_and_grows_the_annex(engine, golden_dir):
    """An outlying first observation puts nearly all the weight on a few particles: whole shards are repopulated from
    another rank (more immigrants than the initial annex holds), real-valued lineages."""
    obs = np.array(_obs(golden_dir, "lgssm100")[:12])
    obs[0] = 7.5
    n_pers = [40000, 40000, 40000]
    n = sum(n_pers)
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=5, ess_threshold=0.5)
    engine.run()
    ref_paths, ref_stats, ref_sum = engine.paths(), engine.stats().copy(), engine.summary()
    stats, s, paths, _, moved = _run_exchange_virtual(cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n_pers, 5, 0.5)
    got = np.concatenate(paths, axis=1)
    assert (got != ref_paths).any(axis=0).sum() <= 2
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=1e-4 if (got != ref_paths).any() else 1e-11)
    assert abs(s["log_evidence"] - ref_sum["log_evidence"]) < 1e-11
    assert max(moved) > 40000 // 16 + 4096                                      # beyond the initial annex: it grew


def test_caller_driven_run_is_told_when_the_fixed_point_weights_lost_their_bits(engine, golden_dir):
    """The step protocol cannot be repeated by the library: a run whose heaviest particle sat more than 6 nats below the reference
    (an observation ~30 standard deviations from every particle) returns CPPROB_HIP_EPRECISION from the first call that reads its
    results; with CPPROB_HIP_FLAG_FLOATING_POINT_STEP it runs, and matches the single context repeated in that form."""
    obs = np.array(_obs(golden_dir, "lgssm100")[:10])
    obs[4] = 40.0
    n_pers = [3000, 2000]
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, obs, sum(n_pers), seed=5, ess_threshold=0.5, flags=cp.capi.FLAG_REPEAT_IN_FLOATING_POINT)
    engine.run()
    ref_stats, ref_sum = engine.stats().copy(), engine.summary()
    assert ref_sum["step_form"] == cp.capi.FORM_FLOAT
    with pytest.raises(cp.CpprobHipError) as err:
        _run_exchange_virtual(cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n_pers, 5, 0.5)
    assert err.value.code == cp.capi.EPRECISION
    stats, s, _, _, _ = _run_exchange_virtual(cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n_pers, 5, 0.5, flags=cp.capi.FLAG_FLOATING_POINT_STEP)
    assert abs(s["log_evidence"] - ref_sum["log_evidence"]) < 1e-9
    np.testing.assert_allclose(stats, ref_stats, rtol=0, atol=5e-3)


def test_exchange_scope_world1_is_bit_identical_to_run(engine, golden_dir):
    from cpprob_amd import distributed as D
    obs = _obs(golden_dir, "hmm16")
    n = 30000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=0.5)
    engine.run()
    ref_stats, ref_sum, ref_vals = engine.stats().copy(), engine.summary(), engine.values().copy()
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=0.5, scope=cp.SCOPE_EXCHANGE)
    cnt = {}
    stats, s = D.run_exchange(engine, D.TorchCollective(engine), counters=cnt)
    assert np.array_equal(engine.values(), ref_vals) and cnt["records_sent"] == 0
    np.testing.assert_allclose(stats, ref_stats, rtol=1e-13, atol=1e-15)
    assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    # a generation that loses its bits: the Python host repairs it in the run, in integers, through the same two calls the C++ group driver
    # uses (cpprob_hip_smc_repair_begin / _end) -- the one-context run, bit for bit
    o2 = np.array(_obs(golden_dir, "lgssm100")[:12])
    o2[5] = 40.0
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, o2, n, seed=4, ess_threshold=0.5)
    engine.run()
    r_sum, r_vals, r_anc = engine.summary(), engine.values().copy(), engine.ancestors().copy()
    assert r_sum["n_requantised"] >= 1
    engine.begin(cp.ALG_SMC, cp.MODEL_LINEAR_GAUSSIAN_1D, o2, n, seed=4, ess_threshold=0.5, scope=cp.SCOPE_EXCHANGE)
    stats, s = D.run_exchange(engine, D.TorchCollective(engine))
    assert s["step_form"] == cp.capi.FORM_FIXED and s["n_requantised"] == r_sum["n_requantised"] and s["log_evidence"] == r_sum["log_evidence"]
    assert np.array_equal(engine.values(), r_vals) and np.array_equal(engine.ancestors(), r_anc)
    # multinomial resampling in the exchange scope is the strata form over remote lineages (tests/test_gpu_group.py): the literal form and
    # the synchronising plan / pack / commit calls refuse loudly
    with pytest.raises(cp.CpprobHipError):
        engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, resampler=cp.RESAMPLE_MULTINOMIAL, scope=cp.SCOPE_EXCHANGE, flags=cp.capi.FLAG_MULTINOMIAL_LITERAL)
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, resampler=cp.RESAMPLE_MULTINOMIAL, scope=cp.SCOPE_EXCHANGE)
    with pytest.raises(cp.CpprobHipError):
        D.run_exchange(engine, D.TorchCollective(engine))


def test_contexts_in_flight_do_not_interfere(engine, golden_dir):
    """Independent runs enqueued round-robin on three contexts (their kernels overlap on the GPU: bench.py's `pipelined`
    figure) give bit-identical results to the same runs one at a time."""
    obs = _obs(golden_dir, "hmm16")
    n = 400_000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=6, ess_threshold=2.0)
    ref = []
    for i in range(6):
        engine.run(i)
        ref.append((engine.stats().copy(), engine.summary()["log_evidence"]))
    engs = [cp.Engine(0) for _ in range(3)]
    for e in engs:
        e.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=6, ess_threshold=2.0)
    for rnd in range(2):
        for k, e in enumerate(engs):
            e.run(rnd * 3 + k)                      # no synchronisation in between: up to three runs in flight
        for k, e in enumerate(engs):
            st, lz = e.stats(), e.summary()["log_evidence"]
            assert np.array_equal(st, ref[rnd * 3 + k][0]) and lz == ref[rnd * 3 + k][1]
    for e in engs:
        e.close()


def test_results_device_and_island_batch_match_the_host_read_out(engine, golden_dir):
    """cpprob_hip_infer_results_device leaves summary + stats on the device without a host sync; IslandBatch chains
    runs on it (world = 1 here: the all-gather degenerates to a copy)."""
    import torch
    from cpprob_amd import distributed as D
    obs = _obs(golden_dir, "hmm16")
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, 50000, seed=3, ess_threshold=0.5, scope=cp.SCOPE_ISLAND)
    out = dzeros(4 + 16 * 3, dtype=torch.float64)
    engine.run(5)
    engine.results_device(out)
    engine.sync()
    s, st = engine.summary(), engine.stats()
    h = out.cpu().numpy()
    assert h[0] == s["log_evidence"] and h[1] == s["ess_final"] and abs(h[2] - s["log_norm"]) < 1e-14 and h[3] == s["max_logw"]
    assert np.array_equal(h[4:].reshape(16, 3), st)
    batch = D.IslandBatch(engine, 4)
    for i in range(6):
        batch.run(i % 4, i)
    stats, lz, iess = batch.results(5 % 4)
    assert np.array_equal(stats, st) and abs(lz - s["log_evidence"]) < 1e-14 and iess == 1.0
    with pytest.raises(cp.CpprobHipError):
        engine.results_device(dzeros(3, dtype=torch.float64))


def test_step_protocol_world1_is_bit_identical_to_run(engine, golden_dir):
    import torch
    from cpprob_amd import distributed as D
    obs = _obs(golden_dir, "hmm16")
    n = 30000
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=4, ess_threshold=0.5)
    engine.run()
    ref_stats, ref_sum, ref_vals = engine.stats().copy(), engine.summary(), engine.values().copy()
    coll = D.TorchCollective(engine)
    stats, s = D.run_joint(engine, coll)
    assert np.array_equal(engine.values(), ref_vals)
    np.testing.assert_allclose(stats, ref_stats, rtol=1e-13, atol=1e-15)
    assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"]
    # the same with the results left on the device (no host synchronisation inside run_joint)
    import torch
    slot = dzeros(4 + engine.T * engine.K, dtype=torch.float64)
    assert D.run_joint(engine, coll, slot=slot) is None
    stats2, s2 = D.joint_results(engine, slot)
    np.testing.assert_allclose(stats2, stats, rtol=1e-15, atol=0)
    assert s2["log_evidence"] == s["log_evidence"]


@pytest.mark.parametrize("world", [2, 4])
def test_joint_population_over_virtual_ranks_smc(golden_dir, world):
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    n_per = 100000
    stats, s, traces = _run_joint_virtual(cp.MODEL_HMM3, cp.ALG_SMC, z["hmm16"], n_per, world, 21, 2.0)
    assert s["n_resampled"] == 15
    np.testing.assert_allclose(stats.sum(axis=1), 1.0, rtol=1e-10)
    assert np.abs(stats - z["hmm16_smooth"]).max() < 0.03
    assert np.abs(stats[-1] - z["hmm16_smooth"][-1]).max() < 5e-3
    assert abs(s["log_evidence"] - float(z["hmm16_logz"])) < 0.02
    for ess, res in traces:                       # joint ESS / decisions agree on all ranks
        assert np.array_equal(res, traces[0][1]) and np.allclose(ess, traces[0][0], rtol=1e-12)
    # ESS-triggered LGSSM
    stats, s, _ = _run_joint_virtual(cp.MODEL_LINEAR_GAUSSIAN_1D, cp.ALG_SMC, z["lgssm100"][:30], n_per, world, 22, 0.5)
    from oracle import exact as E
    ms, ps, _, _, ll = E.kalman_rts(z["lgssm100"][:30])
    assert abs(stats[-1, 0] - ms[-1]) < 0.02 and abs(stats[-1, 1] - ps[-1]) < 0.02 and abs(s["log_evidence"] - ll) < 0.05
    assert 0 < s["n_resampled"] < 29


def test_joint_population_sis_equals_single_shard(engine):
    """SIS shards: ids are global, so 4 shards of 25000 hold exactly the particles of one run of 100000."""
    obs = [3.0, 4.0]
    engine.begin(cp.ALG_SIS, cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, obs, 100000, seed=8)
    engine.run()
    ref, rs = engine.stats().copy(), engine.summary()
    stats, s, _ = _run_joint_virtual(cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, cp.ALG_SIS, obs, 25000, 4, 8, 2.0)
    np.testing.assert_allclose(stats, ref, rtol=1e-10)
    assert abs(s["log_evidence"] - rs["log_evidence"]) < 1e-12 and abs(s["ess_final"] - rs["ess_final"]) < 1e-6 * rs["ess_final"]


@pytest.mark.parametrize("model,key", [(cp.MODEL_HMM3, "hmm16"), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100"), (cp.MODEL_GAUSSIAN_UNKNOWN_MEAN, None)])
@pytest.mark.parametrize("offset", [1, 2, 3, 5001, 4098])
def test_unaligned_particle_offsets_share_blocks_correctly(engine, golden_dir, model, key, offset):
    """Neighbouring particles share Philox blocks (groups of 4 / 2): a shard whose first global id is not a
    multiple of 4 must still draw exactly the variates of those ids (oracle: per-particle, any offset)."""
    obs = [3.0, 4.0] if key is None else _obs(golden_dir, key)[:6]
    n = 3001
    engine.begin(cp.ALG_SIS, model, obs, n, seed=17, particle_offset=offset, n_global=offset + n, scope=cp.SCOPE_ISLAND)
    engine.run()
    vals, logw = O.sis(model, obs, n, 17, pid0=offset)
    got = engine.values()
    if got.dtype == np.int32:
        assert np.array_equal(got, vals)
    else:
        np.testing.assert_allclose(got, vals, rtol=FP_TOL, atol=FP_TOL)
    np.testing.assert_allclose(engine.logw(), logw, rtol=1e-10, atol=1e-10)


def test_joint_population_with_static_schedule_equals_evidence_weighted_islands(golden_dir):
    """Resampling after EVERY step: local resampling + mass carry (joint, per-step all-gather) is the same
    estimator as independent shards combined once by their evidence (no per-step collective)."""
    from cpprob_amd import distributed as D
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    world, n_per, seed = 4, 50000, 31
    joint, s, _ = _run_joint_virtual(cp.MODEL_HMM3, cp.ALG_SMC, z["hmm16"], n_per, world, seed, 2.0)
    log_z, stats = [], []
    for r in range(world):
        e = cp.Engine(0)
        e.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm16"], n_per, seed=seed, ess_threshold=2.0, particle_offset=r * n_per, n_global=world * n_per,
                scope=cp.SCOPE_ISLAND)
        e.run()
        log_z.append(e.summary()["log_evidence"]); stats.append(e.stats().copy())
        e.close()
    comb, lz, w, _ = D.combine_islands(np.array(log_z), np.array(stats), True)
    np.testing.assert_allclose(comb, joint, rtol=1e-9, atol=1e-12)
    assert abs(lz - s["log_evidence"]) < 1e-9


def test_large_population_two_level_normalisation(engine, golden_dir):
    """> 4096 tiles: the normalisation runs as two multi-workgroup launches over slabs (and the step kernel
    reads ctrl / bc / bf instead of normalising in its prologue).  Same answers as the small-population path."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    n = 5_000_000 + 37                      # 4883 tiles of 1024, ragged last tile, 5 slabs
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm16"], n, seed=77, ess_threshold=2.0)
    engine.run()
    st, s = engine.stats(), engine.summary()
    assert s["n_resampled"] == 15
    np.testing.assert_allclose(st.sum(axis=1), 1.0, rtol=1e-12)
    assert np.abs(st[-1] - z["hmm16_smooth"][-1]).max() < 2e-3
    assert np.abs(st - z["hmm16_smooth"]).max() < 1.5e-2
    assert abs(s["log_evidence"] - float(z["hmm16_logz"])) < 3e-3
    anc = engine.ancestors()
    assert np.all(np.diff(anc[1:].astype(np.int64), axis=1) >= 0) and anc.max() < n and anc.min() >= 0
    # offspring counts of systematic resampling differ from N*W by less than one: check one step against the weights
    ess, res = engine.step_trace()
    assert np.all(res[:-1] == 1) and res[-1] == 0
    # ESS-triggered on the same population (carry path through the generic partial)
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, z["hmm16"], n, seed=78, ess_threshold=0.5)
    engine.run()
    st2, s2 = engine.stats(), engine.summary()
    assert 0 < s2["n_resampled"] < 15
    assert np.abs(st2 - z["hmm16_smooth"]).max() < 1.5e-2 and abs(s2["log_evidence"] - float(z["hmm16_logz"])) < 1e-2   # fewer resampling steps: larger evidence variance


@pytest.mark.parametrize("n", [300_000, 1_000_000, 1_500_000, 3_000_000, 6_000_000])
def test_back_to_back_runs_are_bitwise_reproducible(engine, golden_dir, n):
    """Every code path of the step kernel (prologue variants for <= 512 / 1024 / 1664 tiles, ctrl-reading form above,
    two-level normalisation above 4096 tiles): the same run index must give bit-identical results however the
    launches interleave -- any race in the inter-step hand-offs (partials ping-pong, u0 hand-off, ctrl) would show here."""
    obs = _obs(golden_dir, "hmm16")
    for ess in (2.0, 0.5):
        engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=99, ess_threshold=ess)
        ref = {}
        for rep in range(3):
            for idx in (0, 1, 2, 3):
                engine.run(idx)
                if rep == 0:
                    ref[idx] = (engine.stats().copy(), engine.summary()["log_evidence"])
                elif rep == 2:
                    st, lz = engine.stats(), engine.summary()["log_evidence"]
                    assert np.array_equal(st, ref[idx][0]) and lz == ref[idx][1]
        assert not np.array_equal(ref[0][0], ref[1][0])


@pytest.mark.parametrize("scope,workload", [("auto", "hmm16_smc"), ("global", "hmm16_smc"), ("exchange", "hmm16_smc"), ("exchange", "hmm128_smc_ess"),
                                            ("exchange", "lgssm100_smc")])
def test_bench_two_ranks_on_one_gpu(scope, workload):
    """The N > 1 path of bench.py end to end -- torchrun, one process per rank, IslandBatch / run_joint / run_exchange --
    with both ranks sharing cuda:0 and the collectives going through gloo (test hook CPPROB_DIST_BACKEND / CPPROB_FORCE_DEVICE:
    RCCL refuses two ranks on one device).  Everything but the transport is the production code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CPPROB_DIST_BACKEND="gloo", CPPROB_FORCE_DEVICE="0")
    import socket
    for attempt in range(2):                                        # a rendezvous port can be taken between probing and use: one retry
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--particles", "200000", "--scope", scope, "--no-cpu-baseline",
               "--workload", workload, "--no-extras"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                          # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["n_global"] == 400000 and d["value"] > 0
    if workload == "hmm16_smc":
        assert d["posterior_max_abs_err_vs_exact"] < 0.01 and abs(d["log_evidence"] + 26.326) < 0.02
        assert d["config"]["scope"] == ("exchange" if scope in ("auto", "exchange") else scope)      # exact global resampling is the default
    else:
        z = np.load(os.path.join(root, "tests", "golden", "observations.npz"))
        key = "hmm128_logz" if workload == "hmm128_smc_ess" else "lgssm100_logz"
        assert abs(d["log_evidence"] - float(z[key])) < 0.05 and 0 < d["n_resampled"] < len(z[key.replace("_logz", "")]) - 1
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    if d["config"]["scope"] == "exchange":
        # the library's own driver, one rank per process: set-up and the final reduction through torch.distributed (gloo here), the
        # steps through mailboxes and direct stores between the two processes
        assert d["config"]["host"].startswith("C++ (cpprob_hip_group_run: torch.distributed gloo"), d["config"]
        assert d["exchange_traffic_per_run"]["transport"] == "direct" and d["exchange_traffic_per_run"]["records"] > 0
        assert d["config"]["exchange_reruns"]["timed_batch"] == 0
        # the preflight in front of the timed region: a 2e5-particle-per-rank run on the default transport, every rank's traces equal
        # to the same population run on one GPU alone
        assert d["preflight"]["settled_on"] == "default" and d["preflight"]["rungs"][0]["traces_equal_single_gpu_run_on_every_rank"] is True, d["preflight"]
        # where a rank-step's time goes, and which transports carried it (what the first run over real links has to explain)
        bd = d["rank_step_breakdown_us"]
        for key in ("step_and_totals", "allgather", "totals_handover", "pack", "barrier", "commit", "mailbox_wait", "steps", "transport_note"):
            assert key in bd, bd
        assert bd["steps"] > 0 and bd["step_and_totals"] > 0 and bd["pack"] > 0 and "migrants: direct stores" in d["transport_note"]
        assert d["roofline"]["frac_layout"] <= 1.0 and d["roofline"]["layout_bytes_per_unit"] > 0 and "step_form" in d["roofline"]


HMM2 = ([-1.5, 1.0], [[0.85, 0.15], [0.3, 0.7]])
HMM5 = ([-2.0, -1.0, 0.0, 1.0, 2.5], [[4, 2, 1, 1, 2], [1, 5, 2, 1, 1], [1, 1, 6, 1, 1], [2, 1, 1, 5, 1], [1, 1, 2, 2, 4]])


@pytest.mark.parametrize("table", [HMM2, HMM5], ids=["2-state", "5-state"])
@pytest.mark.parametrize("ess", [2.0, 0.5])
@pytest.mark.parametrize("n", [30_000, 1_300_000])
def test_table_hmm_of_any_size_is_bit_exact_against_the_oracle(engine, golden_dir, table, ess, n):
    """CPPROB_HIP_MODEL_HMM_TABLE: the model body of models.hpp:114-141 over a caller-given table (here 2 and 5 states; transition
    rows as un-normalised weights).  Its steps run on fixed-point weights, so the index work is integer arithmetic whatever the number
    of states and the schedule: states and ancestors equal the oracle's, bit for bit, the estimator agrees with the oracle's applied
    to the same integers, and the posterior matches forward-backward for that table."""
    means, trans = np.array(table[0], float), np.array(table[1], float)
    rng = np.random.default_rng(5)
    T = 12
    P = trans / trans.sum(1, keepdims=True)
    st = rng.integers(0, len(means))
    obs = np.zeros(T)
    for t in range(T):
        if t:
            st = rng.choice(len(means), p=P[st])
        obs[t] = means[st] + rng.standard_normal()
    engine.set_hmm(means, trans)
    O.set_hmm(means, trans)
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM_TABLE, obs, n, seed=19, ess_threshold=ess)
    engine.run()
    s, stats = engine.summary(), engine.stats()
    assert s["step_form"] == cp.capi.FORM_FIXED and s["stats_per_predict"] == 8
    ref = O.smc(O.MODEL_HMM_TABLE, obs, n, 19, O.RESAMPLE_SYSTEMATIC, ess)
    vals, anc = engine.values(), engine.ancestors()
    assert np.array_equal(vals, ref["hist"]) and np.array_equal(anc, ref["anc"])
    gess, gres = engine.step_trace()
    assert np.array_equal(gres, ref["resampled"]) and abs(s["log_evidence"] - ref["log_z"]) < 1e-9
    np.testing.assert_allclose(gess, ref["ess"], rtol=1e-9)
    k = len(means)
    assert np.all(stats[:, k:] == 0.0)
    q = O.fix_weights(engine.logw(), s["max_logw"]).astype(np.float64)
    np.testing.assert_allclose(stats[:, :k], O.smoothing_linear(vals, anc, q, k=k), rtol=1e-11, atol=1e-13)
    # exact smoothing marginals of this table (forward-backward)
    lik = np.exp(-0.5 * ((obs[:, None] - means[None, :]) ** 2 + np.log(2 * np.pi)))
    alpha = np.zeros((T, k)); c = np.zeros(T)
    a = np.full(k, 1.0 / k) * lik[0]; c[0] = a.sum(); alpha[0] = a / c[0]
    for t in range(1, T):
        a = (alpha[t - 1] @ P) * lik[t]; c[t] = a.sum(); alpha[t] = a / c[t]
    beta = np.ones((T, k))
    for t in range(T - 2, -1, -1):
        beta[t] = (P @ (lik[t + 1] * beta[t + 1])) / c[t + 1]
    gamma = alpha * beta
    gamma /= gamma.sum(1, keepdims=True)
    assert np.abs(stats[:, :k] - gamma).max() < (4e-2 if n < 100_000 else 8e-3)
    assert abs(s["log_evidence"] - np.log(c).sum()) < (5e-2 if n < 100_000 else 1e-2)
    # SIS of the same model: every particle's trace and weight against the oracle's
    engine.begin(cp.ALG_SIS, cp.MODEL_HMM_TABLE, obs, 20_000, seed=4)
    engine.run()
    v0, lw0 = O.sis(O.MODEL_HMM_TABLE, obs, 20_000, 4)
    assert np.array_equal(engine.values(), v0)
    np.testing.assert_allclose(engine.logw(), lw0, rtol=1e-12, atol=1e-12)
    with pytest.raises(cp.CpprobHipError):
        engine.set_hmm([0.0], [[1.0]])


@pytest.mark.parametrize("n,T", [(100000, 16), (1000003, 16), (5000, 1), (3, 7), (40000, 9)])
def test_trace_word_readout_is_the_lineage_walk_in_integers(engine, golden_dir, n, T):
    """Short discrete traces ride with the particles (csrc/trace_words.hpp): the default read-out of hmm<T <= 16> counts
    (x_t, x_T-1) pairs over the final particles' trace words -- integers -- where CPPROB_HIP_FLAG_WALK_READOUT walks anc[]
    backwards and sums weights in floating point.  Same traces, same evidence; the statistics are those the materialised
    traces give, to the last bits of the final division."""
    obs = _obs(golden_dir, "hmm16")[:T]
    out = {}
    for name, flags in (("words", 0), ("walk", cp.capi.FLAG_WALK_READOUT)):
        engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=77, ess_threshold=2.0, flags=flags)
        engine.run()
        stats = engine.stats().copy()
        s = engine.summary()
        paths = engine.paths()
        assert np.array_equal(engine.stats(), stats)                   # (materialising the traces leaves the statistics alone)
        out[name] = (stats, s, paths, engine.logw())
    assert out["words"][1]["step_form"] == cp.capi.FORM_COUNTS
    assert np.array_equal(out["words"][2], out["walk"][2]) and out["words"][1]["log_evidence"] == out["walk"][1]["log_evidence"]
    np.testing.assert_allclose(out["words"][0], out["walk"][0], rtol=0, atol=2e-15)
    # from the traces themselves
    paths, lw = out["words"][2], out["words"][3]
    w = np.exp(lw - lw.max())
    ref = np.stack([[w[paths[t] == s].sum() for s in range(3)] for t in range(T)]) / w.sum()
    np.testing.assert_allclose(out["words"][0], ref, rtol=0, atol=1e-13)
    # twice the same run: the same bits
    engine.begin(cp.ALG_SMC, cp.MODEL_HMM3, obs, n, seed=77, ess_threshold=2.0)
    engine.run()
    assert np.array_equal(engine.stats(), out["words"][0])
    engine.run(0)
    assert np.array_equal(engine.stats(), out["words"][0])
