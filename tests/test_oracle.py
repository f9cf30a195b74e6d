"""CPU tests: the oracle against every golden vector / known answer the reference holds for the path.

  * normal logpdf on the grid of the reference's own test (tests/cpprob/logpdf.cpp:23-35, eps 1e-8 :16)
  * uniform_real logpdf on the grid of tests/cpprob/logpdf.cpp:61-78; the other functors vs closed forms
  * Philox4x32-10 / Box-Muller against rocRAND's host engine (tests/golden/philox_rocrand.json)
  * estimators against the analytic posteriors (README.md:118, thesis p.85)
  * dump grammar against lines printed by the reference's own serialization.hpp
  * SMC against exact forward-backward / Kalman-RTS posteriors
"""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import exact as E
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_normal_logpdf_reference_grid():
    g = np.load(os.path.join(GOLD, "logpdf_grid.npz"))
    L = O.lib()
    grid, exp = g["normal_grid"], g["normal_expected"]
    got = np.array([L.orc_normal_logpdf(*row) for row in grid])
    eps = 1e-8                                   # tests/cpprob/logpdf.cpp:16
    assert np.max(np.abs(got - exp)) < eps
    assert np.max(np.abs(np.exp(got) - np.exp(exp))) < eps
    assert np.max(np.abs(got - exp)) < 1e-12     # and far tighter in fact


def test_normal_logpdf_branches():
    L = O.lib()
    inf = float("inf")
    assert L.orc_normal_logpdf(1.0, 1.0, 0.0) == 0.0            # Dirac delta, utils_normal_distribution.hpp:28-32
    assert L.orc_normal_logpdf(2.0, 1.0, 0.0) == -inf
    assert L.orc_normal_logpdf(inf, 0.0, 1.0) == -inf           # :34-36
    assert L.orc_normal_logpdf(-inf, 0.0, 1.0) == -inf


def test_other_logpdfs_closed_forms():
    g = np.load(os.path.join(GOLD, "logpdf_grid.npz"))
    L = O.lib()
    got = np.array([L.orc_uniform_real_logpdf(*row) for row in g["uniform_grid"]])
    exp = g["uniform_expected"]
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.max(np.abs(got[fin] - exp[fin])) < 1e-8
    got = np.array([L.orc_poisson_logpdf(int(k), l) for k, l in g["poisson_grid"]])
    assert np.max(np.abs(got - g["poisson_expected"])) < 1e-10
    assert L.orc_uniform_smallint_logpdf(1, 0, 2) == pytest.approx(-np.log(3.0))
    assert L.orc_uniform_smallint_logpdf(3, 0, 2) == -float("inf")
    w = np.array([0.1, 0.5, 0.4])
    assert L.orc_discrete_logpdf(1, w, 3) == pytest.approx(np.log(0.5))
    assert L.orc_discrete_logpdf(-1, w, 3) == -float("inf")


def test_philox_matches_rocrand_host_engine():
    with open(os.path.join(GOLD, "philox_rocrand.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 48
    for c in cases:
        assert O.draw_block(c["seed"], c["pid"], c["draw"]).tolist() == c["words"]
        z = O.box_muller(c["words"])
        assert abs(z[0] - float.fromhex(c["normal_x"])) < 1e-13
        assert abs(z[1] - float.fromhex(c["normal_y"])) < 1e-13
        u = O.lib().orc_u01_53(c["words"][0], c["words"][1])     # rocRAND's is (0,1]: one ulp of 2^-53 higher
        assert abs((u + 2.0 ** -53) - float.fromhex(c["u01"])) < 1e-18
    # Random123 known-answer test
    assert O.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_generators_have_the_right_laws():
    L = O.lib()
    n = 200000
    z = np.array([L.orc_draw_std_normal(3, i, 0) for i in range(n)])
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3) < 0.08
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 0.01       # the two Box-Muller outputs of a block
    s = np.array([L.orc_draw_smallint(3, i, 1, 0, 2) for i in range(n)])
    assert np.all(np.abs(np.bincount(s, minlength=3) / n - 1 / 3) < 0.005)
    w = np.array([0.15, 0.15, 0.7])
    d = np.array([L.orc_draw_discrete(3, i, 2, w, 3) for i in range(n)])
    assert np.all(np.abs(np.bincount(d, minlength=3) / n - w) < 0.005)
    k = np.array([L.orc_draw_poisson(3, i, 4, 3.7) for i in range(n)])
    assert abs(k.mean() - 3.7) < 0.03 and abs(k.var() - 3.7) < 0.08
    u = np.array([L.orc_draw_uniform_real(3, i, 3, -1.0, 3.0) for i in range(n)])
    assert u.min() >= -1.0 and u.max() < 3.0 and abs(u.mean() - 1.0) < 0.02


def test_estimators_against_published_posteriors():
    with open(os.path.join(GOLD, "posteriors.json")) as f:
        post = json.load(f)
    n = 400000
    # README.md:118 -- src/models/gaussian.cpp, observes (3,4): mean 2.32353, variance 1.05882
    v, lw = O.sis(O.MODEL_GAUSSIAN_README, [3.0, 4.0], n, 99)
    m = O.weighted_moments(v[0], lw)
    assert abs(m[0] - 2.32353) < 0.01 and abs(m[1] - 1.05882) < 0.015
    assert abs(post["readme_gaussian_obs_3_4"]["derived"][0] - 2.32353) < 5e-6
    # thesis p.85 -- models.hpp gaussian, observes (8,9): N(7.25, 5/6).  ESS/N ~ 0.02: wide tolerance
    v, lw = O.sis(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [8.0, 9.0], n, 99)
    m = O.weighted_moments(v[0], lw)
    assert abs(m[0] - 7.25) < 0.06 and abs(m[1] - 5.0 / 6.0) < 0.08
    # BASELINE.json configs[0]: observes (3,4), 10^4 particles
    v, lw = O.sis(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], 10000, 1)
    m = O.weighted_moments(v[0], lw)
    assert abs(m[0] - post["models_gaussian_obs_3_4"]["mean"]) < 0.08
    assert abs(m[1] - post["models_gaussian_obs_3_4"]["variance"]) < 0.12
    assert abs((m[2] - np.log(10000)) - post["models_gaussian_obs_3_4"]["log_evidence"]) < 0.05
    assert abs(m[3] / 10000 - 0.344) < 0.03                     # importance-sampling efficiency, SURVEY 8(d)


def test_logsumexp_and_moments_definitions():
    rng = np.random.default_rng(0)
    lw = rng.normal(size=1000) * 5 - 1000
    x = rng.normal(size=1000)
    lse = O.logsumexp(lw)
    assert abs(lse - (np.log(np.sum(np.exp(lw - lw.max()))) + lw.max())) < 1e-12
    w = np.exp(lw - lse)
    m = O.weighted_moments(x, lw)
    assert abs(m[0] - np.sum(w * x)) < 1e-12 and abs(m[1] - (np.sum(w * x * x) - np.sum(w * x) ** 2)) < 1e-12
    assert O.logsumexp(np.zeros(0)) == 0.0                      # empirical_distribution.hpp:131-133
    xi = rng.integers(0, 3, 1000).astype(np.int32)
    h = O.weighted_hist(xi, lw, 3)
    assert abs(h.sum() - 1) < 1e-12 and abs(h[1] - w[xi == 1].sum()) < 1e-12


def test_faithful_dump_matches_reference_grammar(tmp_path):
    """orc_sis_faithful writes what StateInfer::dump_predicts would (state.cpp:262-267); the
    expected text comes from the reference's own serialization.hpp (tests/golden/serialization.json)."""
    with open(os.path.join(GOLD, "serialization.json")) as f:
        gold = json.load(f)
    # 1. the C formatter of the oracle reproduces the reference's lines for the golden inputs
    import ctypes
    libc = ctypes.CDLL(None)
    libc.snprintf.restype = ctypes.c_int
    for case in gold["real"]:
        vals = [float.fromhex(v) for v in case["values"]]
        lw = float.fromhex(case["logw"])
        line = "([" + " ".join("(0 %.15e)" % v for v in vals) + "] %.15e)" % lw
        assert line == case["line"]
    for case in gold["int"]:
        lw = float.fromhex(case["logw"])
        line = "([" + " ".join("(0 %d)" % v for v in case["values"]) + "] %.15e)" % lw
        assert line == case["line"]
    # 2. a faithful run: three append-mode files per particle, all-empty ones removed, ids written
    prefix = str(tmp_path / "posterior")
    O.sis_faithful(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], 50, 7, prefix, "Mu")
    assert os.path.exists(prefix + ".real") and os.path.exists(prefix + ".ids")
    assert not os.path.exists(prefix + ".int") and not os.path.exists(prefix + ".any")
    assert open(prefix + ".ids").read() == "Mu\n"
    lines = open(prefix + ".real").read().splitlines()
    vals, lw = O.sis(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], 50, 7)
    assert len(lines) == 50
    assert lines[0] == "([(0 %.15e)] %.15e)" % (vals[0, 0], lw[0])
    # 3. when the reference-built parser is available, it reads the file back
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_serialization")
    if os.path.exists(ref):
        out = subprocess.check_output([ref, "parse-real"], input="\n".join(lines).encode()).decode().splitlines()
        assert len(out) == 50 and not any(o == "BAD" for o in out)
        f = out[3].split()
        assert int(f[0]) == 1 and abs(float(f[2]) - vals[0, 3]) < 1e-14 * max(1, abs(vals[0, 3])) and abs(float(f[3]) - lw[3]) < 1e-13
    O.sis_faithful(O.MODEL_HMM3, E.simulate_hmm(4, 1), 10, 7, prefix + "_h", "State")
    assert os.path.exists(prefix + "_h.int") and not os.path.exists(prefix + "_h.real")
    assert open(prefix + "_h.int").readline().count("(0 ") == 4


@pytest.mark.parametrize("kind", [O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL])
def test_resamplers_are_unbiased(kind):
    rng = np.random.default_rng(1)
    n = 2000
    lw = rng.normal(size=n)
    w = np.exp(lw - lw.max()); w /= w.sum()
    counts = np.zeros(n)
    reps = 200
    for r in range(reps):
        counts += np.bincount(O.resample(kind, lw, 5, r), minlength=n)
    z = (counts / reps - n * w) / np.sqrt(np.maximum(n * w, 1e-9) / reps)
    assert np.abs(z).max() < 6.0
    if kind != O.RESAMPLE_MULTINOMIAL:
        assert np.all(np.diff(O.resample(kind, lw, 5, 0)) >= 0)


def test_smc_against_exact_posteriors():
    z = np.load(os.path.join(GOLD, "observations.npz"))
    r = O.smc(O.MODEL_HMM3, z["hmm16"], 200000, 3, O.RESAMPLE_SYSTEMATIC, 2.0)
    sm = O.smoothing(r["hist"], r["anc"], r["logw"])
    assert np.abs(sm - z["hmm16_smooth"]).max() < 0.02
    assert np.abs(sm[-1] - z["hmm16_smooth"][-1]).max() < 5e-3          # last step = filtering quality
    assert abs(r["log_z"] - float(z["hmm16_logz"])) < 0.02
    assert r["resampled"].sum() == 15 and r["resampled"][-1] == 0
    r = O.smc(O.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"][:30], 100000, 3, O.RESAMPLE_SYSTEMATIC, 0.5)
    sm = O.smoothing(r["hist"], r["anc"], r["logw"])
    ms, ps, _, _, ll = E.kalman_rts(z["lgssm100"][:30])
    assert abs(sm[-1, 0] - ms[-1]) < 0.02 and abs(sm[-1, 1] - ps[-1]) < 0.02
    assert abs(r["log_z"] - ll) < 0.05
    assert 0 < r["resampled"].sum() < 29                                  # ESS-triggered: not every step


def test_sis_equals_smc_without_resampling():
    obs = E.simulate_hmm(6, 5)
    v, lw = O.sis(O.MODEL_HMM3, obs, 5000, 11)
    r = O.smc(O.MODEL_HMM3, obs, 5000, 11, O.RESAMPLE_SYSTEMATIC, 0.0)   # ESS < 0 never holds
    assert np.array_equal(v, r["hist"]) and np.allclose(lw, r["logw"], rtol=0, atol=1e-12)
    assert np.array_equal(r["anc"], np.tile(np.arange(5000, dtype=np.int32), (6, 1)))


def test_exact_solvers_self_consistency():
    obs = E.simulate_hmm(5, 2)
    g, a, ll = E.hmm_forward_backward(obs)
    # brute force over 3^5 paths
    import itertools
    tot = 0.0
    marg = np.zeros((5, 3))
    for path in itertools.product(range(3), repeat=5):
        p = 1.0 / 3.0
        for t, s in enumerate(path):
            if t > 0:
                p *= E.HMM_T[path[t - 1], s]
            p *= np.exp(E.normal_logpdf(obs[t], E.HMM_MEAN[s], 1.0))
        tot += p
        for t, s in enumerate(path):
            marg[t, s] += p
    assert abs(np.log(tot) - ll) < 1e-12 and np.abs(marg / tot - g).max() < 1e-12
    m, v = E.gaussian_posterior(1, 1.5, 2, [3, 4])
    assert abs(m - 2.323529411764706) < 1e-12 and abs(v - 1.0588235294117647) < 1e-12


def test_oracle_under_sanitizers():
    """run the CPU oracle under ASan/UBSan (the GPU pool offers no sanitizer)."""
    src = os.path.join(ROOT, "oracle", "cpprob_oracle.c")
    exe = "/tmp/orc_asan_test"
    drv = "/tmp/orc_asan_drv.c"
    with open(drv, "w") as f:
        f.write('''
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
int orc_smc(int, const double*, size_t, uint64_t, uint64_t, int, double, double*, int32_t*, int32_t*, double*, double*, double*, int32_t*);
int orc_sis(int, const double*, size_t, uint64_t, uint64_t, uint64_t, double*, int32_t*, double*);
void orc_smoothing_int(const int32_t*, const int32_t*, const double*, size_t, uint64_t, int, double*);
int main(void) {
  double obs[7] = {0.3,-1.2,0.8,1.5,-0.1,0.4,2.0}; size_t T = 7; uint64_t n = 1531;
  int32_t* hist = malloc(T*n*4); int32_t* anc = malloc(T*n*4); double* lw = malloc(n*8); double lz, ess[7]; int32_t rs[7];
  double out[21];
  for (int k = 0; k < 3; ++k) { if (orc_smc(3, obs, T, n, 5, k, k == 0 ? 2.0 : 0.5, NULL, hist, anc, lw, &lz, ess, rs)) return 1;
    orc_smoothing_int(hist, anc, lw, T, n, 3, out); }
  double* vr = malloc(n*8); if (orc_sis(0, obs, 2, n, 1, 0, vr, NULL, lw)) return 2;
  free(hist); free(anc); free(lw); free(vr); printf("ok\\n"); return 0; }
''')
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-o", exe, drv, src, "-lm"])
    out = subprocess.check_output([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1")).decode()
    assert out.strip() == "ok"


def test_table_weight_resampler_is_order_independent_and_matches_the_cdf_form():
    """The systematic resampler the index-work parity rests on (orc_resample_table_systematic: integer prefix counts,
    C_k = fma(c_2, e_2, fma(c_1, e_1, c_0 e_0)), ancestor of j = min{k : ceil(fma(C_k, N/W, -u0)) > j}): equal to the
    sequential-CDF form (thesis p.36 remark: positions (j + u0) W/N) wherever that form's sums are exact, and independent
    of how the population is cut into shards -- the property that lets tiles, wavefronts and GPUs evaluate it in any order."""
    rng = np.random.default_rng(1)
    x = rng.integers(0, 3, 60000).astype(np.int32)
    n = len(x)
    for e in (np.array([1.0, 0.5, 0.25]), np.array([0.125, 1.0, 0.5])):            # exactly summable: both forms must agree to the last index
        a = O.resample_table_systematic(x, e, 7, 3)
        b = O.resample(O.RESAMPLE_SYSTEMATIC, np.log(e[x]), 7, 3)
        assert np.array_equal(a, b)
    e = np.array([0.3123, 1.0, 0.0712345])
    full = O.resample_table_systematic(x, e, 11, 5)
    assert np.all(np.diff(full) >= 0) and full[0] >= 0 and full[-1] <= n - 1
    counts = np.bincount(full, minlength=n)
    expect = n * e[x] / e[x].sum()
    assert np.abs(counts - expect).max() < 1.0 + 1e-9                              # systematic: every offspring count within one of its expectation
    total = np.bincount(x, minlength=3).astype(np.uint64)
    for cuts in ([0, 12000, 30001, n], [0, 1, 2, 59999, n], [0, 20000, 20001, 40000, n]):
        got = np.full(n, -1, np.int64)
        for r in range(len(cuts) - 1):
            before = np.bincount(x[:cuts[r]], minlength=3).astype(np.uint64)
            ar = O.resample_table_systematic(x[cuts[r]:cuts[r + 1]], e, 11, 5, before=before, total=total, last_shard=(r == len(cuts) - 2),
                                             j0=0, n_out=n, n_total_out=n)
            m = ar >= 0
            assert (got[m] == -1).all()                                            # no output is claimed by two shards
            got[m] = ar[m] + cuts[r]
        assert np.array_equal(got, full)                                           # ... and none is left out; the ancestors are the single-shard ones


def test_fixed_point_resampler_is_order_independent_and_matches_the_cdf_form(golden_dir):
    """The systematic resampler of continuous weights and ESS-triggered schedules (orc_resample_fixed_systematic: integer weights
    q_i = rint(exp(lw_i - R) 2^32), exact 64-bit prefix masses, ancestor of j = min{k : ceil(fma(double(C_k), N / double(C_N), -u0)) > j}):
    equal to the sequential-CDF form on the same weights wherever that form's sums are exact (integers below 2^53 are), independent
    of how the population is cut into shards, every offspring count within one of its expectation; the integer weight itself is
    exp to within one unit of 2^-32, monotone, 0 for -inf and saturating at the reference."""
    rng = np.random.default_rng(2)
    n = 50000
    lw = rng.normal(size=n) * 2.5 - 3.0
    ref = lw.max() + 0.25
    q = O.fix_weights(lw, ref)
    assert np.abs(q.astype(np.float64) - np.exp(lw - ref) * 2.0 ** 32).max() <= 0.5 + 1e-3
    assert O.fix_weights(np.array([-np.inf, ref, ref - 1e-300, ref - 800.0]), ref).tolist() == [0, 2 ** 32 - 1, 2 ** 32 - 1, 0]
    srt = np.sort(lw)
    assert np.all(np.diff(O.fix_weights(srt, ref).astype(np.int64)) >= 0)
    full = O.resample_fixed_systematic(q, 11, 5)
    # the floating-point CDF form on the SAME weights: its running sums are integers below 2^53, i.e. exact
    assert np.array_equal(full, O.resample(O.RESAMPLE_SYSTEMATIC, np.log(q.astype(np.float64)), 11, 5))
    assert np.all(np.diff(full) >= 0) and full[0] >= 0 and full[-1] <= n - 1
    counts = np.bincount(full, minlength=n)
    qd = q.astype(np.float64)
    assert np.abs(counts - n * qd / qd.sum()).max() < 1.0 + 1e-9
    tot = int(q.astype(np.uint64).sum())
    for cuts in ([0, 12000, 30001, n], [0, 1, 2, n - 1, n], [0, 20000, 20001, 40000, n]):
        got = np.full(n, -1, np.int64)
        for r in range(len(cuts) - 1):
            ar = O.resample_fixed_systematic(q[cuts[r]:cuts[r + 1]], 11, 5, before=int(q[:cuts[r]].astype(np.uint64).sum()), total=tot,
                                             last_shard=(r == len(cuts) - 2), j0=0, n_out=n, n_total_out=n)
            m = ar >= 0
            assert (got[m] == -1).all()
            got[m] = ar[m] + cuts[r]
        assert np.array_equal(got, full)
    # the SMC driver on this form against the exact posteriors (Kalman / RTS, forward-backward)
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    r = O.smc(O.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"][:30], 100_000, 5, O.RESAMPLE_SYSTEMATIC, 0.5)
    assert 5 <= r["resampled"].sum() <= 25 and abs(r["log_z"] - E.kalman_rts(z["lgssm100"][:30])[4]) < 3e-2
    r = O.smc(O.MODEL_HMM3, z["hmm128"][:40], 100_000, 5, O.RESAMPLE_SYSTEMATIC, 0.5)
    assert abs(r["log_z"] - E.hmm_forward_backward(z["hmm128"][:40])[2]) < 3e-2


def test_oracle_filtering_statistics_against_the_exact_filters(golden_dir):
    """What a filtering-only run reports: predict hit t under generation t's own weights.  Forward algorithm (HMM) and Kalman
    filter (LGSSM) within Monte-Carlo error; at the last hit filtering and smoothing are the same numbers."""
    z = np.load(os.path.join(golden_dir, "observations.npz"))
    r = O.smc(O.MODEL_HMM3, z["hmm16"], 200_000, 3, O.RESAMPLE_SYSTEMATIC, 0.5)
    assert np.abs(r["filter"] - z["hmm16_filter"]).max() < 8e-3
    np.testing.assert_allclose(r["filter"][-1], O.smoothing(r["hist"], r["anc"], r["logw"])[-1], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r["filter"].sum(axis=1), 1.0, rtol=1e-10)
    r = O.smc(O.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"][:12], 300_000, 3, O.RESAMPLE_SYSTEMATIC, 2.0)
    assert np.abs(r["filter"][:, 0] - z["lgssm100_filter_mean"][:12]).max() < 8e-3
    # (the ESS drops to ~3e4 at the outlying ninth observation: sd of the variance estimate there 0.62 sqrt(2 / ESS) = 5e-3)
    assert np.abs(r["filter"][:, 1] - z["lgssm100_filter_var"][:12]).max() < 1.5e-2
    np.testing.assert_allclose(r["filter"][-1], O.smoothing(r["hist"], r["anc"], r["logw"])[-1], rtol=1e-8, atol=1e-10)



def test_integer_stratified_and_literal_multinomial_match_the_cdf_forms_and_are_order_independent():
    """The integer forms every resampler runs in on the device (stratified: A_k = F + [u_F < H_k - F], H_k = double(C_k) N / double(C_N);
    literal multinomial: tau_j = floor(u_j C_N), ancestor = min{k : C_k > tau_j}; stratified on prefix counts for table weights):
    equal to the sequential floating-point CDF forms (orc_resample) wherever those sums are exact, and independent of how the
    population is cut into shards."""
    rng = np.random.default_rng(2)
    n = 50000
    lw = rng.normal(size=n) * 2.5 - 3.0
    q = O.fix_weights(lw, lw.max() + 0.25)
    tot = int(q.astype(np.uint64).sum())
    for f, kind in ((O.resample_fixed_stratified, O.RESAMPLE_STRATIFIED), (O.resample_fixed_multinomial, O.RESAMPLE_MULTINOMIAL)):
        full = f(q, 11, 5)
        assert np.array_equal(full, O.resample(kind, np.log(q.astype(np.float64)), 11, 5))
        assert full.min() >= 0 and full.max() <= n - 1 and q[full].min() > 0                  # a weightless particle is nobody's ancestor
        if kind == O.RESAMPLE_STRATIFIED:
            assert np.all(np.diff(full) >= 0)
            counts = np.bincount(full, minlength=n)
            qd = q.astype(np.float64)
            assert np.abs(counts - n * qd / qd.sum()).max() < 2.0 + 1e-9                        # stratified: every offspring count within two of its expectation
        for cuts in ([0, 12000, 30001, n], [0, 1, 2, n - 1, n], [0, 20000, 20001, 40000, n]):
            got = np.full(n, -1, np.int64)
            for r in range(len(cuts) - 1):
                kw = dict(before=int(q[:cuts[r]].astype(np.uint64).sum()), total=tot, j0=0, n_out=n)
                if kind == O.RESAMPLE_STRATIFIED:
                    kw.update(last_shard=(r == len(cuts) - 2), n_total_out=n)
                ar = f(q[cuts[r]:cuts[r + 1]], 11, 5, **kw)
                m = ar >= 0
                assert (got[m] == -1).all()
                got[m] = ar[m] + cuts[r]
            assert np.array_equal(got, full)
    x = rng.integers(0, 3, 60000).astype(np.int32)
    for e in (np.array([1.0, 0.5, 0.25]), np.array([0.125, 1.0, 0.5])):
        assert np.array_equal(O.resample_table_stratified(x, e, 7, 3), O.resample(O.RESAMPLE_STRATIFIED, np.log(e[x]), 7, 3))
    e = np.array([0.3123, 1.0, 0.0712345])
    full = O.resample_table_stratified(x, e, 11, 5)
    total = np.bincount(x, minlength=3).astype(np.uint64)
    for cuts in ([0, 12000, 30001, len(x)], [0, 1, 2, 59999, len(x)]):
        got = np.full(len(x), -1, np.int64)
        for r in range(len(cuts) - 1):
            ar = O.resample_table_stratified(x[cuts[r]:cuts[r + 1]], e, 11, 5, before=np.bincount(x[:cuts[r]], minlength=3).astype(np.uint64), total=total,
                                             last_shard=(r == len(cuts) - 2), j0=0, n_out=len(x), n_total_out=len(x))
            m = ar >= 0
            assert (got[m] == -1).all()
            got[m] = ar[m] + cuts[r]
        assert np.array_equal(got, full)


def _strata_offsets_in_python(seed, step, n_out, k):
    """orc_multinomial_strata restated with numpy + the oracle's Philox blocks only: a binary tree over 2^k strata, the n thresholds of
    a node go left with probability 1/2 each (popcount of the first n bits of the node's stream); above 6 levels the outputs are
    dealt to groups that run the top six levels each and add up, and every level-6 node splits its total further."""
    base3 = (1 << 40) + (1 << 38)

    def left_of(node_key, n):
        left = 0
        for chunk in range((n + 127) // 128):
            r = O.draw_block(seed, (node_key << 32) | chunk, base3 + step)
            bits = int(r[0]) | (int(r[1]) << 32) | (int(r[2]) << 64) | (int(r[3]) << 96)
            rem = min(128, n - chunk * 128)
            left += bin(bits & ((1 << rem) - 1)).count("1")
        return left

    def split(cnt, key_hi, heap0, levels, stride):
        for l in range(levels):
            span = stride >> l
            for i in range(1 << l):
                nn = cnt[i * span]
                le = left_of(key_hi | ((heap0 << l) + i), nn)
                cnt[i * span], cnt[i * span + span // 2] = le, nn - le
    K = 1 << k
    cnt = [0] * K
    if k <= 6:
        cnt[0] = n_out
        split(cnt, 0, 1, k, K)
    else:
        G = max(1, min(64, K // 128))
        acc = [0] * 64
        for g in range(G):
            top = [0] * 64
            top[0] = n_out * (g + 1) // G - n_out * g // G
            split(top, (1 << 31) | (g << 8), 1, 6, 64)
            acc = [a + b for a, b in zip(acc, top)]
        sub = K >> 6
        for i in range(64):
            part = [0] * sub
            part[0] = acc[i]
            split(part, 0, 64 + i, k - 6, sub)
            cnt[i * sub:(i + 1) * sub] = part
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)


def test_strata_form_of_multinomial_resampling_is_multinomial_and_nearly_sorted():
    """Multinomial resampling as the device runs it by default (orc_resample_fixed_multinomial_strata): N iid uniform thresholds =
    counts per equal stratum ~ Multinomial(N; 1/K ..) -- popcounts of Philox bits down a binary tree, independent of the weights --
    and iid uniforms inside each stratum.  The counts against an independent restatement (both the one-tree and the two-part form);
    their law (chi-square); the ancestors against the definition evaluated in numpy; offspring counts with multinomial, not
    systematic, variance; ancestors sorted stratum by stratum."""
    for n_out, k in ((900, 2), (5000, 5), (16000, 6), (40000, 8), (150_000, 10)):           # (K = the power of two >= four times the tiles)
        assert O.lib().orc_strata_levels(n_out) == k
        assert np.array_equal(O.multinomial_strata(5, 3, n_out), _strata_offsets_in_python(5, 3, n_out, k))
    ch = []
    for st in range(40):
        m = np.diff(O.multinomial_strata(9, st, 200000).astype(np.int64))
        ex = 200000 / len(m)
        ch.append(((m - ex) ** 2 / ex).sum() / (len(m) - 1))
    assert abs(np.mean(ch) - 1.0) < 0.05                                             # (sd of the mean of 40 reduced chi-squares with 1023 dof: 0.007)
    rng = np.random.default_rng(3)
    n = 20000
    lw = rng.normal(size=n) * 2.0
    q = O.fix_weights(lw, lw.max() + 0.1)
    anc = O.resample_fixed_multinomial_strata(q, 11, 5)
    # the definition, in numpy: output s of stratum w takes tau = B_w + floor(v_s (B_w+1 - B_w)), B_w = floor(S w / K)
    k = O.lib().orc_strata_levels(n)
    offs = O.multinomial_strata(11, 5, n)
    S = int(q.astype(np.uint64).sum())
    cdf = np.cumsum(q.astype(np.uint64)).astype(object)
    base2 = (1 << 40) + (1 << 39)
    want = np.zeros(n, np.int64)
    for w in range(1 << k):
        b0, b1 = (S * w) >> k, (S * (w + 1)) >> k
        for s2 in range(int(offs[w]), int(offs[w + 1])):
            r = O.draw_block(11, s2 >> 1, base2 + 5)
            lo, hi = (int(r[2]), int(r[3])) if s2 & 1 else (int(r[0]), int(r[1]))
            v = (lo | ((hi >> 11) << 32)) << 11
            tau = b0 + ((v * (b1 - b0)) >> 64)
            lo_i, hi_i = 0, n
            while lo_i < hi_i:
                mid = (lo_i + hi_i) // 2
                if cdf[mid] > tau:
                    hi_i = mid
                else:
                    lo_i = mid + 1
            want[s2] = lo_i
    assert np.array_equal(anc, want)
    assert np.max(np.maximum.accumulate(anc) - anc) < 4096                              # nearly sorted: only inside a stratum's range
    counts = np.zeros(n)
    reps = 120
    for r in range(reps):
        counts += np.bincount(O.resample_fixed_multinomial_strata(q, 5, r), minlength=n)
    wgt = q / q.sum()
    big = n * wgt > 0.5
    zsc = (counts[big] / reps - n * wgt[big]) / np.sqrt(n * wgt[big] * (1 - wgt[big]) / reps)
    assert abs(zsc.std() - 1.0) < 0.05 and abs(zsc.mean()) < 0.05 and np.abs(zsc).max() < 6.0    # (systematic resampling would give sd << 1)


def test_table_weight_multinomial_is_the_integer_mass_strata_form_on_representable_weights():
    """orc_resample_table_multinomial (strata form on the table CDF of integer prefix counts: what the 3-state HMM on an every-step
    schedule resamples with) against orc_resample_fixed_multinomial_strata on weights both represent exactly (powers of two: B_w,
    tau_s and every C_k are then the same numbers in double and in 64-bit integers), and its law on generic table values."""
    rng = np.random.default_rng(1)
    x = rng.integers(0, 3, 60000).astype(np.int32)
    n = len(x)
    e = np.array([1.0, 0.5, 0.25])
    q = (e[x] * 2 ** 20).astype(np.uint32)
    for step in (3, 4):
        assert np.array_equal(O.resample_table_multinomial(x, e, 7, step), O.resample_fixed_multinomial_strata(q, 7, step))
    e2 = np.array([0.3123, 1.0, 0.0712345])
    counts = np.zeros(n)
    reps = 100
    for r in range(reps):
        a = O.resample_table_multinomial(x, e2, 5, r)
        assert a.min() >= 0 and a.max() < n
        counts += np.bincount(a, minlength=n)
    w = e2[x] / e2[x].sum()
    z = (counts / reps - n * w) / np.sqrt(n * w * (1 - w) / reps)
    assert abs(z.std() - 1.0) < 0.05 and abs(z.mean()) < 0.05 and np.abs(z).max() < 6.0


def test_oracle_repairs_a_generation_that_loses_its_bits():
    """The rule both sides state for the fixed-point form: a generation whose heaviest particle sits more than 6 nats below the
    reference known in advance is weighed against its exact maximum instead.  With an observation ~30 sd from every particle the
    evidence stays finite and equals the floating-point form's to the weights' resolution; without one nothing changes."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = np.array(z["lgssm100"][:14])
    obs[6] = 40.0
    a = O.smc(O.MODEL_LINEAR_GAUSSIAN_1D, obs, 5000, 8, O.RESAMPLE_SYSTEMATIC, 0.5)
    b = O.smc_ref(O.MODEL_LINEAR_GAUSSIAN_1D, obs, 5000, 8, O.REF_FLOATING_POINT, O.RESAMPLE_SYSTEMATIC, 0.5)
    assert np.isfinite(a["log_z"]) and abs(a["log_z"] - b["log_z"]) < 1e-6 and np.array_equal(a["resampled"], b["resampled"])
    assert np.mean(a["anc"] != b["anc"]) < 1e-3


def test_strata_form_is_independent_of_how_the_population_is_cut_into_shards():
    """Multinomial resampling, strata form, over shards of ONE population (orc_resample_fixed_multinomial_strata_shard /
    orc_resample_table_multinomial_shard: population-wide strata and output ids, each shard searching only the thresholds inside
    its own mass range): every output is claimed by exactly one shard, and the ancestors are the single-population form's."""
    rng = np.random.default_rng(1)
    for n in (700, 5000, 20000):
        lw = rng.normal(size=n) * 2
        q = O.fix_weights(lw, lw.max() + 0.3)
        tot = int(q.astype(np.uint64).sum())
        ref = O.resample_fixed_multinomial_strata(q, 5, 3)
        for cuts in ([n // 2], [n // 3, n // 3 + 5, n - 100], [10, 20, 30]):
            b = [0] + cuts + [n]
            got = np.full(n, -1, np.int64)
            for r in range(len(b) - 1):
                a = O.resample_fixed_multinomial_strata_shard(q[b[r]:b[r + 1]], 5, 3, int(q[:b[r]].astype(np.uint64).sum()), tot, n)
                m = a >= 0
                assert (got[m] == -1).all()
                got[m] = a[m] + b[r]
            assert np.array_equal(got, ref)
        x = rng.integers(0, 3, n).astype(np.int32)
        e = np.array([0.3, 1.0, 0.05])
        ref = O.resample_table_multinomial(x, e, 5, 3)
        b = [0, n // 3, n // 3 + 5, n - 100, n]
        total = np.bincount(x, minlength=3).astype(np.uint64)
        got = np.full(n, -1, np.int64)
        for r in range(4):
            a = O.resample_table_multinomial_shard(x[b[r]:b[r + 1]], e, 5, 3, np.bincount(x[:b[r]], minlength=3).astype(np.uint64), total, r == 3, n)
            m = a >= 0
            assert (got[m] == -1).all()
            got[m] = a[m] + b[r]
        assert np.array_equal(got, ref)


def _check_cut_plan(P, sizes, seed, step, table=None):
    from cpprob_amd import distributed as D
    world, N = len(sizes), sum(sizes)
    begins = np.concatenate([[0], np.cumsum(sizes)])
    k = O.lib().orc_strata_levels(N)
    K = 1 << k
    offs = O.multinomial_strata(seed, step, N)
    if table is None:
        B = [(P[-1] * w) >> k for w in range(K + 1)]
        tau = [int(t) for t in O.strata_thresholds_fixed(P[-1], seed, step, N)[0]]
    else:
        unit = np.ldexp(P[-1], -k)
        B = [float(w) * unit for w in range(K + 1)]
        tau = [float(t) for t in O.strata_thresholds_table(P[-1], seed, step, N)[0]]
    plan = D.StrataCutPlan(P, B, offs, begins, lambda a, b: tau[a:b], table_form=table is not None)
    src = [max(r for r in range(world) if P[r] <= t) for t in tau]                 # the definition: whose mass range holds the threshold
    dst = [max(r for r in range(world) if begins[r] <= s) for s in range(N)]
    for r in range(world):
        assert all(src[s] == r for s in range(plan.A[r], plan.Z[r + 1]))          # a rank's regular interval is its sources' alone
    n_mig = 0
    for d in range(world):
        col = 0
        for s in range(begins[d], begins[d + 1]):
            if src[s] != d:
                assert plan.column(d, s) == col                                    # immigrants take the annex columns in output order
                col += 1
        assert plan.arrivals(d) == col
        n_mig += col
    sends = sorted(sum([[(s, d, r) for s, d in plan.sends(r)] for r in range(world)], []))
    assert sends == sorted((s, dst[s], src[s]) for s in range(N) if src[s] != dst[s])
    return n_mig


def test_strata_cut_plan_equals_its_definition():
    """The exchange plan of a multinomial resampling step (cpprob_amd/distributed.py: StrataCutPlan = the host statement of
    csrc/strata_cut.hpp + exchange_cut_kernel): regular intervals + tables over the strata the ranks' boundaries cut, against the
    definition (source rank of every threshold by comparison with the ranks' bounds; annex columns by counting) -- even and wildly
    uneven masses, ranks without mass, several boundaries inside one stratum, boundaries exactly ON strata bounds, both forms."""
    rng = np.random.default_rng(3)
    moved = 0
    for trial in range(40):
        world = int(rng.integers(2, 7))
        sizes = [int(x) for x in rng.integers(1, 3000, world)]
        kind = trial % 4
        if kind == 0:
            masses = [int(s * 2**30 * rng.uniform(0.7, 1.3)) for s in sizes]
        elif kind == 1:
            masses = [int(rng.integers(1, 2**20)) if rng.random() < 0.5 else int(2**40 * rng.random()) for _ in sizes]
            masses[0] += 2**24
        elif kind == 2:
            masses = [0 if rng.random() < 0.4 else int(2**36 * rng.random()) + 1 for _ in sizes]
            masses[int(rng.integers(0, world))] += 2**24
        else:
            K = 1 << O.lib().orc_strata_levels(sum(sizes))
            cuts = sorted(rng.integers(0, K + 1, world - 1))
            Pb = [0] + [int(c) * 2**20 for c in cuts] + [K * 2**20]
            masses = [Pb[i + 1] - Pb[i] for i in range(world)]
        P = [0]
        for m in masses:
            P.append(P[-1] + int(m))
        moved += _check_cut_plan(P, sizes, 11 + trial, 1 + trial % 5)
    assert moved > 10000
    moved = 0
    for trial in range(16):
        world = int(rng.integers(2, 6))
        sizes = [int(x) for x in rng.integers(1, 3000, world)]
        x = rng.integers(0, 3, sum(sizes))
        e = np.array([1.0, 0.5, 0.25]) if trial % 3 == 0 else np.exp(-rng.random(3) * 3)          # (exactly representable: bounds may coincide)
        begins = np.concatenate([[0], np.cumsum(sizes)])
        P = [O.table_cdf(np.bincount(x[:b], minlength=3), e) for b in begins]
        moved += _check_cut_plan(P, sizes, 100 + trial, 1 + trial % 3, table=e)
    assert moved > 1000
