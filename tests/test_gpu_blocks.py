"""GPU parity of the building blocks, through the C ABI, against the CPU oracle.

Bars: raw Philox words, discrete draws, ancestor indices on exactly-summable weights: bit-exact.
Floating point (normal draws, logpdf, logsumexp, moments): relative/absolute 1e-12 (device libm
vs glibc differ by <= 2 ulp in log/exp/sinpi; FMA contraction by 1 ulp).
"""
import json
import os

import numpy as np
import pytest

from devmem import dcat, dtensor, dzeros, dzeros_like  # noqa: F401

from oracle import oracle as O
import cpprob_amd as cp  # noqa: E402

pytestmark = pytest.mark.gpu

FP_TOL = 1e-12


def _t(a, dtype=None):
    return dtensor(a, dtype)


def test_philox_words_bit_exact_vs_oracle_and_rocrand(engine, golden_dir):
    import torch
    with open(os.path.join(golden_dir, "philox_rocrand.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        out = dzeros(4, dtype=torch.int32)
        engine.philox_blocks(c["seed"], c["pid"], c["draw"], out)
        engine.sync()
        words = out.cpu().numpy().view(np.uint32)
        assert words.tolist() == c["words"]
        assert O.draw_block(c["seed"], c["pid"], c["draw"]).tolist() == c["words"]
        # Box-Muller of that block: both rocRAND outputs (host libm there: not bit-exact)
        z = O.box_muller(c["words"])
        assert abs(z[0] - float.fromhex(c["normal_x"])) < 1e-13 and abs(z[1] - float.fromhex(c["normal_y"])) < 1e-13
    n = 100003
    out = dzeros(4 * n, dtype=torch.int32)
    engine.philox_blocks(777, 5, 9, out)
    engine.sync()
    got = out.cpu().numpy().view(np.uint32).reshape(n, 4)
    for i in (0, 1, 63, 64, 4095, n - 1):
        assert got[i].tolist() == O.draw_block(777, 5 + i, 9).tolist()


def test_draws_match_oracle(engine):
    import torch
    n = 20000
    L = O.lib()
    out = dzeros(n, dtype=torch.float64)
    engine.draw_normal(42, 1000, 3, 1.0, np.sqrt(5), out)
    engine.sync()
    ref = np.array([L.orc_draw_normal(42, 1000 + i, 3, 1.0, np.sqrt(5)) for i in range(n)])
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=FP_TOL, atol=FP_TOL)

    outi = dzeros(n, dtype=torch.int32)
    engine.draw_uniform_smallint(42, 0, 0, 0, 2, outi)
    engine.sync()
    ref = np.array([L.orc_draw_smallint(42, i, 0, 0, 2) for i in range(n)])
    assert np.array_equal(outi.cpu().numpy(), ref)

    for row in ([0.1, 0.5, 0.4], [0.2, 0.2, 0.6], [0.15, 0.15, 0.7], [3.0, 1.0, 2.0, 2.0]):
        engine.draw_discrete(9, 10, 7, row, outi)
        engine.sync()
        w = np.array(row)
        ref = np.array([L.orc_draw_discrete(9, 10 + i, 7, w, len(row)) for i in range(n)])
        assert np.array_equal(outi.cpu().numpy(), ref)

    engine.draw_uniform_real(1, 0, 2, -3.0, 5.0, out)
    engine.sync()
    ref = np.array([L.orc_draw_uniform_real(1, i, 2, -3.0, 5.0) for i in range(n)])
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=FP_TOL, atol=FP_TOL)

    for lam in (0.3, 4.0, 25.0):
        engine.draw_poisson(5, 3, 1, lam, outi)
        engine.sync()
        ref = np.array([L.orc_draw_poisson(5, 3 + i, 1, lam) for i in range(n)])
        got = outi.cpu().numpy()
        assert np.mean(got != ref) < 1e-4          # exp / running products differ by ulps: a draw on a CDF boundary may flip
        assert abs(got.mean() - lam) < 5 * np.sqrt(lam / n) and abs(got.var() - lam) < 0.1 * lam + 0.05


def test_logpdf_normal_reference_grid(engine, golden_dir):
    """The grid of the reference's own test (tests/cpprob/logpdf.cpp:23-35), eps 1e-8 there."""
    import torch
    g = np.load(os.path.join(golden_dir, "logpdf_grid.npz"))
    grid = g["normal_grid"]
    x, mean, sigma = (_t(grid[:, k].copy()) for k in range(3))
    out = dzeros_like(x)
    engine.logpdf_normal(x, mean, sigma, out)
    engine.sync()
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, g["normal_expected"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.exp(got), np.exp(g["normal_expected"]), rtol=0, atol=1e-8)
    L = O.lib()
    ref = np.array([L.orc_normal_logpdf(*r) for r in grid[::7]])
    np.testing.assert_allclose(got[::7], ref, rtol=FP_TOL, atol=FP_TOL)


def test_logpdf_edge_cases(engine):
    import torch
    inf = float("inf")
    x = _t([1.0, 2.0, inf, -inf, 0.0])
    mean = _t([1.0, 1.0, 0.0, 0.0, 0.0])
    sigma = _t([0.0, 0.0, 1.0, 1.0, 2.0])
    out = dzeros_like(x)
    engine.logpdf_normal(x, mean, sigma, out)
    engine.sync()
    got = out.cpu().numpy()
    L = O.lib()
    ref = [L.orc_normal_logpdf(a, b, c) for a, b, c in zip(x.cpu().numpy(), mean.cpu().numpy(), sigma.cpu().numpy())]
    assert got[0] == 0.0 and got[1] == -inf and got[2] == -inf and got[3] == -inf
    np.testing.assert_allclose(got, ref, rtol=FP_TOL)


def test_logpdf_other_functors(engine, golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, "logpdf_grid.npz"))
    u = g["uniform_grid"]
    x, a, b = (_t(u[:, k].copy()) for k in range(3))
    out = dzeros_like(x)
    engine.logpdf_uniform_real(x, a, b, out)
    engine.sync()
    got = out.cpu().numpy()
    exp = g["uniform_expected"]
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], exp[fin], atol=1e-8)
    p = g["poisson_grid"]
    xi = _t(p[:, 0].astype(np.int32))
    lam = _t(p[:, 1].copy())
    out = dzeros(len(p), dtype=torch.float64)
    engine.logpdf_poisson(xi, lam, out)
    engine.sync()
    np.testing.assert_allclose(out.cpu().numpy(), g["poisson_expected"], atol=1e-9)
    xs = _t(np.array([-1, 0, 1, 2, 3], np.int32))
    out = dzeros(5, dtype=torch.float64)
    engine.logpdf_uniform_smallint(xs, 0, 2, out)
    engine.sync()
    L = O.lib()
    np.testing.assert_allclose(out.cpu().numpy(), [L.orc_uniform_smallint_logpdf(int(v), 0, 2) for v in xs.cpu().numpy()], rtol=FP_TOL)
    engine.logpdf_discrete(xs, [0.1, 0.5, 0.4], out)
    engine.sync()
    w = np.array([0.1, 0.5, 0.4])
    np.testing.assert_allclose(out.cpu().numpy(), [L.orc_discrete_logpdf(int(v), w, 3) for v in xs.cpu().numpy()], rtol=FP_TOL)


@pytest.mark.parametrize("n", [1, 3, 64, 1023, 1024, 1025, 4095, 4096, 4097, 8193, 300001])
def test_logsumexp_ess_and_moments(engine, n):
    rng = np.random.default_rng(n)
    logw = rng.normal(size=n) * 3 - 700.0      # far from 0: needs the max shift
    x = rng.normal(size=n) * 2 + 1
    m, lse, ess = engine.logsumexp_ess(_t(logw))
    assert m == logw.max()
    assert abs(lse - O.logsumexp(logw)) < 1e-11
    ref = O.weighted_moments(x, logw)
    got = engine.weighted_moments(_t(x), _t(logw))
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-11)
    xi = rng.integers(0, 3, n).astype(np.int32)
    np.testing.assert_allclose(engine.weighted_hist(_t(xi), _t(logw), 3), O.weighted_hist(xi, logw, 3), rtol=1e-10, atol=1e-12)


def test_logsumexp_with_minus_inf_entries(engine):
    logw = np.array([-np.inf, 0.0, -np.inf, np.log(3.0)] * 600)
    m, lse, ess = engine.logsumexp_ess(_t(logw))
    assert abs(lse - O.logsumexp(logw)) < 1e-12
    assert abs(ess - 1.0 / (600 * (0.25 ** 2 + 0.75 ** 2) / 600 ** 2)) < 1e-6


@pytest.mark.parametrize("kind", [O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL])
@pytest.mark.parametrize("n,alive", [(1, 1.0), (5, 0.6), (1024, 0.3), (1025, 0.3), (4096, 0.3), (4097, 0.7), (5000, 0.01), (8192 + 3, 0.2),
                                     (262144 + 17, 0.5), (262144 + 17, 1.0)])
def test_resample_exact_weights_bit_exact(engine, kind, n, alive):
    """log-weights in {0, -inf}: exp(lw - max) is exactly 1 or 0 on both sides and every partial
    sum is an exact integer in any summation order, so ancestors must match the sequential
    oracle bit for bit (index work)."""
    import torch
    rng = np.random.default_rng(1000 + n)
    logw = np.where(rng.random(n) < alive, 0.0, -np.inf)
    logw[rng.integers(0, n)] = 0.0
    anc = dzeros(n, dtype=torch.int32)
    engine.resample(kind, _t(logw), 31337, 4, anc)
    engine.sync()
    ref = O.resample(kind, logw, 31337, 4)
    got = anc.cpu().numpy()
    assert np.array_equal(got, ref)
    assert np.all(logw[got] == 0.0)


@pytest.mark.parametrize("n,spread", [(1_200_000, 1.5), (4_300_000, 0.3), (2_000_000, 4.0)])
def test_resample_continuous_weights_flip_only_across_a_boundary_within_1e9(engine, n, spread):
    """Continuous weights keep a tolerance on index work (the parallel scan sums the CDF in another order than the sequential
    oracle): every ancestor that differs must be the ADJACENT source, and the output's threshold must sit within 1e-9 * W of the
    CDF value at the boundary between the two -- i.e. only a rounding-sized difference of the CDF can have moved it.  The flip
    rate itself is documented in DESIGN.md (about 1e-6 per output at these sizes)."""
    import torch
    rng = np.random.default_rng(n)
    logw = rng.normal(size=n) * spread
    anc = dzeros(n, dtype=torch.int32)
    engine.resample(O.RESAMPLE_SYSTEMATIC, _t(logw), 4242, 3, anc)
    engine.sync()
    got = anc.cpu().numpy().astype(np.int64)
    ref = O.resample(O.RESAMPLE_SYSTEMATIC, logw, 4242, 3).astype(np.int64)
    flips = np.nonzero(got != ref)[0]
    assert len(flips) < 1e-5 * n + 3, len(flips)
    assert np.all(np.abs(got[flips] - ref[flips]) == 1)
    # the CDF in extended precision, and the systematic offset recovered from the oracle's own ancestors:
    # ref[j] = a  <=>  C[a-1] <= (j + u) W / N < C[a]
    w = np.exp((logw - logw.max()).astype(np.longdouble))
    C = np.cumsum(w)
    W = C[-1]
    j = np.arange(n, dtype=np.longdouble)
    lo = np.where(ref > 0, C[np.maximum(ref - 1, 0)], 0) * n / W - j
    hi = C[ref] * n / W - j
    u_lo, u_hi = lo.max(), hi.min()
    assert u_lo - u_hi < 1e-6 and 0.0 <= float(u_hi) and float(u_lo) < 1.0, (u_lo, u_hi)     # (a consistent offset, up to the oracle's own rounding)
    u = (u_lo + u_hi) / 2
    b = np.minimum(got[flips], ref[flips])                                  # the boundary lies behind source b
    gap = np.abs(C[b] - (flips.astype(np.longdouble) + u) * W / n)
    assert np.all(gap < 1e-9 * W), (float(gap.max() / W) if len(gap) else 0.0)


@pytest.mark.parametrize("kind", [O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL])
def test_resample_generic_weights(engine, kind):
    import torch
    n = 200000
    rng = np.random.default_rng(5)
    logw = rng.normal(size=n) * 2.0
    anc = dzeros(n, dtype=torch.int32)
    engine.resample(kind, _t(logw), 99, 1, anc)
    engine.sync()
    got = anc.cpu().numpy()
    ref = O.resample(kind, logw, 99, 1)
    mism = got != ref
    assert mism.mean() < 1e-4
    assert np.all(np.abs(got[mism].astype(np.int64) - ref[mism]) <= 1)
    if kind != O.RESAMPLE_MULTINOMIAL:
        assert np.all(np.diff(got) >= 0)           # sortedness
    # offspring counts follow the weights: |count_k - N W_k| < 1 for systematic
    if kind == O.RESAMPLE_SYSTEMATIC:
        w = np.exp(logw - logw.max()); w /= w.sum()
        cnt = np.bincount(got, minlength=n)
        assert np.all(np.abs(cnt - n * w) < 1.0 + 1e-6)


@pytest.mark.parametrize("kind", [O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL])
def test_smc_bookkeep_decides_and_resamples_on_the_device(engine, kind):
    """cpprob_hip_smc_bookkeep = logsumexp/ESS + ESS test + evidence + ancestors without a host round trip: against the
    oracle's estimators and resampler for a resampling step, identity ancestors for a non-resampling one."""
    import torch
    n = 50000
    rng = np.random.default_rng(9)
    ess = dzeros(3, dtype=torch.float64)
    res = dzeros(3, dtype=torch.int32)
    lz = dzeros(1, dtype=torch.float64)
    anc = dzeros(n, dtype=torch.int32)
    want_lz = 0.0
    for step, (spread, last) in enumerate([(2.0, False), (0.01, False), (1.0, True)]):
        logw = rng.normal(size=n) * spread                      # spread 2: ESS << N/2 -> resample; 0.01: ESS ~ N -> keep
        engine.smc_bookkeep(kind, _t(logw), 77, step, last, 0.5, ess, res, lz, anc)
        engine.sync()
        ref_lse, ref_ess = O.logsumexp(logw), O.weighted_moments(np.zeros(n), logw)[3]
        got_anc = anc.cpu().numpy()
        assert abs(float(ess[step]) - ref_ess) < 1e-6 * ref_ess
        if step == 0:
            assert int(res[0]) == 1
            ref = O.resample(kind, logw, 77, 1)
            mism = got_anc != ref
            assert mism.mean() < 1e-4 and np.all(np.abs(got_anc[mism].astype(np.int64) - ref[mism]) <= 1)
            want_lz += ref_lse - np.log(n)
        elif step == 1:
            assert int(res[1]) == 0 and np.array_equal(got_anc, np.arange(n))
        else:
            assert int(res[2]) == 0                              # the last step never resamples, but closes the evidence
            want_lz += ref_lse - np.log(n)
        assert abs(float(lz[0]) - want_lz) < 1e-10


def test_resample_degenerate_weight(engine):
    import torch
    n = 10000
    logw = np.full(n, -np.inf)
    logw[1234] = 0.0
    anc = dzeros(n, dtype=torch.int32)
    for kind in (O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL):
        engine.resample(kind, _t(logw), 1, 0, anc)
        engine.sync()
        assert np.all(anc.cpu().numpy() == 1234)


def test_resample_subrange_matches_full(engine):
    """Sharded use: outputs [j0, j0+n_out) of n_total positions."""
    import torch
    n = 50000
    rng = np.random.default_rng(8)
    logw = rng.normal(size=n)
    full = dzeros(n, dtype=torch.int32)
    engine.resample(O.RESAMPLE_SYSTEMATIC, _t(logw), 3, 2, full)
    part = dzeros(7777, dtype=torch.int32)
    engine.resample(O.RESAMPLE_SYSTEMATIC, _t(logw), 3, 2, part, j0=12345, n_total_out=n)
    engine.sync()
    assert np.array_equal(part.cpu().numpy(), full.cpu().numpy()[12345:12345 + 7777])


def test_gather(engine):
    import torch
    n = 100000
    rng = np.random.default_rng(0)
    idx = rng.integers(0, n, n).astype(np.int32)
    src = rng.normal(size=n)
    dst = dzeros(n, dtype=torch.float64)
    engine.gather(_t(src), _t(idx), dst)
    engine.sync()
    assert np.array_equal(dst.cpu().numpy(), src[idx])
    srci = rng.integers(0, 3, n).astype(np.int32)
    dsti = dzeros(n, dtype=torch.int32)
    engine.gather(_t(srci), _t(idx), dsti)
    engine.sync()
    assert np.array_equal(dsti.cpu().numpy(), srci[idx])


def _ulp_err(got, ref_ld):
    """|got - ref| in units of the fp64 ulp of ref (ref in 80-bit long double: 11 more bits than the result it judges)."""
    ref_ld = np.asarray(ref_ld, np.longdouble)
    got_ld = np.asarray(got, np.float64).astype(np.longdouble)
    ulp = np.spacing(np.abs(ref_ld.astype(np.float64))).astype(np.longdouble)
    return np.abs(got_ld - ref_ld) / ulp


def test_fastmath_log01_edges_and_random_points(engine):
    """cpprob/detail/fastmath.hpp::log01 on its whole domain [2^-53, 1]: the edges, the binade boundaries, the neighbourhood of
    sqrt(1/2) (the reduction's branch) and 10^6 random points, against 80-bit logl: <= 1 ulp (tools/fit_math.py: 0.67)."""
    import torch
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision for the reference"
    rng = np.random.default_rng(11)
    edges = [2.0 ** -53, np.nextafter(2.0 ** -53, 1), 1.0, np.nextafter(1.0, 0), 0.5, np.nextafter(0.5, 0), np.nextafter(0.5, 1),
             np.sqrt(0.5), np.nextafter(np.sqrt(0.5), 0), np.nextafter(np.sqrt(0.5), 1)] + [2.0 ** -k for k in range(1, 54)]
    u = np.concatenate([np.array(edges), (rng.integers(1, 2 ** 53, 600000) * 2.0 ** -53), 2.0 ** -rng.uniform(0, 53, 400000)])
    u = np.clip(u, 2.0 ** -53, 1.0)
    x = _t(u)
    out = dzeros_like(x)
    engine.fastmath(0, x, out)
    engine.sync()
    got = out.cpu().numpy()
    assert got[2] == 0.0                                        # log(1) is exactly 0
    ref = np.log(u.astype(np.longdouble))
    err = _ulp_err(got[ref != 0], ref[ref != 0])
    assert err.max() <= 1.0, err.max()


def test_fastmath_sincospi02_edges_and_random_points(engine):
    """sincospi02 on (0, 2]: the quadrant boundaries k/4 and their neighbours, 0+ and 2, 10^6 random points, against 80-bit
    sinl / cosl of the exactly reduced argument: sin <= 1 ulp of the result, cos <= 1.05 (its worst case, 1.03, sits at the quadrant
    edges |t| -> 1/4: cpprob/detail/fastmath.hpp), exact zeros / ones where the reference has them."""
    import torch
    rng = np.random.default_rng(12)
    q = np.arange(1, 9) / 4.0
    edges = np.concatenate([[2.0 ** -53, 2.0 ** -60, 2.0], q, np.nextafter(q, 0), np.nextafter(q[:-1], 3)])
    w = np.concatenate([edges, rng.uniform(0, 2, 700000), (rng.integers(1, 2 ** 32, 300000) * 2.0 ** -31)])
    w = np.clip(w, 2.0 ** -60, 2.0)
    x = _t(w)
    sn, cs = dzeros_like(x), dzeros_like(x)
    engine.fastmath(1, x, sn, cs)
    engine.sync()
    sn, cs = sn.cpu().numpy(), cs.cpu().numpy()
    r = np.rint(w + w)
    t = (w - 0.5 * r).astype(np.longdouble)                     # exact in fp64
    pi = np.longdouble("3.14159265358979323846264338327950288")
    s0, c0 = np.sin(pi * t), np.cos(pi * t)
    i = r.astype(np.int64) & 3
    ref_s = np.where(i == 0, s0, np.where(i == 1, c0, np.where(i == 2, -s0, -c0)))
    ref_c = np.where(i == 0, c0, np.where(i == 1, -s0, np.where(i == 2, -c0, s0)))
    for got, ref in ((sn, ref_s), (cs, ref_c)):
        nz = t != 0                                              # (where the reduced argument is exactly 0 the values are exactly 0 / +-1)
        big = np.abs(ref) > 1e-300
        assert _ulp_err(got[nz & big], ref[nz & big]).max() <= 1.05
    z = t == 0
    assert np.all(np.abs(sn[z]) + np.abs(cs[z]) == 1.0) and np.all((sn[z] == 0) | (cs[z] == 0))
    assert sn[2] == 0.0 and cs[2] == 1.0                        # w = 2


def test_fix_weight_against_extended_precision_and_the_oracle(engine):
    """The integer weight q = min(rint(exp(lw - R) 2^32), 2^32 - 1) (cpprob/detail/fixed_mass.hpp: fix_weight).  Two references: the
    oracle's restatement -- which shares the kernel's exp polynomial, so the two must agree bit for bit -- and, independently of that
    polynomial, rint(expl(x) 2^32) in x87 extended precision: exp_nonpos is faithfully rounded (<= 0.66 ulp), so the integers can
    differ only where exp(x) 2^32 lies within ~2^-21 of a half-integer, and then by ONE unit.  10^6 random points over the 23 nats
    the form resolves, the clamps at both ends, exact zeros and the neighbourhoods of small half-integers."""
    import torch
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision for the reference"
    rng = np.random.default_rng(14)
    halves = np.log((np.arange(0, 4000) + 0.5) * 2.0 ** -32)            # exp(x) 2^32 = k + 1/2: the rounding boundaries of the small integers
    edges = np.concatenate([[0.0, -0.0, -1e-300, -2.0 ** -53, -2.0 ** -33, -2.0 ** -32, -22.18, -22.1807, -23.0, -24.0, -40.0, -745.0, -1000.0, -1e9, -np.inf],
                            halves, np.nextafter(halves, 0), np.nextafter(halves, -1e9)])
    xv = np.concatenate([edges, -rng.uniform(0, 23, 700000), -rng.uniform(0, 6, 200000), -(2.0 ** -rng.uniform(0, 40, 100000))])
    x = _t(xv)
    out = dzeros_like(x)
    engine.fastmath(3, x, out)
    engine.sync()
    got = out.cpu().numpy()
    assert np.array_equal(got, O.fix_weights(xv, 0.0).astype(np.float64))                      # the oracle states the same arithmetic
    assert got[0] == 4294967295.0 and got[1] == 4294967295.0 and got[13] == 0.0 and got[14] == 0.0       # 0, -0 | -1e9, -inf
    ref = np.rint(np.exp(np.maximum(xv, -1000.0).astype(np.longdouble)) * np.longdouble(4294967296.0))
    ref = np.minimum(ref, np.longdouble(4294967295.0)).astype(np.float64)
    d = np.abs(got - ref)
    assert d.max() <= 1.0, d.max()
    # how many: among the points placed ON the rounding boundaries (x = log((k + 1/2) 2^-32) and its neighbours) a tie may fall either
    # way; among the 10^6 random points only where exp(x) 2^32 happens to lie within the polynomial's error of a half-integer
    n_edge, n_rand = int(np.count_nonzero(d[: len(edges)])), int(np.count_nonzero(d[len(edges):]))
    assert n_rand <= 20, (n_rand, n_edge)
    print("fix_weight: %d of %d random points and %d of %d boundary points differ by one unit from the x87 reference" % (n_rand, len(xv) - len(edges), n_edge, len(edges)))


def test_fastmath_exp_nonpos_edges_and_random_points(engine):
    """exp_nonpos on [-745, 0]: 0, -0, the underflow edge, multiples of ln 2 / 2 (the reduction's boundaries) and 10^6 random
    points, against 80-bit expl: <= 1 ulp in the normal range (0.66); denormal results within one denormal spacing."""
    import torch
    rng = np.random.default_rng(13)
    ln2 = np.log(2.0)
    ks = np.arange(0, 2140)
    edges = np.concatenate([[0.0, -0.0, -745.0, -744.44, -708.39, -708.4, -1e-300, -2.0 ** -60], -ks * ln2 / 2, np.nextafter(-ks * ln2 / 2, 0), np.nextafter(-ks * ln2 / 2, -1e9)])
    xv = np.concatenate([edges, -rng.uniform(0, 745, 500000), -rng.uniform(0, 40, 400000), -(2.0 ** -rng.uniform(0, 60, 100000))])
    xv = np.clip(xv, -745.0, 0.0)
    x = _t(xv)
    out = dzeros_like(x)
    engine.fastmath(2, x, out)
    engine.sync()
    got = out.cpu().numpy()
    assert got[0] == 1.0 and got[1] == 1.0
    ref = np.exp(xv.astype(np.longdouble))
    normal = ref >= np.longdouble(2.0) ** -1022
    assert _ulp_err(got[normal], ref[normal]).max() <= 1.0
    den = ~normal
    assert np.all(np.abs(got[den].astype(np.longdouble) - ref[den]) <= np.longdouble(2.0) ** -1074)
    assert np.all(got >= 0) and np.all(got <= 1.0)


@pytest.mark.parametrize("n,n_cols", [(1, 1), (1000, 3), (70001, 16), (4097, 130)])
def test_weighted_columns_equal_the_per_column_calls(engine, n, n_cols):
    """EmpiricalDistribution on the device for many predict hits at once (cpprob_hip_weighted_moments_columns / _hist_columns: one
    normalisation, one read-out launch, one synchronisation): the same numbers, bit for bit, as one call per column, and the
    oracle's (empirical_distribution.hpp:30-81) to rounding.  130 columns: more than one chunk of 64."""
    rng = np.random.default_rng(n + n_cols)
    logw = rng.normal(size=n) * 3.0 - 50.0
    x = rng.normal(size=(n_cols, n)) * 2.0 + 1.0
    xi = rng.integers(0, 5, size=(n_cols, n)).astype(np.int32)
    got = engine.weighted_moments_columns(_t(x), _t(logw))
    goth = engine.weighted_hist_columns(_t(xi), _t(logw), 5)
    for k in range(n_cols):
        one = engine.weighted_moments(_t(x[k]), _t(logw))
        assert np.array_equal(got[k], np.array(one))
        assert np.array_equal(goth[k], engine.weighted_hist(_t(xi[k]), _t(logw), 5))
    ref = np.array([O.weighted_moments(x[k], logw) for k in range(min(n_cols, 4))])
    np.testing.assert_allclose(got[:len(ref), :2], ref[:, :2], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(goth[0], O.weighted_hist(xi[0], logw, 5), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n,T,hits", [(1, 1, [0]), (1000, 5, [0, 1, 1, 3, 4]), (70001, 16, list(range(16))), (4097, 7, [0] * 40 + [2] * 50 + [6] * 40)])
def test_lineage_statistics_equal_gather_then_columns(engine, n, T, hits):
    """cpprob_hip_lineage_moments / _hist: the statistics of per-step records along the final particles' lineages, taken on the walk.
    Bit for bit what gathering the traces (cpprob_hip_lineage_gather) and reading the gathered columns returns, and the traces
    themselves are a numpy walk's.  Records per generation: none, one, several; 130 records: more than one chunk of 64; one of the
    ancestor rows does not count (its step did not resample)."""
    rng = np.random.default_rng(n + T)
    H = len(hits)
    anc = rng.integers(0, n, size=(T, n)).astype(np.int32)
    res = np.ones(T, dtype=np.int32)
    if T > 2:
        res[1] = 0
    logw = rng.normal(size=n) * 3.0 - 50.0
    x = rng.normal(size=(H, n)) * 2.0 + 1.0
    xi = rng.integers(0, 5, size=(H, n)).astype(np.int32)
    d_anc, d_res, d_x, d_xi, d_lw = _t(anc), _t(res), _t(x), _t(xi), _t(logw)
    out_x, out_xi = dzeros_like(d_x), dzeros_like(d_xi)
    engine.lineage_gather(d_anc, d_res, d_x, hits, out_x)
    engine.lineage_gather(d_anc, d_res, d_xi, hits, out_xi)
    engine.sync()
    # the walk in numpy
    idx = np.arange(n)
    want_x, want_xi = np.empty_like(x), np.empty_like(xi)
    for t in range(T - 1, -1, -1):
        for h in range(H):
            if hits[h] == t:
                want_x[h] = x[h, idx]
                want_xi[h] = xi[h, idx]
        if t > 0 and res[t - 1]:
            idx = anc[t, idx]
    assert np.array_equal(out_x.cpu().numpy(), want_x) and np.array_equal(out_xi.cpu().numpy(), want_xi)
    got = engine.lineage_moments(d_anc, d_res, d_x, hits, d_lw)
    goth = engine.lineage_hist(d_anc, d_res, d_xi, hits, d_lw, 5)
    assert np.array_equal(got, engine.weighted_moments_columns(out_x, d_lw))
    assert np.array_equal(goth, engine.weighted_hist_columns(out_xi, d_lw, 5))
    # a read-back of the caller's rides the read-out's result (one launch into pinned memory, one wait): the bytes arrive with it
    rider = np.zeros(min(n, 4096), np.float64)
    engine.readback_with_next_result(d_lw, rider)
    assert np.array_equal(engine.lineage_hist(d_anc, d_res, d_xi, hits, d_lw, 5), goth) and np.array_equal(rider, logw[:rider.size])
    rider[:] = 0.0
    assert np.array_equal(engine.lineage_moments(d_anc, d_res, d_x, hits, d_lw), got) and not rider.any()      # (it rode once)
    with pytest.raises(cp.CpprobHipError):
        engine.readback_with_next_result(d_lw, np.zeros(3, np.uint8))                                             # whole 4-byte words only
    np.testing.assert_allclose(got[0, :2], np.array(O.weighted_moments(want_x[0], logw))[:2], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("n", [1, 1000, 70001, 1_000_000, 5_000_000])
def test_fixed_point_bookkeeping_matches_the_integer_comb(engine, n):
    """cpprob_hip_smc_bookkeep_fixed (what the unchanged-model path runs between two launches of the model body): decisions as the
    ESS test asks, the identity where the step does not resample, and the ancestors of a resampling step `array_equal` to the oracle's
    integer comb on q_i = min(rint(exp(lw_i - max lw) 2^32), 2^32 - 1).  One, two and three levels of the hierarchy of sums."""
    import torch
    rng = np.random.default_rng(n)
    ess = dzeros(3, dtype=torch.float64)
    res = dzeros(3, dtype=torch.int32)
    lz = dzeros(1, dtype=torch.float64)
    anc = dzeros(n, dtype=torch.int32)
    for step, (spread, last) in enumerate([(2.0, False), (0.01, False), (1.0, True)]):
        logw = rng.normal(size=n) * spread - 30.0
        engine.smc_bookkeep_fixed(_t(logw), 77, step, last, 0.5, ess, res, lz, anc)
        engine.sync()
        got = anc.cpu().numpy()
        if step == 0:
            assert np.array_equal(got, O.resample_fixed_systematic(O.fix_weights(logw, logw.max()), 77, 1))
            assert n == 1 or int(res[0]) == 1
        elif step == 1 and n > 1:
            assert int(res[1]) == 0 and np.array_equal(got, np.arange(n))
    assert int(res[2]) == 0
