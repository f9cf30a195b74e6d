# Randomised building blocks against the oracle: extended coverage of tests/test_gpu_blocks.py (not part of the suite; needs the GPU):
#   python tools/fuzz_blocks.py FIRST LAST                  e.g. 0 6000: ~2 min on the box
import os, sys, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import cpprob_amd as cp
from oracle import oracle as O
from devmem import dtensor, dzeros
eng = cp.Engine(0)
def _t(a): return dtensor(a)
bad = 0
t0 = time.time()
for it in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(it)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 1023, 1024, 1025, 4097, 65535, 65536, 65537, 262145, 1000003, int(rng.integers(1, 3_000_000))]))
    kind_w = int(rng.integers(0, 5))
    logw = rng.normal(size=n) * float(rng.choice([0.0, 1e-3, 1.0, 5.0, 30.0])) - float(rng.choice([0.0, 50.0, 700.0]))
    if kind_w == 1 and n > 2: logw[rng.integers(0, n, size=max(1, n // 3))] = -np.inf
    if kind_w == 2: logw[:] = -np.inf; logw[int(rng.integers(0, n))] = -3.0
    if kind_w == 3 and n > 1: logw[int(rng.integers(0, n))] += 40.0
    seed, step = int(rng.integers(0, 2**31)), int(rng.integers(0, 100))
    tag = "it %d n %d kind_w %d" % (it, n, kind_w)
    try:
        # fixed-point bookkeeping: exact
        ess = dzeros(1, dtype=torch.float64); res = dzeros(1, dtype=torch.int32); lz = dzeros(1, dtype=torch.float64); anc = dzeros(n, dtype=torch.int32)
        eng.smc_bookkeep_fixed(_t(logw), seed, 0, False, 2.0, ess, res, lz, anc); eng.sync()
        ref = O.resample_fixed_systematic(O.fix_weights(logw, logw.max()), seed, 1)
        assert np.array_equal(anc.cpu().numpy(), ref), tag + " bookkeep_fixed"
        # floating-point resamplers
        for kind in (O.RESAMPLE_SYSTEMATIC, O.RESAMPLE_STRATIFIED, O.RESAMPLE_MULTINOMIAL):
            if kind == O.RESAMPLE_MULTINOMIAL and n > 400000: continue
            eng.resample(kind, _t(logw), seed, step, anc); eng.sync()
            got = anc.cpu().numpy(); ref = O.resample(kind, logw, seed, step)
            mism = np.nonzero(got != ref)[0]
            assert len(mism) < 1e-4 * n + 3, tag + " resample %d: %d flips" % (kind, len(mism))
            w = np.exp(logw - logw.max())
            for j in mism:                                           # a boundary flip: neighbours, or across sources whose weight vanishes in the CDF
                a, b = sorted((int(got[j]), int(ref[j])))
                # (a sequential fp64 CDF of n terms carries up to n eps W of rounding: mass below that is not resolved by either side)
                assert b - a == 1 or w[a + 1:b].sum() < 1e-9 * w.sum(), tag + " resample %d: output %d %d vs %d" % (kind, j, got[j], ref[j])
            assert np.all(np.isfinite(logw[got])), tag + " resample %d picked a zero-weight source" % kind
        # weighted moments / logsumexp
        x = rng.normal(size=n) * 3.0
        got = np.array(eng.weighted_moments(_t(x), _t(logw))); ref = np.array(O.weighted_moments(x, logw))
        # (variance = raw moment - mean^2, stats_printer.hpp:78-81: cancellation against mean^2 when one particle holds the mass)
        assert np.allclose(got[[0, 2, 3]], ref[[0, 2, 3]], rtol=1e-8, atol=1e-9) and abs(got[1] - ref[1]) < 1e-9 * max(1.0, ref[0] ** 2) + 1e-8 * abs(ref[1]), tag + " moments %s %s" % (got, ref)
    except Exception as e:
        bad += 1; print("FAIL", tag); traceback.print_exc()
    if it % 20 == 0: print("it", it, round(time.time() - t0, 1), flush=True)
print("failures:", bad)
