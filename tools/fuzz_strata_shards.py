# Multinomial resampling (strata form) of ONE population over loopback shards against the one-context run: random models, trace lengths,
# schedules, worlds and shard layouts -- tiny shards (several rank boundaries inside one stratum), shards that start at odd particles,
# populations around the tile and stratum edges -- every surviving trace, the evidence and the decisions must be the one-GPU run's.
# Also stratified / systematic through the same layouts, and outlier observations (a generation repaired in the run).
# Not part of the suite; needs the GPU:   python tools/fuzz_strata_shards.py SEED COUNT       e.g. 1 80: ~1.5 min on the box
import os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch  # noqa
import cpprob_amd as cp
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "observations.npz"))
eng = cp.Engine(0)
bad = 0
rng = np.random.default_rng(int(sys.argv[1]))


def ctx_paths(g, r, n, T, is_int):
    e = g.context(r)
    e.n, e.T = n, T
    return e.paths()


for it in range(int(sys.argv[2])):
    model, key = [(cp.MODEL_HMM3, "hmm128"), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100")][int(rng.integers(0, 2))]
    T = int(rng.integers(2, 30))
    obs = np.array(z[key][:T])
    if rng.random() < 0.15:
        obs[int(rng.integers(0, T))] = 30.0 if model == cp.MODEL_LINEAR_GAUSSIAN_1D else 9.0       # an outlier: a generation loses its bits
    world = int(rng.integers(2, 1 + int(os.environ.get("FUZZ_WORLD_MAX", "8"))))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        sizes = [int(rng.integers(1, 40)) for _ in range(world)]                                     # everything inside one stratum
        sizes[int(rng.integers(0, world))] += int(rng.integers(0, 5000))
    elif kind == 1:
        sizes = [int(rng.integers(1, 3000)) for _ in range(world)]
    elif kind == 2:
        sizes = [int(rng.choice([1023, 1024, 1025, 2047, 2049, 511])) for _ in range(world)]
    else:
        sizes = [int(rng.integers(1, 60000)) for _ in range(world)]
    n = int(sum(sizes))
    ess = float(rng.choice([2.0, 0.5, 0.9]))
    rs = int(rng.choice([cp.RESAMPLE_MULTINOMIAL, cp.RESAMPLE_MULTINOMIAL, cp.RESAMPLE_STRATIFIED, cp.RESAMPLE_SYSTEMATIC]))
    seed = int(rng.integers(0, 2**31))
    tag = "it %d model %d T %d sizes %s ess %.1f rs %d seed %d" % (it, model, T, sizes, ess, rs, seed)
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != it:
        continue
    if os.environ.get("FUZZ_ONLY"):
        print("obs", list(obs), flush=True)
    try:
        eng.begin(cp.ALG_SMC, model, obs, n, seed=seed, resampler=rs, ess_threshold=ess)
        eng.run()
        ref_sum, ref_paths, ref_stats = eng.summary(), eng.paths(), eng.stats().copy()
        g = cp.Group([0] * world)
        g.begin(cp.ALG_SMC, model, obs, n, seed=seed, resampler=rs, ess_threshold=ess, shard_sizes=sizes)
        g.run()
        stats, s, reruns = g.results()
        paths = np.concatenate([ctx_paths(g, r, sizes[r], T, model == cp.MODEL_HMM3) for r in range(world)], axis=1)
        g.close()
        assert s["step_form"] == ref_sum["step_form"] != cp.capi.FORM_FLOAT, tag + " forms %d %d" % (s["step_form"], ref_sum["step_form"])
        assert np.array_equal(paths, ref_paths), tag + " traces differ in %d particles" % int((paths != ref_paths).any(axis=0).sum())
        assert s["log_evidence"] == ref_sum["log_evidence"] and s["n_resampled"] == ref_sum["n_resampled"] and s["n_requantised"] == ref_sum["n_requantised"], tag
        np.testing.assert_allclose(stats, ref_stats, rtol=1e-9, atol=1e-9, err_msg=tag)
        print("ok", tag, "reruns", reruns, "requantised", s["n_requantised"], flush=True)
    except Exception as e:      # noqa
        bad += 1
        print("FAIL", tag, e, flush=True)
        traceback.print_exc()
print("failures:", bad)
