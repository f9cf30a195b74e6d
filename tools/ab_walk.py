"""A/B of the lineage walk's forms (builds of the library with other CPPROB_SMOOTH_ROWS / _TILES): us per read-out launch, HIP events.
usage (through gpurun): python tools/ab_walk.py   -- runs itself once per library in cpprob_amd/lib/ab/ as a child process"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch  # noqa
    import cpprob_amd as cp
    obs = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))
    eng = cp.Engine(0)
    out = []
    for key, n, model in (("hmm128", 12_500_000, cp.MODEL_HMM3), ("lgssm100", 10_000_000, cp.MODEL_LINEAR_GAUSSIAN_1D), ("lgssm100", 1_250_000, cp.MODEL_LINEAR_GAUSSIAN_1D), ("hmm128", 1_250_000, cp.MODEL_HMM3)):
        eng.begin(cp.ALG_SMC, model, obs[key], n, seed=12345, ess_threshold=0.5)
        eng.run(); eng.sync()
        eng.profile_enable(True); eng.profile_read(reset=True)
        for r in range(4):
            eng.run(r)
        p = eng.profile_read(reset=True); eng.profile_enable(False)
        out.append("%s@%d %.1f" % (key, n, p["smooth"][0] * 1e3 / max(p["smooth"][1], 1)))
    print(sys.argv[1], " | ".join(out), flush=True)
else:
    d = os.path.join(ROOT, "cpprob_amd", "lib", "ab")
    for f in sorted(os.listdir(d)):
        subprocess.call([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, CPPROB_HIP_LIB=os.path.join(d, f)))
