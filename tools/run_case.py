"""A few runs of one built-in case (for profilers): python tools/run_case.py hmm128 12500000 0.5 [flags] [resampler] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa
import cpprob_amd as cp
key, n, ess = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rs = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
obs = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))[key]
eng = cp.Engine(0)
eng.begin(cp.ALG_SMC, cp.MODEL_HMM3 if key.startswith("hmm") else cp.MODEL_LINEAR_GAUSSIAN_1D, obs, n, seed=12345, resampler=rs, ess_threshold=ess, flags=flags)
for r in range(reps):
    eng.run(r)
eng.sync()
print(eng.summary())
