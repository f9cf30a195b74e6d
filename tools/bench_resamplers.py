"""ms per run of the three resamplers on the built-in models (one GPU).  python tools/bench_resamplers.py [--reps 20] [--flags F]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import cpprob_amd as cp  # noqa: E402


def time_runs(eng, reps):
    eng.run(); eng.sync()
    eng.run(); eng.sync()
    t0 = time.perf_counter()
    for r in range(reps):
        eng.run(r)
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--cases", default="hmm16:1000000:2.0,lgssm100:1250000:2.0,lgssm100:1250000:0.5,hmm128:1250000:0.5")
    a = ap.parse_args()
    obs = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))
    eng = cp.Engine(0)
    out = {}
    for case in a.cases.split(","):
        key, n, ess = case.split(":")
        n, ess = int(n), float(ess)
        model = cp.MODEL_HMM3 if key.startswith("hmm") else cp.MODEL_LINEAR_GAUSSIAN_1D
        row = {}
        for name, rs in (("systematic", cp.RESAMPLE_SYSTEMATIC), ("stratified", cp.RESAMPLE_STRATIFIED), ("multinomial", cp.RESAMPLE_MULTINOMIAL)):
            eng.begin(cp.ALG_SMC, model, obs[key], n, seed=12345, resampler=rs, ess_threshold=ess, flags=a.flags)
            ms = time_runs(eng, a.reps)
            s = eng.summary()
            row[name] = dict(ms_per_run=round(ms, 4), step_form=s["step_form"], n_resampled=s["n_resampled"], log_evidence=s["log_evidence"])
        for k in ("stratified", "multinomial"):
            row[k]["vs_systematic"] = round(row[k]["ms_per_run"] / row["systematic"]["ms_per_run"], 3)
        out[case] = row
        print(case, json.dumps(row), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
