# Random sizes / seeds / schedules through the C++ host: the unchanged-model path against the built-in kernels (not part of the suite; needs the GPU):
#   python tools/fuzz_cpp_host.py SEED COUNT               e.g. 1 60: ~1 min on the box
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
MAIN = os.path.join(ROOT, "cpprob_amd", "bin", "cpprob_main")
z = np.load(os.path.join(ROOT, "tests", "golden", "observations.npz"))
def obs_str(v): return "[" + " ".join(repr(float(x)) for x in v) + "]"
def run(td, *args):
    p = subprocess.run([MAIN, "--model_folder", td] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    res = None
    for line in p.stdout.splitlines():
        if line.startswith("{"): res = json.loads(line)
    return p.returncode, res, p.stderr[-500:]
bad = 0
rng = np.random.default_rng(int(sys.argv[1]))
for it in range(int(sys.argv[2])):
    model, obs = [("hmm16", z["hmm16"]), ("linear_gaussian_1d25", z["lgssm100"][:25])][int(rng.integers(0, 2))]
    n = int(rng.choice([1, 2, 3, 255, 256, 257, 1023, 1025, 4097, 70001, 262145, int(rng.integers(1, 1_500_000))]))
    ess = float(rng.choice([2.0, 0.5, 0.9, 0.0]))
    seed = int(rng.integers(0, 2**31))
    extra = [["--no_dump"], ["--estimate"], []][int(rng.integers(0, 3))] if n <= 70001 else ["--no_dump"]
    tag = "%s n %d ess %.1f seed %d %s" % (model, n, ess, seed, extra)
    with tempfile.TemporaryDirectory() as td:
        base = ["--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", seed, "--ess_threshold", ess, "--json"] + extra
        rcb, rb, eb = run(td, *base)
        rcg, rg, eg = run(td, *base, "--generic")
    ok = rcb == 0 and rcg == 0 and rb is not None and rg is not None
    if ok:
        ok = np.isfinite(rb["log_evidence"]) and np.isfinite(rg["log_evidence"])
        tol = 0.5 / np.sqrt(max(n, 1)) * 30 + 1e-6
        if model == "hmm16":
            pb = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in rb["predicts"]]); pg = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in rg["predicts"]])
            ok = ok and abs(pb.sum(1) - 1).max() < 1e-9 and abs(pg.sum(1) - 1).max() < 1e-9 and (n < 1000 or np.abs(pb - pg).max() < tol)
        else:
            mb = np.array([[p["mean"], p["variance"]] for p in rb["predicts"]]); mg = np.array([[p["mean"], p["variance"]] for p in rg["predicts"]])
            ok = ok and np.isfinite(mb).all() and np.isfinite(mg).all() and (n < 1000 or np.abs(mb - mg).max() < tol * 3)
        ok = ok and (n < 1000 or abs(rb["log_evidence"] - rg["log_evidence"]) < 1e-3 + tol)
    if not ok:
        bad += 1
        print("FAIL", tag, rcb, rcg, (rb or {}).get("log_evidence"), (rg or {}).get("log_evidence"), eb[-200:], eg[-200:], flush=True)
    else:
        print("ok", tag, rb["log_evidence"], rg["log_evidence"], rb.get("n_resampled"), rg.get("n_resampled"), flush=True)
print("failures:", bad)
