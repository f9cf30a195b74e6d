# A/B of the carrying step's inputs staged through LDS by direct loads (LaneStage: tools/dropped/lds_staged_carry_inputs.patch -- apply it first,
# the shipped library has no such form) against plain register loads behind the decision (CPPROB_HIP_NO_STAGE=1), same box, alternating:
#   python tools/ab_stage.py [rounds]        (profiles/r06_ab_stage.txt, profiles/r06_notes.md section 5)
import os, sys, time, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, cpprob_amd as cp
    obs = np.load(os.path.join(ROOT, "tests/golden/observations.npz"))
    eng = cp.Engine(0)
    out = {}
    for model, key, n, ess, rs in ((cp.MODEL_HMM3, "hmm128", 12_500_000, 0.5, cp.RESAMPLE_SYSTEMATIC), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 10_000_000, 0.5, cp.RESAMPLE_SYSTEMATIC),
                                   (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 1_250_000, 0.5, cp.RESAMPLE_SYSTEMATIC), (cp.MODEL_HMM3, "hmm128", 1_250_000, 0.5, cp.RESAMPLE_SYSTEMATIC),
                                   (cp.MODEL_HMM3, "hmm128", 12_500_000, 0.5, cp.RESAMPLE_STRATIFIED), (cp.MODEL_HMM3, "hmm16", 1_000_000, 0.5, cp.RESAMPLE_SYSTEMATIC),
                                   (cp.MODEL_HMM3, "hmm128", 12_500_000, 1e-9, cp.RESAMPLE_SYSTEMATIC), (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 10_000_000, 1e-9, cp.RESAMPLE_SYSTEMATIC)):   # (never resamples: carry launches only)
        eng.begin(cp.ALG_SMC, model, obs[key], n, seed=12345, resampler=rs, ess_threshold=ess)
        eng.run(); eng.sync(); eng.run(); eng.sync()
        t0 = time.perf_counter()
        reps = 5
        for r in range(reps): eng.run(r)
        eng.sync()
        s = eng.summary()
        out["%s@%d/rs%d/ess%g" % (key, n, rs, ess)] = [(time.perf_counter() - t0) / reps * 1e3, s["log_evidence"], s["n_resampled"]]
    print("RESULT " + json.dumps(out))
    sys.exit(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
acc = {}
for r in range(rounds):
    for name, env in (("staged", {}), ("plain", {"CPPROB_HIP_NO_STAGE": "1"})):
        e = dict(os.environ); e.update(env)
        o = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(o.stdout[-2000:], o.stderr[-2000:]); sys.exit(1)
        for k, v in json.loads(line[0][7:]).items():
            acc.setdefault(k, {}).setdefault(name, []).append(v)
for k, d in acc.items():
    st, pl = d["staged"], d["plain"]
    same = all(a[1] == b[1] and a[2] == b[2] for a, b in zip(st, pl))
    print("%-28s staged %s ms   plain %s ms   best %.3f / %.3f = %.3f   same evidence and schedule: %s" % (
        k, " ".join("%.3f" % a[0] for a in st), " ".join("%.3f" % a[0] for a in pl), min(a[0] for a in st), min(a[0] for a in pl), min(a[0] for a in st) / min(a[0] for a in pl), same))
