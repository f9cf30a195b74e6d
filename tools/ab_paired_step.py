import os, sys, time, json
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch, cpprob_amd as cp
obs=np.load(os.path.join(ROOT,"tests/golden/observations.npz"))
eng=cp.Engine(0)
def t(model,key,n,ess,flags,reps=5, rs=cp.RESAMPLE_SYSTEMATIC):
    eng.begin(cp.ALG_SMC, model, obs[key], n, seed=12345, resampler=rs, ess_threshold=ess, flags=flags)
    eng.run(); eng.sync(); eng.run(); eng.sync()
    t0=time.perf_counter()
    for r in range(reps): eng.run(r)
    eng.sync()
    return (time.perf_counter()-t0)/reps*1e3, eng.summary()["log_evidence"], eng.summary()["n_resampled"]
for model,key,n in ((cp.MODEL_HMM3,"hmm128",12_500_000),(cp.MODEL_LINEAR_GAUSSIAN_1D,"lgssm100",10_000_000),(cp.MODEL_LINEAR_GAUSSIAN_1D,"lgssm100",1_250_000),(cp.MODEL_HMM3,"hmm128",1_250_000)):
    for name,fl in (("single",0),("paired",512)):
        print(key,n,name,t(model,key,n,0.5,fl), flush=True)
