#!/usr/bin/env python3
"""Is an exchange-scope run bound by the host's enqueue rate (the case for capturing it as a HIP graph)?  For the group driver the
whole run is enqueued by cpprob_hip_group_run without a host synchronisation inside: the call's own duration is the host's share,
the time until cpprob_hip_group_sync returns is the run.  Prints one JSON line per shape; profiles/r03_notes.md quotes it."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cpprob_amd as cp

z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "observations.npz"))
shapes = [("hmm16, 10^6, world 1", cp.MODEL_HMM3, z["hmm16"], 1_000_000, 1, 2.0),
          ("hmm16, 8 x 10^6, 8 loopback ranks", cp.MODEL_HMM3, z["hmm16"], 8_000_000, 8, 2.0),
          ("linear_gaussian_1d<100>, 10^7, 8 loopback ranks", cp.MODEL_LINEAR_GAUSSIAN_1D, z["lgssm100"], 10_000_000, 8, 0.5)]
for name, model, obs, n, world, ess in shapes:
    g = cp.Group([0] * world)
    g.begin(cp.ALG_SMC, model, obs, n, seed=12345, ess_threshold=ess)
    g.run(0); g.results()
    host, total = [], []
    for i in range(1, 6):
        t0 = time.perf_counter(); g.run(i); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    g.close()
    print(json.dumps({"shape": name, "host_enqueue_ms": round(float(np.median(host)), 3), "run_ms": round(float(np.median(total)), 3),
                      "host_share": round(float(np.median(host) / np.median(total)), 3)}))
