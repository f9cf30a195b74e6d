# Premise check for "capture the run as a HIP graph": one whole run (its step launches + read-out) on the stream against a replay of the same
# launches captured from that stream (the replay repeats ONE seed: timing only).  Needs the GPU:  python tools/graph_replay_vs_stream.py
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import cpprob_amd as cp
hip = C.CDLL("libamdhip64.so")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "observations.npz"))
def chk(rc, what):
    if rc != 0: raise RuntimeError("%s -> %d" % (what, rc))
for key, model, T, n, ess in (("hmm16", cp.MODEL_HMM3, 16, 1_000_000, 2.0), ("lgssm100", cp.MODEL_LINEAR_GAUSSIAN_1D, 100, 1_250_000, 0.5), ("hmm128", cp.MODEL_HMM3, 128, 1_000_000, 0.5)):
    eng = cp.Engine(0)
    eng.begin(cp.ALG_SMC, model, z[key][:T], n, seed=1, ess_threshold=ess)
    for i in range(5): eng.run(i)
    eng.sync()
    reps = 50
    t0 = time.perf_counter()
    for i in range(reps): eng.run(i)
    eng.sync()
    dt_stream = (time.perf_counter() - t0) / reps * 1e3
    st = C.c_void_p(eng.stream_ptr)
    graph = C.c_void_p(); gexec = C.c_void_p()
    chk(hip.hipStreamBeginCapture(st, 2), "begin capture")          # hipStreamCaptureModeRelaxed
    eng.run(7)
    chk(hip.hipStreamEndCapture(st, C.byref(graph)), "end capture")
    chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0), "instantiate")
    for i in range(5): chk(hip.hipGraphLaunch(gexec, st), "launch")
    eng.sync()
    t0 = time.perf_counter()
    for i in range(reps): chk(hip.hipGraphLaunch(gexec, st), "launch")
    eng.sync()
    dt_graph = (time.perf_counter() - t0) / reps * 1e3
    print("%s n %d T %d: %.4f ms per run on the stream, %.4f ms per replay of the captured run" % (key, n, T, dt_stream, dt_graph), flush=True)
    eng.close()
