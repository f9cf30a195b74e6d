# Dependent-launch cost of a trivial kernel: back-to-back launches on a stream (torch: host-bound) against a replayed graph (GPU-side).
# Needs the GPU:  python tools/graph_floor.py
import time, torch
x = torch.zeros(1 << 20, device="cuda")        # 1M floats: ~1000 workgroups, like a step launch's grid
y = torch.zeros(64, device="cuda")             # one workgroup
for name, t in (("1M-element add_", x), ("64-element add_", y)):
    K = 17
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(50):
            for _ in range(K): t.add_(1.0)
        s.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            for _ in range(K): t.add_(1.0)
        s.synchronize()
        dt_stream = (time.perf_counter() - t0) / reps / K * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(K): t.add_(1.0)
        for _ in range(20): g.replay()
        s.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        dt_graph = (time.perf_counter() - t0) / reps / K * 1e6
    print("%s: %.2f us per dependent launch on a stream, %.2f us in a replayed graph of %d" % (name, dt_stream, dt_graph, K))
