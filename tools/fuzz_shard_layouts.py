import os, sys, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, cpprob_amd
import test_gpu_fuzz as F
eng = cpprob_amd.Engine(0); bad = 0
for sweep in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        F.test_random_shard_layouts_exchange_scope(eng, "tests/golden", sweep)
    except Exception:
        print("sweep", sweep); traceback.print_exc(limit=3); bad += 1
print("failures:", bad)
