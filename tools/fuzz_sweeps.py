# Extended seeds for the randomised sweeps of tests/test_gpu_fuzz.py (not part of the suite; needs the GPU):
#   python tools/fuzz_sweeps.py FIRST_SEED LAST_SEED        e.g. 100 600: ~8 min on the box (profiles/r03_notes.md section 6b)
import os, sys, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa
import cpprob_amd
import test_gpu_fuzz as F
golden = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
eng = cpprob_amd.Engine(0)
t0 = time.time(); bad = 0
for sweep in range(int(sys.argv[1]), int(sys.argv[2])):
    for fn in (F.test_random_smc_runs_keep_their_invariants, F.test_random_shard_layouts_exchange_scope, F.test_random_groups_transports_and_collectives):
        try:
            fn(eng, golden, sweep)
        except cpprob_amd.capi.CpprobHipError as e:
            print("sweep", sweep, fn.__name__, "CpprobHipError", e); bad += 1
        except Exception:
            print("sweep", sweep, fn.__name__); traceback.print_exc(); bad += 1
    print("sweep", sweep, "done", round(time.time() - t0, 1), "s", flush=True)
print("failures:", bad)
