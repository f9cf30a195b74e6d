# kernel timeline of the last cpprob::inference call on the built-in path (cpprob_main, hmm<16>, 10^6 particles) -> gpurun_out/builtin_trace/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
mkdir -p /tmp/mf
D=$R/gpurun_out/builtin_trace
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -o b -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --no_dump --json --repeat 4 > $D/stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$D/b_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-40:]
t0=int(last[0]['Start_Timestamp']); pe=None
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print('%9.1f dur %6.1f gap %6.1f %s' % ((s-t0)/1e3,(e-s)/1e3,((s-pe)/1e3 if pe else 0),r['Kernel_Name'][:60])); pe=e
PY
