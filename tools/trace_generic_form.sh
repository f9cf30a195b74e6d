# per-launch kernel times of the unchanged-model step kernel: bash tools/trace_generic_form.sh <form> [model] [obs-key] [T] [n] [ess]  ->  gpurun_out/gen_form<form>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FORM=${1:-3}; MODEL=${2:-hmm16}; KEY=${3:-hmm16}; T=${4:-16}; N=${5:-1000000}; ESS=${6:-2.0}
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['$KEY'][:$T])+']')")
mkdir -p /tmp/mf
D=$R/gpurun_out/gen_form$FORM
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model $MODEL --smc --observes "$OBS" --n_samples $N --seed 7 --ess_threshold $ESS --generic --no_dump --json --repeat 4 --step_form $FORM > $D/stdout.log 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$D/gen_kernel_trace.csv')) if 'model_step_kernel' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
T=$T
last=d[-T:]
print('form $FORM $MODEL n=$N: last call step launches (us):', ' '.join('%.1f'%x for x in last), 'sum %.1f'%sum(last))
PY
