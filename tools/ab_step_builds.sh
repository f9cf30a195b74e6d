# unchanged-model SMC: the step kernels built per step (model_step_kernel_at) against the run-time kernel: ms per run, un-profiled.
# run on the GPU box as   bash tools/ab_step_builds.sh   ->  gpurun_out/ab_step_builds.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
obs() { python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['$1'][:$2])+']')"; }
mkdir -p /tmp/mf
OUT=$R/gpurun_out/ab_step_builds.txt
: > $OUT
run() {  # model obs-key T n ess
  for SW in "" "--no_step_builds"; do
    L=$($R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model $1 --smc --observes "$(obs $2 $3)" --n_samples $4 --seed 7 --ess_threshold $5 --generic --no_dump --json --repeat 8 --step_form 1 $SW | tail -1)
    echo "$1 n=$4 ess=$5 ${SW:-step_builds} $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run_ms=%.4f step_form=%d builds_used=%d log_evidence=%.12f n_resampled=%d' % (1e3*d['run_seconds'], d['step_form'], d['step_builds_used'], d['log_evidence'], d['n_resampled']))")" >> $OUT
  done
}
run hmm16 hmm16 16 1000000 2.0
run hmm16 hmm16 16 1000000 0.5
run linear_gaussian_1d25 lgssm100 25 1000000 0.5
run linear_gaussian_1d100 lgssm100 100 1250000 0.5
run hmm128 hmm128 128 1250000 0.5
cat $OUT
