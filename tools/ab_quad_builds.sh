# unchanged-model SMC with the step kernels built per step: one particle a lane (form 1) against four a lane (form 3), by population size
# run on the GPU box as   bash tools/ab_quad_builds.sh   ->  gpurun_out/ab_quad_builds.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
obs() { python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['$1'][:$2])+']')"; }
mkdir -p /tmp/mf
OUT=$R/gpurun_out/ab_quad_builds.txt
: > $OUT
run() {  # model obs-key T n ess
  for FORM in 1 3; do
    L=$($R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model $1 --smc --observes "$(obs $2 $3)" --n_samples $4 --seed 7 --ess_threshold $5 --generic --no_dump --json --repeat 8 --step_form $FORM | tail -1)
    echo "$1 n=$4 ess=$5 form=$FORM $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run_ms=%.4f builds_used=%d log_evidence=%.12f' % (1e3*d['run_seconds'], d['step_builds_used'], d['log_evidence']))")" >> $OUT
  done
}
for N in 100000 300000 1000000 3000000 10000000; do run hmm16 hmm16 16 $N 2.0; done
run hmm16 hmm16 16 1000000 0.5
for N in 300000 1000000 3000000; do run linear_gaussian_1d25 lgssm100 25 $N 0.5; done
cat $OUT
