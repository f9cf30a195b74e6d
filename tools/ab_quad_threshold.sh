# where the four-a-lane step form starts to pay (hmm<16> and linear_gaussian_1d<25>, every size with both forms): ms per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
obs() { python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['$1'][:$2])+']')"; }
mkdir -p /tmp/mf
for N in 1500000 2000000 3000000 4000000 6000000; do
for M in "hmm16 hmm16 16 2.0" "linear_gaussian_1d25 lgssm100 25 0.5"; do
  set -- $M
  for FORM in 1 3; do
    L=$($R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model $1 --smc --observes "$(obs $2 $3)" --n_samples $N --seed 7 --ess_threshold $4 --generic --no_dump --json --repeat 6 --step_form $FORM | tail -1)
    echo "$1 n=$N form=$FORM $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run_ms=%.4f' % (1e3*d['run_seconds']))")"
  done
done; done
