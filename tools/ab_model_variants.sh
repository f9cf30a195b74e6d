# A/B of model translation units built differently (scratch/variants/<V>/libcpprob_models.so), form 1, same box, alternating: ms per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
obs() { python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['$1'][:$2])+']')"; }
mkdir -p /tmp/mf
OUT=$R/gpurun_out/ab_model_variants.txt
: > $OUT
run() {  # variant model obs-key T n ess form
  L=$(LD_LIBRARY_PATH=$R/scratch/variants/$1:$R/cpprob_amd/lib $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model $2 --smc --observes "$(obs $3 $4)" --n_samples $5 --seed 7 --ess_threshold $6 --generic --no_dump --json --repeat 8 --step_form $7 | tail -1)
  echo "$1 $2 n=$5 ess=$6 form=$7 $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run_ms=%.4f log_evidence=%.12f' % (1e3*d['run_seconds'], d['log_evidence']))")" >> $OUT
}
for rep in 1 2; do
for V in ${VARIANTS:-A B C}; do
run $V hmm16 hmm16 16 1000000 2.0 1
run $V linear_gaussian_1d100 lgssm100 100 1250000 0.5 1
run $V hmm128 hmm128 128 1250000 0.5 1
run $V linear_gaussian_1d25 lgssm100 25 1000000 0.5 1
done; done
sort -k2,2 -k1,1 -s $OUT
