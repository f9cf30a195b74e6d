# kernel trace of the unchanged-model path (cpprob_main --generic, hmm<16>, 10^6 particles, golden observations, 8 calls + the pilot), once
# per step form: run on the GPU box as   bash tools/profile_generic.sh <tag>   ->  gpurun_out/<tag>_form{1,0}/gen_kernel_{stats,trace}.csv, stdout.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
mkdir -p /tmp/mf
for FORM in 1 0; do
  D=$R/gpurun_out/${1:-r04_generic}_form$FORM
  rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 --step_form $FORM > $D/stdout.log 2>&1
  $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 --step_form $FORM | tail -1 > $D/unprofiled.json
done
