cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
mkdir -p /tmp/mf
for G in "" "--generic"; do
$R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 $G --no_dump --json --repeat 8 | grep "^run" | tr '\n' ' '; echo
done
