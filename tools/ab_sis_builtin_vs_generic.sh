cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/mf
for G in "" "--generic"; do
echo "gaussian_unknown_mean sis 1e7 $G"
$R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model gaussian_unknown_mean --sis --observes "3 4" --n_samples 10000000 --seed 7 $G --no_dump --json --repeat 6 | grep "^run" | tr '\n' ' '; echo
done
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
for G in "" "--generic"; do
echo "hmm16 sis 1e6 $G"
$R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --sis --observes "$OBS" --n_samples 1000000 --seed 7 $G --no_dump --json --repeat 6 | grep "^run" | tr '\n' ' '; echo
done
