# kernel trace of the unchanged-model path (cpprob_main --generic, hmm<16>, 10^6 particles, golden observations, 8 calls + the pilot), once
# per step form: run on the GPU box as   bash tools/profile_generic.sh <tag>   ->  gpurun_out/<tag>_form{1,0}/gen_kernel_{stats,trace}.csv, stdout.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
mkdir -p /tmp/mf
for FORM in 1 0 3; do
  D=$R/gpurun_out/${1:-r04_generic}_form$FORM
  rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 --step_form $FORM > $D/stdout.log 2>&1
  $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 --step_form $FORM | tail -1 > $D/unprofiled.json
done
# issue counters of the fused step kernel (form 1): one --pmc pass, nothing else on the command line
D=$R/gpurun_out/${1:-r04_generic}_form1_pmc
rm -rf $D; mkdir -p $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 4 --step_form 1 > $D/stdout.log 2>&1
D=$R/gpurun_out/${1:-r04_generic}_form1_pmc_lds
rm -rf $D; mkdir -p $D
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 4 --step_form 1 > $D/stdout.log 2>&1
