# kernel trace of the unchanged-model path (cpprob_main --generic, hmm<16>, 10^6 particles, 8 runs + the pilot): run on the GPU box as
#   bash tools/profile_generic.sh <tag>      ->  gpurun_out/<tag>/gen_kernel_stats.csv, stdout.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBS="[0.9 0.8 0.7 0.0 -0.025 -5.0 -2.0 -0.1 0.0 0.13 0.45 6.0 0.2 0.3 -1.0 -1.0]"
mkdir -p $R/gpurun_out/${1:-r03_generic} /tmp/mf
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${1:-r03_generic} -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 > $R/gpurun_out/${1:-r03_generic}/stdout.log 2>&1
ls -R $R/gpurun_out/${1:-r03_generic} | head -20
$R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 8 | tail -1 | cut -c1-300
