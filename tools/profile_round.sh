#!/bin/bash
# Collects one round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r04
# bench line(s), rocprofv3 kernel stats of the headline and the other BASELINE configs, PMC traffic / SQ counters of the step
# kernels -- each counter set in a run of its own (kernel-trace / stats only next to --pmc, as the pool requires).  Raw output goes to
# gpurun_out/<tag>_*; tools/summarise_profiles.py turns it into profiles/<tag>_* on the build machine.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
for W in "hmm16_smc 1000000" "hmm16_smc 10000000" "lgssm100_smc 1250000" "lgssm100_smc 10000000" "hmm128_smc_ess 12500000" "gaussian_sis 10000000"; do
  set -- $W
  D=$O/${TAG}_prof_$1_$2
  rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --workload $1 --particles $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.log 2>&1
  python3 $R/bench.py --workload $1 --particles $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.json 2>> $D.log
done
# the other resamplers (thesis Alg. 1 p.36 is multinomial): kernel stats of the headline workload and of configs[3]'s per-GPU shape
for W in "hmm16_smc 1000000" "lgssm100_smc 1250000"; do
  set -- $W
  for RS in stratified multinomial multinomial_literal; do
    D=$O/${TAG}_prof_$1_$2_$RS
    rm -rf $D
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --workload $1 --particles $2 --resampler $RS --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.log 2>&1
    python3 $R/bench.py --workload $1 --particles $2 --resampler $RS --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.json 2>> $D.log
  done
done
# every workload / size pair bench.py quotes a `traffic` figure for (its roofline block reads profiles/<tag>_pmc_traffic.json)
for W in "hmm16_smc 1000000" "hmm16_smc 10000000" "lgssm100_smc 1250000" "lgssm100_smc 10000000" "hmm128_smc_ess 12500000"; do
  set -- $W
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES"; do
    N=$(echo $C | cut -d' ' -f1)
    D=$O/${TAG}_pmc_$1_$2_$N
    rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py --workload $1 --particles $2 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc > $D.log 2>&1
  done
done
# the two 8-GPU configs whole, eight loopback ranks on this one GPU (the full exchange protocol, program order instead of collectives)
python3 $R/bench.py --workload lgssm100_smc --particles 10000000 --loopback-ranks 8 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_loop_c4.json 2> $O/${TAG}_loop_c4.err
python3 $R/bench.py --workload hmm128_smc_ess --particles 100000000 --loopback-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_loop_c5.json 2> $O/${TAG}_loop_c5.err
python3 $R/bench.py --workload hmm16_smc --particles 8000000 --loopback-ranks 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_loop_c3x8.json 2> $O/${TAG}_loop_c3x8.err
# multinomial resampling (thesis Alg. 1) as ONE population over eight loopback shards, beside systematic: kernel stats + the un-profiled line
for RS in multinomial systematic; do
  D=$O/${TAG}_prof_loop8_$RS
  rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --resampler $RS --loopback-ranks 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.log 2>&1
  python3 $R/bench.py --resampler $RS --loopback-ranks 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $D.json 2>> $D.log
done
# the unchanged-model path: step kernels built per step against the run-time kernel, one / four particles a lane
bash $R/tools/ab_step_builds.sh > /dev/null 2>&1
bash $R/tools/ab_quad_builds.sh > /dev/null 2>&1
# A/B of the headline's read-out (trace words vs the lineage walk), and the host's share of an exchange-scope run
python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --flags 64 > $O/${TAG}_bench_walk_readout.json 2>> $O/${TAG}_bench.err
python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/${TAG}_bench_trace_words.json 2>> $O/${TAG}_bench.err
python3 $R/tools/enqueue_vs_gpu.py > $O/${TAG}_enqueue_vs_gpu.jsonl 2>> $O/${TAG}_bench.err
# the unchanged-model path: kernel trace of cpprob_main --generic (hmm<16>, 10^6), fused step and the r03 form
bash $R/tools/profile_generic.sh ${TAG}_generic > $O/${TAG}_generic.log 2>&1
ls $O | grep ${TAG}_ | head -80
cut -c1-900 $O/${TAG}_bench.json
