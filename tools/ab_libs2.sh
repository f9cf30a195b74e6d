# A/B of builds of libcpprob_hip.so (scratch/libs/lib_<tag>.so through CPPROB_HIP_LIB), alternating on one box: ms per run.
# usage on the GPU box: TAGS="base f8" CASES="hmm128_smc_ess:12500000 hmm16_smc:10000000" bash tools/ab_libs2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_libs2.txt
: > $OUT
for rep in 1 2; do
for TAG in ${TAGS:-base}; do
for W in ${CASES:-hmm16_smc:1000000 hmm16_smc:10000000 hmm128_smc_ess:12500000 hmm128_smc_ess:1250000 lgssm100_smc:1250000}; do
  WL=${W%%:*}; N=${W##*:}
  L=$(CPPROB_HIP_LIB=$R/scratch/libs/lib_$TAG.so python3 $R/bench.py --workload $WL --particles $N --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc 2>/dev/null | tail -1)
  echo "$TAG $WL $N $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_run=%.4f step_us=%.2f' % (d['ms_per_step'], d['roofline']['avg_launch_us']))")" >> $OUT
done; done; done
sort -k2,3 -k1,1 -s $OUT
