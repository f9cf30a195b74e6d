# A/B of two builds of libcpprob_hip.so (scratch/libs/lib_<tag>.so, loaded through CPPROB_HIP_LIB), alternating on one box: ms per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_libs.txt
: > $OUT
for rep in 1 2; do
for TAG in ${TAGS:-orig sgpr}; do
for W in "hmm16_smc 1000000" "lgssm100_smc 1250000" "lgssm100_smc 10000000" "hmm128_smc_ess 12500000" "gaussian_sis 10000000"; do
  set -- $W
  L=$(CPPROB_HIP_LIB=$R/scratch/libs/lib_$TAG.so python3 $R/bench.py --workload $1 --particles $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc 2>/dev/null | tail -1)
  echo "$TAG $1 $2 $(echo "$L" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_run=%.4f' % d['ms_per_step'])")" >> $OUT
done; done; done
sort -k2,3 -k1,1 -s $OUT
