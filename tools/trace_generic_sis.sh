# kernel timeline of the last cpprob::inference(sis) call on the unchanged-model path (gaussian_unknown_mean, 10^7 particles) -> gpurun_out/gen_sis/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/mf
D=$R/gpurun_out/gen_sis
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o g -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model gaussian_unknown_mean --sis --observes "3 4" --n_samples 10000000 --seed 7 --generic --no_dump --json --repeat 4 > $D/stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$D/g_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-14:]
t0=int(last[0]['Start_Timestamp']); pe=None
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print('%9.1f dur %6.1f gap %6.1f %s' % ((s-t0)/1e3,(e-s)/1e3,((s-pe)/1e3 if pe else 0),r['Kernel_Name'][:70])); pe=e
PY
