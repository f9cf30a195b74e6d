# issue counters of the unchanged-model step kernel of one step form: bash tools/pmc_generic_form.sh <form>  ->  gpurun_out/gen_form<form>_pmc/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FORM=${1:-3}
OBS=$(python3 -c "
import numpy as np
z=np.load('$R/tests/golden/observations.npz'); print('['+' '.join(repr(float(x)) for x in z['hmm16'])+']')")
mkdir -p /tmp/mf
D=$R/gpurun_out/gen_form${FORM}_pmc
rm -rf $D; mkdir -p $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $D -o gen -- $R/cpprob_amd/bin/cpprob_main --model_folder /tmp/mf --model hmm16 --smc --observes "$OBS" --n_samples 1000000 --seed 7 --ess_threshold 2.0 --generic --no_dump --json --repeat 4 --step_form $FORM > $D/stdout.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob('$D/*counter_collection.csv') + glob.glob('$D/*/*counter_collection.csv')
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'model_step_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) >= 100000:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
c = {k: sum(v) / len(v) for k, v in acc.items()}
w = c['SQ_WAVES']
print('form $FORM: launches %d, wavefronts %.0f; per wavefront: VALU %.0f SALU %.0f wave-cycles %.0f wait %.0f; wait_frac %.2f valu_issue_frac %.3f' % (len(acc['SQ_WAVES']), w, c['SQ_INSTS_VALU'] / w, c['SQ_INSTS_SALU'] / w, c['SQ_WAVE_CYCLES'] / w, c['SQ_WAIT_ANY'] / w, c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']))
PY
