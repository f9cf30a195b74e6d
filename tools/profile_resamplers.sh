#!/bin/bash
# rocprofv3 kernel trace of the three resamplers on one case: tools/profile_resamplers.sh <tag> <case> (run through gpurun from the repo root)
TAG=${1:-r05}
CASE=${2:-hmm16:1000000:2.0}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
N=$(echo $CASE | tr ':' '_')
D=$O/${TAG}_resamplers_$N
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/bench_resamplers.py --reps 10 --cases $CASE > $D.log 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | avg us | total ms | % |")
print("|---|---|---|---|---|")
for r in rows[:24]:
    print("| %s | %s | %.2f | %.3f | %s |" % (r["Name"][:150], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -3 $D.log
