#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command: tools/profile_cmd.sh <tag> <python script + args...>   (through gpurun, from the repo root)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$O/$TAG
rm -rf $D
S=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/$S "$@" > $D.log 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | avg us | total ms | % |")
print("|---|---|---|---|---|")
for r in rows[:16]:
    print("| %s | %s | %.2f | %.3f | %s |" % (r["Name"][:150], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
