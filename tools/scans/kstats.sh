#!/bin/bash
# kernel-trace stats of one command, msc kernels only: tools/scans/kstats.sh <tag> <python script and args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG.log 2>&1
f=$(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "msc::" in n:
        print("%-60s calls %5s avg %10.1f us  min %10.1f  max %10.1f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
