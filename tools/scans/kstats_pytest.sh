#!/bin/bash
# which msc kernels a pytest selection launches (and how often): tools/scans/kstats_pytest.sh <tag> <pytest args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python3 -m pytest -p no:cacheprovider -x -q "$@" > $GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG.log 2>&1
f=$(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "msc::" in n:
        print("%-70s calls %5s avg %10.1f us" % (n.split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep -E "passed|failed" $GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG.log
