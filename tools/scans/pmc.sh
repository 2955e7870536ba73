#!/bin/bash
# PMC counters (one pass per group) for kernels matching a substring: tools/scans/pmc.sh <tag> <substr> "<counters>" <script args...>
TAG=$1; SUB=$2; CNT=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d /tmp/pmc_$TAG -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
for sub in ${SUB//,/ }; do echo "-- $sub"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$sub" /tmp/pmc_$TAG; done
