#!/bin/bash
# One round's rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag> [bench.py args...]
# 1. --kernel-trace --stats of the bench command  -> gpurun_out/<tag>/<tag>_kernel_stats.csv
# 2. PMC passes, one counter group each (separate runs, as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE
#    do not fit one pass)                           -> gpurun_out/<tag>/<tag>_pmc.json
# The raw rocprofv3 output is deleted afterwards (gpurun_out/ must stay small); copy the two summaries into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $*"            # (bench.py's own default steps / warm-up: the summary must describe the same run)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -- python3 $ROOT/bench.py $ARGS > $OUT/kt.log 2>&1
echo "kernel trace rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $RAW/pmc$i -- python3 $ROOT/bench.py $ARGS --steps 50 --warmup 5 > $OUT/pmc$i.log 2>&1   # (counters per launch do not depend on the step count)
  echo "pmc pass $i rc=$?"
done
cd $ROOT
MSC_PROFILES_DIR=$OUT python3 tools/summarize_prof.py $TAG $RAW/kt $RAW/pmc*
rm -rf $RAW
ls -la $OUT
