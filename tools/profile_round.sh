#!/bin/bash
# One round's rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag> [bench.py args...]
# 1. --kernel-trace --stats of the bench command  -> gpurun_out/<tag>/<tag>_kernel_stats.csv, <tag>_timed_region.txt
# 2. the same trace under the other allocation regimes of the C2 score matrix: a caller's torch.empty (--alloc torch: plain
#    stores) and a placed buffer with the accept mark lowered to 1 GB/s (MSC_ALLOC_ACCEPT_GBPS=1: the first candidate is
#    "fast", NON-TEMPORAL stores) -- whichever regime the driver's box offers, its kernel instantiation is in the file
# 3. PMC passes, one counter group each (separate runs, as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE
#    do not fit one pass); the two HBM groups once more with the accept mark lowered, so that both instantiations of the
#    headline kernel have their traffic                                -> gpurun_out/<tag>/<tag>_pmc.json
# The raw rocprofv3 output is deleted afterwards (gpurun_out/ must stay small); copy the summaries into profiles/.
# (python3 directly after `--`: no env / bash -c hop under the profiler)
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
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $RAW/kt_torch -- python3 $ROOT/bench.py $ARGS --alloc torch --no-extra --no-sweep > $OUT/kt_caller_torch_empty.log 2>&1
echo "kernel trace (--alloc torch) rc=$?"
export MSC_ALLOC_ACCEPT_GBPS=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $RAW/kt_nt -- python3 $ROOT/bench.py $ARGS --no-extra --no-sweep > $OUT/kt_placed_nt_forced.log 2>&1
echo "kernel trace (accept mark 1 GB/s: non-temporal) rc=$?"
unset MSC_ALLOC_ACCEPT_GBPS
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $RAW/pmc$i -- python3 $ROOT/bench.py $ARGS --steps 50 --warmup 5 > $OUT/pmc$i.log 2>&1   # (counters per launch do not depend on the step count)
  echo "pmc pass $i rc=$?"
done
export MSC_ALLOC_ACCEPT_GBPS=1
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $RAW/pmc$i -- python3 $ROOT/bench.py $ARGS --steps 50 --warmup 5 --no-extra --no-sweep > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i (non-temporal forced) rc=$?"
done
unset MSC_ALLOC_ACCEPT_GBPS
cd $ROOT
MSC_EXTRA_TRACES="caller_torch_empty=$RAW/kt_torch placed_nt_forced=$RAW/kt_nt" MSC_PROFILES_DIR=$OUT python3 tools/summarize_prof.py $TAG $RAW/kt $RAW/pmc*
rm -rf $RAW
ls -la $OUT
