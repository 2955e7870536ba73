#!/bin/bash
# PMC counters of the C4 scoring kernel (NIW dim 32 on the f64 matrix pipe): tools/scans/pmc_c4.sh <outdir> [dim]
# (two rocprofv3 --pmc passes; python3 directly after --)
OUT=$1; DIM=${2:-32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c4a /tmp/pmc_c4b
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d /tmp/pmc_c4a -- python3 $ROOT/tools/scans/c4_case.py $DIM > $ROOT/$OUT/pmc_c4a.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_c4b -- python3 $ROOT/tools/scans/c4_case.py $DIM > $ROOT/$OUT/pmc_c4b.log 2>&1
python3 $ROOT/tools/pmc_summary.py niw64 /tmp/pmc_c4a /tmp/pmc_c4b > $ROOT/$OUT/pmc_c4.txt 2>&1
rm -rf /tmp/pmc_c4a /tmp/pmc_c4b
