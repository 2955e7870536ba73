#!/bin/bash
# PMC counters of the C3 scoring kernels for several library builds: tools/scans/pmc_variants.sh <outdir> <variant>...   ("cur" = the product)
# (separate rocprofv3 --pmc run per variant; python3 directly after --)
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = cur ]; then unset MSC_LIB_PATH; else export MSC_LIB_PATH=$ROOT/common_amd/lib/variants/$v.so; fi
  rm -rf /tmp/pmc_$v
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d /tmp/pmc_$v -- python3 $ROOT/tools/scans/c3_pieces.py 256 > $ROOT/$OUT/pmc_$v.log 2>&1
  echo "== $v" >> $ROOT/$OUT/pmc.txt
  python3 $ROOT/tools/pmc_summary.py tile_roles /tmp/pmc_$v >> $ROOT/$OUT/pmc.txt 2>&1
  rm -rf /tmp/pmc_$v
done
