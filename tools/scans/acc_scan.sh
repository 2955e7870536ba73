# (round 5: MSC_ACC_WGS_PER_CU lives in tools/microbench/r05_experiment_switches.patch, not in the product)
for w in 2 4 8 12 16 24; do echo "== wgs_per_cu $w"; MSC_ACC_WGS_PER_CU=$w python tools/scans/c3_pieces.py 256 2>&1 | grep -o '"accumulate_ms": [0-9.]*'; done
