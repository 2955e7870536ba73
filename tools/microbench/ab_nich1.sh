#!/bin/bash
for rep in 1 2 3; do
for cfg in "4 0" "1 1" "2 1" "3 0"; do set -- $cfg
  MSC_NICH1_ITERS=$1 MSC_NICH1_NT=$2 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('iters=$1 nt=$2', 'C2', round(d['roofline']['achieved'],1), 'GB/s', round(d['ms_per_step'],4),'ms/step sweep', d.get('sweep'))"
done; done
