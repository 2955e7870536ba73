#!/bin/bash
# A/B of the k_score_nich1 launch shapes on ONE box: MSC_NICH1_SHAPE fixes the shape (index into kNich1Shapes,
# common_amd/csrc/kernels_score.hip), -1 lets the library select it at the first large pass.
# (Earlier tables of profiles/r01_nich1_variants.txt were made with MSC_NICH1_ITERS / MSC_NICH1_NT knobs that no longer exist.)
for q in -1 0 1 2 3 4 5 6 7 -1; do
  MSC_NICH1_SHAPE=$q python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('shape=$q', round(r['achieved'],1), 'GB/s avg', round(r['kernel_avg_ms']*1e3,1), 'us; value', '%.3e' % d['value'])"
done
