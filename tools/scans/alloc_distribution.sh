#!/bin/bash
# roofline.frac of the C2 pass in N fresh processes, the library's default allocator (and the caller-alloc figure beside it)
N=${1:-10}
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --no-extra --no-sweep --steps 500 --warmup 200 2>/dev/null | python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
ro=r['roofline']; sm=r['config']['score_matrix']
print(json.dumps({'run': $i, 'frac': round(ro['frac'],4), 'frac_caller_alloc': round(ro['frac_caller_alloc'],4), 'kernel_avg_ms': round(ro['kernel_avg_ms'],5), 'stores': sm.get('stores'), 'candidates_fill_GBps': sm.get('candidates_fill_GBps'), 'kept': sm.get('kept')}))"
done
