#!/usr/bin/env python
"""which entries of a leave-one-out pass differ between the role-split kernel (40k rows) and short slices (the kernels
that run the phases one after the other): row within the wave pair, own-group entry or not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import common_amd
from common_amd import BB, GP, DD, NICH

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = common_amd.Context(0)
dev = ctx.torch_device
g = torch.Generator(device=dev); g.manual_seed(5)
N = 40_000
spec = [(BB, 0), (GP, 0), (NICH, 0), (DD, 7), (NICH, 0), (BB, 0), (NICH, 0)]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench_configs import make_columns
cols, z = make_columns(ctx, spec, N, K, 5)
view = common_amd.DataView.from_tensors(ctx, cols)
st = common_amd.State(ctx, spec, K)
st.accumulate(view, z)
whole = st.score_value(view, z=z).clone()
plain = st.score_value(view).clone()
bad_rows, bad_own, bad_other = [], 0, 0
for row0 in range(0, 4096, 64):
    part = st.score_value(view, row0=row0, nrows=64, z=z[row0:row0 + 64].contiguous())
    d = (part != whole[row0:row0 + 64])
    if d.any():
        idx = d.nonzero().cpu().numpy()
        for r, k in idx:
            own = int(z[row0 + r]) == int(k)
            bad_own += own; bad_other += (not own)
            if len(bad_rows) < 40:
                bad_rows.append((row0 + int(r), int(k), own, float(part[r, k]), float(whole[row0 + r, k]), float(plain[row0 + r, k])))
print("K", K, "bad own-group entries", bad_own, "other entries", bad_other)
for b in bad_rows: print(b)
