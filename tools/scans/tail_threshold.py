import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, common_amd
from tools.bench_configs import make_columns, timed
ctx = common_amd.Context(0)
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
for K in (16, 64, 128):
    for N in (16384, 32768, 65536, 131072, 262144):
        cols, z = make_columns(ctx, spec, N, K, 73)
        view = common_amd.DataView.from_tensors(ctx, cols)
        st = common_amd.State(ctx, spec, K); st.set_alpha(1.0); st.accumulate(view, z)
        out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
        r = {}
        for name, env in (("rows", "1"), ("tile", "1000000000")):
            os.environ["MSC_TAIL_MIN_ROWS"] = env
            r[name] = round(timed(lambda: st.score_value(view, out=out), 20)[1], 4)
        os.environ.pop("MSC_TAIL_MIN_ROWS")
        r["default"] = round(timed(lambda: st.score_value(view, out=out), 20)[1], 4)
        print(K, N, r, flush=True)
