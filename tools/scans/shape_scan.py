"""Shapes beside the BASELINE configs: one line of score / sweep-step time each.  Written to find cliffs (a shape
whose rate falls far below its neighbours'); profiles/r01_shape_scan.txt keeps the before / after of the ones found."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, common_amd
from common_amd import BB, DD, GP, NICH, NIW
import common_amd as CA
from bench_configs import make_columns
ctx = common_amd.Context(0)
def run(name, spec, N, K, boost=0, loo=False):
    cols, z = make_columns(ctx, spec, N, K, 73)
    if boost:
        cols = [(c.view(torch.int32) + torch.randint(0, boost, c.shape, device="cuda", dtype=torch.int32)).contiguous().view(torch.uint32)
                if s[0] == GP else c for c, s in zip(cols, spec)]
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    st.set_alpha(1.0)
    out = torch.empty((N, K), dtype=torch.float32, device="cuda")
    def timeit(fn, n=5):
        for i in range(2): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(2 + i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_score = timeit(lambda i: st.score_value(view, out=out))
    zs = z.clone()
    t_sweep = timeit(lambda i: st.sweep_step(view, zs, seed=1, sweep=i))
    D = len(spec)
    print("%-34s N=%-8d K=%-5d score %8.3f ms (%.2e evals/s)  sweep %8.3f ms (%.2e rows/s)" %
          (name, N, K, t_score, N * K * D / t_score * 1e3, t_sweep, N / t_sweep * 1e3), flush=True)
    del out
def run_np(name, fams, N, K):
    """families make_columns does not know (dm, bnb, bbnc): columns built here"""
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    cols, spec = [], []
    for fam, dim in fams:
        if fam == CA.DM:
            cols.append(torch.randint(0, 6, (N, dim), generator=g, device="cuda", dtype=torch.int32).contiguous())
        elif fam == CA.BNB:
            cols.append(torch.randint(0, 20, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.uint32).contiguous())
        elif fam == CA.BBNC:
            cols.append((torch.rand(N, generator=g, device="cuda") < 0.4).contiguous())
        spec.append((fam, dim))
    z = torch.randint(0, K, (N,), generator=g, device="cuda", dtype=torch.int32)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    for i, (fam, dim) in enumerate(spec):
        if fam == CA.DM: st.set_hp(i, {"alphas": [1.0] * dim})
        if fam == CA.BBNC: pass
    st.accumulate(view, z)
    st.set_alpha(1.0)
    out = torch.empty((N, K), dtype=torch.float32, device="cuda")
    def timeit(fn, n=5):
        for i in range(2): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(2 + i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_score = timeit(lambda i: st.score_value(view, out=out))
    t_loo = timeit(lambda i: st.score_value(view, out=out, z=z, crp_prior=True))
    zs = z.clone()
    t_sweep = timeit(lambda i: st.sweep_step(view, zs, seed=1, sweep=i))
    print("%-30s N=%-8d K=%-4d score %7.3f ms  loo-score %7.3f ms  sweep %7.3f ms" % (name, N, K, t_score, t_loo, t_sweep), flush=True)
run("bb x8", [(BB, 0)] * 8, 1_000_000, 256)
run("gp x8 small counts", [(GP, 0)] * 8, 1_000_000, 256)
run("gp x8 counts to 900", [(GP, 0)] * 8, 1_000_000, 256, boost=900)
run("dd128 x8", [(DD, 128)] * 8, 1_000_000, 256)
run("dd8 x8", [(DD, 8)] * 8, 1_000_000, 256)
run("nich x8", [(NICH, 0)] * 8, 1_000_000, 256)
run("nich x2 K=2048", [(NICH, 0)] * 2, 500_000, 2048)
run("mixed x8 K=1000", [(BB, 0), (GP, 0), (DD, 16), (NICH, 0)] * 2, 500_000, 1000)
run("nich x1 K=5000", [(NICH, 0)], 200_000, 5000)
run("nich x1 K=40", [(NICH, 0)], 1_000_000, 40)
run("bb x1 K=16 (C1 shape x100)", [(BB, 0)] * 8, 1_000_000, 16)
for d in (2, 3, 8, 16, 32):
    run("niw d=%d" % d, [(NIW, d)], 262_144, 128)
run("niw d=8 x2 + bb", [(NIW, 8), (NIW, 8), (BB, 0)], 262_144, 64)
run("bb x8 K=16", [(BB, 0)] * 8, 1_000_000, 16)
run("mixed x12 (8 bb + 4 nich) K=32", [(BB, 0)] * 8 + [(NICH, 0)] * 4, 1_000_000, 32)
run("mixed x8 (bb, gp, dd16, nich) K=64", [(BB, 0), (GP, 0), (DD, 16), (NICH, 0)] * 2, 1_000_000, 64)
run("mixed x12 K=100", [(BB, 0)] * 8 + [(NICH, 0)] * 4, 1_000_000, 100)
run("niw d=3 K=64", [(NIW, 3)], 262_144, 64)
run_np("dm4 x4", [(CA.DM, 4)] * 4, 1_000_000, 256)
run_np("dm16 x1", [(CA.DM, 16)], 1_000_000, 256)
run_np("bnb x8", [(CA.BNB, 0)] * 8, 1_000_000, 256)
run_np("bbnc x8", [(CA.BBNC, 0)] * 8, 1_000_000, 256)
