#!/usr/bin/env python
"""Measure the BASELINE.json configs that fit one GPU (C2, C3, C4, a C5 per-GPU shard); one JSON
line per measurement.  Not the driver's bench (that is bench.py) -- this feeds DESIGN.md.
usage: tools/bench_configs.py [c2] [c3] [c4] [c5] [--steps K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from common_amd import BB, DD, GP, NICH, NIW  # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return wall, sum(ms) / len(ms), ms[0]


def make_columns(ctx, spec, N, K, seed):
    g = torch.Generator(device=ctx.torch_device)
    g.manual_seed(seed)
    dev = ctx.torch_device
    z = torch.randint(0, K, (N,), generator=g, device=dev, dtype=torch.int32)
    zl = z.long()
    cols = []
    for fam, dim in spec:
        if fam == BB:
            p = torch.rand(K, generator=g, device=dev)
            cols.append((torch.rand(N, generator=g, device=dev) < p[zl]).contiguous())
        elif fam == GP:
            lam = torch.distributions.Gamma(2.0, 0.5).sample((K,)).to(dev)
            cols.append(torch.poisson(lam[zl], generator=g).to(torch.int32).view(torch.uint32).contiguous()
                        if hasattr(torch, "uint32") else torch.poisson(lam[zl], generator=g).to(torch.int32))
        elif fam == DD:
            cols.append(((torch.randint(0, dim, (N,), generator=g, device=dev) + zl) % dim).to(torch.int32).contiguous())
        elif fam == NICH:
            c = torch.randn(K, generator=g, device=dev) * 10
            cols.append((c[zl] + torch.randn(N, generator=g, device=dev)).float().contiguous())
        elif fam == common_amd.DM:                         # int32 [N, dim] counts: small (the tables cover them: tens of rows)
            lam = (torch.rand(K, dim, generator=g, device=dev) * 3.0 + 0.5)
            cols.append(torch.poisson(lam[zl], generator=g).to(torch.int32).contiguous())
        elif fam == NIW:
            c = torch.randn(K, dim, generator=g, device=dev) * 3
            A = torch.randn(K, dim, dim, generator=g, device=dev) / dim ** 0.5
            e = torch.randn(N, dim, generator=g, device=dev)
            cols.append((c[zl] + torch.einsum("nij,nj->ni", A[zl], e)).float().contiguous())
    return cols, z


def run(name, ctx, spec, N, K, steps, sweep=True, **kw):
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    D = len(spec)
    wall, avg, mn = timed(lambda: st.score_value(view, out=out, **kw), steps)
    rec = {"config": name, "N": N, "K": K, "D": D, "score_ms_avg": avg, "score_ms_min": mn,
           "evals_per_s": N * K * D / (avg * 1e-3), "out_GBps": 4.0 * N * K / (avg * 1e-3) / 1e9}
    if sweep:
        zs = z.clone()
        drv = common_amd.dist.ShardedSweep(st, view, zs, 0)
        it = [0]

        def one():
            it[0] += 1
            drv.sweep(seed=73, sweep_index=it[0])
        w, a, m = timed(one, max(3, steps // 3), warmup=2)
        rec.update({"sweep_ms_avg": a, "sweep_rows_per_s": N / (a * 1e-3)})
        w, a, m = timed(lambda: st.sweep_assign(view, zs, seed=1, sweep=it[0]), max(3, steps // 3), warmup=1)
        rec.update({"sweep_assign_only_ms": a})
        w, a, m = timed(lambda: st.accumulate(view, zs), max(3, steps // 3), warmup=1)
        rec.update({"accumulate_ms": a})
    print(json.dumps(rec), flush=True)
    del out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = 20
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
    which = args or ["c2", "c3", "c4", "c5"]
    ctx = common_amd.Context(0)
    if "c2" in which:
        run("C2 nich N=1M K=256 D=1", ctx, [(NICH, 0)], 1_000_000, 256, steps)
    if "c3" in which:
        run("C3 mixed bb+gp+dd32+nich x16 N=1M K=256 D=64", ctx, [(BB, 0), (GP, 0), (DD, 32), (NICH, 0)] * 16,
            1_000_000, 256, max(5, steps // 2))
    if "c4" in which:
        run("C4 niw d=32 N=256k K=128 (f64 MFMA)", ctx, [(NIW, 32)], 262_144, 128, max(5, steps // 2))
        run("C4 niw d=32 N=256k K=128 (f32 MFMA, MSC_SCORE_NIW_F32)", ctx, [(NIW, 32)], 262_144, 128,
            max(5, steps // 2), sweep=False, niw_f32=True)
    if "c5" in which:
        run("C5 shard nich N=12.5M K=1024 D=1 (one of 8 GPUs; scores not materialised: sweep only)", ctx,
            [(NICH, 0)], 2_000_000, 1024, 5)


if __name__ == "__main__":
    main()
