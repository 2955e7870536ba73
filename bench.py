#!/usr/bin/env python
"""bench.py -- one JSON line for the driver (see the contract in the task statement).

Workload at N=1: BASELINE.json configs[1] ("C2"): NICH scalar-Gaussian, N=1M rows,
K=256 groups, D=1 feature.  One *step* = one full scoring pass: msc_score_value
over every row x every group, the [N, K] float matrix materialised in HBM (the
inner loop of entity_based_state_object::inplace_score_value for the whole
dataset).  Inputs (column, group tables) are resident in HBM before timing.

metric  score_value evals/sec = N*K*D*world / seconds-per-step    (weak scaling:
        every rank owns its own N-row shard; the scoring pass has no collective)
sweep   (extra object) rows/sec of one synchronous Gibbs sweep = fused
        leave-one-out score + CRP prior + sample, suff-stat accumulate, and the
        sum all-reduce of the additive tables across ranks (RCCL) + commit.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--groups", type=int, default=256)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    return ap.parse_args()


def cpu_baseline(K, sample_rows):
    """oracle (float restatement, kind 'port') through the virtual group API on this host's cores."""
    exe = os.path.join(ROOT, "oracle", "perf_group_cpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = subprocess.check_output([exe, "c2", str(sample_rows), str(K), str(threads)]).decode()
    r = json.loads(out.strip().splitlines()[-1])
    return {
        "value": r["score_evals_per_s_1core"], "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": "C2 rows 0..%d (N=%d x K=%d x D=1 = %.3g score_value evals) through the virtual "
                  "group API, float libm restatement; linear in N" % (sample_rows, sample_rows, K, r["evals"]),
        "perf_group_shape_evals_per_s": r["perf_group_evals_per_s"],
        "noop_api_overhead_s": r["noop_s"],
        "all_cores": {"value": r["score_evals_per_s_ncore"], "cores": threads},
    }


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_pmc.json, written by tools/summarize_prof.py from separate rocprofv3 --pmc passes)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for name, e in d.items():
            if kernel_substr in name and "hbm_bytes_per_launch" in e:
                best = (e["hbm_bytes_per_launch"]["total"], os.path.basename(f))
    return best


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import common_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knob for a 1-GPU box: MSC_BENCH_BACKEND=gloo puts every rank on cuda:0
        backend = os.environ.get("MSC_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if a.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)

    ctx = common_amd.Context(device=local)
    dev = ctx.torch_device
    N, K = a.rows, a.groups

    # synthetic C2 data: mixture of K unit-variance normals, centres N(0, 10^2)
    g = torch.Generator(device=dev)
    g.manual_seed(73 + rank)
    centres = torch.randn(K, generator=g, device=dev) * 10.0
    z = torch.randint(0, K, (N,), generator=g, device=dev, dtype=torch.int32)
    x = (centres[z.long()] + torch.randn(N, generator=g, device=dev)).to(torch.float32).contiguous()
    view = common_amd.DataView.from_tensors(ctx, [x])
    st = common_amd.State(ctx, [(common_amd.NICH, 0)], K)   # default hp mu=0,kappa=1,sigmasq=1,nu=1
    st.accumulate(view, z)                                   # suff-stats from the true components
    out = torch.empty((N, K), dtype=torch.float32, device=dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # one-off setup outside warmup and timing: derived tables, and the launch-shape selection the library does at the
    # first large pass of a context (abi.cpp run_score; ~10 ms)
    st.score_value(view, out=out)
    for _ in range(a.warmup):
        st.score_value(view, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    sync_all()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        st.score_value(view, out=out)
        ev[i][1].record()
    sync_all()
    dt = time.perf_counter() - t0
    kern_ms = sorted(s.elapsed_time(e) for s, e in ev)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    sweep = None
    if not a.no_sweep:
        try:
            sweep = run_sweep(a, ctx, st, view, z, world, rank, dist, sync_all)
        except common_amd.MicroscopesHipError as e:
            if e.code != -4:
                raise
            sweep = {"error": str(e)}

    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ms = dt / a.steps * 1e3
        evals = float(N) * K * 1 * world
        alg_bytes = 4.0 * N + 4.0 * N * K          # 4.016 B per eval for C2
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
        line = {
            "metric": "score_value evals/sec", "value": evals / (dt / a.steps), "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2 NICH scalar-Gaussian score_value pass, N=%d rows/GPU x K=%d groups x D=1, "
                                   "[N,K] f32 scores materialised" % (N, K),
                       "rows_per_gpu": N, "groups": K, "features": 1, "parallelism": "row-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (pmc_traffic("k_score_nich1") or (None, None))[0] if (N, K) == (1_000_000, 256) else None,
                         "traffic_source": (pmc_traffic("k_score_nich1") or (None, None))[1],
                         "kernel": "k_score_nich1", "kernel_avg_ms": kern_avg_ms,
                         "kernel_min_ms": kern_ms[0], "algorithmic_bytes_per_launch": alg_bytes},
        }
        if sweep is not None:
            line["sweep"] = sweep
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, a.cpu_sample_rows)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_sweep(a, ctx, st, view, z, world, rank, dist, sync_all):
    """rows/sec of one synchronous Gibbs sweep incl. the suff-stat all-reduce."""
    import torch
    import common_amd
    N = view.nrows
    zs = z.clone()
    st.set_alpha(1.0)
    drv = common_amd.dist.ShardedSweep(st, view, zs, first_global_row=rank * N)
    drv.rebuild_tables()          # suff-stats of the global assignment (a no-op exchange on one rank)

    def one(sweep_idx):
        drv.sweep(seed=73, sweep_index=sweep_idx)

    steps = max(1, min(a.steps, 20))
    for i in range(2):
        one(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        one(2 + i)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=ctx.torch_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"metric": "Gibbs-sweep rows/sec", "value": N * world / (dt / steps), "unit": "rows/s",
            "ms_per_sweep": dt / steps * 1e3, "steps": steps,
            "includes": "leave-one-out score + CRP prior + sample, accumulate, all-reduce(i64,f64), commit"}


if __name__ == "__main__":
    main()
