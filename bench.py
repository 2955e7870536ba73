#!/usr/bin/env python
"""bench.py -- one JSON line for the driver (see the contract in the task statement).

N = 1 (`python bench.py`): BASELINE.json configs[1] ("C2"): NICH scalar-Gaussian, N=1M rows, K=256 groups,
  D=1 feature.  One *step* = one full scoring pass: msc_score_value over every row x every group, the [N, K] float
  matrix materialised in HBM (the inner loop of entity_based_state_object::inplace_score_value for the whole dataset).
  Inputs (column, group tables) are resident in HBM before timing; the derived tables are current (k_prepare, 9 us,
  runs when suff-stats change, not per pass).  metric = score_value evals/sec = N*K*D / seconds per step.
  Beside it, as extra objects on the same line: `sweep` (C2 as a Gibbs sweep), `c3`, `c4`, `c5_shard` (the other
  BASELINE configs that fit one GPU, each with its own roofline figure), `cpu_baseline`.

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[4] ("C5"): NICH, K=1024,
  12.5 M rows per rank (N = 100 M at 8 ranks; weak scaling), rows block-sharded, group tables replicated.  One *step*
  = one synchronous Gibbs sweep: fused leave-one-out score + CRP prior + sample (nothing materialised), suff-stat
  accumulate, ONE sum all-reduce of the additive tables across ranks (RCCL over xGMI), commit + prepare -- all inside
  the timed region.  metric = Gibbs-sweep rows/sec = rows of all ranks / seconds per step.  The N = 1 line carries the
  same workload at one rank as `c5_shard`, so a weak-scaling ratio is value(N) / (N * c5_shard.value).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F64_PEAK_TF = 78.6      # dense f64 matrix peak (MI355X spec; the table's dense figures, no sparsity)
# vector-pipe issue roof for the kernels that are bound by it (C3 and the fused sweep): one wave
# instruction per SIMD every 4 cycles (v_fma_f32 "one wave alone: 4", MI355X_MICROARCH.md per-instruction table),
# 256 CUs x 4 SIMDs at the 2.4 GHz the guide's constants are quoted at = 6.14e11 wave-instructions/s
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0
# ... which is what ONE wave on a SIMD sustains.  With two or more waves on it a SIMD issues a plain f32 / integer vector
# instruction every 2 cycles (it is 32 lanes wide), a transcendental every 8, an f64 or packed-f32 one every 4
# (tools/microbench/valu_rate.hip, profiles/r05_valu_rate.txt: 2.0 / 8.1 / 4.1-4.3 cycles at the clock the run held).  So the
# lines also carry the kernel's instructions priced at those costs against the SIMD-cycles it had: frac_weighted.
VALU_PLAIN_CYCLES, VALU_TRANS_CYCLES = 2.0, 8.0
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9


def valu_weighted(insts, busy_slots, ms):
    """SQ_ACTIVE_INST_VALU counts one slot a plain instruction and two a transcendental: busy - insts = transcendentals"""
    trans = max(0.0, busy_slots - insts)
    plain = max(0.0, insts - trans)
    cyc = plain * VALU_PLAIN_CYCLES + trans * VALU_TRANS_CYCLES
    return {"frac_weighted": cyc / (ms * 1e-3 * SIMD_CYCLES_PER_S), "weighted_simd_cycles_per_launch": cyc,
            "issue_cost_cycles": {"plain": VALU_PLAIN_CYCLES, "transcendental": VALU_TRANS_CYCLES,
                                  "source": "tools/microbench/valu_rate.hip, two or more waves a SIMD (profiles/r05_valu_rate.txt)"},
            "note": "frac / busy_frac are against ONE wave's issue rate (an instruction per 4 cycles); frac_weighted prices "
                    "plain instructions at the SIMD's 2 cycles and transcendentals at 8, f64 / packed ones (4) counted as plain"}
C5_ROWS_PER_RANK = 12_500_000
C5_GROUPS = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults (None): 500 / 200 for the C2 pass at N = 1 (0.15 ms a step), 100 / 20 for the C5 sweep at N > 1 (5 ms a step).
    # The first ~100 launches after an idle stretch run up to 20 % slower while the clocks come up
    # (profiles/r02_launch_transient.txt): the warm-up has to be longer than that
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--groups", type=int, default=256)
    ap.add_argument("--c5-rows", type=int, default=C5_ROWS_PER_RANK, help="rows per rank of the C5 sweep")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c3 / c4 / c5_shard objects at N = 1")
    ap.add_argument("--tune", action="store_true", help="msc_score_tune before the C2 pass (off: the default launch shape)")
    ap.add_argument("--alloc", choices=("default", "probed", "torch"), default="default",
                    help="where the C2 score matrix comes from: msc_device_alloc (the library's default, placed), "
                         "msc_device_alloc_probed (--probe-alloc candidates, all probed), or a plain torch.empty")
    ap.add_argument("--probe-alloc", type=int, default=24, help="candidates of msc_device_alloc_probed (--alloc probed)")
    ap.add_argument("--min-region-ms", type=float, default=10.0,
                    help="a timed region shorter than this gets a longer one measured beside it (config.long_region); "
                         "`value` is always the --steps region")
    return ap.parse_args()


def cpu_baseline(K, sample_rows):
    """oracle (float restatement, kind 'port') through the virtual group API on this host's cores."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])     # (a no-op when everything is current)
    exe = os.path.join(ROOT, "oracle", "perf_group_cpu")
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


class PmcLookupError(RuntimeError):
    pass


def pmc_entry(kernel, key):
    """(value per launch, file) of counter `key` for the kernel INSTANTIATION `kernel` -- the exact name, as the library reports
    it (Context.last_kernel) and rocprofv3 spells it, e.g. "k_score_tile_roles<false, false, false>" -- from the newest
    committed PMC summary that holds it (profiles/*_pmc.json, written by tools/summarize_prof.py from separate rocprofv3
    --pmc passes).  (None, None) when no committed summary has that instantiation; PmcLookupError when a summary holds it
    more than once -- never a neighbouring instantiation's counters (round 4 matched by substring and printed the PAIR
    kernel's instruction count for C3: VERDICT r04)."""
    import glob
    want = kernel if kernel.startswith("msc::") else "msc::" + kernel
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        hits = [e for name, e in d.items() if name.replace("void ", "").strip() == want and key in e]
        if len(hits) > 1:
            raise PmcLookupError("%s holds %d entries named %s" % (f, len(hits), want))
        if hits:
            v = hits[0][key]
            best = (v["total"] if "total" in v else v["avg"], os.path.basename(f))
    return best


def traffic_of(kernel, alg_bytes):
    """{"traffic": HBM bytes per launch from the committed counters (corrected as MI355X_MICROARCH.md prescribes), "traffic_ratio":
    traffic / algorithmic bytes, "traffic_source": file} for the kernel instantiation, or traffic None"""
    t, src = pmc_entry(kernel, "hbm_bytes_per_launch")
    return {"traffic": t, "traffic_ratio": (t / alg_bytes) if (t is not None and alg_bytes) else None, "traffic_source": src,
            "traffic_kernel": kernel}


def timed(torch, fn, steps, warmup):
    """HIP events on the stream the library launches on (torch's current stream is the context's stream)."""
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return wall * 1e3, sum(ms) / len(ms), ms[0]


def c2_data(torch, dev, N, K, seed):
    """mixture of K unit-variance normals, centres N(0, 10^2); z = the true component"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.randn(K, generator=g, device=dev) * 10.0
    z = torch.randint(0, K, (N,), generator=g, device=dev, dtype=torch.int32)
    x = (centres[z.long()] + torch.randn(N, generator=g, device=dev)).to(torch.float32).contiguous()
    return x, z


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: become the launcher.  N child ranks under torch.distributed.run
    (fresh processes; this one has made no HIP call -- torch.cuda.device_count() does not initialise the GPU on this
    image -- and only waits for them), rank 0's JSON line goes straight to our stdout, the exit code is the child's."""
    import torch
    backend = os.environ.get("MSC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < a.gpus:
        print("bench.py: --gpus %d but %d GPU(s) visible; one rank per GPU over RCCL needs %d (MSC_BENCH_BACKEND=gloo "
              "puts every rank on cuda:0 as a rehearsal and says so in the line)" % (a.gpus, ndev, a.gpus), file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MSC_BENCH_SPAWNED="1")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is relayed: the JSON line to our stdout, anything else a library prints there (gloo's connection
    # notices) to stderr -- the driver reads ONE line
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    for ln in p.stdout:
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    return p.wait()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    import torch
    import torch.distributed as dist
    import common_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "none"
    # MSC_BENCH_FORCE_C5=1: the N > 1 code path (process group, C5 sweep, all-reduce inside the step) with whatever world
    # there is -- one rank over nccl on a one-GPU box checks the RCCL branch on hardware
    force_c5 = os.environ.get("MSC_BENCH_FORCE_C5", "0") not in ("", "0")
    if force_c5:
        os.environ.setdefault("MSC_DIST_FORCE_EXCHANGE", "1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if a.steps is None:
        a.steps = 100 if (world > 1 or force_c5) else 500
    if a.warmup is None:
        a.warmup = 20 if (world > 1 or force_c5) else 200
    if a.gpus != world and not force_c5:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d: launch %d ranks, or run `python bench.py --gpus %d` and let it "
                  "launch them" % (a.gpus, world, a.gpus, a.gpus), file=sys.stderr)
        sys.exit(2)
    one_rank = None
    stdout_fd = None
    if world > 1 or force_c5:
        # the collective library announces itself on STDOUT when its communicator is created (RCCL: five lines of
        # versions; gloo: its connection notices): file descriptor 1 points at stderr until the JSON line is due
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knob for a 1-GPU box: MSC_BENCH_BACKEND=gloo puts every rank on cuda:0
        backend = os.environ.get("MSC_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = 0
        dry = os.environ.get("MSC_BENCH_DRYRUN", "0") not in ("", "0")
        if not dry:
            torch.cuda.set_device(local)
    if (world > 1 or force_c5) and dry:
        # the launch plumbing alone, on the CPU (tests/test_bench_launch_cpu.py): ranks, rendezvous, one sum over the
        # process group, the line on stdout and nothing else there -- no GPU call, no measurement, and the line says so
        dist.init_process_group("gloo")
        seen = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        names = [None] * world
        dist.all_gather_object(names, "rank %d pid %d" % (rank, os.getpid()))
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
        if rank == 0:
            print(json.dumps({"dry_run": True, "metric": "Gibbs-sweep rows/sec", "value": None, "n_gpus": world,
                              "ranks_seen": int(seen.item()), "rank_devices": names, "steps": a.steps, "warmup": a.warmup,
                              "config": {"backend": "gloo", "workload": "none (MSC_BENCH_DRYRUN: launch plumbing only)"}}), flush=True)
        os.dup2(2, 1)
        dist.barrier()
        dist.destroy_process_group()
        return
    ctx = common_amd.Context(device=local)
    if world > 1 or force_c5:
        # the one-rank reference of the weak-scaling ratio, in this very job: rank 0 runs its shard's sweep step alone,
        # before the process group forms (the other ranks wait in the rendezvous, their GPUs idle)
        if rank == 0:
            one_rank = extra_c5(a, torch, common_amd, ctx)
            torch.cuda.empty_cache()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1 or force_c5:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1 or force_c5:
        line = run_c5(a, torch, dist, common_amd, ctx, world, rank, backend, sync_all, one_rank)
    else:
        line = run_c2(a, torch, dist, common_amd, ctx, sync_all)
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1 or force_c5:
        os.dup2(2, 1)                                       # (whatever the teardown prints is not the line)
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: config C5, the row-sharded sweep
# ---------------------------------------------------------------------------------------------------------------------
def c5_setup(a, torch, common_amd, ctx, world, rank, nrows):
    x, z = c2_data(torch, ctx.torch_device, nrows, C5_GROUPS, 73 + rank)
    view = common_amd.DataView.from_tensors(ctx, [x])
    st = common_amd.State(ctx, [(common_amd.NICH, 0)], C5_GROUPS)        # default hp mu=0,kappa=1,sigmasq=1,nu=1
    st.set_alpha(1.0)
    drv = common_amd.dist.ShardedSweep(st, view, z, first_global_row=rank * nrows)
    drv.rebuild_tables()                     # suff-stats of the GLOBAL assignment: accumulate, all-reduce, commit
    return x, z, view, st, drv


def sweep_kernel_ms(torch, st, view, z, steps=20):
    """average duration of the fused sweep kernel alone (k_sweep_nich1_t), HIP events around msc_sweep_assign on a copy
    of z -- outside the timed region; the rocprof summary in profiles/ must agree"""
    zc = z.clone()
    _, avg, mn = timed(torch, lambda: st.sweep_assign(view, zc, seed=11, sweep=0), steps, 5)
    return avg, mn, st.ctx.last_kernel("sweep")


def sweep_roofline(nrows, K, kern_ms, kernel):
    """`kernel`: the instantiation the library launched (Context.last_kernel("sweep")), e.g. "k_sweep_nich1_t<16>" """
    evals = float(nrows) * K
    alg_bytes = 12.0 * nrows                                  # x, z in, z out
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # (the committed counters are per launch of this very shape and instantiation: <16> = K 1024 on the 12.5 M-row shard,
    # <4> = C2's 1M rows; other row counts get no counter-derived figure)
    standard = (nrows, K) in ((C5_ROWS_PER_RANK, C5_GROUPS), (1_000_000, 256))
    insts, src = pmc_entry(kernel, "SQ_INSTS_VALU") if standard else (None, None)
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "kernel": kernel, "kernel_avg_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
         "note": "the fused sweep materialises nothing: 12 B per row of HBM traffic, so HBM is not what binds it "
                 "(SURVEY 8d: transcendental / vector issue rate); valu_issue = SQ_INSTS_VALU of the committed PMC pass "
                 "over this run's kernel time, against one wave instruction per SIMD per 4 cycles",
         "evals_per_s": evals / (kern_ms * 1e-3)}
    r.update(traffic_of(kernel, alg_bytes) if standard else {"traffic": None})
    if insts is not None:
        # SQ_INSTS_VALU counts wave instructions; the roof is one per SIMD every 4 cycles
        r["valu_issue"] = {"wave_insts_per_launch": insts, "source": src, "achieved": insts / (kern_ms * 1e-3),
                           "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
                           "frac": insts / (kern_ms * 1e-3) / VALU_ISSUE_PEAK}
        # SQ_ACTIVE_INST_VALU counts the 4-cycle slots the vector ALUs were held (a transcendental holds two): the same
        # roof with every instruction at its own issue cost
        busy, _ = pmc_entry(kernel, "SQ_ACTIVE_INST_VALU")
        if busy is not None:
            r["valu_issue"]["busy_slots_per_launch"] = busy
            r["valu_issue"]["busy_frac"] = busy / (kern_ms * 1e-3) / VALU_ISSUE_PEAK
            r["valu_issue"].update(valu_weighted(insts, busy, kern_ms))
    return r


def rccl_version(torch, backend):
    """the collective library the all-reduce ran on, as the process group itself reports it"""
    if backend != "nccl":
        return None
    try:
        import torch.distributed as dist
        pg = dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda"))
        ver = lambda v: ".".join(str(i) for i in v) if isinstance(v, (tuple, list)) else str(v)   # noqa: E731
        return {"runtime": ver(pg.get_runtime_nccl_version()), "build": ver(pg.get_build_nccl_version()),
                "hip": torch.version.hip, "note": 'torch.distributed backend "nccl" on ROCm is RCCL'}
    except Exception:
        v = torch.cuda.nccl.version()
        return {"runtime": ".".join(str(i) for i in v), "hip": torch.version.hip}


def run_c5(a, torch, dist, common_amd, ctx, world, rank, backend, sync_all, one_rank=None):
    nrows = a.c5_rows
    x, z, view, st, drv = c5_setup(a, torch, common_amd, ctx, world, rank, nrows)
    idx = [0]

    def one():
        drv.sweep(seed=73, sweep_index=idx[0])
        idx[0] += 1

    prewarm = max(0, 5 - a.warmup)                           # (25 ms of work before anything counts: the clocks, as in run_c2)
    for _ in range(prewarm + a.warmup):
        one()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    sync_all()
    dt_own = time.perf_counter() - t0
    # max over ranks = the job's time; min beside it (the spread between ranks), and a count of the ranks that took part:
    # every rank adds 1 over the same process group the sweeps' all-reduce used
    t = torch.tensor([dt_own, -dt_own], device=ctx.torch_device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt_min = float(t[0].item()), -float(t[1].item())
    seen = torch.ones(1, device=ctx.torch_device, dtype=torch.int64)
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    ranks_seen = int(seen.item())
    devices = [None] * world
    dist.all_gather_object(devices, "%s cuda:%d %s" % (os.uname().nodename, ctx.device,
                                                        torch.cuda.get_device_properties(ctx.device).name))
    kern_ms, kern_min, kern_name = sweep_kernel_ms(torch, st, view, z)
    # every rank holds the same tables after the exchange: group sizes sum to the global row count
    total = int(st.get_group_counts().astype("int64").sum())
    assert total == nrows * world, (total, nrows * world)
    if rank != 0:
        return None
    ms = dt / a.steps * 1e3
    value = float(nrows) * world / (dt / a.steps)
    ref = None
    if one_rank is not None:
        ref = {"value": one_rank["value"], "unit": "rows/s", "ms_per_sweep": one_rank["ms_per_sweep"],
               "how": "rank 0 alone, its own %d-row shard, msc_sweep_step (no exchange), before the process group formed" % nrows}
    return {
        "metric": "Gibbs-sweep rows/sec", "value": value, "unit": "rows/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "rccl_version": rccl_version(torch, backend),
        "rank_devices": devices,
        "ms_per_step_ranks": {"max": ms, "min": dt_min / a.steps * 1e3},
        "one_rank_reference": ref,
        "weak_scaling_eff": (value / (world * ref["value"])) if ref else None,
        "steps": a.steps, "warmup": a.warmup, "preconditioning_steps": prewarm, "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C5 NICH N=%d rows (%d per GPU) x K=%d groups, row-sharded synchronous Gibbs sweep: fused "
                               "leave-one-out score + CRP prior + sample, accumulate, ONE sum all-reduce of the additive "
                               "suff-stat tables (%s), commit + prepare" % (nrows * world, nrows, C5_GROUPS,
                                                                            "RCCL" if backend == "nccl" else backend),
                   "rows_per_gpu": nrows, "groups": C5_GROUPS, "features": 1, "parallelism": "row-shard x%d" % world,
                   "collective": "1 x all_reduce(sum, f64[%d]) per sweep" % (drv.red_i64.numel() + drv.red_f64.numel()),
                   "backend": backend},
        "evals_per_s": float(nrows) * world * C5_GROUPS / (dt / a.steps),
        "note": "weak_scaling_eff = value / (n_gpus * one_rank_reference.value); the N = 1 line reports the same one-rank "
                "workload as c5_shard, its own `value` is the C2 scoring pass (evals/s), the metric BASELINE.json quotes "
                "for one GPU",
        "roofline": sweep_roofline(nrows, C5_GROUPS, kern_ms, kern_name),
    }


# ---------------------------------------------------------------------------------------------------------------------
# N = 1: config C2 (the headline), plus the other single-GPU configs as extra objects
# ---------------------------------------------------------------------------------------------------------------------
def run_c2(a, torch, dist, common_amd, ctx, sync_all):
    dev = ctx.torch_device
    N, K = a.rows, a.groups
    x, z = c2_data(torch, dev, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, [x])
    st = common_amd.State(ctx, [(common_amd.NICH, 0)], K)   # default hp mu=0,kappa=1,sigmasq=1,nu=1
    st.accumulate(view, z)                                   # suff-stats from the true components
    # the [N, K] matrix is caller-owned; where the driver places it decides between ~5.5 and ~7.0 TB/s for the same
    # kernel (profiles/r02_placement_study.txt).  The headline uses the library's DEFAULT allocator (msc_device_alloc:
    # from 64 MiB on a buffer mapped from 32 MiB chunks, up to twelve candidates tried until one fills at 6.65 TB/s, six
    # when they all fill alike -- include/microscopes_hip.h); the
    # same pass into a plain torch.empty is measured beside it (roofline.frac_caller_alloc)
    if a.alloc == "torch":
        out = torch.empty((N, K), dtype=torch.float32, device=dev)
        placement = {"allocator": "torch.empty"}
    elif a.alloc == "probed":
        out, rates, kept = ctx.alloc_probed((N, K), torch.float32, candidates=a.probe_alloc)
        placement = {"allocator": "msc_device_alloc_probed", "candidates_fill_GBps": [round(r, 1) for r in rates], "kept": kept}
    else:
        out = ctx.alloc((N, K), torch.float32)
        rates, kept = ctx.alloc_stats()
        placement = {"allocator": "msc_device_alloc (library default)", "candidates_fill_GBps": [round(r, 1) for r in rates],
                     "kept": kept,
                     # (include/microscopes_hip.h: non-temporal stores into a buffer probed at >= 6.65 TB/s, plain ones otherwise)
                     "stores": "non-temporal" if rates and rates[kept] >= 6650.0 else "plain"}
    tuned = None
    if a.tune:
        tuned = st.score_tune(view, out)                     # explicit and synchronous; never inside msc_score_value
    st.score_value(view, out=out)                            # derived tables (k_prepare) are built here, once
    # the clocks need ~100 launches (~15 ms) after an idle stretch (profiles/r02_launch_transient.txt).  That is the
    # device's state, not the bench's warm-up: 200 conditioning passes run first and are reported as such
    # (config.preconditioning_passes); then EXACTLY --warmup untimed passes and EXACTLY --steps timed ones, as asked --
    # `steps` / `warmup` in the line are the command line's
    precondition = 200
    for _ in range(precondition + a.warmup):
        st.score_value(view, out=out)
    # launches of the headline kernel so far (for the trace summary): msc_score_tune times seven shapes x eight passes and then
    # non-temporal against plain stores with the winning shape, 2 x 7 more
    launched = 1 + precondition + a.warmup + (7 * 8 + 2 * 7 if a.tune else 0)

    def region(buf, steps):
        """HIP events over the timed region, on the stream the library launches on: ONE pair around the launches (a
        pair per launch puts two event packets between consecutive kernels and reads ~5 % long against the rocprof
        trace) -> (wall seconds, kernel ms per step)"""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            st.score_value(view, out=buf)
        ev1.record()
        sync_all()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps
    dt, kern_avg_ms = region(out, a.steps)
    headline_kernel = ctx.last_kernel("score")              # the instantiation the timed region ran, as rocprofv3 spells it
    timed_from = launched
    launched += a.steps
    long_region = None
    if dt * 1e3 < a.min_region_ms:
        # a 3 ms region is a handful of clock ticks of whoever samples it from outside: a region of >= 200 steps is
        # measured BESIDE the one that was asked for (config.long_region); `value` stays with the steps asked for
        lsteps = max(200, int(a.min_region_ms / (dt / a.steps * 1e3)) + 1)
        ldt, lkern = region(out, lsteps)
        long_region = {"steps": lsteps, "ms_per_step": ldt / lsteps * 1e3, "kernel_avg_ms": lkern,
                       "value": float(N) * K / (ldt / lsteps),
                       "frac": (4.0 * N + 4.0 * N * K) / (lkern * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # (per-launch spread, outside the timed region)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(a.steps, 50))]
    for s_, e_ in ev:
        s_.record()
        st.score_value(view, out=out)
        e_.record()
    torch.cuda.synchronize()
    kern_ms = sorted(s_.elapsed_time(e_) for s_, e_ in ev)

    ms = dt / a.steps * 1e3
    evals = float(N) * K
    alg_bytes = 4.0 * N + 4.0 * N * K          # 4.016 B per eval for C2
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_entry(headline_kernel, "hbm_bytes_per_launch")
    # the same pass into a caller-owned torch.empty (what a caller gets who does not ask the library for the matrix)
    caller = None
    if "stores" in placement:                               # (what the library did, not what the probe's figure suggests)
        placement["stores"] = "non-temporal" if headline_kernel.endswith("true>") else "plain"
    if a.alloc != "torch":
        tbuf = torch.empty((N, K), dtype=torch.float32, device=dev)
        for _ in range(50):
            st.score_value(view, out=tbuf)
        _, c_ms = region(tbuf, max(200, min(a.steps, 500)))
        caller = {"allocator": "torch.empty", "stores": "plain", "kernel": ctx.last_kernel("score"), "kernel_avg_ms": c_ms, "achieved": alg_bytes / (c_ms * 1e-3) / 1e9,
                  "frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del tbuf
    line = {
        "metric": "score_value evals/sec", "value": evals / (dt / a.steps), "unit": "evals/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C2 NICH scalar-Gaussian score_value pass, N=%d rows/GPU x K=%d groups x D=1, "
                               "[N,K] f32 scores materialised; group tables prepared once before timing (k_prepare, "
                               "9 us, runs when suff-stats change)" % (N, K),
                   "rows_per_gpu": N, "groups": K, "features": 1, "parallelism": "row-shard x1",
                   "launch_shape": "msc_score_tune -> %s" % (tuned,) if tuned else "default (4 rows x 2 visits)",
                   "preconditioning_passes": precondition, "long_region": long_region,
                   "score_matrix": placement},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "frac_caller_alloc": caller["frac"] if caller else None, "caller_alloc": caller,
                     "traffic": traffic if (N, K) == (1_000_000, 256) else None,
                     "traffic_ratio": (traffic / alg_bytes) if (traffic is not None and (N, K) == (1_000_000, 256)) else None,
                     "traffic_source": traffic_src,
                     "kernel": headline_kernel, "kernel_avg_ms": kern_avg_ms,
                     "timed_region_launches": [timed_from, timed_from + a.steps],   # of this kernel, in launch order

                     "kernel_min_ms_single_launch_events": kern_ms[0], "algorithmic_bytes_per_launch": alg_bytes,
                     "out_va": "%#x" % out.data_ptr()},
    }
    if not a.no_sweep:
        line["sweep"] = c2_sweep(a, torch, common_amd, ctx, st, view, z)
    del out
    if not a.no_extra:
        for name, fn in (("c3", extra_c3), ("c4", extra_c4), ("c5_shard", extra_c5)):
            torch.cuda.empty_cache()
            try:
                line[name] = fn(a, torch, common_amd, ctx)
            except Exception as e:                            # an extra object never takes the headline down
                line[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(K, a.cpu_sample_rows)
    return line


def c2_sweep(a, torch, common_amd, ctx, st, view, z):
    """C2 as one synchronous Gibbs sweep on one rank (no exchange to do): rows/sec of msc_sweep_step."""
    N = view.nrows
    zs = z.clone()
    st.set_alpha(1.0)
    drv = common_amd.dist.ShardedSweep(st, view, zs, first_global_row=0)
    drv.rebuild_tables()
    idx = [0]

    def one():
        drv.sweep(seed=73, sweep_index=idx[0])
        idx[0] += 1
    steps = max(1, min(a.steps, 100))
    wall_ms, avg, mn = timed(torch, one, steps, min(20, steps))
    kern_ms, _, kern_name = sweep_kernel_ms(torch, st, view, zs)
    return {"metric": "Gibbs-sweep rows/sec", "value": N / (wall_ms * 1e-3), "unit": "rows/s",
            "ms_per_sweep": wall_ms, "steps": steps, "kernel": kern_name, "kernel_avg_ms": kern_ms,
            "roofline": sweep_roofline(N, st.K, kern_ms, kern_name),
            "includes": "leave-one-out score + CRP prior + sample (fused, nothing materialised), accumulate, "
                        "commit + prepare; one rank: no exchange (msc_sweep_step)"}


def extra_c3(a, torch, common_amd, ctx):
    """BASELINE configs[2]: mixed bb+gp+dd32+nich x16, N=1M, K=256, D=64; scoring pass + sweep."""
    from tools.bench_configs import make_columns
    N, K = 1_000_000, 256
    spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    steps = max(3, min(a.steps, 20))
    wall, avg, mn = timed(torch, lambda: st.score_value(view, out=out), steps, 10)
    kern = ctx.last_kernel("score")                          # "k_score_tile_roles<false, false, false>"
    rowbytes = sum(c.element_size() * (c.shape[1] if c.dim() > 1 else 1) for c in cols)
    alg = float(N) * rowbytes + 4.0 * N * K

    def valu(kernel, ms):
        """vector-issue figures of the kernel instantiation from the committed counters, over this run's time"""
        insts, src = pmc_entry(kernel, "SQ_INSTS_VALU")
        if insts is None:
            return {"error": "no committed counters for %s" % kernel}
        v = {"kernel": kernel, "wave_insts_per_launch": insts, "source": src, "peak": VALU_ISSUE_PEAK,
             "achieved": insts / (ms * 1e-3), "frac": insts / (ms * 1e-3) / VALU_ISSUE_PEAK, "unit": "wave-instructions/s"}
        busy, _ = pmc_entry(kernel, "SQ_ACTIVE_INST_VALU")
        if busy is not None:
            v["busy_slots_per_launch"] = busy
            v["busy_frac"] = busy / (ms * 1e-3) / VALU_ISSUE_PEAK
            v.update(valu_weighted(insts, busy, ms))
        return v
    r = {"workload": "C3 mixed bb+gp+dd32+nich x16, N=1M, K=256, D=64, scoring pass", "ms": avg, "ms_min": mn,
         "evals_per_s": float(N) * K * len(spec) / (avg * 1e-3), "kernel": kern,
         "roofline": {"bound": "hbm", "achieved": alg / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
                      "note": "not what binds it (SURVEY 8d): vector issue rate, see valu_issue"}}
    r["roofline"].update(traffic_of(kern, alg))
    r["roofline"]["valu_issue"] = valu(kern, avg)
    st.set_alpha(1.0)
    zs = z.clone()
    idx = [0]

    def one():
        st.sweep_step(view, zs, seed=73, sweep=idx[0])
        idx[0] += 1
    w, savg, smn = timed(torch, one, max(3, steps // 2), 1)
    r["sweep_ms"] = savg
    r["sweep_rows_per_s"] = N / (savg * 1e-3)
    # the fused assignment kernel of the step alone (leave-one-out pass, accumulate and commit are the step's other launches)
    zc = zs.clone()
    _, kavg, _ = timed(torch, lambda: st.sweep_assign(view, zc, seed=11, sweep=0), max(3, steps // 2), 2)
    skern = ctx.last_kernel("sweep")                         # "k_sweep_tile_roles<0, false>"
    salg = float(N) * (rowbytes + 8.0)                       # N (rowbytes + 4 + 4): the rows, z in, z out
    r["sweep"] = {"kernel": skern, "assign_ms": kavg, "note": "assign = k_loo_own_lds + this kernel; step = assign + k_accumulate + commit",
                  "roofline": dict({"bound": "hbm", "achieved": salg / (kavg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": salg / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": salg,
                                    "valu_issue": valu(skern, kavg)}, **traffic_of(skern, salg))}
    # the same columns at other table sizes (a CRP table is not a multiple of 256): at most 128 groups and the groups
    # beyond a tile up to 384 run on the lane <-> row kernel (k_score_tail_rows), whose cost follows the groups
    del out, st
    other = []
    for k in (32, 128, 300):
        z2 = z % k
        s2 = common_amd.State(ctx, spec, k)
        s2.set_alpha(1.0)
        s2.accumulate(view, z2)
        o2 = torch.empty((N, k), dtype=torch.float32, device=ctx.torch_device)
        _, sc_avg, _ = timed(torch, lambda: s2.score_value(view, out=o2), max(3, steps // 2), 2)
        zz, it = z2.clone(), [0]

        def step():
            s2.sweep_step(view, zz, seed=73, sweep=it[0])
            it[0] += 1
        _, sw_avg, _ = timed(torch, step, max(3, steps // 2), 2)
        other.append({"K": k, "score_ms": sc_avg, "sweep_ms": sw_avg, "evals_per_s": float(N) * k * len(spec) / (sc_avg * 1e-3),
                      "sweep_rows_per_s": N / (sw_avg * 1e-3)})
        del o2, s2
    r["other_table_sizes"] = other
    # other feature lists on the same rows (not BASELINE configs; the kernels round 4 added for them): categoricals only
    # (k_score_lookups), independent Gaussians only (k_score_nich_pack), mostly Gaussians (the same with a few lookups from L2)
    del view
    plans = []
    BB, NICH = common_amd.BB, common_amd.NICH
    for name, sp in (("32 bb", [(BB, 0)] * 32), ("16 nich", [(NICH, 0)] * 16), ("8 bb + 8 nich", [(BB, 0)] * 8 + [(NICH, 0)] * 8)):
        cols2, z3 = make_columns(ctx, sp, N, K, 79)
        v2 = common_amd.DataView.from_tensors(ctx, cols2)
        s3 = common_amd.State(ctx, sp, K)
        s3.set_alpha(1.0)
        s3.accumulate(v2, z3)
        o3 = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
        _, sc_avg, _ = timed(torch, lambda: s3.score_value(v2, out=o3), max(3, steps // 2), 2)
        zz3, it3 = z3.clone(), [0]

        def step3():
            s3.sweep_step(v2, zz3, seed=79, sweep=it3[0])
            it3[0] += 1
        _, sw_avg, _ = timed(torch, step3, max(3, steps // 2), 2)
        plans.append({"features": name, "K": K, "score_ms": sc_avg, "sweep_ms": sw_avg, "evals_per_s": float(N) * K * len(sp) / (sc_avg * 1e-3),
                      "sweep_rows_per_s": N / (sw_avg * 1e-3)})
        del o3, s3, v2, cols2
    r["other_feature_lists"] = plans
    return r


MFMA_F32_PEAK_TF = 157.3     # dense f32 matrix peak (the roof BASELINE.md names for C4)


def extra_c4(a, torch, common_amd, ctx):
    """BASELINE configs[3]: NIW dim 32, N=256k, K=128; the scoring pass on the f64 matrix pipe (the default: the 1e-6
    path) and on the f32 matrix pipe (MSC_SCORE_NIW_F32, the roof BASELINE.md names; ~2e-5 of the twin), each with the
    fraction on SURVEY 8d's flop count and on the flops the kernel EXECUTES (it skips the zero triangle of the factor)."""
    from tools.bench_configs import make_columns
    N, K, d = 262_144, 128, 32
    spec = [(common_amd.NIW, d)]
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    steps = max(3, min(a.steps, 20))
    flops = 2.0 * d * d * N * K                 # SURVEY 8d's count (the full d x d contraction per pair)
    nb = (d + 15) // 16
    executed = flops * (nb + 1) / (2.0 * nb)    # blocks on or below the diagonal of the triangular factor (3/4 at dim 32)

    def leg(f32, peak):
        wall, avg, mn = timed(torch, lambda: st.score_value(view, out=out, niw_f32=f32), steps, 10)
        kernel = ctx.last_kernel("score")                    # "k_score_niw64<2, 4, false, false>" / "k_score_niw<2, false, false>"
        tf = flops / (avg * 1e-3) / 1e12
        busy, src = pmc_entry(kernel, "SQ_VALU_MFMA_BUSY_CYCLES")
        cycles, _ = pmc_entry(kernel, "GRBM_GUI_ACTIVE")
        r = {"ms": avg, "ms_min": mn, "kernel": kernel, "tflops_survey_count": tf, "peak": peak, "unit": "TFLOP/s",
             "frac_survey_count": tf / peak,
             "frac_executed": (executed if not f32 else flops) / (avg * 1e-3) / 1e12 / peak,
             "flops_executed_per_launch": executed if not f32 else flops}
        r.update(traffic_of(kernel, 4.0 * d * N + 4.0 * N * K))
        if busy is not None:
            # (counters of a COMMITTED profile of this instantiation -- profiles/<source>, its tag names the round --, not of
            # this run: GRBM_GUI_ACTIVE sums the eight XCDs' busy cycles, the matrix pipe's over 4 SIMDs x 256 CUs)
            r["mfma_busy"] = {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": cycles, "source": src,
                              "frac": (busy / (1024.0 * cycles / 8.0)) if cycles else None,
                              "note": "committed PMC pass of this instantiation; busy cycles / (4 SIMDs x 256 CUs x kernel cycles)"}
        return r
    f64 = leg(False, MFMA_F64_PEAK_TF)
    ref = out.clone()
    f32 = leg(True, MFMA_F32_PEAK_TF)
    # what the f32 pipe costs in accuracy, against the f64 kernel's own output on the same state (the f64 kernel is the
    # one held to the twin at 1e-6: tests/test_gpu_score.py)
    diff = (out - ref).abs() / ref.abs().clamp_min(1.0)
    f32["max_rel_err_vs_f64_kernel"] = float(diff.max().item())
    return {"workload": "C4 NIW dim=32, N=256k, K=128, scoring pass", "ms": f64["ms"], "ms_min": f64["ms_min"],
            "evals_per_s": float(N) * K / (f64["ms"] * 1e-3), "kernel": f64["kernel"],
            "roofline": {"bound": "mfma", "achieved": f64["tflops_survey_count"], "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                         "frac": f64["frac_survey_count"], "frac_executed": f64["frac_executed"],
                         "algorithmic_flops_per_launch": flops, "traffic": f64["traffic"], "traffic_ratio": f64["traffic_ratio"],
                         "note": "frac counts flops as SURVEY 8d does (2 d^2 per pair); the kernel skips the zero "
                                 "upper-right block of the triangular factor and executes 3/4 of them: frac_executed"},
            "f64": f64, "f32": f32}


def extra_c5(a, torch, common_amd, ctx):
    """BASELINE configs[4] at one rank: the 12.5 M-row shard, K=1024, the sweep step of the N > 1 bench."""
    nrows = a.c5_rows
    x, z, view, st, drv = c5_setup(a, torch, common_amd, ctx, 1, 0, nrows)
    idx = [0]

    def one():
        drv.sweep(seed=73, sweep_index=idx[0])
        idx[0] += 1
    steps = max(3, min(a.steps, 20))
    wall_ms, avg, mn = timed(torch, one, steps, 5)
    kern_ms, _, kern_name = sweep_kernel_ms(torch, st, view, z)
    return {"workload": "C5 shard: NICH %d rows x K=%d, one rank's sweep step (no exchange at one rank)" % (nrows, C5_GROUPS),
            "metric": "Gibbs-sweep rows/sec", "value": nrows / (wall_ms * 1e-3), "unit": "rows/s", "ms_per_sweep": wall_ms,
            "evals_per_s": float(nrows) * C5_GROUPS / (wall_ms * 1e-3),
            "roofline": sweep_roofline(nrows, C5_GROUPS, kern_ms, kern_name)}


if __name__ == "__main__":
    main()
