"""`python bench.py --gpus 8` as the launcher of its own ranks, rehearsed WITHOUT a GPU (MSC_BENCH_DRYRUN=1: the ranks form
the process group over gloo on the CPU, sum a one per rank and rank 0 prints the line; nothing is measured and the line
says so).  What it pins: the port / rendezvous on 127.0.0.1, the fan-out to eight ranks, `ranks_seen == 8`, ONE line on
stdout and only that.  The measured N > 1 path runs on the GPU box (tests/test_gpu_bench.py, at most four ranks on the one
card: the box allows six processes on it)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_ranks_launch_and_report_over_gloo():
    env = dict(os.environ, MSC_BENCH_BACKEND="gloo", MSC_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = p.stdout.decode().splitlines()
    assert len(out) == 1, out                                 # the line, nothing else on stdout
    r = json.loads(out[0])
    assert r["dry_run"] is True and r["value"] is None
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and len(r["rank_devices"]) == 8
    assert sorted(int(d.split()[1]) for d in r["rank_devices"]) == list(range(8))
    assert r["steps"] == 3 and r["warmup"] == 1


def test_gpus_must_agree_with_world_size():
    env = dict(os.environ, MSC_BENCH_BACKEND="gloo", MSC_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2 and b"WORLD_SIZE=2" in p.stderr and not p.stdout.strip()


def test_the_drivers_own_launch_line_over_gloo():
    """the command the driver uses for N > 1 -- torch.distributed.run around bench.py, one rank per GPU -- as a dry run of
    four ranks: rank 0 alone prints, and prints one line"""
    env = dict(os.environ, MSC_BENCH_BACKEND="gloo", MSC_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["ranks_seen"] == 4 and r["steps"] == 2 and r["warmup"] == 1 and r["dry_run"] is True
