"""bench.py as the driver runs it: the N = 1 line carries every object without an "error" key, `--gpus N` launches its
own N ranks (rehearsed on the one-GPU box over gloo) and proves it in the line, and refuses when it cannot."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    e.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_every_object_and_no_error(gpu_ctx):
    rc, out, err = _run(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--c5-rows", "1000000"])
    assert rc == 0, err[-2000:]
    r = _line(out)
    assert r["metric"] == "score_value evals/sec" and r["n_gpus"] == 1
    for name in ("sweep", "c3", "c4", "c5_shard"):
        assert name in r and "error" not in r[name], (name, r.get(name))
    # the driver's short command is stretched to a region the sampler can see, and says so
    assert r["steps_requested"] == 20 and r["steps"] >= 200 and r["warmup"] >= 200 and r["warmup_requested"] == 5
    assert r["config"]["short_region"]["steps"] == 20
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and 0.4 < roof["frac"] < 1.0 and 0.4 < roof["frac_caller_alloc"] < 1.0
    assert r["config"]["score_matrix"]["allocator"].startswith("msc_device_alloc") and r["config"]["score_matrix"]["candidates_fill_GBps"]


def test_gpus_2_launches_two_ranks_by_itself(gpu_ctx):
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--c5-rows", "20000"], MSC_BENCH_BACKEND="gloo")
    assert rc == 0, err[-2000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and len(r["rank_devices"]) == 2
    assert r["metric"] == "Gibbs-sweep rows/sec" and r["scaling"] == "weak" and r["config"]["backend"] == "gloo"
    assert r["one_rank_reference"]["value"] > 0 and r["weak_scaling_eff"] > 0


def test_gpus_2_over_rccl_on_a_one_gpu_box_fails_loudly(gpu_ctx):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here")
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--c5-rows", "20000"])
    assert rc != 0 and "GPU(s) visible" in err and not [ln for ln in out.splitlines() if ln.startswith("{")]
