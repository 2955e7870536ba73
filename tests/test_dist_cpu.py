"""CPU, world_size 2 over gloo: the N>1 path of a sweep -- shard the rows, build the additive
tables per rank, sum all-reduce (the product's own allreduce_tables), compare with the oracle's
suff-stats of the whole data.  (The kernels themselves need a GPU; this covers the collective.)"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common_amd.dist import allreduce_tables, shard_rows


def test_shard_rows_partitions_exactly():
    for n in (0, 1, 7, 1000, 10**8):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _worker(rank, world, port, x, z, K, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, n = shard_rows(len(x), world, rank)
    xs, zs = x[lo:lo + n].astype(np.float64), z[lo:lo + n]
    # additive tables of one NICH feature, laid out as msc_state_reduce_buffers documents:
    # int64 [group sizes | counts], float64 [sum x | sum x^2]
    cnt = np.bincount(zs, minlength=K).astype(np.int64)
    red_i = torch.from_numpy(np.concatenate([cnt, cnt, np.array([2**45 + 3 * rank + 1], dtype=np.int64)]))   # (+ a count far beyond float32)
    red_f = torch.from_numpy(np.concatenate([np.bincount(zs, weights=xs, minlength=K),
                                             np.bincount(zs, weights=xs * xs, minlength=K)]))
    allreduce_tables(red_i, red_f)
    if rank == 0:
        np.save(out + "_i.npy", red_i.numpy())
        np.save(out + "_f.npy", red_f.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_of_sharded_tables_equals_whole_data(tmp_path):
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    N, K, world = 5001, 17, 2
    x = rng.normal(3, 2, N).astype(np.float32)
    z = rng.integers(0, K, N).astype(np.int32)
    out = str(tmp_path / "red")
    port = 29500 + int(rng.integers(0, 2000))
    mp.spawn(_worker, args=(world, port, x, z, K, out), nprocs=world, join=True)
    red_i, red_f = np.load(out + "_i.npy"), np.load(out + "_f.npy")
    F = orc.Family(orc.NICH, dict(mu=0., kappa=1., sigmasq=1., nu=1.), 0, "f64")
    ss = F.accumulate(K, x, z)
    assert np.array_equal(red_i[:K], ss["count"]) and np.array_equal(red_i[K:2 * K], ss["count"])  # bit-exact
    assert int(red_i[2 * K]) == 2 * 2**45 + 5      # the counts ride the float64 all-reduce: exact below 2**53
    n = ss["count"].astype(np.float64)
    mean = red_f[:K] / n
    ctv = red_f[K:] - n * mean * mean
    assert np.all(np.abs(mean - ss["mean"]) <= 1e-9 * np.maximum(1, np.abs(ss["mean"])))
    assert np.all(np.abs(ctv - ss["count_times_variance"]) <= 1e-9 * np.maximum(1, ss["count_times_variance"]))


def test_allreduce_is_a_noop_without_a_process_group():
    a, b = torch.arange(4), torch.ones(3, dtype=torch.float64)
    allreduce_tables(a, b)
    assert a.tolist() == [0, 1, 2, 3] and b.tolist() == [1.0, 1.0, 1.0]
