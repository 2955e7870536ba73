"""Worker of tests/test_gpu_dist.py, one process per rank under torch.distributed.run.  Every rank builds the same
seeded dataset, keeps its contiguous row shard, and runs sharded Gibbs sweeps (msc_sweep_step_begin -> all-reduce of the
additive tables -> msc_state_commit_reduce, common_amd.dist.ShardedSweep).  Rank 0 also runs the same sweeps unsharded
(msc_sweep_step) in its own process and compares: assignments, counts, float suff-stats.

On a one-GPU box every rank uses cuda:0 and the collective is gloo (MSC_DIST_BACKEND=gloo); on a multi-GPU node the
same script runs with nccl (= RCCL), one GPU per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import common_amd  # noqa: E402
from common_amd.dist import ShardedSweep, shard_rows  # noqa: E402


def dataset(dev, spec, N, K, seed):
    from tools.bench_configs import make_columns

    class _Ctx(object):
        torch_device = dev
    return make_columns(_Ctx, spec, N, K, seed)


def main():
    out_path, which = sys.argv[1], sys.argv[2]
    N, K, nsweeps = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    backend = os.environ.get("MSC_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    ctx = common_amd.Context(device=local)
    dev = ctx.torch_device
    spec = {"nich": [(common_amd.NICH, 0)],
            "mixed": [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 5), (common_amd.NICH, 0)]}[which]
    torch.manual_seed(1234)                                 # (make_columns draws the gp rates from the global generator)
    cols, z_all = dataset(dev, spec, N, K, 7)               # same seeds on every rank: the same global dataset
    lo, n = shard_rows(N, world, rank)
    shard_cols = [c[lo:lo + n].contiguous() for c in cols]
    view = common_amd.DataView.from_tensors(ctx, shard_cols)
    st = common_amd.State(ctx, spec, K)
    st.set_alpha(1.3)
    z = z_all[lo:lo + n].clone()
    drv = ShardedSweep(st, view, z, first_global_row=lo)
    drv.rebuild_tables()
    z_after = []
    for s in range(nsweeps):
        drv.sweep(seed=5, sweep_index=s)
        z_after.append(z.clone())
    torch.cuda.synchronize()
    # gather the shards' assignments on rank 0 (through the host: works for gloo and nccl alike)
    gathered = [None] * world
    dist.all_gather_object(gathered, [t.cpu().numpy() for t in z_after])
    counts = st.get_group_counts()
    ss = [st.get_ss(f) for f in range(len(spec))]
    seen = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)             # every rank adds 1 over the group the sweeps' all-reduce used
    res = None
    if rank == 0:
        z_sharded = [np.concatenate([gathered[r][s] for r in range(world)]) for s in range(nsweeps)]
        # the unsharded run of the same thing, in this process
        view1 = common_amd.DataView.from_tensors(ctx, cols)
        st1 = common_amd.State(ctx, spec, K)
        st1.set_alpha(1.3)
        z1 = z_all.clone()
        st1.accumulate(view1, z1)
        same = []
        for s in range(nsweeps):
            st1.sweep_step(view1, z1, seed=5, sweep=s)
            same.append(float((z1.cpu().numpy() == z_sharded[s]).mean()))
        zf = z_sharded[-1]
        res = {"world": world, "ranks_seen": int(seen.item()), "backend": backend, "N": N, "K": K, "which": which,
               "same_fraction_per_sweep": same,
               "counts_equal_bincount": bool(np.array_equal(counts, np.bincount(zf, minlength=K))),
               "counts_equal_unsharded": bool(np.array_equal(counts, st1.get_group_counts())),
               "moved_fraction": float((zf != z_all.cpu().numpy()).mean())}
        worst = 0.0
        for f in range(len(spec)):
            a, b = ss[f], st1.get_ss(f)
            for name in a.dtype.names:
                x, y = a[name].astype(np.float64), b[name].astype(np.float64)
                if np.issubdtype(a.dtype[name].base, np.integer):
                    res.setdefault("int_fields_equal", True)
                    res["int_fields_equal"] = res["int_fields_equal"] and bool(np.array_equal(x, y))
                else:
                    worst = max(worst, float((np.abs(x - y) / np.maximum(1.0, np.abs(y))).max()))
        res["float_fields_max_rel_diff"] = worst
        with open(out_path, "w") as fh:
            json.dump(res, fh)
        print(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
