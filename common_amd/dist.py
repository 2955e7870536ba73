"""Row-sharded multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm), rows block-sharded, group tables replicated.

The scoring pass has no collective.  A Gibbs sweep has exactly one exchange: the sum all-reduce
of the additive suff-stat tables (int64 counts -> bit-exact, float64 sums) between sweeps; the
payload is K * O(10) * 8 bytes, i.e. latency-bound, so both tables travel in one float64 all-reduce
(counts below 2**53 add exactly as doubles) and nothing is bucketed or overlapped.
"""
import os

import torch
import torch.distributed as dist


def _force_exchange():
    """MSC_DIST_FORCE_EXCHANGE=1: run the collective path with one rank too (hardware check of the nccl = RCCL branch
    on a one-GPU box; a one-rank all-reduce returns its input)"""
    return os.environ.get("MSC_DIST_FORCE_EXCHANGE", "0") not in ("", "0")


def shard_rows(nrows, world, rank):
    """Contiguous block partition: (first global row, number of rows) of `rank`."""
    base, rem = divmod(int(nrows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def allreduce_tables(red_i64, red_f64, group=None, pack=None):
    """In-place SUM of the additive tables across ranks (no-op when not distributed).

    The payload is a few KB, so the cost is the collective's latency: both tables travel in ONE float64
    all-reduce.  The counts ride along as doubles -- integers below 2**53 add exactly and in any order, so they
    come back bit-exact -- and are copied into the int64 table again.  `pack`: a float64 scratch tensor of
    red_i64.numel() + red_f64.numel() elements to reuse (one is allocated otherwise)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _force_exchange()):
        return
    ni, nf = red_i64.numel(), red_f64.numel()
    if ni == 0 or nf == 0:                      # only one kind of table: nothing to merge
        if ni:
            dist.all_reduce(red_i64, op=dist.ReduceOp.SUM, group=group)
        if nf:
            dist.all_reduce(red_f64, op=dist.ReduceOp.SUM, group=group)
        return
    if pack is None or pack.numel() != ni + nf or pack.device != red_f64.device:
        pack = torch.empty(ni + nf, dtype=torch.float64, device=red_f64.device)
    pack[:ni].copy_(red_i64)                    # int64 -> float64, exact for |count| < 2**53
    pack[ni:].copy_(red_f64)
    dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
    red_i64.copy_(pack[:ni])                    # integer-valued doubles -> int64, exact
    red_f64.copy_(pack[ni:])


class ShardedSweep(object):
    """state + this rank's shard of the rows; sweep() = assign -> accumulate -> all-reduce -> commit.

    The exchange is pack (one launch of the library: both tables into one float64 buffer) -> ONE all_reduce -> unpack
    (one launch) -> commit: the launch count of the C / C++ driver (msc_sweep_step_sharded) plus the two copies a
    one-dtype collective needs.  The kernels a sweep takes are chosen by the rows of the WHOLE dataset (the shards'
    sum, told to the state once), so that the shards draw what an unsharded sweep draws."""

    def __init__(self, state, view, z, first_global_row, group=None, global_rows=None):
        self.state, self.view, self.z = state, view, z
        self.row_id0 = int(first_global_row)
        self.group = group
        self.red_i64, self.red_f64 = state.reduce_buffers()
        if global_rows is None:
            global_rows = int(view.nrows)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                t = torch.tensor([global_rows], dtype=torch.int64, device=self.red_f64.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                global_rows = int(t.item())
        self.global_rows = int(global_rows)
        state.set_sweep_rows(self.global_rows)

    def _exchange(self):
        pack = self.state.reduce_pack()
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.group)
        self.state.reduce_unpack()

    def rebuild_tables(self):
        """suff-stats of the global assignment: local accumulate, sum across ranks, commit."""
        self.state.accumulate(self.view, self.z, reset=True, commit=False)
        if not self._alone():
            self._exchange()
        self.state.commit_reduce()

    def _alone(self):
        if not (dist.is_available() and dist.is_initialized()):
            return True
        return dist.get_world_size(self.group) == 1 and not _force_exchange()

    def sweep(self, seed, sweep_index):
        if self._alone():         # nothing to exchange: the whole step is one library call (graph-replayed when it repeats)
            self.state.sweep_step(self.view, self.z, seed=seed, sweep=sweep_index, row_id0=self.row_id0)
            return
        self.state.sweep_step_begin(self.view, self.z, seed=seed, sweep=sweep_index, row_id0=self.row_id0)
        self._exchange()
        self.state.commit_reduce()
