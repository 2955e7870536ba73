"""msc_sweep_step (assign + rebuild in one call with fused launches; optionally replayed as a HIP graph once it
repeats) must give exactly what the two separate calls give, sweep after sweep, through every way a replay
can be interrupted."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import make_feature, recarray_of

pytestmark = pytest.mark.gpu


def _pair(gpu_ctx, specs, N, K, seed):
    """two states over the same data and starting assignment: one stepped eagerly, one through sweep_step"""
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z0 = rng.integers(0, max(1, K - 2), N).astype(np.int32)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    out = []
    for _ in range(2):
        st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
        for i, f in enumerate(feats):
            st.set_hp(i, orc.Family(f["family"], f["hp"], f["dim"], "f64").hp)
        st.set_alpha(1.1)
        z = torch.from_numpy(z0).to(gpu_ctx.torch_device)
        st.accumulate(view, z)
        out.append((st, z))
    return view, out[0], out[1], len(feats)


def _same_tables(a, b, nfeat):
    assert np.array_equal(a.get_group_counts(), b.get_group_counts())
    for f in range(nfeat):
        ra, rb = a.get_ss(f), b.get_ss(f)
        for name in ra.dtype.names:
            assert np.array_equal(ra[name], rb[name]), (f, name)


@pytest.mark.parametrize("specs,N,K", [
    ([(orc.NICH, 0)], 5000, 64),                                   # the single-feature kernel
    ([(orc.BB, 0), (orc.GP, 0), (orc.DD, 5), (orc.NICH, 0)], 3000, 40),
    ([(orc.DM, 4), (orc.BNB, 0)], 2000, 30),
    ([(orc.BB, 0), (orc.NICH, 0)], 1200, 400),                     # materialised path: never captured, still equal
    ([(orc.NIW, 3), (orc.BB, 0)], 1500, 12),
])
@pytest.mark.parametrize("graph", [False, True])
def test_steps_equal_the_two_calls(gpu_ctx, specs, N, K, graph, monkeypatch):
    monkeypatch.setenv("MSC_SWEEP_GRAPH", "1" if graph else "0")
    view, (ea, za), (gr, zg), nfeat = _pair(gpu_ctx, specs, N, K, seed=N + K)
    for sweep in range(8):                     # steps 0-1 eager, 2 captured, 3.. replayed
        ea.sweep_assign(view, za, seed=77, sweep=sweep)
        ea.accumulate(view, za)
        gr.sweep_step(view, zg, seed=77, sweep=sweep)
        assert torch.equal(za, zg), sweep
    _same_tables(ea, gr, nfeat)
    fused = not any(f == orc.NIW for f, _ in specs) and K <= 256
    assert gr.sweep_step_stats() == ((2, 6) if fused and graph else (8, 0))       # the graph really is what ran
    # interruptions: a call in between that changes what is current on the device, a jump in the sweep index,
    # a new seed, new hyperparameters, a new alpha -- each must fall back or re-capture, never replay stale work
    gr.score_value(view, z=zg, crp_prior=True)
    ea.sweep_assign(view, za, seed=77, sweep=8); ea.accumulate(view, za)
    gr.sweep_step(view, zg, seed=77, sweep=8)
    assert torch.equal(za, zg)
    for seed, sweep in [(77, 20), (77, 21), (5, 22), (5, 23), (5, 24)]:
        ea.sweep_assign(view, za, seed=seed, sweep=sweep); ea.accumulate(view, za)
        gr.sweep_step(view, zg, seed=seed, sweep=sweep)
        assert torch.equal(za, zg), (seed, sweep)
    for st in (ea, gr):
        st.set_alpha(0.4)
    for sweep in range(25, 30):
        ea.sweep_assign(view, za, seed=5, sweep=sweep); ea.accumulate(view, za)
        gr.sweep_step(view, zg, seed=5, sweep=sweep)
        assert torch.equal(za, zg), sweep
    _same_tables(ea, gr, nfeat)


@pytest.mark.parametrize("graph", [False, True])
def test_step_after_rebinding_another_view(gpu_ctx, graph, monkeypatch):
    """the captured step reads the bound columns through the descriptors: binding other data in between must not leak in"""
    import common_amd
    monkeypatch.setenv("MSC_SWEEP_GRAPH", "1" if graph else "0")
    specs = [(orc.BB, 0), (orc.NICH, 0)]
    view, (ea, za), (gr, zg), nfeat = _pair(gpu_ctx, specs, 2000, 25, seed=3)
    rng = np.random.default_rng(99)
    other = common_amd.DataView.from_recarray(
        gpu_ctx, recarray_of([make_feature(f, 500, 25, rng, d) for f, d in specs]))
    for sweep in range(5):
        ea.sweep_assign(view, za, seed=1, sweep=sweep); ea.accumulate(view, za)
        gr.sweep_step(view, zg, seed=1, sweep=sweep)
    gr.score_value(other)                                   # rebinds the state to `other`
    for sweep in range(5, 9):
        ea.sweep_assign(view, za, seed=1, sweep=sweep); ea.accumulate(view, za)
        gr.sweep_step(view, zg, seed=1, sweep=sweep)
        assert torch.equal(za, zg), sweep
    _same_tables(ea, gr, nfeat)


def test_sharded_driver_uses_the_step_when_alone(gpu_ctx):
    from common_amd.dist import ShardedSweep
    view, (ea, za), (gr, zg), nfeat = _pair(gpu_ctx, [(orc.NICH, 0)], 4000, 32, seed=12)
    drv = ShardedSweep(gr, view, zg, 0)
    for sweep in range(6):
        ea.sweep_assign(view, za, seed=9, sweep=sweep); ea.accumulate(view, za)
        drv.sweep(9, sweep)
        assert torch.equal(za, zg), sweep
    _same_tables(ea, gr, nfeat)


@pytest.mark.parametrize("specs,N,K", [([(orc.NICH, 0)], 4000, 48), ([(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0)], 3000, 100),
                                       ([(orc.NIW, 3)], 2000, 20), ([(orc.BB, 0), (orc.NICH, 0)], 1200, 400)])
def test_sharded_form_of_the_step_equals_the_two_calls(gpu_ctx, specs, N, K):
    """msc_sweep_step_begin + (all-reduce) + msc_state_commit_reduce: the step as a row shard runs it"""
    view, (ea, za), (sh, zs), nfeat = _pair(gpu_ctx, specs, N, K, seed=7 * N + K)
    for sweep in range(6):
        ea.sweep_assign(view, za, seed=3, sweep=sweep)
        ea.accumulate(view, za)
        sh.sweep_step_begin(view, zs, seed=3, sweep=sweep)
        sh.commit_reduce()
        assert torch.equal(za, zs), sweep
    _same_tables(ea, sh, nfeat)
    # a jump in the sweep index and a plain two-call sweep in between are picked up
    ea.sweep_assign(view, za, seed=3, sweep=40); ea.accumulate(view, za)
    sh.sweep_assign(view, zs, seed=3, sweep=40); sh.accumulate(view, zs)
    ea.sweep_assign(view, za, seed=3, sweep=41); ea.accumulate(view, za)
    sh.sweep_step_begin(view, zs, seed=3, sweep=41); sh.commit_reduce()
    assert torch.equal(za, zs)
    _same_tables(ea, sh, nfeat)
