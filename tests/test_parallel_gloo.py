"""CPU (-m "not gpu"): the N > 1 sampling path -- unit sharding + result gather -- with world_size 2 over gloo. The per-volume
"sampler" here is a cheap deterministic stand-in (the real one needs an MI355X); what is tested is that every volume is produced
exactly once, from rank-independent noise, and arrives in global order."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generativemodels_amd import parallel


def _retry_once(fn):
    """The two spawn tests rendezvous over a TCP port picked just before the workers start: on a busy host the port can be taken in between
    (or a worker can miss the store's timeout).  One retry with a fresh port; a real failure fails twice."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except AssertionError:
            raise
        except Exception:
            return fn(*a, **k)
    return wrapper


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(z):
    return torch.tanh(z) * 2.0 + z.mean()


def _worker(rank, world, port, n_units, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids, outs = parallel.sample_units(_fake_sampler, n_units, (1, 1, 4, 4, 4), base_seed=7, rank=rank, world_size=world)
        allv = parallel.gather_units(ids, outs, n_units)
        if rank == 0:
            # by value (numpy): a tensor goes through the queue as a shared-memory handle, and this process may be gone before the parent opens it
            q.put((ids, [v.numpy().copy() for v in allv]))
        else:
            assert allv is None
            q.put((ids, None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


@_retry_once
def test_two_rank_sampling_matches_single_process():
    n_units, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_ids = sorted(i for ids, _ in results for i in ids)
    assert all_ids == list(range(n_units))  # every volume exactly once
    gathered = [torch.from_numpy(a) for a in next(v for _, v in results if v is not None)]
    ids1, ref = parallel.sample_units(_fake_sampler, n_units, (1, 1, 4, 4, 4), base_seed=7, rank=0, world_size=1)
    assert ids1 == list(range(n_units))
    for a, b in zip(gathered, ref):
        assert torch.equal(a, b)  # same volumes as a single-process run, in global order


@_retry_once
@pytest.mark.parametrize("n_units", [13, 5])
def test_eight_rank_sampling_with_uneven_unit_counts_matches_single_process(n_units):
    """world_size 8 (the driver's largest run), unit counts that do not divide: 13 volumes -> shards of 2 and 1; 5 volumes -> three ranks
    sample nothing and still take part in the gather."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = sorted(len(ids) for ids, _ in results)
    assert sizes[-1] - sizes[0] <= 1 and sum(sizes) == n_units
    assert sorted(i for ids, _ in results for i in ids) == list(range(n_units))
    gathered = [torch.from_numpy(a) for a in next(v for _, v in results if v is not None)]
    _, ref = parallel.sample_units(_fake_sampler, n_units, (1, 1, 4, 4, 4), base_seed=7, rank=0, world_size=1)
    assert len(gathered) == n_units
    for a, b in zip(gathered, ref):
        assert torch.equal(a, b)


def _reducer_worker(rank, world, port, q, nrows=8):
    import torch.distributed as dist
    import torch.nn as nn
    from generativemodels_amd.parallel import GradientReducer, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
        unused = nn.Linear(4, 4)  # never applied: like the reference's proj_attn it gets no gradient
        late = nn.Linear(3, 3)    # applied from the third step on only: joins the exchange when its first gradient appears
        side = nn.Linear(3, 3)    # applied on rank 0 only: every rank must still end up with the same averaged gradient
        params = list(model.parameters()) + list(unused.parameters()) + list(late.parameters()) + list(side.parameters())
        red = GradientReducer(params, bucket_mb=0.0005)  # ~500-byte buckets: several exchanges per step
        assert red.active and len(red.buckets) >= 3
        data = torch.randn(nrows, 6, generator=torch.Generator().manual_seed(5))
        target = torch.randn(nrows, 3, generator=torch.Generator().manual_seed(6))
        lo, hi = shard_range(nrows, rank, world)
        overlapped = []

        def loss_of(step, rows):
            y = model(data[rows])
            if step >= 2:
                y = late(y)
            if rank == 0 and step >= 4:
                y = y + 0.5 * side(y.detach())
            return torch.nn.functional.mse_loss(y, target[rows])

        for step in range(6):  # the reducer re-arms itself; step 0 learns the used set, steps 1+ overlap every bucket
            if step % 2 == 0:
                red.zero_grad()  # in-place fill: .grad stays a bucket view
            else:
                for p in params:
                    p.grad = None  # optimizer.zero_grad(set_to_none=True): the next gradients are adopted back into the views
            loss_of(step, slice(lo, hi)).backward()
            red.finish()
            overlapped.append(red.launched_in_backward)
        grads = [None if p.grad is None else p.grad.detach().numpy().copy() for p in params]  # numpy: no shared-memory handles in the queue
        # gradient accumulation: two micro-batches, the first under no_sync, equals one backward over both
        mid = (lo + hi) // 2
        red.zero_grad()
        wa, wb = (mid - lo) / (hi - lo), (hi - mid) / (hi - lo)  # row-weighted halves: their sum is the shard's mean loss
        with red.no_sync():
            (wa * loss_of(5, slice(lo, mid))).backward()
        (wb * loss_of(5, slice(mid, hi))).backward()
        red.finish()
        acc = [None if p.grad is None else p.grad.detach().numpy().copy() for p in params]
        # a second backward outside no_sync is refused
        red.zero_grad()
        loss_of(5, slice(lo, hi)).backward()
        try:
            loss_of(5, slice(lo, hi)).backward()
            refused = False
        except RuntimeError:
            refused = True
        red.finish()
        q.put((rank, grads, acc, overlapped, len(red.buckets), refused))
    finally:
        dist.destroy_process_group()


@_retry_once
def test_two_rank_gradient_all_reduce_matches_the_full_batch():
    """GradientReducer over gloo, world_size 2: gradients live in persistent flat buckets, buckets are exchanged from the grad-ready
    hooks in bucket order; the averaged shard gradients equal the single-process gradients of the full batch; a never-used parameter
    keeps grad = None and -- once the used set is learned on the first step -- stops delaying its bucket (every bucket overlaps from
    step 2); a parameter used on one rank only ends with the same averaged gradient on both; a parameter whose first gradient shows up
    later joins; gradient accumulation under no_sync matches; a second backward outside no_sync raises."""
    import torch.multiprocessing as mp
    import torch.nn as nn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
    unused, late, side = nn.Linear(4, 4), nn.Linear(3, 3), nn.Linear(3, 3)
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))
    target = torch.randn(8, 3, generator=torch.Generator().manual_seed(6))
    # single-process equivalent of the average over the two ranks' shard losses (rank 0 alone carries the `side` term)
    y0, y1 = late(model(data[0:4])), late(model(data[4:8]))
    l0 = torch.nn.functional.mse_loss(y0 + 0.5 * side(y0.detach()), target[0:4])
    l1 = torch.nn.functional.mse_loss(y1, target[4:8])
    (0.5 * (l0 + l1)).backward()
    want = [p.grad for p in list(model.parameters())] + [None, None] + [p.grad for p in list(late.parameters()) + list(side.parameters())]
    for rank in (0, 1):
        grads, acc, overlapped, nbuckets, refused = got[rank]
        assert refused
        assert overlapped[0] < nbuckets              # step 0: the never-used parameters hold their buckets back until finish()
        assert overlapped[1] == nbuckets             # step 1: used set learned -> every bucket is exchanged during backward
        assert overlapped[3] == nbuckets             # step 3: `late` joined at step 2, everything overlaps again
        # steps 4-5: `side` is applied on rank 0 only -- rank 1 cannot complete that bucket during backward (nothing arrives), the
        # exchange still happens in finish() and both ranks hold the same averaged gradient (checked below)
        for got_list in (grads, acc):
            assert got_list[len(list(model.parameters()))] is None and got_list[len(list(model.parameters())) + 1] is None
            for g, w in zip(got_list, want):
                if w is None:
                    assert g is None
                else:
                    assert g is not None and torch.allclose(torch.from_numpy(g), w, atol=1e-6), (torch.from_numpy(g) - w).abs().max()


@_retry_once
def test_eight_rank_gradient_all_reduce_with_uneven_shards_matches_the_average_of_the_shard_losses():
    """GradientReducer over gloo at world_size 8 (the driver's largest run) with 20 rows = shards of 3 and 2: every rank ends with the
    average over ranks of its shard-mean-loss gradient (what DDP computes for uneven shards), the rank-0-only parameter included; the learned
    unused set, the late joiner and accumulation under no_sync behave as at world_size 2."""
    import torch.multiprocessing as mp
    import torch.nn as nn
    world, nrows = 8, 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q, nrows)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
    unused, late, side = nn.Linear(4, 4), nn.Linear(3, 3), nn.Linear(3, 3)
    data = torch.randn(nrows, 6, generator=torch.Generator().manual_seed(5))
    target = torch.randn(nrows, 3, generator=torch.Generator().manual_seed(6))
    total = 0.0
    for r in range(world):
        lo, hi = parallel.shard_range(nrows, r, world)
        y = late(model(data[lo:hi]))
        if r == 0:
            y = y + 0.5 * side(y.detach())
        total = total + torch.nn.functional.mse_loss(y, target[lo:hi]) / world
    total.backward()
    nmodel = len(list(model.parameters()))
    want = [p.grad for p in model.parameters()] + [None, None] + [p.grad for p in list(late.parameters()) + list(side.parameters())]
    for rank in range(world):
        grads, acc, overlapped, nbuckets, refused = got[rank]
        assert refused and overlapped[1] == nbuckets and overlapped[3] == nbuckets
        for got_list in (grads, acc):
            assert got_list[nmodel] is None and got_list[nmodel + 1] is None
            for g, w in zip(got_list, want):
                if w is None:
                    assert g is None
                else:
                    assert g is not None and torch.allclose(torch.from_numpy(g), w, atol=1e-6), (rank, (torch.from_numpy(g) - w).abs().max())


def _static_worker(rank, world, port, q):
    import torch.distributed as dist
    import torch.nn as nn
    from generativemodels_amd.parallel import GradientReducer, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
        unused = nn.Linear(4, 4)  # never applied
        params = list(model.parameters()) + list(unused.parameters())
        red = GradientReducer(params, bucket_mb=0.0005, static_graph=True)
        nrows = 8
        data = torch.randn(nrows, 6, generator=torch.Generator().manual_seed(5))
        target = torch.randn(nrows, 3, generator=torch.Generator().manual_seed(6))
        lo, hi = shard_range(nrows, rank, world)
        hooks, overlapped, grads = [], [], []
        for step in range(5):
            red.zero_grad()
            with torch.no_grad():  # the parameters move between steps: a stale bucket would show
                for p in model.parameters():
                    p.add_(0.01 * (step + 1))
            torch.nn.functional.mse_loss(model(data[lo:hi]), target[lo:hi]).backward()
            red.finish()
            hooks.append(len(red._hook_handles))
            overlapped.append(red.launched_in_backward)
            grads.append([None if p.grad is None else p.grad.detach().numpy().copy() for p in params])
        # accumulation under no_sync still works with one hook per bucket
        mid = (lo + hi) // 2
        red.zero_grad()
        wa, wb = (mid - lo) / (hi - lo), (hi - mid) / (hi - lo)
        with red.no_sync():
            (wa * torch.nn.functional.mse_loss(model(data[lo:mid]), target[lo:mid])).backward()
        (wb * torch.nn.functional.mse_loss(model(data[mid:hi]), target[mid:hi])).backward()
        red.finish()
        acc = [None if p.grad is None else p.grad.detach().numpy().copy() for p in params]
        # a replaced .grad (set_to_none) is adopted back by the bucket hook
        for p in params:
            p.grad = None
        torch.nn.functional.mse_loss(model(data[lo:hi]), target[lo:hi]).backward()
        red.finish()
        none_grads = [None if p.grad is None else p.grad.detach().numpy().copy() for p in params]
        # a replayed step that carried its exchange (graphs.GraphedForwardBackward.exchange_captured marks the reducer): the loop's finish() has
        # nothing left to do -- no second exchange, gradients untouched, the mark cleared
        red.zero_grad()
        torch.nn.functional.mse_loss(model(data[lo:hi]), target[lo:hi]).backward()
        red.finish()  # (what the graph's last nodes did)
        before = [None if p.grad is None else p.grad.detach().clone() for p in params]
        red._exchanged_in_graph = True
        red.finish()
        noop = (not red._exchanged_in_graph and red._work == [None] * len(red.buckets)
                and all((a is None and p.grad is None) or torch.equal(a, p.grad) for a, p in zip(before, params)))
        # the promise broken: the never-used layer suddenly takes part
        red.zero_grad()
        (torch.nn.functional.mse_loss(model(data[lo:hi]), target[lo:hi]) + unused(torch.ones(1, 4)).sum()).backward()
        try:
            red.finish()
            raised = False
        except RuntimeError:
            raised = True
        q.put((rank, hooks, overlapped, grads, acc, none_grads, raised, len(red.buckets), noop))
    finally:
        dist.destroy_process_group()


@_retry_once
@pytest.mark.parametrize("world", [2, 8])
def test_static_graph_reducer_keeps_one_hook_per_bucket(world):
    """GradientReducer(static_graph=True): step 1 learns the used set, step 2 records the arrival order, from step 3 on one hook per bucket
    is left and no usage mask is exchanged -- the gradients stay the full-batch gradients at every step, accumulation under no_sync and a
    `.grad = None` reset still work, and a parameter outside the learned set that produces a gradient makes finish() raise."""
    import numpy as np
    import torch.multiprocessing as mp
    import torch.nn as nn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_static_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))
    target = torch.randn(8, 3, generator=torch.Generator().manual_seed(6))
    nparams = len(list(model.parameters()))
    want = []
    for step in range(5):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * (step + 1))
        model.zero_grad()
        torch.nn.functional.mse_loss(model(data), target).backward()
        want.append([p.grad.numpy().copy() for p in model.parameters()])
    for rank in range(world):
        hooks, overlapped, grads, acc, none_grads, raised, nbuckets, noop = got[rank]
        # (counted after finish(): the first step ends with the per-parameter hooks, the second -- the recorded one -- already with one per non-empty bucket)
        assert hooks[0] == nparams + 2 and hooks[1] == hooks[2] == hooks[3] == hooks[4] < nparams
        assert hooks[1] <= nbuckets and overlapped[1] == overlapped[2] == overlapped[3] == overlapped[4] == nbuckets
        for step in range(5):
            for g, w in zip(grads[step][:nparams], want[step]):
                assert np.allclose(g, w, rtol=1e-5, atol=1e-6), (rank, step)
            assert all(g is None for g in grads[step][nparams:])  # the never-used layer keeps grad = None
        for g, w in zip(acc[:nparams], want[4]):
            assert np.allclose(g, w, rtol=1e-5, atol=1e-6)
        for g, w in zip(none_grads[:nparams], want[4]):
            assert np.allclose(g, w, rtol=1e-5, atol=1e-6)
        assert raised and noop
