"""CPU (-m "not gpu"): the N > 1 sampling path -- unit sharding + result gather -- with world_size 2 over gloo. The per-volume
"sampler" here is a cheap deterministic stand-in (the real one needs an MI355X); what is tested is that every volume is produced
exactly once, from rank-independent noise, and arrives in global order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generativemodels_amd import parallel


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
            q.put((ids, [v.clone() for v in allv]))
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
    gathered = next(v for _, v in results if v is not None)
    ids1, ref = parallel.sample_units(_fake_sampler, n_units, (1, 1, 4, 4, 4), base_seed=7, rank=0, world_size=1)
    assert ids1 == list(range(n_units))
    for a, b in zip(gathered, ref):
        assert torch.equal(a, b)  # same volumes as a single-process run, in global order


def _reducer_worker(rank, world, port, q):
    import torch.distributed as dist
    import torch.nn as nn
    from generativemodels_amd.parallel import GradientReducer, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
        unused = nn.Linear(4, 4)  # never applied: like the reference's proj_attn it gets no gradient
        params = list(model.parameters()) + list(unused.parameters())
        red = GradientReducer(params, bucket_mb=0.0005)  # ~500-byte buckets: several exchanges per step
        assert red.active and len(red.buckets) >= 3
        data = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))
        target = torch.randn(8, 3, generator=torch.Generator().manual_seed(6))
        lo, hi = shard_range(8, rank, world)
        for _ in range(2):  # two steps: the reducer re-arms itself
            for p in params:
                p.grad = None
            loss = torch.nn.functional.mse_loss(model(data[lo:hi]), target[lo:hi])
            loss.backward()
            red.finish()
        q.put((rank, [None if p.grad is None else p.grad.clone() for p in params]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_matches_the_full_batch():
    """GradientReducer (bucketed, hook-driven all-reduce + averaging) over gloo, world_size 2: the averaged shard gradients equal the
    single-process gradients of the full batch; a parameter without a gradient stays None."""
    import torch.multiprocessing as mp
    import torch.nn as nn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    model = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))
    target = torch.randn(8, 3, generator=torch.Generator().manual_seed(6))
    torch.nn.functional.mse_loss(model(data), target).backward()
    want = [p.grad for p in model.parameters()]
    for rank in (0, 1):
        grads = got[rank]
        assert grads[-1] is None and grads[-2] is None
        for g, w in zip(grads[:len(want)], want):
            assert torch.allclose(g, w, atol=1e-6), (g - w).abs().max()
