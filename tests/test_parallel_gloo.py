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
