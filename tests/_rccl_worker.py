"""Worker of tests/test_gpu_distributed.py, launched by `python -m torch.distributed.run --nproc-per-node 1`: GradientReducer on backend
"nccl" (= RCCL on ROCm) with ONE rank and force=True -- every line of the multi-GPU training path (persistent flat buckets, hook-driven
bucket launches on the side HIP stream, the usage-mask exchange, finish) executes on the GPU; with one rank the averaged gradients must
equal the plain single-process gradients (reference: ddpm_training_ddp.py:125,199,249-270)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def main():
    import restatement as R
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDPMScheduler
    from generativemodels_amd.parallel import GradientReducer

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 64), attention_levels=(False, True),
               num_head_channels=32, norm_num_groups=32)

    def build():
        torch.manual_seed(11)
        m = DiffusionModelUNet(**cfg)
        R.derandomize_zeros(m, seed=5)
        return m.to(dev).train()

    plain, sharded = build(), build()
    red = GradientReducer(sharded.parameters(), bucket_mb=0.25, force=True)  # ~0.25 MB buckets: several exchanges per backward
    assert red.active and red.world == 1 and len(red.buckets) >= 3
    inferer = DiffusionInferer(DDPMScheduler(1000))
    g = torch.Generator().manual_seed(3)
    overlapped = []
    for step in range(3):
        x = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        noise = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        t = torch.randint(0, 1000, (2,), generator=g).to(dev)
        for m in (plain, sharded):
            for p in m.parameters():
                p.grad = None
        if step == 2:
            red.zero_grad()  # the in-place path: .grad stays a view of the flat bucket
        for m in (plain, sharded):
            pred = inferer(inputs=x, diffusion_model=m, noise=noise, timesteps=t)
            assert pred.requires_grad
            F.mse_loss(pred.float(), noise.float()).backward()
        red.finish()
        torch.cuda.synchronize()
        overlapped.append(red.launched_in_backward)
        checked = 0
        for (name, a), b in zip(plain.named_parameters(), sharded.parameters()):
            if a.grad is None:
                assert b.grad is None and "proj_attn" in name, name
                continue
            # bitwise: the backward pass has no atomics (stored per-block partials, fixed-order sums), the sum over ONE rank and the division
            # by 1 are exact, and a gradient accumulated by the kernel into a zeroed bucket view is the kernel's value
            assert b.grad is not None and torch.equal(a.grad, b.grad), f"step {step}: gradient of {name} differs after the RCCL exchange"
            assert b.grad.data_ptr() == red._view[id(b)].data_ptr()  # the gradient lives in its bucket
            checked += 1
        assert checked > 40
    assert red._side is not None, "the exchange did not run on a side HIP stream"
    assert overlapped[0] < len(red.buckets) and overlapped[1] == len(red.buckets) == overlapped[2], (overlapped, len(red.buckets))
    # ---- the same step replayed from ONE HIP graph (generativemodels_amd.GraphedForwardBackward with the reducer): bucket fills + forward +
    #      backward inside the graph, gradients accumulated straight into the bucket views, the RCCL exchange in finish() after the replay ----
    import generativemodels_amd as gm

    def loss_fn(m):
        return lambda x, noise, t: F.mse_loss(inferer(inputs=x, diffusion_model=m, noise=noise, timesteps=t).float(), noise.float())

    x0 = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
    n0 = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
    t0 = torch.randint(0, 1000, (2,), generator=g).to(dev)
    # (a fresh replica: a graphed step is set up before the model's first eager step -- the gradient accumulators of parameters that have
    #  already run a backward on the default stream remember that stream, which a capture on a side stream cannot follow)
    replica = build()
    red2 = GradientReducer(replica.parameters(), bucket_mb=0.25, force=True)
    graphed = gm.GraphedForwardBackward(loss_fn(replica), (x0, n0, t0), replica.parameters(), reducer=red2)
    for step in range(2):
        x = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        noise = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        t = torch.randint(0, 1000, (2,), generator=g).to(dev)
        for p_ in plain.parameters():
            p_.grad = None
        want = loss_fn(plain)(x, noise, t)
        want.backward()
        got = graphed(x, noise, t)
        red2.finish()
        torch.cuda.synchronize()
        assert torch.equal(got, want.detach()), (step, float(got), float(want))
        for (name, a), b in zip(plain.named_parameters(), replica.parameters()):
            if a.grad is None:
                assert b.grad is None, name
                continue
            assert torch.equal(a.grad, b.grad), f"graphed step {step}: gradient of {name} differs"
            assert b.grad.data_ptr() == red2._view[id(b)].data_ptr()
    # ---- (round 5) the replayed step CARRIES its exchange: GradientReducer(static_graph=True) records the arrival order in the warm-up steps, the
    #      capture then contains one RCCL all-reduce per bucket on the reducer's side stream, forked from the hook of the bucket's last gradient and
    #      joined by finish()'s waits inside the graph; the loop's finish() after a replay returns at once ----
    replica3 = build()
    red4 = GradientReducer(replica3.parameters(), bucket_mb=0.25, force=True, static_graph=True)
    graphed3 = gm.GraphedForwardBackward(loss_fn(replica3), (x0, n0, t0), replica3.parameters(), reducer=red4)
    assert graphed3.exchange_captured and red4._static_stage == 2 and len(red4.buckets) >= 3
    for step in range(3):
        x = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        noise = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        t = torch.randint(0, 1000, (2,), generator=g).to(dev)
        for p_ in plain.parameters():
            p_.grad = None
        want = loss_fn(plain)(x, noise, t)
        want.backward()
        got = graphed3(x, noise, t)
        red4.finish()  # nothing left to do: the exchange ran inside the graph
        assert red4._work == [None] * len(red4.buckets) and not red4._exchanged_in_graph
        torch.cuda.synchronize()
        assert torch.equal(got, want.detach()), (step, float(got), float(want))
        for (name, a), b in zip(plain.named_parameters(), replica3.parameters()):
            if a.grad is None:
                assert b.grad is None, name
                continue
            assert torch.equal(a.grad, b.grad), f"step {step} (exchange inside the graph): gradient of {name} differs"
            assert b.grad.data_ptr() == red4._view[id(b)].data_ptr()
    try:
        gm.GraphedForwardBackward(loss_fn(replica), (x0, n0, t0), replica.parameters(), reducer=red2, capture_exchange=True)
        raise AssertionError("capture_exchange=True without static_graph must raise")
    except ValueError:
        pass
    # ---- a weight used TWICE in one backward: with fp32 parameters the weight-gradient kernel of each use adds straight into the
    #      bucket view; the bucket's exchange may only start after BOTH (readiness comes from the engine's post-accumulate hook, which fires once
    #      per parameter after all of its uses).  Step 0 learns the used set (gradients arrive as tensors), steps 1-2 take the in-place path ----
    from generativemodels_amd import autograd as A

    torch.manual_seed(21)
    w = torch.nn.Parameter(torch.randn(16, 16, 3, 3, 3, device=dev) * 0.05)
    bb = torch.nn.Parameter(torch.randn(16, device=dev) * 0.1)
    w0, b0 = w.detach().clone().requires_grad_(), bb.detach().clone().requires_grad_()
    xs = torch.randn((2, 6, 6, 8, 16), device=dev)

    def twice(wt, bt):
        h = A.conv(xs, wt, bt, kernel=3, padding=1)
        return (A.conv(h, wt, bt, kernel=3, padding=1).float() ** 2).mean()

    twice(w0, b0).backward()
    red3 = GradientReducer([w, bb], bucket_mb=0.001, force=True)
    for step in range(3):
        red3.zero_grad()
        twice(w, bb).backward()
        red3.finish()
        torch.cuda.synchronize()
        assert torch.allclose(w.grad, w0.grad, rtol=1e-5, atol=1e-7) and torch.allclose(bb.grad, b0.grad, rtol=1e-5, atol=1e-7), f"shared weight, step {step}"
        if step > 0:
            assert w.grad.data_ptr() == red3._view[id(w)].data_ptr()
    # torch.autograd.grad outside an armed step returns tensors and leaves .grad alone
    before = w.grad.clone()
    gw, = torch.autograd.grad(twice(w, bb), [w])
    assert gw is not None and torch.allclose(gw, w0.grad, rtol=1e-5, atol=1e-7) and torch.equal(w.grad, before)
    red3.close()
    from generativemodels_amd import parallel as P
    assert id(w) not in P._DIRECT_GRAD
    # ---- static_graph=True: step 1 learns the used set, step 2 records the arrival order, from step 3 on one hook per bucket is left and no
    #      usage mask is exchanged; gradients stay bitwise those of the plain replica (the in-place fp32 path from step 2 on) ----
    red.close()
    stat = build()
    red4 = GradientReducer(stat.parameters(), bucket_mb=0.25, force=True, static_graph=True)
    nparam = sum(1 for p_ in stat.parameters() if p_.requires_grad)
    hooks = []
    for step in range(4):
        x = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        noise = torch.randn((2, 1, 8, 8, 8), generator=g).to(dev)
        t = torch.randint(0, 1000, (2,), generator=g).to(dev)
        for p_ in plain.parameters():
            p_.grad = None
        red4.zero_grad()
        for m in (plain, stat):
            F.mse_loss(inferer(inputs=x, diffusion_model=m, noise=noise, timesteps=t).float(), noise.float()).backward()
        red4.finish()
        torch.cuda.synchronize()
        hooks.append(len(red4._hook_handles))
        for (name, a), b in zip(plain.named_parameters(), stat.parameters()):
            if a.grad is None:
                assert b.grad is None, name
                continue
            assert torch.equal(a.grad, b.grad), f"static graph, step {step}: gradient of {name} differs"
        if step >= 1:
            assert red4.launched_in_backward == len(red4.buckets), (step, red4.launched_in_backward)
    assert hooks[0] == nparam and hooks[1] == hooks[2] == hooks[3] <= len(red4.buckets), hooks
    red4.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"RCCL_WORKER_OK buckets={len(red.buckets)} overlapped_per_step={overlapped} graphed_steps=2 static_graph_hooks={hooks}")


if __name__ == "__main__":
    main()
