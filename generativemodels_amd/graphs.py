"""HIP-graph replay of a whole training step's forward + backward (MI355X runtime support for the training path).

The reference's training loop (tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270) issues a forward and a backward of the
41.7 M-parameter latent UNet per step: ~620 kernel launches on 32^3 latents.  Their kernels add up to ~19 ms on MI355X, but issued one by one
from Python (descriptor set-up, autograd bookkeeping: ~50 us per launch) the step takes 31-34 ms and the GPU idles a third of the time
(tools/prof_host.py).  Every kernel of this package takes fixed device addresses and reads nothing back to the host, so the whole
forward + loss + backward can be captured ONCE into a HIP graph and replayed per step: 18.2 ms, gradients bit-identical to the eager step
(tools/try_graph_train.py, tests/test_gpu_backward.py).

    step = GraphedForwardBackward(lambda x, t, eps: F.mse_loss(inferer(x, ae, unet, eps, t).float(), eps.float()), (x0, t0, eps0), unet.parameters())
    for x, t, eps in batches:
        loss = step(x, t, eps)          # copies the batch into the static inputs, replays, re-attaches the static .grad tensors
        reducer.finish(); optimizer.step()

Set the step up BEFORE the model's first eager training step: the capture runs on a side stream, and the gradient accumulators of parameters that
have already run a backward on the default stream stay bound to that stream.  Constraints (those of any stream capture): fixed shapes and dtypes; `fn` must not read device values on the host (`.item()`, printing the loss)
nor draw host-side random numbers -- draw noise / timesteps outside and pass them in; parameters are updated IN PLACE by the optimizer (every
torch optimizer does), so a replay always reads the current weights: during the capture every derivative of a trainable parameter (packed MFMA
panel, fp32 copy, pre-summed sub-pixel kernels) is re-made inside the graph instead of served from the cache (ops.refresh_trainable_derivatives)."""
from __future__ import annotations

import contextlib
from typing import Callable, Iterable, Optional, Sequence

import torch

__all__ = ["GraphedForwardBackward"]


class GraphedForwardBackward:
    """Capture `loss = fn(*inputs); loss.backward()` for fixed-shape device inputs and replay it on every call.

    fn:      callable of device tensors returning a scalar loss (run under whatever autocast context it enters itself).
    example_inputs: tensors of the shapes / dtypes / device every later call uses (their values are used by the warm-up steps).
    params:  the parameters whose `.grad` the step produces (e.g. `model.parameters()`); the graph owns their gradient tensors.
    reducer: optional `parallel.GradientReducer` (data-parallel training).  The gradients are accumulated straight into the reducer's bucket
             views and the bucket fills run inside the graph.  With `GradientReducer(static_graph=True)` on an RCCL group (round 5) the EXCHANGE
             is captured too: once the warm-up steps have recorded the arrival order, the captured backward runs with the reducer's one hook
             per bucket, each bucket's all-reduce is launched on the reducer's side stream from the hook of its last gradient (a fork of the
             capture: event -> side stream -> RCCL kernel), and `finish()`'s waits, join and division are the graph's last nodes -- the
             replayed step overlaps its exchange with the rest of the backward like the eager step does (`exchange_captured` is True; the
             `reducer.finish()` a training loop calls after the step returns at once).  Otherwise the captured backward runs under
             `reducer.no_sync()` and `reducer.finish()` after each step starts the exchange after the replay (a ring all-reduce of the 167 MB
             of C4 gradients is ~2 ms over xGMI, an eager backward costs 13 ms more than the replayed one).
    capture_exchange: None = whenever the reducer allows it (above), False = never.
    warmup:  eager steps run before the capture (weight packing, kernel attribute set-up, allocator sizing, the reducer's first-step discovery
             of unused parameters).  They accumulate nothing: gradients are cleared before the capture.
    """

    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor], params: Iterable[torch.nn.Parameter],
                 reducer=None, warmup: int = 2, capture_exchange: Optional[bool] = None) -> None:
        self.fn = fn
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GraphedForwardBackward: no parameter requires a gradient")
        self.inputs = [t.clone() for t in example_inputs]
        if any(not t.is_cuda for t in self.inputs):
            raise ValueError("GraphedForwardBackward: the inputs must live on the GPU (the step is replayed from device memory)")
        self.reducer = reducer if (reducer is not None and getattr(reducer, "active", False)) else None
        red = self.reducer
        if capture_exchange and not (red is not None and getattr(red, "static_graph", False)):
            raise ValueError("GraphedForwardBackward(capture_exchange=True) needs an active GradientReducer(static_graph=True)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self._clear()
                fn(*self.inputs).backward()
                if red is not None:
                    red.finish()  # (a real exchange: the first one teaches the reducer which parameters ever receive a gradient)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._clear()
        self.graph = torch.cuda.CUDAGraph()
        # the exchange inside the graph: only with the recorded bucket order (stage 2: finish() then has no host synchronisation -- no usage-mask
        # exchange, no first-gradient admissions) and a backend whose collectives are stream work (RCCL)
        can = (red is not None and getattr(red, "static_graph", False) and getattr(red, "_static_stage", 0) == 2
               and str(red._dist.get_backend(red.group)) == "nccl")
        if capture_exchange and not can:
            raise ValueError("GraphedForwardBackward(capture_exchange=True) needs GradientReducer(static_graph=True) on an RCCL (\"nccl\") group whose "
                             "warm-up steps recorded the gradient arrival order (warmup >= 2)")
        self.exchange_captured = bool(can and capture_exchange is not False)
        sync_off = red.no_sync() if (red is not None and not self.exchange_captured) else contextlib.nullcontext()
        # thread-local capture mode: HIP calls of other host threads (DataLoader pin-memory, the RCCL watchdog) do not invalidate the capture;
        # the autograd engine issues this backward's kernels on the capturing stream
        from . import ops

        with ops.refresh_trainable_derivatives(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if red is not None:
                red.zero_grad()  # one fill per bucket, inside the graph; .grad of every expected parameter = its bucket view
            with sync_off:
                self.loss = fn(*self.inputs)
                self.loss.backward()
            if self.exchange_captured:
                red.finish()  # remaining launches, the waits, the join of the side stream, the division: stream work only in stage 2
        # the gradient tensors the captured step writes: graph-pool allocations (or the reducer's bucket views); re-attached after every replay,
        # so an optimizer.zero_grad(set_to_none=True) between steps is harmless
        self.grads = [p.grad for p in self.params]
        self.loss = self.loss.detach()

    def _clear(self) -> None:
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            for p in self.params:
                p.grad = None

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        """One step on a new batch: -> the (static, device-resident) loss tensor; `.grad` of every parameter holds this step's gradient."""
        if len(inputs) != len(self.inputs):
            raise ValueError(f"GraphedForwardBackward: expected {len(self.inputs)} inputs")
        for dst, src in zip(self.inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("GraphedForwardBackward: the captured step has fixed input shapes and dtypes")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        if self.exchange_captured:
            self.reducer._exchanged_in_graph = True  # the loop's reducer.finish() after this step has nothing left to do
        return self.loss
