"""Multi-GPU execution of the sampling path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm).

Sampling chains are independent per volume (GroupNorm and attention are per-sample: reference diffusion_model_unet.py:623,
407-415), so the path shards by *unit = one volume*: every rank samples its own contiguous slice of the unit list and there is
NO collective on the data path. The only communication is the optional gather of finished volumes to one rank. (The reference
ships no multi-GPU sampling code at all; its only distributed example is the DDP training tutorial.)"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_units: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `n_units` for `rank` (the first n_units % world ranks get one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_units, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def unit_seed(base_seed: int, unit: int) -> int:
    """Per-volume RNG seed: a volume's noise depends on its global index only, never on the rank that samples it."""
    return (int(base_seed) * 1_000_003 + int(unit)) % (2**63 - 1)


def unit_noise(shape: Sequence[int], base_seed: int, unit: int, dtype=torch.float32) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(unit_seed(base_seed, unit))).to(dtype)


def sample_units(sample_one: Callable[[torch.Tensor], torch.Tensor], n_units: int, noise_shape: Sequence[int], base_seed: int,
                 rank: int, world_size: int, device=None, dtype=torch.float32) -> Tuple[List[int], List[torch.Tensor]]:
    """Run `sample_one(noise)` (e.g. `lambda z: inferer.sample(z, model, scheduler, verbose=False)`) on this rank's units."""
    import contextlib

    begin, end = shard_range(n_units, rank, world_size)
    ids, outs = [], []
    dev = None if device is None else torch.device(device)
    # the kernels run on the CURRENT device's stream (ops.require_device): make the rank's device current while it samples
    ctx = torch.cuda.device(dev) if (dev is not None and dev.type == "cuda") else contextlib.nullcontext()
    with ctx:
        for u in range(begin, end):
            z = unit_noise(noise_shape, base_seed, u, dtype)
            if dev is not None:
                z = z.to(dev)
            ids.append(u)
            outs.append(sample_one(z))
    return ids, outs


def gather_units(ids: List[int], outs: List[torch.Tensor], n_units: int, group=None, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Collect every rank's finished volumes on rank `dst`, ordered by global unit index (None on the other ranks).
    Works with any backend (tensors are exchanged as objects on the host: this is result collection, not the data path)."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return list(outs)
    world = dist.get_world_size(group)
    payload = [(i, o.detach().cpu()) for i, o in zip(ids, outs)]
    gathered: Optional[List] = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(payload, gathered, dst=dst, group=group)
    if dist.get_rank(group) != dst:
        return None
    table = {}
    for part in gathered:
        for i, o in part:
            if i in table:
                raise RuntimeError(f"unit {i} was produced twice")
            table[i] = o
    if sorted(table) != list(range(n_units)):
        raise RuntimeError("some units are missing after the gather")
    return [table[i] for i in range(n_units)]


# Parameters whose weight-gradient kernels may ACCUMULATE STRAIGHT INTO `.grad` (a GradientReducer's persistent bucket view) instead of
# returning a fresh tensor for autograd's AccumulateGrad node to add in -- one `add_` launch per parameter and step otherwise, which is what
# a launch-bound backward (the 41.7 M-parameter latent UNet on 32^3 latents: 320 parameters) pays for having its gradients live in buckets.
# id(param) -> (weakref to the parameter, data_ptr of its bucket view, weakref to the reducer).  Nothing here keeps a reducer (or its flat
# buckets) alive: entries die with their reducer (`GradientReducer.close`, also run by its finaliser).
_DIRECT_GRAD: dict = {}


_UNDEFINED_GRAD_HOOK_PROBE = None


def engine_fires_hooks_for_undefined_grads() -> bool:
    """One-time runtime probe (CPU, microseconds): does this torch run a leaf's post-accumulate-grad hook when the backward Function returned
    None for it?  torch 2.10 does; an older AccumulateGrad returns before the hook -- there the in-place accumulation path would never signal
    readiness (in-backward overlap would silently disappear and `static_graph` would never leave its recording stage), so `direct_grad_hook`
    stays off and every gradient goes back to autograd as a tensor."""
    global _UNDEFINED_GRAD_HOOK_PROBE
    if _UNDEFINED_GRAD_HOOK_PROBE is None:
        class _ReturnsNone(torch.autograd.Function):
            @staticmethod
            def forward(ctx, w, x):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                return None, g

        fired = []
        with torch.enable_grad():
            w = torch.zeros(1, requires_grad=True)
            x = torch.ones(1, requires_grad=True)
            h = w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
            try:
                _ReturnsNone.apply(w, x).sum().backward()
            finally:
                h.remove()
        _UNDEFINED_GRAD_HOOK_PROBE = bool(fired)
        if not fired:
            import warnings

            warnings.warn("generativemodels_amd: this torch does not run post-accumulate-grad hooks for undefined gradients; weight gradients are "
                          "returned to autograd instead of being accumulated in place (one extra add per parameter)")
    return _UNDEFINED_GRAD_HOOK_PROBE


def direct_grad_hook(param: torch.Tensor):
    """True when a backward kernel may accumulate `param`'s gradient IN PLACE into `param.grad` and hand autograd None: the parameter belongs
    to a live GradientReducer that is armed for a `.backward()` pass (`reducer.zero_grad()` arms it, `finish()` disarms it) and `.grad` is its
    fp32 bucket view.  Otherwise None: the gradient goes back to autograd as a tensor -- in particular for `torch.autograd.grad(loss, params)`
    outside an armed step (gradient penalties, diagnostics), which must not touch `.grad`.  Readiness is NOT signalled from here: the
    engine's post-accumulate hook fires once per parameter after ALL of its uses in the graph (also for an undefined gradient), so a weight
    used twice in one backward is complete before its bucket's exchange starts."""
    ent = _DIRECT_GRAD.get(id(param))
    if ent is None or not engine_fires_hooks_for_undefined_grads():
        return None
    ref, view_ptr, red = ent
    reducer = red()
    g = param.grad
    if reducer is None or not reducer._armed or ref() is not param or g is None or g.dtype != torch.float32 or g.data_ptr() != view_ptr:
        return None
    return True


class GradientReducer:
    """Batch-sharded data-parallel training (BASELINE config C4): every rank holds a replica, runs forward / backward on
    its own shard of the batch, and the gradients are averaged with ONE exchange per step -- a bucketed all-reduce over
    torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; gloo on CPU for the tests).  It replaces what the reference gets
    from torch DistributedDataParallel(find_unused_parameters=True) (tutorials/generative/distributed_training/ddpm_training_ddp.py:199).

    * Buckets of ~`bucket_mb` MB in reverse parameter order (the order gradients become ready in backward).  Each bucket owns ONE
      persistent flat buffer and every parameter's `.grad` is a VIEW into it: autograd accumulates straight into the buffer, nothing is
      packed or unpacked and nothing is allocated per step (`reducer.zero_grad()` = one memset per bucket).  A `.grad` that was replaced
      (optimizer.zero_grad(set_to_none=True)) is copied back into its view on arrival and re-pointed.
    * A bucket is all-reduced as soon as its last EXPECTED gradient has arrived (post-accumulate-grad hooks), in bucket order (the same
      collective sequence on every rank), on a SIDE stream that waits for the producing stream's event: the exchange overlaps the rest of
      backward.  xGMI is point-to-point (7 links x ~153 GB/s): a ring all-reduce of the 167 MB of fp32 gradients of the 41.7 M-parameter
      UNet is ~1.9 ms per-link bound, far below one backward pass -- few, large buckets are the right shape.
    * Unused parameters (the reference's never-applied `proj_attn`, SURVEY.md fact 4 -- the reason the tutorial needs
      find_unused_parameters): the set of parameters that received a gradient on ANY rank is learned on the first step (one small MAX
      all-reduce of a usage mask per step, inside `finish`) and only those are waited for afterwards, so from the second step on every
      bucket launches during backward.  A parameter that produces its first gradient later joins the set through a one-off exchange of
      its own gradient.  Never-used parameters keep `grad = None`, like DDP.
    * `finish()` (before optimizer.step) launches what is left, waits, averages and makes every globally-used parameter's `.grad` the
      averaged view on EVERY rank (also where the local gradient was None), so replicas never diverge.
    * Gradient accumulation: run the extra backward passes under `with reducer.no_sync():`; a second backward outside it raises.
    With world_size 1 (or torch.distributed uninitialised) every call is a no-op unless `force=True` (single-rank exercise of the whole
    path: the `-m gpu` test runs it on backend nccl = RCCL with one rank)."""

    def __init__(self, params, bucket_mb: float = 25.0, group=None, force: bool = False, usage_check_every: int = 1,
                 static_graph: bool = False) -> None:
        """usage_check_every: 1 (default) = DDP's find_unused_parameters semantics exactly, the usage mask is exchanged and read every step
        (one host synchronisation per step); k > 1 = only every k-th step after the first (a late-joining parameter starts training up to
        k - 1 steps late, identically on every rank).
        static_graph: the caller's promise DDP(static_graph=True) asks for -- every step uses the same parameters and their gradients become
        ready in the same order -- and that gradients are cleared with `reducer.zero_grad()`.  The first step learns the used set, the second
        the arrival order; from the third step on ONE hook per bucket is left (on the parameter whose gradient arrives last) and the usage
        mask is no longer exchanged: the host cost of a step drops from ~320 Python hook calls + one synchronisation to one call per bucket
        (the C4 step: VERDICT r3 item 8).  A parameter outside the learned set that produces a gradient later makes `finish()` raise."""
        import torch.distributed as dist

        engine_fires_hooks_for_undefined_grads()  # the one-time probe runs HERE, outside any backward: `direct_grad_hook` (called from inside a custom
                                                  # Function.backward on the engine's device thread, possibly under a HIP-graph capture) only reads it
        self.usage_check_every = int(usage_check_every)
        self.static_graph = bool(static_graph)
        self._static_stage = 0   # 0: learning the used set; 1: recording the arrival order (full hooks); 2: one hook per bucket
        self._exchanged_in_graph = False  # set by a replayed step that carries its exchange (graphs.GraphedForwardBackward.exchange_captured)
        self._arrival: List[List[int]] = []
        self._step = 0

        self._dist = dist
        self.group = group
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.active = ready and (self.world > 1 or force)
        self.params = [p for p in params if p.requires_grad]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.buckets: List[List[torch.nn.Parameter]] = []
        cap = int(bucket_mb * (1 << 20))
        cur, cur_bytes, cur_key = [], 0, None
        for p in reversed(self.params):
            key = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (key != cur_key or cur_bytes + nbytes > cap):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat: List[torch.Tensor] = []
        self._view = {}
        self._side = None
        self._sync = True
        self._armed = False       # True between zero_grad() and finish(): backward kernels may accumulate fp32 weight gradients in place
        self._closed = False
        self._hook_handles: list = []
        self._direct_ids: list = []
        self.launched_in_backward = 0  # diagnostics: buckets whose exchange started from a hook (overlapped) in the last step
        if self.active:
            for b in self.buckets:
                flat = torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device)
                self._flat.append(flat)
                off = 0
                for p in b:
                    self._view[id(p)] = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
            import weakref

            me = weakref.ref(self)

            def make_hook(j, i, view, me=me):
                # one closure per parameter with its slots captured (round 4: the hook body is the reducer's whole host cost inside a host-bound
                # backward -- 320 calls per C4 step; id() / dict lookups and two data_ptr() calls per call were 2/3 of it).  Holds no strong
                # reference to the reducer: a dropped reducer's hooks become no-ops until close() removes them
                def hook(p):
                    r = me()
                    if r is not None and not r._closed:
                        r._on_grad(p, j, i, view)
                return hook

            self._slots = [(j, p, self._view[id(p)]) for j, p in enumerate(self.params)]
            self._bucket_slots = [[(self._index[id(p)], p, self._view[id(p)]) for p in b] for b in self.buckets]
            self._make_hook = make_hook
            for j, p in enumerate(self.params):
                self._hook_handles.append(p.register_post_accumulate_grad_hook(make_hook(j, self._bucket_of[id(p)], self._view[id(p)])))
                if p.dtype == torch.float32:  # fp32 master parameters (mixed precision): the weight-gradient kernels write fp32
                    _DIRECT_GRAD[id(p)] = (weakref.ref(p), self._view[id(p)].data_ptr(), me)
                    self._direct_ids.append(id(p))
            weakref.finalize(self, GradientReducer._drop_entries, list(self._direct_ids), me)
        # parameters waited for before a bucket launches: all of them until the first step has shown which ones ever get a gradient
        self._expected = [True] * len(self.params)
        self._learned = False
        self.reset()

    # ---- per-step state -----------------------------------------------------------------------------------------------------
    def reset(self) -> None:
        """Re-arm for the next backward pass (done by `finish`)."""
        if self._static_stage == 2:  # one hook per bucket: a bucket waits for that one call (none where nothing is expected)
            self._pending = [1 if h else 0 for h in self._bucket_hooked]
        else:
            self._pending = [sum(1 for p in b if self._expected[self._index[id(p)]]) for b in self.buckets]
        self._seen = [False] * len(self.params)
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._next = 0
        self._late = {}
        self._armed = False
        self._in_backward_launches = 0

    @staticmethod
    def _drop_entries(ids, me) -> None:
        for i in ids:
            ent = _DIRECT_GRAD.get(i)
            if ent is not None and ent[2] is me:
                del _DIRECT_GRAD[i]

    def close(self) -> None:
        """Detach from the parameters: removes the grad hooks and the in-place-accumulation entries (a later reducer over the same parameters
        starts clean) and releases the flat buckets.  Gradients that are bucket views stay valid tensors."""
        if self._closed:
            return
        self._closed = True
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        for i in self._direct_ids:
            ent = _DIRECT_GRAD.get(i)
            if ent is not None and ent[2]() is self:
                del _DIRECT_GRAD[i]
        self._direct_ids = []
        self._armed = False

    def zero_grad(self) -> None:
        """Zero every gradient with one fill per bucket and make `.grad` of every expected parameter its bucket view BEFORE backward: autograd
        (or, for fp32 parameters, the weight-gradient kernel itself: `direct_grad_hook`) then accumulates in place and no gradient is ever
        copied into a bucket.  An optimizer's zero_grad(set_to_none=True) also works -- the next gradient is then copied into its view on arrival."""
        self._exchanged_in_graph = False  # a new step begins: a replay whose finish() was skipped must not excuse THIS step's exchange
        if not self.active:
            for p in self.params:
                p.grad = None
            return
        for flat in self._flat:
            flat.zero_()
        for j, p in enumerate(self.params):
            if self._expected[j] and self._learned:  # (before the first step nobody knows which parameters ever get a gradient: a view installed
                p.grad = self._view[id(p)]           #  on a never-used one would make the optimizer step it with zeros -- DDP leaves it None)
        self._armed = True  # a `.backward()` follows: fp32 weight-gradient kernels may add straight into the views (parallel.direct_grad_hook)

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate locally (like DDP.no_sync)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            keep, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = keep
        return ctx()

    # ---- hooks --------------------------------------------------------------------------------------------------------------
    def _side_ctx(self, dev):
        import contextlib

        if dev.type != "cuda":
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        self._side.wait_event(torch.cuda.current_stream(dev).record_event())
        return torch.cuda.stream(self._side)

    def _adopt(self, p, view=None) -> None:
        """Make p.grad the bucket view (copying a gradient tensor autograd allocated itself into it first)."""
        if view is None:
            view = self._view[id(p)]
        g = p.grad
        if g is None or g is view or g.data_ptr() == view.data_ptr():  # (`is`: the common case after zero_grad() -- no data_ptr() call)
            return
        with torch.no_grad():
            view.copy_(g)
        p.grad = view

    def _on_grad(self, p, j=None, i=None, view=None) -> None:
        """The engine's post-accumulate hook: fires once per parameter and backward pass, after every use of the parameter in the graph has
        delivered its gradient -- also when the backward kernels accumulated in place and handed autograd None (torch >= 2.10 runs the
        AccumulateGrad node with an undefined gradient)."""
        self._exchanged_in_graph = False  # an eager backward is running: its exchange is finish()'s to join, whatever a replay left behind
        if j is None:
            j = self._index[id(p)]
            i, view = self._bucket_of[id(p)], self._view[id(p)]
        if not self._expected[j]:
            # first gradient of a parameter that had none so far: it is not part of its bucket's exchange this step (the bucket may be
            # in flight already); finish() exchanges it on its own once every rank knows, and it is expected from then on
            # (under no_sync() several backward passes may deliver it: the contributions add up -- ADVICE r2)
            self._late[j] = p.grad if j not in self._late else self._late[j] + p.grad
            p.grad = None
            self._seen[j] = True
            return
        self._adopt(p, view)
        if not self._sync:
            return
        if self._seen[j] or self._work[i] is not None:
            raise RuntimeError("GradientReducer: a second backward pass reached a parameter before finish(); accumulate gradients under "
                               "`with reducer.no_sync():` and run only the last backward outside it")
        self._seen[j] = True
        if self._static_stage == 1:
            self._arrival[i].append(j)
        self._pending[i] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:  # in bucket order: the same sequence on every rank
            self._launch(self._next)
            self._in_backward_launches += 1
            self._next += 1

    def _on_bucket(self, i: int) -> None:
        """static_graph, third step on: the hook of bucket i's last-arriving parameter.  Every expected gradient of the bucket is in by now
        (same graph, same order as the recorded step); a `.grad` that is not the bucket view any more is copied back in first."""
        self._exchanged_in_graph = False
        if not self._sync:
            return
        if self._work[i] is not None:
            raise RuntimeError("GradientReducer: a second backward pass reached a parameter before finish(); accumulate gradients under "
                               "`with reducer.no_sync():` and run only the last backward outside it")
        expected, seen = self._expected, self._seen
        for j, p, view in self._bucket_slots[i]:
            if expected[j]:
                g = p.grad
                if g is None:
                    view.zero_()
                elif g is not view:
                    self._adopt(p, view)
                seen[j] = True
        self._pending[i] = 0
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._in_backward_launches += 1
            self._next += 1

    def _enter_static(self) -> None:
        """End of the recorded step: drop the per-parameter hooks, keep one per bucket on its last arrival."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        import weakref

        me = weakref.ref(self)

        def make_bucket_hook(i):
            def hook(p):
                r = me()
                if r is not None and not r._closed:
                    r._on_bucket(i)
            return hook

        self._bucket_hooked = []
        for i, order in enumerate(self._arrival):
            if order:
                self._hook_handles.append(self.params[order[-1]].register_post_accumulate_grad_hook(make_bucket_hook(i)))
            self._bucket_hooked.append(bool(order))
        self._static_stage = 2

    def _launch(self, i: int) -> None:
        flat = self._flat[i]
        with self._side_ctx(flat.device):
            self._work[i] = self._dist.all_reduce(flat, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ---- join ---------------------------------------------------------------------------------------------------------------
    def finish(self) -> None:
        """Launch the buckets that did not complete during backward, wait for every exchange, average, and leave the averaged gradient in
        `.grad` of every parameter that is used on any rank."""
        if not self.active:
            return
        if self._exchanged_in_graph:  # the step was a replay of graphs.GraphedForwardBackward with the exchange captured: averaged gradients are in place
            self._exchanged_in_graph = False
            if self._next == 0 and not any(self._seen) and all(w is None for w in self._work):
                return
            # (unreachable through the hooks, which clear the flag; kept as the explicit statement of when the early return is legal)
            raise RuntimeError("GradientReducer.finish(): an eager backward ran after a graph-replayed step whose exchange was captured; its gradients are "
                               "not reduced yet -- call finish() once per step")
        if not self._sync:
            raise RuntimeError("GradientReducer.finish() inside no_sync()")
        dist = self._dist
        # gradients that arrived under no_sync (or replaced views) are adopted now; stale regions of expected parameters that produced
        # nothing this step (and whose .grad the optimizer set to None) are cleared so they contribute zeros
        expected, seen = self._expected, self._seen
        for j, p, view in self._slots:
            if not expected[j]:
                continue
            g = p.grad
            if g is None:
                if not seen[j]:
                    view.zero_()
            else:
                if g is not view:
                    if self._static_stage == 2 and self._work[self._bucket_of[id(p)]] is not None:
                        # the bucket's exchange was launched from its recorded last arrival and this gradient came AFTER it (a changed graph
                        # order, a conditional branch, a checkpointing toggle): copying it in now would write into a flat buffer whose all-reduce
                        # is in flight on the side stream and leave the gradient unreduced
                        raise RuntimeError("GradientReducer(static_graph=True): a gradient arrived after its bucket's exchange had been launched -- "
                                           "the backward order differs from the recorded step; build the reducer without static_graph for this model")
                    self._adopt(p, view)
                seen[j] = True
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        # usage mask: which parameters got a gradient on ANY rank this step (DDP's unused-parameter bitmap)
        # Exchanged (and read on the host: the one synchronisation of a step, behind the bucket exchanges the optimizer waits for anyway) every
        # `usage_check_every`-th step once the expected set is learned; in between every rank assumes the set unchanged -- a parameter that
        # starts producing gradients in such a step keeps `.grad = None` on EVERY rank (replicas stay identical) until the next check admits it.
        dev = self._flat[0].device if self._flat else torch.device("cpu")
        self._step += 1
        check = (not self._learned) or self.usage_check_every <= 1 or self._step % self.usage_check_every == 0
        if self._static_stage == 2:
            check = False  # the used set is the caller's promise; a violation is caught below instead of being admitted
            for j, p, view in self._slots:
                if not self._expected[j] and p.grad is not None:
                    raise RuntimeError("GradientReducer(static_graph=True): a parameter outside the set learned in the first step produced a gradient; "
                                       "build the reducer without static_graph for models whose used parameters change")
        mwork = mask = None
        if check:
            mask = torch.tensor([1 if s else 0 for s in self._seen], dtype=torch.int32).to(dev)
            with self._side_ctx(dev):
                mwork = dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        for i, flat in enumerate(self._flat):
            self._work[i].wait()  # on CUDA: the current stream waits for the collective
        if mwork is not None:
            mwork.wait()
        if self._side is not None:
            torch.cuda.current_stream(dev).wait_stream(self._side)
        # (steps without a mask exchange: every EXPECTED parameter counts as used on every rank -- a local `_seen` would leave `.grad = None` on
        #  the one rank whose shard produced no gradient for it while the others step it with the averaged view: replicas would diverge)
        used = [bool(v) for v in mask.cpu().tolist()] if check else list(self._expected)
        if not self._learned:  # first step: from now on only parameters that ever produced a gradient (on any rank) are waited for
            self._expected = list(used)
        for flat in self._flat:
            flat.div_(self.world)
        # parameters whose FIRST gradient arrived this step (on some rank): one exchange each, in index order on every rank
        for j, p in enumerate(self.params):
            if used[j] and not self._expected[j]:
                view = self._view[id(p)]
                g = self._late.get(j)
                with torch.no_grad():
                    if g is None:
                        view.zero_()
                    else:
                        view.copy_(g)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                view.div_(self.world)
                self._expected[j] = True
        expected = self._expected  # (re-read: the first step replaces the list)
        for j, p, view in self._slots:
            if expected[j] and p.grad is not view and (used[j] or p.grad is not None):
                p.grad = view  # also where this rank had no gradient: replicas apply the same update
        if self.static_graph:
            if self._static_stage == 0 and self._learned:   # (never: stage 0 ends with the first finish())
                pass
            if self._static_stage == 1:
                complete = all(len(order) == sum(1 for p in b if self._expected[self._index[id(p)]]) for order, b in zip(self._arrival, self.buckets))
                if complete:  # (a step whose backward ran under no_sync only, or skipped parameters, records nothing usable: try again next step)
                    self._enter_static()
                else:
                    self._arrival = [[] for _ in self.buckets]
            elif self._static_stage == 0:
                self._static_stage = 1
                self._arrival = [[] for _ in self.buckets]
        self._learned = True
        self.launched_in_backward = self._in_backward_launches
        self.reset()
