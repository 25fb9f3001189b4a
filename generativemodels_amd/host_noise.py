"""The Gaussian noise of an ancestral sampling step, drawn as the reference draws it -- `torch.randn(shape, dtype=model_output.dtype, generator=generator)` on the
CPU generator, then copied to the device (generative/networks/schedulers/ddpm.py:244-248, ddim.py:231-234) -- without paying for torch's bf16 fill.

For bf16, torch's CPU `normal_` (ATen/native/cpu/DistributionTemplates.h: normal_fill / normal_fill_16, contiguous tensors of >= 16 elements) first fills the tensor
with uniforms -- ONE 32-bit Mersenne-Twister draw per element, of which bf16's 8 significand bits are kept: u = (draw & 0xFF) / 256 -- and then transforms every
block of 16 in place, Box-Muller in bf16 arithmetic: for j < 8, (u1, u2) = (1 - x[j], x[j + 8]), r = sqrt(-2 log u1), theta = 2 pi u2, x[j] = r cos theta * std + mean,
x[j + 8] = r sin theta * std + mean.  Every output is therefore a pure function of a PAIR OF BYTES: 65 536 (cos, sin) pairs.  The scalar bf16 transform costs 0.9 ms per
16 x 1 x 64 x 64 draw on the MI355X box's host -- twice the replayed bf16 forward of the BASELINE configs[0] UNet -- while the 65 536 generator draws themselves
(`Tensor.random_()` on a uint8 tensor: the same one-draw-per-element stream, `draw % 256`) cost 0.2 ms.  So: draw the bytes on the host, copy 1 byte per
element, and look the pairs up on the device (gm_normal_bf16_from_bits).  Same generator, same stream position afterwards, the same bf16 values bit for bit.

The table is built by restating the transform with torch's own bf16 CPU operators and is CHECKED against `torch.randn` itself (2^19 values from a private
generator, generator state compared as well) the first time it is used; if anything differs -- another torch build, another fill -- the fast path stays off and
the draw is torch.randn's.  fp32 / fp16 draws and tensors whose size is not a multiple of 16 always are."""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

ENABLED = os.environ.get("GM_HOST_NOISE_TABLE", "1") != "0"
_table_cpu: Optional[torch.Tensor] = None
_table_dev: dict = {}
_verified: Optional[bool] = None


def bf16_normal_table() -> torch.Tensor:
    """[256][256] int32: for the byte pair (b1, b2), the bf16 bits of (r cos theta) in the low half and (r sin theta) in the high half."""
    global _table_cpu
    if _table_cpu is None:
        b = torch.arange(256, dtype=torch.float32) * (1.0 / 256)      # exact in bf16
        x1 = b.to(torch.bfloat16).reshape(256, 1).expand(256, 256)
        x2 = b.to(torch.bfloat16).reshape(1, 256).expand(256, 256)
        u1 = 1 - x1                                                     # every operator below rounds to bf16, as c10::BFloat16's scalar operators do
        radius = torch.sqrt(-2 * torch.log(u1))
        theta = (2.0 * math.pi * x2.float()).to(torch.bfloat16)        # `2.0f * c10::pi<double> * u2`: one rounding
        one, zero = torch.tensor(1.0, dtype=torch.bfloat16), torch.tensor(0.0, dtype=torch.bfloat16)
        # `... * std + mean` with (std, mean) = (1, 0): the addition turns the -0 of u1 = 1 (r = sqrt(-0)) into +0
        lo = (radius * torch.cos(theta) * one + zero).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        hi = (radius * torch.sin(theta) * one + zero).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        _table_cpu = (lo | (hi << 16)).contiguous()
    return _table_cpu


def normal_bf16_from_bits_host(bits: torch.Tensor) -> torch.Tensor:
    """The table lookup on the host (torch indexing): the CHECKER of the device kernel and of the table itself, not a product path."""
    n = bits.numel()
    if n % 16 != 0:
        raise ValueError("whole blocks of 16 values")
    b = bits.reshape(-1, 16).to(torch.int64)
    e = bf16_normal_table().reshape(-1)[b[:, :8] * 256 + b[:, 8:]]
    out = torch.cat([(e & 0xFFFF), (e >> 16) & 0xFFFF], 1).to(torch.int16)
    return out.reshape(-1).view(torch.bfloat16)


def table_matches_torch() -> bool:
    """The table against torch.randn itself: 2^19 bf16 values from a private generator (every byte pair is expected 4 times), and the generator must end in the
    same state as the byte draw's."""
    global _verified
    if _verified is None:
        try:
            g1, g2 = torch.Generator().manual_seed(0x6D616D64), torch.Generator().manual_seed(0x6D616D64)
            n = 1 << 19
            want = torch.randn(n, dtype=torch.bfloat16, generator=g1)
            bits = torch.empty(n, dtype=torch.uint8).random_(generator=g2)
            got = normal_bf16_from_bits_host(bits)
            _verified = bool(torch.equal(want.view(torch.int16), got.view(torch.int16)) and torch.equal(g1.get_state(), g2.get_state()))
        except Exception:  # an unexpected torch build: the plain draw
            _verified = False
    return _verified


# Page-locked staging, so that the host never waits for the device inside a sampling loop: `tensor.to(device)` from pageable memory returns only when the copy
# has run, i.e. after everything queued in front of it -- the forward of the step -- which serialises the host's work of the NEXT step behind the device's of this
# one.  A draw goes into the next slot of a small ring of pinned buffers (per size and dtype), is copied with non_blocking=True, and an event guards the slot's
# reuse RING steps later.
RING = 4
_ring: dict = {}


def _staged_copy(draw, n: int, dtype: torch.dtype, dev: torch.device) -> torch.Tensor:
    """draw(buffer) fills a pinned [n] buffer of `dtype` on the host; returns its device copy (stream-ordered, the host does not wait)."""
    key = (n, dtype, dev.index)
    ent = _ring.get(key)
    if ent is None:
        ent = _ring[key] = dict(slots=[torch.empty(n, dtype=dtype).pin_memory() for _ in range(RING)], events=[None] * RING, i=0)
        if len(_ring) > 16:  # (a few sizes per process: chains of one shape)
            _ring.pop(next(iter(_ring)))
    i = ent["i"]
    ent["i"] = (i + 1) % RING
    if ent["events"][i] is not None:
        ent["events"][i].synchronize()  # the copy that read this slot RING draws ago
    buf = ent["slots"][i]
    draw(buf)
    out = torch.empty(n, dtype=dtype, device=dev)
    out.copy_(buf, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    ent["events"][i] = ev
    return out


def randn(shape, dtype: torch.dtype, generator: Optional[torch.Generator], device, as_bits: bool = False):
    """`torch.randn(shape, dtype=dtype, generator=generator).to(device)`, bit for bit, the generator left in the same state.  as_bits: a bf16 draw may come back as
    ops.NoiseBits (the bytes + the table) for ops.sched_step, which then does the lookup in the step's kernel: one launch less per step."""
    shape = tuple(int(s) for s in shape)
    n = math.prod(shape)
    dev = torch.device(device)
    if dev.type != "cuda" or n == 0:
        return torch.randn(shape, dtype=dtype, generator=generator).to(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if torch.cuda.is_current_stream_capturing():
        # the draw is HOST work (the reference's CPU generator): a capture would bake one draw's staging buffer into the graph and replay stale bytes
        raise RuntimeError("a noise draw from the CPU generator cannot be captured into a HIP graph: capture the model forward, run the scheduler step outside")
    if ENABLED and dtype == torch.bfloat16 and n >= 16 and n % 16 == 0 and table_matches_torch():
        from . import ops

        bits = _staged_copy(lambda buf: buf.random_(generator=generator), n, torch.uint8, dev)
        tab = _table_dev.get(dev.index)
        if tab is None:
            tab = _table_dev[dev.index] = bf16_normal_table().to(dev)
        if as_bits:  # (a scheduler step: ops.sched_step expands the bytes inside its own kernel)
            return ops.NoiseBits(bits, tab, shape)
        return ops.normal_bf16_from_bits(bits, tab).reshape(shape)
    if not ENABLED:
        return torch.randn(shape, dtype=dtype, generator=generator).to(dev)
    return _staged_copy(lambda buf: torch.randn((n,), dtype=dtype, generator=generator, out=buf), n, dtype, dev).reshape(shape)
