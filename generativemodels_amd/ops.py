"""Tensor-level wrappers over the C-ABI kernels (include/gm_amd.h). torch supplies device memory and the current HIP stream;
all arithmetic happens in libgmamd.so. Activations are "arena" tensors in channels-last layout: shape (N, *spatial, C) with
unit stride on C and an arbitrary leading dimension (so a channel slice of a wider buffer is a valid operand).

No fallback: a CPU tensor, a missing library or an unsupported geometry raises."""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import threading
import weakref
from typing import Optional, Sequence

import torch

from . import _native as nat
from ._native import GmAttnBwdDesc, GmAttnDesc, GmConvDesc, GmGnTables, GmKlParams, GmStepParams, GmWgradDesc, check, lib

_DT = {torch.float32: 0, torch.bfloat16: 1}
ACT = {"none": 0, "silu": 1, "relu": 2}
ACT_BWD = {"none": 0, "silu": 1, "relu": 2, "leakyrelu": 3}  # activation codes of the backward kernels (act' evaluated from the pre-activation)
POST_ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "silu": 4, "swish": 4, "leakyrelu": 5, "gelu": 6}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"generativemodels_amd kernels support float32 and bfloat16, got {dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- mixed precision: fp32 master parameters, bf16 compute -------------------------------------------------------------------------------
# The reference's training loops keep fp32 parameters and run the forward under torch.autocast + GradScaler
# (tutorials/generative/distributed_training/ddpm_training_ddp.py:129,253-270; generative/engines/trainer.py:155-156,258).  torch.autocast wraps
# torch ops and cannot see a ctypes kernel, so the switch is explicit: inside `with generativemodels_amd.autocast(torch.bfloat16):` -- or inside
# a `torch.autocast("cuda", dtype=torch.bfloat16)` region, which is honoured too -- a network casts its input to the compute dtype at its
# entry, every activation is stored in it, the MFMA panels are packed from the fp32 parameters once per parameter version
# (gm_pack_conv_weight converts while packing: packed_conv_weight), GroupNorm / bias / timestep vectors are read in fp32 as always, and the
# weight / bias / affine gradients come back in fp32 -- the dtype of the parameters -- so the optimizer steps the fp32 master copy.
_AUTOCAST = threading.local()


class autocast:
    """Context manager: run the networks of this package with `dtype` activations / MFMA operands over parameters kept in their own
    (fp32) dtype.  Re-entrant and thread-local; `enabled=False` switches an enclosing region (this one or torch.autocast's) off."""

    def __init__(self, dtype: torch.dtype = torch.bfloat16, enabled: bool = True) -> None:
        if enabled:
            dt_code(dtype)  # fp32 / bf16 only
        self.dtype, self.enabled = dtype, enabled

    def __enter__(self):
        stack = getattr(_AUTOCAST, "stack", None)
        if stack is None:
            stack = _AUTOCAST.stack = []
        stack.append(self.dtype if self.enabled else False)
        return self

    def __exit__(self, *exc):
        _AUTOCAST.stack.pop()
        return False


def autocast_dtype() -> Optional[torch.dtype]:
    """The active mixed-precision compute dtype, or None: the innermost generativemodels_amd.autocast region, else torch.autocast("cuda")."""
    stack = getattr(_AUTOCAST, "stack", None)
    if stack:
        return stack[-1] or None
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        dt_code(dt)  # torch.autocast defaults to float16 on "cuda": not a dtype of these kernels -- say so instead of computing in another one
        return dt
    return None


def compute_dtype(param_dtype: torch.dtype) -> torch.dtype:
    """dtype a network computes in: the autocast dtype when a region is active, else the dtype of its parameters."""
    return autocast_dtype() or param_dtype


def entry_cast(x: torch.Tensor, param_dtype: torch.dtype, what: str = "input") -> torch.Tensor:
    """A network's input in its compute dtype.  Outside an autocast region the input must already have the parameters' dtype (as before);
    inside one it is cast (not differentiable: the differentiable form is autograd.cast)."""
    ac = autocast_dtype()
    if ac is None:
        if x.dtype != param_dtype:
            raise TypeError(f"{what} dtype {x.dtype} does not match the model dtype {param_dtype}")
        return x
    return x if x.dtype == ac else cast(x, ac)


def require_device(*ts: Optional[torch.Tensor]) -> None:
    """Every operand is a HIP tensor on the CURRENT device.  The kernels are enqueued on torch's current stream of the current device
    (`_stream`); an operand living on another GPU would be read through a stream that is unordered against that GPU's own work -- silent
    garbage -- so it is refused here instead (`with torch.cuda.device(t.device):` / `torch.cuda.set_device` select the device to run on)."""
    cur = -1
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "generativemodels_amd runs on MI355X (HIP) tensors only: got a tensor on "
                f"'{t.device}'. There is no CPU / eager fallback (move the module and its inputs to 'cuda').")
        if cur < 0:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(
                f"generativemodels_amd: operand on '{t.device}' but the current HIP device is cuda:{cur}; kernels run on the current device's "
                "stream -- wrap the call in `with torch.cuda.device(tensor.device):` (or torch.cuda.set_device) so data and stream agree")


# ---- optional per-launch timing (bench.py / profiling only): HIP events on the launch stream around selected kernels --------
_PROFILE: Optional[list] = None


def start_profile() -> None:
    global _PROFILE
    _PROFILE = []


def stop_profile() -> list:
    """-> [(kernel label, meta dict, elapsed ms)] for every launch since start_profile()."""
    global _PROFILE
    rec, _PROFILE = _PROFILE or [], None
    torch.cuda.synchronize()
    return [(name, meta, e0.elapsed_time(e1)) for name, meta, e0, e1 in rec]


def _timed(name: str, meta: dict, fn):
    if _PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _PROFILE.append((name, meta, e0, e1))
    return r


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def arena_ld(t: torch.Tensor) -> int:
    """Leading dimension (elements between consecutive voxels) of a channels-last arena tensor; validates the layout."""
    if t.dim() < 2:
        raise ValueError("arena tensors are (N, *spatial, C)")
    c = t.shape[-1]
    if c > 1 and t.stride(-1) != 1:
        raise ValueError("arena tensor must have unit stride on the channel dim")
    ld = expect = None
    for sz, st in zip(reversed(t.shape[:-1]), reversed(t.stride()[:-1])):
        if sz == 1:
            continue
        if expect is None:
            ld = st
        elif st != expect:
            raise ValueError(f"arena tensor is not row-dense: shape {tuple(t.shape)}, strides {t.stride()}")
        expect = st * sz
    if ld is None:
        ld = c
    if ld < c:
        raise ValueError("arena leading dim smaller than the channel count")
    return int(ld)


def rows_of(t: torch.Tensor) -> int:
    return int(math.prod(t.shape[:-1]))


def new_arena(n: int, spatial: Sequence[int], c: int, dtype, device) -> torch.Tensor:
    return torch.empty((n, *spatial, c), dtype=dtype, device=device)


# ------------------------------------------------------------------------------------------------------------------------
# parameter-derived caches (packed conv weights, fp32 copies of bias / norm affine)
# ------------------------------------------------------------------------------------------------------------------------
_param_cache: dict = {}


# While a training step is being captured into a HIP graph (graphs.GraphedForwardBackward) the derivatives of TRAINABLE parameters are re-made
# inside the capture instead of served from the cache: the replays must read the weights the optimizer has updated since, and a panel packed
# before the capture would be baked into the graph as a constant.  (Inference graphs keep the cache: their weights do not change.)
_REFRESH_TRAINABLE = [False]


@contextlib.contextmanager
def refresh_trainable_derivatives():
    keep, _REFRESH_TRAINABLE[0] = _REFRESH_TRAINABLE[0], True
    try:
        yield
    finally:
        _REFRESH_TRAINABLE[0] = keep


def _cached(param: torch.Tensor, tag, make):
    if _REFRESH_TRAINABLE[0] and param.requires_grad:
        return make()
    key = (id(param), tag)
    ent = _param_cache.get(key)
    ver = (param._version, param.data_ptr(), param.dtype, param.device)
    if ent is not None and ent[0]() is param and ent[1] == ver:
        return ent[2]
    val = make()
    _param_cache[key] = (weakref.ref(param), ver, val)
    if len(_param_cache) > 65536:  # drop dead entries
        for k in [k for k, e in _param_cache.items() if e[0]() is None]:
            _param_cache.pop(k, None)
    return val


def invalidate_param_cache(param: torch.Tensor) -> None:
    """Drop every cached derivative (fp32 copy, packed panel) of a parameter a native kernel has just rewritten in place (such writes do
    not bump `param._version`, the key the caches are validated against)."""
    for key in [k for k in _param_cache if k[0] == id(param)]:
        _param_cache.pop(key, None)


def as_f32(param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """fp32 device copy of a small parameter vector (bias / gamma / beta); identity when it already is fp32."""
    if param is None:
        return None
    require_device(param)
    if param.dtype == torch.float32 and param.is_contiguous():
        return param.detach()

    def make():
        out = torch.empty(param.shape, dtype=torch.float32, device=param.device)
        src = param.detach().contiguous()
        check(lib().gm_copy_channels(src.data_ptr(), src.numel(), dt_code(src.dtype), out.data_ptr(), out.numel(), 0, 1,
                                     src.numel(), _stream()), "gm_copy_channels")
        return out

    return _cached(param, "f32", make)


def packed_conv_weight(weight: torch.Tensor, dtype: torch.dtype, transposed: bool = False, cin_range: Optional[tuple] = None) -> torch.Tensor:
    """weight: nn.Conv{1,2,3}d / nn.Linear layout [Cout, Cin, *k] (or nn.ConvTranspose [Cin, Cout, *k]) -> MFMA panel layout.
    cin_range = (lo, hi) packs only that slice of the input channels (a convolution over one part of a VirtualCat)."""
    require_device(weight)

    def make():
        w = weight.detach()
        if cin_range is not None:
            w = w[:, cin_range[0]:cin_range[1]]
        w = w.contiguous()
        k = list(w.shape[2:])
        while len(k) < 3:
            k.insert(0, 1)
        cout, cin = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
        n = lib().gm_packed_conv_weight_elems(cout, cin, k[0], k[1], k[2], dt_code(dtype))
        out = torch.empty(n, dtype=dtype, device=w.device)
        check(lib().gm_pack_conv_weight(w.data_ptr(), dt_code(w.dtype), out.data_ptr(), dt_code(dtype), cout, cin, k[0], k[1],
                                        k[2], int(transposed), _stream()), "gm_pack_conv_weight")
        return out

    return _cached(weight, ("pack", dtype, transposed, cin_range), make)


def packed_cin_weight(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The K-MAJOR image of a C_in <= 4 convolution's 3x3x3 weight for tile configuration 12 (csrc/conv_edge.hip conv_cin_kernel):
    [Cout padded to 64][27 * C_in padded to the MFMA K step] with k = tap * C_in + ci, zero padded -- the taps x inputs ARE the GEMM K, and a
    lane's weight fragment is one 16-byte load of this image (DiffusionModelUNet.conv_in, AutoencoderKL encoder conv_in)."""
    require_device(weight)

    def make():
        w = weight.detach()
        cout, cin = w.shape[0], w.shape[1]
        kb = 64 // (4 if dtype == torch.float32 else 2)            # K values per MFMA step: 16 fp32 / 32 bf16
        k = 27 * cin
        kp, cp = -(-k // kb) * kb, -(-cout // 64) * 64
        out = torch.zeros((cp, kp), dtype=dtype, device=w.device)
        out[:cout, :k] = w.reshape(cout, cin, 27).permute(0, 2, 1).reshape(cout, k).to(dtype)
        return out

    return _cached(weight, ("cin_rows", dtype), make)


SUBPIXEL_UPSAMPLE = True  # nearest-2x + 3x3x3 convolutions run as 8 sub-pixel 2x2x2 convolutions (8/27 of the multiply-adds)


def _pack_subpixel(weight: torch.Tensor, dtype: torch.dtype, cout: int, cin: int, k: int, swap_io: bool, masks) -> torch.Tensor:
    """The 8 parity images of 2x2x2 kernels in ONE launch straight from the parameter (gm_pack_subpixel_weight): sub-tap s of parity p along an
    axis = the sum of the source taps in bit mask masks[p][s]."""
    w = weight.detach().contiguous()
    n = lib().gm_packed_conv_weight_elems(cout, cin, 2, 2, 2, dt_code(dtype))
    out = torch.empty(8 * n, dtype=dtype, device=w.device)
    check(lib().gm_pack_subpixel_weight(w.data_ptr(), dt_code(w.dtype), out.data_ptr(), dt_code(dtype), cout, cin, k, int(swap_io), masks[0][0], masks[0][1],
                                        masks[1][0], masks[1][1], _stream()), "gm_pack_subpixel_weight")
    return out


def packed_subpixel_weight(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[Cout, Cin, 3, 3, 3] -> the 8 parity images of 2x2x2 kernels (packed back to back) of `Upsample(nearest 2x) -> conv3x3x3`.  Output
    voxel 2i + p sees the up-sampled taps (2i + p - 1, 2i + p, 2i + p + 1) = input voxels (i - 1, i, i) for p = 0 and (i, i, i + 1)
    for p = 1: per axis the three weights collapse to (w0, w1 + w2) resp. (w0 + w1, w2).  Summed in fp32, then rounded to `dtype`."""
    require_device(weight)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError("sub-pixel up-sampling weights: a [Cout, Cin, 3, 3, 3] kernel")
    return _cached(weight, ("subpixel", dtype),
                   lambda: _pack_subpixel(weight, dtype, weight.shape[0], weight.shape[1], 3, False, ((0b001, 0b110), (0b011, 0b100))))


def stride2_subpixel_taps(K: int, pad: int) -> dict:
    """Per axis: {output parity: (tap read at input i - 1 + parity, tap read at input i + parity)} of a stride-2 transposed convolution
    out[u] = sum_o x[o] W[u + pad - 2 o] whose output is twice its input; None = no such tap (a zero weight).  Pure host logic (tested on CPU)."""
    def tap(kk):
        return kk if 0 <= kk < K else None
    return {0: (tap(pad + 2), tap(pad)), 1: (tap(pad + 1), tap(pad - 1))}


def stride2_subpixel_covers(K: int, pad: int) -> bool:
    """True when EVERY tap of a stride-2 transposed convolution with kernel K and low padding `pad` is one of the two taps per parity the
    sub-pixel kernel reads (stride2_subpixel_taps): taps pad - 1 .. pad + 2.  K = 4 with pad = 0 is not: for odd outputs tap W[3] belongs to
    input i - 1, which parity 1 does not read -- the tap would be dropped silently."""
    return K in (3, 4) and 0 <= pad <= 1 and K <= pad + 3


def packed_stride2_dgrad_weight(weight: torch.Tensor, dtype: torch.dtype, pad_lo: int) -> torch.Tensor:
    """The data gradient of a stride-2 3x3x3 convolution as the 8 parity images of 2x2x2 kernels the sub-pixel kernel (configuration 17)
    consumes.  Forward: y[o] = sum_k W[k] x[2 o + k - pad_lo].  Gradient: dx[u] = sum_{o, k: 2 o + k - pad_lo = u} W[k]^T dy[o], i.e. per axis
      pad_lo = 1 (symmetric padding, DiffusionModelUNet Downsample):  dx[2i] = W[1]^T dy[i];  dx[2i + 1] = W[2]^T dy[i] + W[0]^T dy[i + 1]
      pad_lo = 0 (pad-high-only, AutoencoderKL Downsample):          dx[2i] = W[2]^T dy[i - 1] + W[0]^T dy[i];  dx[2i + 1] = W[1]^T dy[i]
    -- exactly the access pattern of that kernel (output parity 0 reads inputs (i - 1, i), parity 1 reads (i, i + 1)) with at most two
    taps per axis, so a stride-2 data gradient is one launch of it on dy with these weights: no zero-insertion, 8 of 8 MFMA taps issued of
    which 27/8 per output voxel on average carry weight (reference: torch autograd through nn.Conv3d(stride=2),
    diffusion_model_unet.py:510-518, autoencoderkl.py:107-121).  [Cout, Cin, 3, 3, 3] -> packed, rounded to `dtype`."""
    K = weight.shape[2]
    if tuple(weight.shape[2:]) != (K, K, K) or not stride2_subpixel_covers(K, pad_lo):
        raise ValueError("stride-2 sub-pixel weights: cubic kernel 3 (padding 0 or 1) or 4 (padding 1): every tap within pad - 1 .. pad + 2")
    require_device(weight)
    # As a transposed convolution with weight [C_in_t, C_out_t, K, K, K] (a forward weight [C_out, C_in, ...] IS its gradient's transposed weight):
    # out[u] = sum x[o] W[k], u = 2 o - pad_lo + k.  Output parity pi reads the inputs i + delta, delta = (-1, 0) for pi = 0 and (0, +1) for
    # pi = 1, through tap k = pi + pad_lo - 2 delta (when it exists): K = 4, pad 1 (the VQ-VAE up-sampling) uses all eight taps of every parity.
    taps = stride2_subpixel_taps(K, pad_lo)
    masks = tuple(tuple(0 if t is None else 1 << t for t in taps[p]) for p in (0, 1))

    def make():
        # the gradient maps Cout channels of dy to Cin channels of dx: packed with (out, in) = (weight.shape[1], weight.shape[0]), the source
        # read as [in][out][K][K][K] (swap_io) -- one launch for the 8 images (round 2: torch slicing + a pack launch per parity)
        return _pack_subpixel(weight, dtype, weight.shape[1], weight.shape[0], K, True, masks)

    return _cached(weight, ("stride2_dgrad", dtype, pad_lo), make)


def packed_cat_weight(weights: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """Linear weights [Cout_i, Cin] stacked along Cout and packed as one panel (fused q/k/v projections etc.)."""

    def make():
        w = torch.cat([x.detach() for x in weights], dim=0).contiguous()
        cout, cin = w.shape[0], w.shape[1]
        k = [1] * (5 - w.dim()) + list(w.shape[2:])  # nn.Linear [Cout, Cin] or nn.ConvNd [Cout, Cin, *kernel] weights
        n = lib().gm_packed_conv_weight_elems(cout, cin, k[0], k[1], k[2], dt_code(dtype))
        out = torch.empty(n, dtype=dtype, device=w.device)
        check(lib().gm_pack_conv_weight(w.data_ptr(), dt_code(w.dtype), out.data_ptr(), dt_code(dtype), cout, cin, k[0], k[1], k[2], 0,
                                        _stream()), "gm_pack_conv_weight")
        return out

    # validity is tied to the first weight's version; the others are folded into the tag
    tag = ("packcat", dtype, tuple((id(x), x._version, x.data_ptr()) for x in weights[1:]))
    return _cached(weights[0], tag, make)


def cat_f32(params: Sequence[Optional[torch.Tensor]], sizes: Sequence[int], device) -> torch.Tensor:
    """fp32 concatenation of bias vectors (None -> zeros)."""
    key_param = next((p for p in params if p is not None), None)
    build = lambda: torch.cat([torch.zeros(s, dtype=torch.float32, device=device) if p is None else p.detach().float()
                               for p, s in zip(params, sizes)]).contiguous()
    if key_param is None:
        return build()
    tag = ("catf32", tuple((id(p), None if p is None else (p._version, p.data_ptr())) for p in params))
    return _cached(key_param, tag, build)


# ------------------------------------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------------------------------------
def to_channels_last(x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Logical NC[D]HW tensor -> arena (N, *spatial, C) in `dtype` (default: x.dtype)."""
    require_device(x)
    dtype = dtype or x.dtype
    x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    v = math.prod(sp)
    if c == 1 and dtype == x.dtype:
        return x.reshape(n, *sp, 1)
    out = torch.empty((n, *sp, c), dtype=dtype, device=x.device)
    check(lib().gm_nchw_to_nhwc(x.data_ptr(), dt_code(x.dtype), out.data_ptr(), dt_code(dtype), n, c, v, c, _stream()),
          "gm_nchw_to_nhwc")
    return out


def to_channels_first(a: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Arena (N, *spatial, C) -> contiguous NC[D]HW tensor."""
    require_device(a)
    dtype = dtype or a.dtype
    n, c = a.shape[0], a.shape[-1]
    sp = tuple(a.shape[1:-1])
    v = math.prod(sp)
    ld = arena_ld(a)
    if c == 1 and ld == 1 and dtype == a.dtype:
        return a.reshape(n, 1, *sp)
    out = torch.empty((n, c, *sp), dtype=dtype, device=a.device)
    check(lib().gm_nhwc_to_nchw(a.data_ptr(), ld, dt_code(a.dtype), out.data_ptr(), dt_code(dtype), n, c, v, _stream()),
          "gm_nhwc_to_nchw")
    return out


def copy_channels(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[..., :C] = src (dst may be a channel slice of a wider arena buffer; dtypes may differ)."""
    require_device(src, dst)
    if src.shape != dst.shape:
        raise ValueError(f"copy_channels shape mismatch {tuple(src.shape)} vs {tuple(dst.shape)}")
    _timed("copy_channels", dict(flops=0.0, bytes=float(src.numel() * (src.element_size() + dst.element_size())), shape=str(tuple(src.shape))),
           lambda: check(lib().gm_copy_channels(src.data_ptr(), arena_ld(src), dt_code(src.dtype), dst.data_ptr(), arena_ld(dst),
                                                dt_code(dst.dtype), rows_of(src), src.shape[-1], _stream()), "gm_copy_channels"))


def concat_channels(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(dim=channel) in the arena (reference: diffusion_model_unet.py:1232,1340,1461; inferer.py:72,127)."""
    c = sum(p.shape[-1] for p in parts)
    out = torch.empty((*parts[0].shape[:-1], c), dtype=parts[0].dtype, device=parts[0].device)
    off = 0
    for p in parts:
        copy_channels(p, out[..., off:off + p.shape[-1]])
        off += p.shape[-1]
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    require_device(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    n = x.numel()
    check(lib().gm_copy_channels(x.data_ptr(), n, dt_code(x.dtype), out.data_ptr(), n, dt_code(dtype), 1, n, _stream()),
          "gm_copy_channels")
    return out


def normal_bf16_from_bits(bits: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """torch.randn(n, dtype=bfloat16) of the CPU generator from that generator's byte draws (host_noise.py; reference ddpm.py:244-248): bits uint8 [n] and the
    [256][256] int32 pair table, both on the device; n % 16 == 0."""
    require_device(bits, table)
    if bits.dtype != torch.uint8 or table.dtype != torch.int32 or table.numel() != 65536 or not bits.is_contiguous() or not table.is_contiguous():
        raise ValueError("bits: contiguous uint8, table: contiguous int32 [256][256]")
    n = bits.numel()
    if n % 16 != 0:
        raise ValueError("whole blocks of 16 values")
    out = torch.empty(n, dtype=torch.bfloat16, device=bits.device)
    check(lib().gm_normal_bf16_from_bits(bits.data_ptr(), table.data_ptr(), out.data_ptr(), n, _stream()), "gm_normal_bf16_from_bits")
    return out


def resample2x(x: torch.Tensor, mode: str) -> torch.Tensor:
    """mode 'up': nearest x2; 'down': 2x average pool, on every spatial axis of an arena tensor."""
    require_device(x)
    sp = list(x.shape[1:-1])
    nd = len(sp)
    d, h, w = ([1] * (3 - nd) + sp)
    act_d = 1 if nd == 3 else 0
    if nd == 1:
        raise ValueError("resample2x needs 2-D or 3-D data")
    if mode == "up":
        osp = [s * 2 for s in sp]
    else:
        osp = [s // 2 for s in sp]
    out = torch.empty((x.shape[0], *osp, x.shape[-1]), dtype=x.dtype, device=x.device)
    check(lib().gm_resample2x(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), x.shape[0], x.shape[-1], d, h, w, act_d,
                              0 if mode == "up" else 1, dt_code(x.dtype), _stream()), "gm_resample2x")
    return out


def phase2x(x: torch.Tensor, phase: Sequence[int]) -> torch.Tensor:
    """x[:, r0::2, r1::2, ...] of an arena tensor (N, *spatial, C) as a dense arena tensor: the phase image `phase` = one parity per spatial axis
    of the 2x sub-lattice (2-D or 3-D)."""
    require_device(x)
    sp = list(x.shape[1:-1])
    nd = len(sp)
    if nd not in (2, 3) or len(phase) != nd or any(r not in (0, 1) for r in phase):
        raise ValueError("phase2x needs 2-D or 3-D data and one parity (0 / 1) per spatial axis")
    d, h, w = ([1] * (3 - nd) + sp)
    r = [0] * (3 - nd) + [int(v) for v in phase]
    osp = [(s_ - r_ + 1) // 2 for s_, r_ in zip(sp, phase)]
    x = x if arena_ld(x) >= x.shape[-1] else x.contiguous()
    out = torch.empty((x.shape[0], *osp, x.shape[-1]), dtype=x.dtype, device=x.device)
    check(lib().gm_phase2x(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), x.shape[0], x.shape[-1], d, h, w, 1 if nd == 3 else 0,
                           (r[0] << 2) | (r[1] << 1) | r[2], dt_code(x.dtype), _stream()), "gm_phase2x")
    return out


# ------------------------------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------------------------------
def nearest_resize(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """F.interpolate(mode="nearest") of an arena tensor (N, *spatial, C) to the spatial `size`."""
    require_device(x)
    nsp = x.dim() - 2
    if len(size) != nsp or nsp < 1 or nsp > 3:
        raise ValueError("size must have one entry per spatial axis (1-3 axes)")
    x = x.contiguous()
    si = (1,) * (3 - nsp) + tuple(x.shape[1:-1])
    so = (1,) * (3 - nsp) + tuple(int(v) for v in size)
    out = torch.empty((x.shape[0], *[int(v) for v in size], x.shape[-1]), dtype=x.dtype, device=x.device)
    check(lib().gm_nearest_resize(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), x.shape[0], *si, *so, x.shape[-1],
                                  dt_code(x.dtype), _stream()), "gm_nearest_resize")
    return out


def gn_scale_shift(x: torch.Tensor, groups: int, eps: float, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                   want_stats: bool = False):
    """GroupNorm statistics of an arena tensor -> fp32 (scale, shift) of shape [N, C] for a consumer's prologue."""
    require_device(x)
    n, c = x.shape[0], x.shape[-1]
    v = rows_of(x) // max(n, 1)
    if c % groups != 0:
        raise ValueError("num_channels must be divisible by num_groups")
    ws_bytes = lib().gm_gn_workspace_bytes(n, v, c, groups, dt_code(x.dtype))
    if ws_bytes < 0:
        raise ValueError(f"GroupNorm over {c} channels is not supported by the gfx950 kernel")
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    scale = torch.empty((n, c), dtype=torch.float32, device=x.device)
    shift = torch.empty((n, c), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if want_stats:
        mean = torch.empty((n, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty((n, groups), dtype=torch.float32, device=x.device)
    g32, b32 = as_f32(gamma), as_f32(beta)
    _timed(f"gn_stats<{str(x.dtype).split('.')[-1]}>", dict(flops=0.0, bytes=float(x.element_size() * n * v * c), shape=f"N{n} V{v} C{c}"),
           lambda: check(lib().gm_gn_scale_shift(x.data_ptr(), arena_ld(x), n, v, c, groups, float(eps), _ptr(g32), _ptr(b32),
                                                 scale.data_ptr(), shift.data_ptr(), _ptr(mean), _ptr(rstd), ws.data_ptr(),
                                                 dt_code(x.dtype), _stream()), "gm_gn_scale_shift"))
    if want_stats:
        return scale, shift, mean, rstd
    return scale, shift


def gn_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, act: str = "none", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x * scale[n, c] + shift[n, c]); scale/shift may be channel slices of a wider [N, C_total] table, `out` a channel
    slice of a wider arena buffer (that is how a concatenated, activated operand is assembled without a copy pass)."""
    require_device(x, scale, shift, out)
    n, c = x.shape[0], x.shape[-1]
    v = rows_of(x) // max(n, 1)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != x.dtype:
        raise ValueError("gn_apply: out must match x")
    if scale.shape != (n, c) or shift.shape != (n, c) or scale.stride(1) != 1 or shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
        raise ValueError("gn_apply: scale/shift must be matching [N, C] (slices of) fp32 tables")
    ss_ld = scale.stride(0) if n > 1 else max(scale.stride(0), c)
    _timed(f"gn_apply<{str(x.dtype).split('.')[-1]}>", dict(flops=0.0, bytes=float(2 * x.element_size() * x.numel()), shape=str(tuple(x.shape))),
           lambda: check(lib().gm_gn_apply(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), scale.data_ptr(), shift.data_ptr(),
                                           ss_ld, n, v, c, ACT[act], dt_code(x.dtype), _stream()), "gm_gn_apply"))
    return out


def spade_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, g: torch.Tensor, bm: torch.Tensor, act: str = "none",
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act((x * scale[n, c] + shift[n, c]) * g + bm): the SPADE modulation of a parameter-free-normalised tensor; g = 1 + gamma(seg) and
    bm = beta(seg) are arena tensors like x (channel slices of wider buffers allowed, as are x / out)."""
    require_device(x, scale, shift, g, bm, out)
    n, c = x.shape[0], x.shape[-1]
    v = rows_of(x) // max(n, 1)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if g.shape != x.shape or bm.shape != x.shape or g.dtype != x.dtype or bm.dtype != x.dtype or arena_ld(g) != arena_ld(bm):
        raise ValueError("spade_apply: g / bm must match x (shape, dtype) and share a row pitch")
    if scale.shape != (n, c) or shift.shape != (n, c) or scale.stride(1) != 1 or scale.stride(0) != shift.stride(0):
        raise ValueError("spade_apply: scale/shift must be matching [N, C] (slices of) fp32 tables")
    ss_ld = scale.stride(0) if n > 1 else max(scale.stride(0), c)
    _timed(f"spade_apply<{str(x.dtype).split('.')[-1]}>", dict(flops=0.0, bytes=float(4 * x.element_size() * x.numel()), shape=str(tuple(x.shape))),
           lambda: check(lib().gm_spade_apply(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), scale.data_ptr(), shift.data_ptr(), ss_ld,
                                              g.data_ptr(), bm.data_ptr(), arena_ld(g), n, v, c, ACT[act], dt_code(x.dtype), _stream()), "gm_spade_apply"))
    return out


class VirtualCat:
    """Channel concatenation that is never materialised (reference: torch.cat([h, skip], dim=1), diffusion_model_unet.py:1232,
    1340,1461): its consumers read the parts directly."""

    def __init__(self, parts: Sequence[torch.Tensor]):
        self.parts = list(parts)
        if len(self.parts) != 2:
            raise ValueError("VirtualCat holds two parts")
        a, b = self.parts
        if a.shape[:-1] != b.shape[:-1] or a.dtype != b.dtype:
            raise ValueError("VirtualCat parts must agree outside the channel dim")
        self.shape = (*a.shape[:-1], a.shape[-1] + b.shape[-1])
        self.dtype, self.device = a.dtype, a.device

    def numel(self) -> int:
        return math.prod(self.shape)

    def materialise(self) -> torch.Tensor:
        return concat_channels(self.parts)


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    """fp64 [S, N, C, 2] per-channel (sum, sum of squares) partials of an arena tensor (sum over dim 0 = the statistics; S = one partial per
    block of rows, each stored exactly once -- no atomics -- so every consumer's fixed-order sum is bit-reproducible); cached on the tensor
    object (the fast convolution kernels attach their per-tile partials to their outputs for free)."""
    cached = getattr(x, "_gm_cstats", None)
    if cached is not None:
        return cached
    st = _fresh_channel_stats(x)
    try:
        x._gm_cstats = st
    except Exception:  # pragma: no cover
        pass
    return st


STATS_COMPACT_ABOVE = int(os.environ.get("GM_STATS_COMPACT_ABOVE", "256"))  # tables with more partials than this are folded to gm_stats_compact_slots() rows right after they are produced


def _compact_stats(st: torch.Tensor) -> torch.Tensor:
    """[S, N, C, 2] -> [256, N, C, 2] (fixed-order fold, gm_stats_compact) when S is large: every later consumer then reads a small table."""
    s_, n, c, _ = st.shape
    if s_ <= STATS_COMPACT_ABOVE:
        return st
    out = torch.empty((int(lib().gm_stats_compact_slots()), n, c, 2), dtype=torch.float64, device=st.device)
    check(lib().gm_stats_compact(st.data_ptr(), s_, n, c, out.data_ptr(), _stream()), "gm_stats_compact")
    return out


def _fresh_channel_stats(x: torch.Tensor) -> torch.Tensor:
    require_device(x)
    n, c = x.shape[0], x.shape[-1]
    v = rows_of(x) // max(n, 1)
    if v == 0 or n == 0:  # an empty tensor: the kernel launches nothing, so the (otherwise uninitialised) table is zeros by construction
        return torch.zeros((1, n, c, 2), dtype=torch.float64, device=x.device)
    slots = lib().gm_gn_channel_stats_slots(x.data_ptr(), arena_ld(x), v, c, dt_code(x.dtype))
    if slots <= 0:
        raise ValueError(f"per-channel statistics over {c} channels are not supported by the gfx950 kernel")
    st = torch.empty((slots, n, c, 2), dtype=torch.float64, device=x.device)
    _timed(f"gn_stats<{str(x.dtype).split('.')[-1]}>", dict(flops=0.0, bytes=float(x.element_size() * x.numel()), shape=f"N{n} V{v} C{c}"),
           lambda: check(lib().gm_gn_channel_stats(x.data_ptr(), arena_ld(x), n, v, c, st.data_ptr(), dt_code(x.dtype), _stream()),
                         "gm_gn_channel_stats"))
    return _compact_stats(st)


GN_IN_CONSUMER = os.environ.get("GM_GN_IN_CONSUMER", "1") != "0"  # short statistic tables: the fold + finalisation in the consumer convolution's prologue (cfg 24 / 25)
GN_IN_TOKEN_GEMM = os.environ.get("GM_GN_IN_TOKEN_GEMM", "1") != "0"    # ... and in the wide token GEMM (the q | k | v projection of an attention block)
GN_IN_SPLIT_SLICES = os.environ.get("GM_GN_IN_SPLIT_SLICES", "1") != "0"  # ... also in the K slices of a split launch (conv_sk.hip, second half of round 6)
GN_IN_CONSUMER_MAX_ROWS = int(os.environ.get("GM_GN_IN_CONSUMER_MAX_ROWS", "128"))  # = GN_SHORT_MAX_ROWS of csrc/gm_common.h (64 until the 32^3 level's 128-row tables were measured)


class GnRecipe:
    """The (scale, shift) of a GroupNorm that has NOT been computed yet: the per-channel statistic tables of the input's producer(s) plus gamma / beta / eps /
    groups.  Tile configurations 24 / 25 (csrc/conv_sn.hip) take it as it is -- GmConvDesc.pre_stats: they fold the partials and form scale / shift in their own
    prologue, bit-identical to gm_gn_finalize_channels, and the finalisation launch of that norm (4.4-4.8 us: 21 per forward of the BASELINE configs[0] UNet, a fifth
    of its kernel time) does not happen.  Every other consumer unpacks or indexes it like the tuple it stands for, which runs the launch then (once: cached)."""

    def __init__(self, stats, cs, n: int, v: int, groups: int, eps: float, gamma, beta, device) -> None:
        self.stats, self.cs, self.n, self.v, self.groups, self.eps, self.device = stats, cs, n, v, groups, float(eps), device
        self.gamma, self.beta = as_f32(gamma), as_f32(beta)
        self._done = None

    def materialise(self):
        if self._done is None:
            c = sum(self.cs)
            scale = torch.empty((self.n, c), dtype=torch.float32, device=self.device)
            shift = torch.empty((self.n, c), dtype=torch.float32, device=self.device)
            st = self.stats
            check(lib().gm_gn_finalize_channels(st[0].data_ptr(), st[0].shape[0], self.cs[0], st[1].data_ptr() if len(st) > 1 else None,
                                                st[1].shape[0] if len(st) > 1 else 0, self.cs[1] if len(st) > 1 else 0, self.n, self.v, self.groups, self.eps,
                                                _ptr(self.gamma), _ptr(self.beta), scale.data_ptr(), shift.data_ptr(), _stream()), "gm_gn_finalize_channels")
            self._done = (scale, shift)
        return self._done

    def __iter__(self):
        return iter(self.materialise())

    def __getitem__(self, i):
        return self.materialise()[i]

    def __len__(self):
        return 2


def gn_scale_shift_composed(x, groups: int, eps: float, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor]):
    """GroupNorm (scale, shift) [N, C] of a tensor or a VirtualCat from per-channel statistics (fused into the producing
    convolution's epilogue when available, else one stats pass per part).  With short statistic tables the result is a GnRecipe (same unpacking / indexing as
    the tuple): a consumer on tile configuration 24 / 25 finalises in its own prologue, anything else triggers the launch on first use."""
    parts = x.parts if isinstance(x, VirtualCat) else [x]
    n = parts[0].shape[0]
    v = rows_of(parts[0]) // max(n, 1)
    cs = [p.shape[-1] for p in parts]
    c = sum(cs)
    if c % groups != 0:
        raise ValueError("num_channels must be divisible by num_groups")
    stats = [channel_stats(p) for p in parts]
    recipe = GnRecipe(stats, cs, n, v, groups, eps, gamma, beta, parts[0].device)
    if GN_IN_CONSUMER and not torch.is_grad_enabled() and all(st.shape[0] <= GN_IN_CONSUMER_MAX_ROWS for st in stats) and c <= 384 and n > 0:
        return recipe
    return recipe.materialise()


# GroupNorm-apply + SiLU placement.  "prologue": inside the consumer convolution's patch staging (no normalised tensor in HBM).
# "pass": one vectorised element-wise pass that writes the activated tensor, the convolution then runs without a prologue.
# Measured on MI355X (tools/bench_conv.py): the in-kernel prologue is applied to every halo row (2.1-2.5x redundant) on the VALU
# while the work-group's MFMAs wait, and costs ~3x the HBM-bound pass; "auto" therefore un-fuses for large bf16 tensors.
GN_APPLY_POLICY = "auto"
# The LDS-DMA 3x3x3 kernels can apply the prologue IN LDS to the landed patch (csrc/conv_dma.hip: transform_patch; they then also read a
# VirtualCat's two parts in place, GmConvDesc.x2): no activated tensor in HBM, no extra pass, one launch less.  Measured on MI355X
# (profiles/r02_gn_prologue_ab.txt): the transform runs between the patch wait and the barrier of every K chunk, where it is exposed --
# both work-groups of a CU are then short of MFMA work -- and costs what the HBM-bound pass costs (64->64 at 128^3: 0.60 ms fused vs
# 0.48 + 0.11 two-pass) or more (deep K: 384->128 at 64^3 0.67 vs 0.54 + 0.08).  It wins where launches, not bytes, set the time: the
# policy "auto" fuses convolutions below DMA_FUSED_PROLOGUE_MAX_FLOP (the 32^3 / 16^3 / 8^3 levels of the latent UNet: 4.95 -> 4.3 ms per
# forward) and keeps the two-pass form above.  "always" / "never" pin it (A/B measurements, tests).
DMA_FUSED_PROLOGUE = os.environ.get("GM_DMA_FUSED_PROLOGUE", "auto")
DMA_FUSED_PROLOGUE_MAX_FLOP = 3.0e10
# Split-K for small grids (GmConvDesc.ksplit): a 3x3x3 convolution over a 32^3 .. 8^3 latent has fewer 256-voxel x 64-channel tiles than the
# chip has CUs, and each tile is a serial chain of K chunks with an exposed LDS-DMA round trip per chunk (256 input channels = 8 chunks:
# ~35 us for 0.3 us of arithmetic).  Below SPLITK_MAX_TILES work-groups the chunks are dealt to up to SPLITK_MAX slices per tile; a combine
# kernel sums the fp32 partials and applies the epilogue.
SPLITK = os.environ.get("GM_CONV_SPLITK", "1") != "0"
SPLITK_MAX_TILES = 256   # split when the unsplit launch has fewer work-groups than this ...
# ... into as many slices as it takes to reach about this many.  One per CU: a slice work-group of conv_sk.hip owns its CU's LDS, so a second wave of
# work-groups runs after the first (round 3, two per CU on the general tile kernel: 512; measured on the C3 latent UNet: 1.657 vs 1.678 ms per forward)
SPLITK_TARGET_WGS = int(os.environ.get("GM_CONV_SPLITK_WGS", "256"))
SPLITK_MAX = int(os.environ.get("GM_CONV_SPLITK_MAX", "8"))
_SK_KERNEL = os.environ.get("GM_CONV_SK")  # "0": the slices on the general cfg 11 tile kernel (round-3 path) instead of conv_sk.hip -- A/B measurements only
DMA_CFGS = (11, 14, 15, 16, 17, 18, 19, 24, 25)
# cfg 24 (csrc/conv_sn.hip, round 6): a small volume's 3x3x3 convolution K-COMPLETE on 16-channel output blocks -- 256 voxels x 16 channels per work-group, the
# epilogue and the GroupNorm statistics in the kernel -- instead of split-K slices + a combine launch.  Taken where the launch would have been split (fewer
# 64-channel tiles than SPLITK_MAX_TILES) and the contraction is at most NARROW_N_MAX_CHUNKS K chunks deep (a work-group walks them one after the other; deeper
# contractions keep the K slices), and for the C_out <= 16 heads the tile kernels do not cover (the latent UNet's 64 -> 4 output convolution).
NARROW_N = os.environ.get("GM_CONV_SN", "1") != "0"
NARROW_N_MAX_CHUNKS = int(os.environ.get("GM_CONV_SN_MAX_CHUNKS", "6"))
# cfg 25: the same kernel over IMAGES (2-D 3x3 stride-1 convolutions, also over a nearest-2x up-sampled input): 16 x 16 pixels x 16 channels per work-group with the
# in-LDS GroupNorm + SiLU prologue, the two-source input, the fused 1x1 shortcut and the statistics -- where the register-staged 2-D kernels took a separate
# gn_stats / gn_apply / shortcut launch each (BASELINE configs[0], the 2-D DDPM UNet).  Up to NARROW_N_2D_MAX_FLOP per convolution (what was measured).
NARROW_N_2D = os.environ.get("GM_CONV_SN2D", "1") != "0"
# conv_in of a 2-D network (C_in <= 4, 3x3, stride 1) on the C_in <= 4 edge kernel (cfg 12) as a depth-1 volume: BASELINE configs[0]'s 1 -> 32 at 16 x 64 x 64 ran
# on the generic tile kernel (27 us, the longest launch of that forward, and a stand-alone statistics pass behind it)
EDGE_2D_AS_3D = os.environ.get("GM_CONV_EDGE2D", "1") != "0"
NARROW_N_2D_MAX_FLOP = float(os.environ.get("GM_CONV_SN2D_MAX_FLOP", "3e10"))
# ... and the stride-2 Downsample convolution of a 2-D UNet on the same kernel (the stride-1 result's even positions: conv_sn.hip), statistics fused
NARROW_N_2D_STRIDE2 = os.environ.get("GM_CONV_SN2D_STRIDE2", "1") != "0"
NARROW_HEAD_MAX_FLOP = 1.0e9
# (Rounds 4-5 built three more tile structures on v_mfma_f32_32x32x16_bf16 -- cfg 21: 16-channel half-chunks, three work-groups per CU; cfg 22: 512-voxel
#  tiles with 16-channel weight panels; cfg 23: cfg 22's image on four waves of 4 x 2 blocks -- each verified bit-level and measured: all tie or lose against
#  cfg 14 in time, and in round 6 in JOULES per launch on every C2 shape (profiles/r06_taploop_energy.txt: +1 ... +16 %).  They live under experiments/
#  with their host replays, not in the library.)
COUT1_MARCH = os.environ.get("GM_CONV_COUT1_MARCH", "1") != "0"  # C_out == 1 heads: the depth-marching kernel (cfg 20) before the tile kernel (cfg 13)
DMA_WIDE_WAVES = os.environ.get("GM_CONV_WIDE_WAVES", "1") != "0"  # prefer cfg 14 (4 waves x 64 voxels) for large prologue-free stride-1 convolutions
DMA_WIDE_WAVES_PRE = os.environ.get("GM_CONV_WIDE_WAVES_PRE", "0") != "0"  # ... also with the fused in-LDS prologue (its cfg 14 instantiation)
DMA_WIDE_WAVE_MIN_TILES = 512                                       # ... from one full wave of work-groups on (2 per CU)


def fuse_gn_prologue(x: torch.Tensor) -> bool:
    if GN_APPLY_POLICY == "prologue":
        return True
    if GN_APPLY_POLICY == "pass":
        return False
    if x.dtype != torch.bfloat16 or x.shape[-1] % 8 != 0:
        return True
    if x.numel() >= (1 << 22):
        return False
    # small 3-D tensors whose convolution the LDS-DMA kernel covers (C % 32 == 0): below ~16^3 voxels the prologue-fused kernels are
    # serial chains of 200+ barriers per work-group (52-172 us, tools/bench_conv_small.py) while gn_apply (launch-bound, 16 us) + the
    # LDS-DMA kernel take 43-64 us; in between (32^3 x 64) the fused kernel still wins (33 vs 41 us)
    vox = x.numel() // max(x.shape[0] * x.shape[-1], 1)
    return not (x.dim() == 5 and x.shape[-1] % 32 == 0 and vox <= 4096)


def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    require_device(x)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(lib().gm_layernorm(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), _ptr(as_f32(gamma)), _ptr(as_f32(beta)),
                             rows_of(x), x.shape[-1], float(eps), dt_code(x.dtype), _stream()), "gm_layernorm")
    return out


def geglu(x: torch.Tensor) -> torch.Tensor:
    require_device(x)
    inner = x.shape[-1] // 2
    out = torch.empty((*x.shape[:-1], inner), dtype=x.dtype, device=x.device)
    check(lib().gm_geglu(x.data_ptr(), arena_ld(x), out.data_ptr(), arena_ld(out), rows_of(x), inner, dt_code(x.dtype), _stream()),
          "gm_geglu")
    return out


# ------------------------------------------------------------------------------------------------------------------------
# convolution / linear
# ------------------------------------------------------------------------------------------------------------------------
def _clog2(v: int) -> int:
    return max(0, (int(v) - 1).bit_length())


def _tile_bits(total_bits: int, out_dims):
    """Distribute log2(voxels per tile) over (d, h, w): grow the smallest tile extent first, never beyond the output."""
    caps = [_clog2(d) for d in out_dims]
    bits = [0, 0, 0]
    for _ in range(total_bits):
        cand = [i for i in (2, 1, 0) if bits[i] < caps[i]]
        if not cand:
            bits[2] += 1
            continue
        i = min(cand, key=lambda j: bits[j])
        bits[i] += 1
    return bits


def _tile_bits_fast(total_bits: int, out_dims):
    """Tiles of the fast kernels are 16 voxels wide (16 consecutive LDS rows per MFMA fragment = conflict-free reads);
    the remaining bits go to h and d, smallest extent first."""
    caps = [_clog2(d) for d in out_dims]
    bits = [0, 0, min(4, caps[2], total_bits)]
    for _ in range(total_bits - bits[2]):
        cand = [i for i in (1, 0) if bits[i] < caps[i]]
        if not cand:
            bits[2] += 1
            continue
        i = min(cand, key=lambda j: bits[j])
        bits[i] += 1
    return bits


_CFG_TILES = {}


def _cfg_tile(cfg: int):
    if cfg not in _CFG_TILES:
        bm, bn = C.c_int(), C.c_int()
        check(lib().gm_conv_cfg_tile(cfg, C.byref(bm), C.byref(bn)), "gm_conv_cfg_tile")
        _CFG_TILES[cfg] = (bm.value, bn.value)
    return _CFG_TILES[cfg]


COUT1_MARCH_LTD = os.environ.get("GM_CONV_COUT1_LTD")  # bench only: pin the depth-segment length (log2 planes) of configuration 20
# log2 of the output columns a configuration-20 work-group walks over 128-byte rows: 5 = 8 x 32 (one work-group per CU, halo 1.33), 4 = 8 x 16 (69 KiB of
# LDS: two work-groups per CU, halo 1.41; round 5 A/B: GM_CONV_COUT1_LTW)
COUT1_MARCH_LTW_128B = int(os.environ.get("GM_CONV_COUT1_LTW", "4"))  # measured (gpurun r5v2): C2 out head 0.197-0.200 -> 0.139 ms


_CONV_DEBUG_FLAGS = 0  # tools/bench_conv.py ablations only
_SK_STAMPS = None  # tools/sk_timeline.py: the stamp table of the last split-K launch (debug_flags bit 12)
_CONV_TIMELINE_BUFFER = None  # tools/conv_timeline.py (bench-only -DGM_CONV_TIMELINE build): int64 [work-groups, 64] stamp table

SMALL_LINEAR_ROWS = 64       # 1x1 "convolutions" over at most this many rows take gm_linear_rows
# 1x1 convolutions over token rows (the GN-prologue q|k|v projections of the latent-resolution attention blocks and their gradients): the tiled
# kernels cost ~20 us per launch there whatever the size (a stage -> barrier -> tap -> barrier chain per K chunk); gm_linear_rows_affine requests
# all K chunks of a 16 x 16 output block straight from L2.  Bounds: what was measured (tools/bench_conv1x1.py)
TOKEN_GEMM = os.environ.get("GM_TOKEN_GEMM", "1") != "0"
TOKEN_GEMM_MAX_ROWS = int(os.environ.get("GM_TOKEN_GEMM_MAX_ROWS", "32768"))  # (8192 until the wide form of round 6: small_ops.hip token_gemm_wide_kernel)
TOKEN_GEMM_MAX_CIN = 512
TOKEN_GEMM_MAX_FLOP = float(os.environ.get("GM_TOKEN_GEMM_MAX_FLOP", "2.0e9"))
DMA_CONV = True              # route eligible 3x3x3 convolutions through conv_dma.hip (cfg 11)
DMA_CONV_MIN_VOXELS = 1 << 8
LDS_SOFT_LIMIT = 80 * 1024   # two workgroups per CU
LDS_HARD_LIMIT = 160 * 1024


def _choose_conv_cfg(desc: GmConvDesc, n_vox_out: int, force_cfg: Optional[int] = None, only: Optional[tuple] = None, exclude: tuple = ()):
    """Pick the tile configuration (desc.cfg + tile bits) for a built descriptor; `only` / `exclude` restrict the candidates."""
    cout = desc.Cout
    if force_cfg is not None:
        order = [force_cfg]
    elif cout <= 16:
        order = [3, 4, 2]
    elif cout <= 64:
        # 16 waves per CU measured best at every C2 shape (tools/bench_conv.py); 512-voxel tiles once there are >= 8 tiles per CU
        order = [10, 8, 5, 0, 4, 2] if n_vox_out * desc.N >= (1 << 20) else [8, 5, 0, 4, 2]
    else:
        order = [10, 9, 6, 1, 4, 2] if n_vox_out * desc.N >= (1 << 20) else [9, 6, 1, 4, 2]
    if force_cfg is None and n_vox_out * desc.N >= DMA_CONV_MIN_VOXELS:
        if cout == 1:
            # taps-as-N kernels of the single-channel output heads (conv_edge.hip): 20 = marching along depth (every input row staged once per
            # work-group by LDS-DMA, planes double-buffered), 13 = one 4x4x16 tile per work-group (what 20 does not cover)
            order = ([20, 13] if COUT1_MARCH else [13]) + order
        if desc.Cin <= 4:
            order = [12] + order  # taps-as-K kernel of the 1..4-channel input convolutions
    if force_cfg is None and cout > 16 and DMA_CONV and n_vox_out * desc.N >= DMA_CONV_MIN_VOXELS:
        # LDS-DMA 3x3x3 kernels: the C side rejects (lds = -1) whatever a configuration does not cover.  Stride 1: the 4-wave x 64-voxel form
        # (cfg 14: two operand register sets, software-pipelined tap loop -- 256 registers per wave) is 2-9 % faster than the 8-wave x 32-voxel
        # form (cfg 11, 128 registers: one operand set) on every C2 / C3 shape once the grid fills the chip (profiles/r02_conv_tile_configs_v2.txt);
        # cfg 11 keeps the fused-prologue instantiation (the cfg 14 one spills) and the small grids (its split-K form).
        tiles = desc.N * -(-desc.Do // 4) * -(-desc.Ho // 4) * -(-desc.Wo // 16) * -(-cout // 64)
        wide = (bool(desc.pre_scale is None or not desc.pre_scale) or DMA_WIDE_WAVES_PRE) and tiles >= DMA_WIDE_WAVE_MIN_TILES and DMA_WIDE_WAVES
        order = ([15] if desc.sd == 2 else ([14, 11] if wide else [11])) + order
        # (512-voxel tiles -- cfg 16 / 18, one work-group per CU, half the weight-panel traffic -- measure within +-5 % of two 256-voxel
        # work-groups in isolation and 5-15 % slower on the 64 -> 64 layers inside the forward: profiles/r02_conv_tile_configs.txt,
        # r02_layer_times_cfg16_rule.txt.  They stay available through force_cfg.)
    if (force_cfg is None and NARROW_N_2D and DMA_CONV and desc.kd == 1 and desc.Ds == 1 and desc.kh == 3 and desc.kw == 3
            and ((desc.sh == 1 and desc.sw == 1) or (NARROW_N_2D_STRIDE2 and desc.sh == 2 and desc.sw == 2))
            and 2.0 * n_vox_out * desc.N * cout * desc.Cin * 9 * desc.sh * desc.sw <= NARROW_N_2D_MAX_FLOP):
        order = [25] + order  # images: the K-complete 16-channel-block kernel (the C side rejects what it does not cover)
    if force_cfg is None and n_vox_out * desc.N <= 256 * 64:  # small problem: favour more, smaller workgroups (the LDS-DMA kernels stay first)
        dma_first = [c for c in order if c in (11, 12, 15, 18, 19)]  # (12: the C_in <= 4 edge kernel -- its cost is the output store, whatever the tile)
        rest = [c for c in order if c not in (11, 12, 15, 18, 19)]
        order = dma_first + [c for c in rest if _cfg_tile(c)[0] <= 64] + [c for c in rest if _cfg_tile(c)[0] > 64]
    order = [c for c in order if (only is None or c in only) and c not in exclude]
    best = None
    for cfg in order:
        bm, _ = _cfg_tile(cfg)
        tb = _tile_bits_fast if cfg >= 5 else _tile_bits
        bits = tb(bm.bit_length() - 1, (desc.Do, desc.Ho, desc.Wo))
        if cfg == 20:
            # rows of 128 bytes walk 8 x 32 output columns, rows of 256 bytes 8 x 16; the depth segment of a work-group (2^ltd planes, halo
            # overhead (2^ltd + 2) / 2^ltd) is the longest that still gives every CU a work-group
            ltw = (COUT1_MARCH_LTW_128B if desc.Cin * (4 if desc.dtype == 0 else 2) == 128 else 4)
            cols = desc.N * -(-desc.Ho // 8) * -(-desc.Wo // (1 << ltw))
            want = 512 if (ltw == 4 and desc.Cin * (4 if desc.dtype == 0 else 2) == 128) else 256  # (8 x 16 columns of 128-byte rows: two work-groups per CU)
            ltd = 5
            while ltd > 2 and cols * -(-desc.Do // (1 << ltd)) < want:
                ltd -= 1
            if COUT1_MARCH_LTD is not None:
                ltd = int(COUT1_MARCH_LTD)
            bits = [ltd, 3, ltw]
        if cfg in (11, 12, 14, 15, 16, 18, 19, 24, 25):  # the LDS-DMA kernels (and the C_in <= 4 edge kernel) are built for fixed tiles; extents below the tile are masked (W = 8 at the
            bits = {11: [2, 2, 4], 12: [2, 2, 4], 14: [2, 2, 4], 15: [1, 2, 4], 16: [3, 2, 4], 18: [3, 2, 4], 19: [3, 2, 4], 24: [2, 2, 4], 25: [0, 4, 4]}[cfg]  # 8^3 level: half the tile idles, still 2x faster than cfg 4)
        desc.cfg, desc.ltd, desc.lth, desc.ltw = cfg, bits[0], bits[1], bits[2]
        lds = lib().gm_conv_lds_bytes(C.byref(desc))  # -1: configuration not applicable to this geometry
        soft = LDS_HARD_LIMIT if cfg >= 5 else LDS_SOFT_LIMIT  # the fast kernels are sized for their own occupancy
        if 0 < lds <= soft:
            return
        if 0 < lds <= LDS_HARD_LIMIT and best is None:
            best = (cfg, bits)
    if best is None:
        raise ValueError("convolution geometry needs more than 160 KiB of LDS per tile (kernel/stride/dilation too large)")
    desc.cfg, (desc.ltd, desc.lth, desc.ltw) = best[0], best[1]


def _attach_conv_stats(d: GmConvDesc, out: torch.Tensor, n: int, cout: int) -> None:
    """Point the descriptor (tile configuration already chosen) at a fresh [S, N, Cout, 2] table of per-tile partials and attach it to the
    output tensor; nothing happens when this configuration does not fuse the statistics (gm_conv_stats_slots = 0)."""
    slots = lib().gm_conv_stats_slots(C.byref(d))
    if slots > 0:
        cst = torch.empty((slots, n, cout, 2), dtype=torch.float64, device=out.device)
        d.stats = cst.data_ptr()
        out._gm_cstats = cst


def _conv_subpixel(x, weight, bias, rowvec, res, post_act, out, want_stats, n, cin, cout, src, packed=None, label="mode3 (executed flops: 8 of the 27 taps)"):
    """Nearest-2x up-sampling + 3x3x3 convolution as 8 sub-pixel 2x2x2 convolutions (GmConvDesc.in_mode 3, configuration 17).
    Returns None when the geometry is not covered (the caller falls back to the folded up-sampling path)."""
    dtype = x.dtype
    out_sp = tuple(2 * v for v in src)
    out_shape = (n, *out_sp, cout)
    if out is None:
        out = torch.empty(out_shape, dtype=dtype, device=x.device)
    elif tuple(out.shape) != out_shape or out.dtype != dtype:
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {out_shape}")
    d = GmConvDesc()
    if packed is None:
        packed = packed_subpixel_weight(weight, dtype)
    d.in_mode, d.fd, d.fh, d.fw = 3, 2, 2, 2
    d.x, d.x_ld, d.w = x.data_ptr(), arena_ld(x), packed.data_ptr()
    b32 = as_f32(bias) if bias is not None else None
    d.bias = _ptr(b32)
    d.pre_scale = d.pre_shift = None
    if rowvec is not None:
        if rowvec.dtype != torch.float32 or rowvec.dim() != 2 or rowvec.shape[1] != cout or rowvec.shape[0] not in (1, n) or rowvec.stride(1) != 1:
            raise ValueError(f"rowvec must be fp32 [1 or N, Cout], got {tuple(rowvec.shape)} {rowvec.dtype}")
        d.rowvec, d.rowvec_bstride = rowvec.data_ptr(), (0 if rowvec.shape[0] == 1 else rowvec.stride(0))
    else:
        d.rowvec, d.rowvec_bstride = None, 0
    if res is not None:
        if tuple(res.shape) != out_shape or res.dtype != dtype:
            raise ValueError(f"residual has shape {tuple(res.shape)} / {res.dtype}, expected {out_shape} / {dtype}")
        d.res, d.res_ld = res.data_ptr(), arena_ld(res)
    else:
        d.res, d.res_ld = None, 0
    d.y, d.y_ld = out.data_ptr(), arena_ld(out)
    d.skip_w = d.skip_bias = None
    d.skip_x[0] = d.skip_x[1] = None
    d.N, d.Cin, d.Cout = n, cin, cout
    d.Ds, d.Hs, d.Ws = src
    d.Do, d.Ho, d.Wo = out_sp
    d.kd = d.kh = d.kw = 2
    d.sd = d.sh = d.sw = 1
    d.pd = d.ph = d.pw = 0
    d.dd = d.dh = d.dw = 1
    d.pre_act, d.post_act, d.dtype = 0, POST_ACT[post_act], dt_code(dtype)
    d.debug_flags = _CONV_DEBUG_FLAGS
    d.cfg, d.ltd, d.lth, d.ltw = 17, 2, 2, 4
    if lib().gm_conv_lds_bytes(C.byref(d)) <= 0:
        return None
    d.stats = None
    if want_stats:
        _attach_conv_stats(d, out, n, cout)
    nvo = n * math.prod(out_sp)
    es = x.element_size()
    _timed(f"conv_igemm<{str(dtype).split('.')[-1]},cfg17>",
           dict(flops=2.0 * nvo * cout * cin * 8, bytes=float(es * (n * math.prod(src) * cin + nvo * cout * (2 if res is not None else 1) + 8 * cout * cin * 8)),
                shape=f"{cin}->{cout} k(3, 3, 3) s(1, 1, 1) out{out_sp} {label}"),
           lambda: check(lib().gm_conv_forward(C.byref(d), _stream()), "gm_conv_forward"))
    if d.stats:
        out._gm_cstats = _compact_stats(out._gm_cstats)
    return out


STRIDE2_DGRAD_SUBPIXEL = os.environ.get("GM_CONV_STRIDE2_DGRAD", "1") != "0"
TRANSPOSED_S2_SUBPIXEL = os.environ.get("GM_CONV_TRANSPOSED_S2_SUBPIXEL", "1") != "0"


def conv_stride2_dgrad(gy: torch.Tensor, weight: torch.Tensor, x_spatial: Sequence[int], pad_lo: int) -> Optional[torch.Tensor]:
    """dx of y = conv3d(x, weight, stride=2, padding=(pad_lo low, 1 high)) for even x extents, as ONE sub-pixel launch on gy
    (packed_stride2_dgrad_weight).  Returns None when the geometry is not covered: the caller takes the transposed-convolution path."""
    if not STRIDE2_DGRAD_SUBPIXEL or gy.dim() != 5 or weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or pad_lo not in (0, 1):
        return None
    n, src, cg = gy.shape[0], tuple(gy.shape[1:4]), gy.shape[4]
    if tuple(x_spatial) != tuple(2 * v for v in src) or cg != weight.shape[0]:
        return None
    vec = 16 // gy.element_size()
    if cg % (64 // gy.element_size()) != 0 or weight.shape[1] % vec != 0 or n * math.prod(x_spatial) >= 2 ** 31:
        return None
    require_device(gy, weight)
    packed = packed_stride2_dgrad_weight(weight, gy.dtype, pad_lo)
    return _conv_subpixel(gy, weight, None, None, None, "none", None, False, n, cg, weight.shape[1], src, packed=packed,
                          label="stride-2 data gradient as 8 sub-pixel 2x2x2 convolutions")


def conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], *, kernel, stride=1, padding=0, dilation=1,
         pad_hi=None, upsample: bool = False, transposed: bool = False, output_padding=0,
         pre: Optional[tuple] = None, pre_act: str = "none", rowvec: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, post_act: str = "none", out: Optional[torch.Tensor] = None,
         packed: Optional[torch.Tensor] = None, cout: Optional[int] = None, force_cfg: Optional[int] = None,
         want_stats: bool = False, skip: Optional[tuple] = None, allow_subpixel: bool = True, ksplit: Optional[int] = None,
         vt: Optional[tuple] = None) -> torch.Tensor:
    """Fused convolution over an arena tensor x = (N, *spatial, Cin) -- or over a VirtualCat of two (their channel concatenation).

    kernel/stride/padding/dilation: int or per-axis tuples (len = number of spatial axes). `padding` is the low-side pad,
    `pad_hi` the high side (default: same as low). upsample: nearest 2x folded into the input indexing.
    transposed: nn.ConvTransposeNd semantics (weight [Cin, Cout, *k]). pre = (scale, shift) fp32 [N, Cin] GroupNorm affine;
    pre_act in ACT; rowvec fp32 [B or 1, Cout]; res arena tensor in the output geometry; post_act in POST_ACT.
    `packed`/`cout` override the weight panel (fused multi-head projections).
    skip = (parts, weight, bias): a ResnetBlock's 1x1 shortcut convolution over cat(parts) (1 or 2 arena tensors in the output
    geometry), added to the result.  Fused into the LDS-DMA kernel as extra K chunks when it covers the geometry; otherwise
    computed by 1x1 launches first and added as the residual (`res` must then be None).
    allow_subpixel: an eligible up-sampling convolution runs as 8 sub-pixel 2x2x2 convolutions on pre-summed weights (cached per weight
    version: right for inference; a training step, whose weights change every iteration, passes False)."""
    x2 = None
    if isinstance(x, VirtualCat):
        x, x2 = x.parts
    require_device(x, x2, weight, bias, rowvec, res, out)
    nsp = x.dim() - 2
    if nsp < 1 or nsp > 3:
        raise ValueError("conv expects (N, *spatial, C) with 1-3 spatial axes")

    def two_pass():
        """The operand as ONE activated tensor (gm_gn_apply per part into channel slices / a channel copy), then the plain convolution:
        the form every kernel covers."""
        parts = [x] if x2 is None else [x, x2]
        if pre is None and x2 is None:
            raise AssertionError
        xa = torch.empty((*x.shape[:-1], sum(t.shape[-1] for t in parts)), dtype=x.dtype, device=x.device)
        off = 0
        for t in parts:
            c_t = t.shape[-1]
            if pre is not None:
                gn_apply(t, pre[0][:, off:off + c_t], pre[1][:, off:off + c_t], pre_act, out=xa[..., off:off + c_t])
            else:
                copy_channels(t, xa[..., off:off + c_t])
            off += c_t
        return conv(xa, weight, bias, kernel=kernel, stride=stride, padding=padding, dilation=dilation, pad_hi=pad_hi, upsample=upsample,
                    transposed=transposed, output_padding=output_padding, rowvec=rowvec, res=res, post_act=post_act, out=out, packed=packed,
                    cout=cout, force_cfg=force_cfg, want_stats=want_stats, skip=skip, allow_subpixel=allow_subpixel, ksplit=ksplit, vt=vt)

    def tup(v):
        v = tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * nsp
        if len(v) != nsp:
            raise ValueError("per-axis argument has the wrong length")
        return v

    def pad3(v, fill):
        return (fill,) * (3 - nsp) + tuple(v)

    k = pad3(tup(kernel), 1)
    s = pad3(tup(stride), 1)
    plo = pad3(tup(padding), 0)
    phi = pad3(tup(pad_hi if pad_hi is not None else padding), 0)
    dil = pad3(tup(dilation), 1)
    opad = pad3(tup(output_padding), 0)
    n, cin = x.shape[0], x.shape[-1] + (0 if x2 is None else x2.shape[-1])
    src = pad3(tuple(x.shape[1:-1]), 1)
    dtype = x.dtype
    if x2 is not None and (x2.shape[:-1] != x.shape[:-1] or x2.dtype != dtype):
        raise ValueError("the two parts of a concatenated input must agree outside the channel dim")
    if packed is None:
        cout = weight.shape[1] if transposed else weight.shape[0]
        wcin = weight.shape[0] if transposed else weight.shape[1]
        if wcin != cin:
            raise ValueError(f"input has {cin} channels but the weight expects {wcin}")

    def panel():  # the generic MFMA panel, packed on first use: the sub-pixel paths below bring their own images (no wasted pack per step)
        return packed if packed is not None else packed_conv_weight(weight, dtype, transposed)
    rows = n * math.prod(src)
    vecw = 16 // x.element_size()
    if (rows <= SMALL_LINEAR_ROWS and x2 is None and k == (1, 1, 1) and s == (1, 1, 1) and not transposed and not upsample and pre is None
            and rowvec is None and not want_stats and skip is None and force_cfg is None and cin % vecw == 0
            and arena_ld(x) % vecw == 0 and x.data_ptr() % 16 == 0 and plo == (0, 0, 0) and phi == (0, 0, 0)):
        # a handful of rows (decode steps, the timestep MLP): the barrier-free small-row GEMM instead of the tiled kernel
        out_shape = (*x.shape[:-1], cout)
        if out is None:
            out = torch.empty(out_shape, dtype=dtype, device=x.device)
        elif tuple(out.shape) != out_shape or out.dtype != dtype:
            raise ValueError(f"out has shape {tuple(out.shape)}, expected {out_shape}")
        if res is not None and (tuple(res.shape) != out_shape or res.dtype != dtype):
            raise ValueError(f"residual has shape {tuple(res.shape)} / {res.dtype}, expected {out_shape} / {dtype}")
        b32 = as_f32(bias) if bias is not None else None
        _timed(f"linear_rows<{str(dtype).split('.')[-1]}>", dict(flops=2.0 * rows * cin * cout, bytes=float(x.element_size() * cin * cout), shape=f"{rows}x{cin}->{cout}"),
               lambda: check(lib().gm_linear_rows(x.data_ptr(), arena_ld(x), panel().data_ptr(), _ptr(b32), _ptr(res), 0 if res is None else arena_ld(res),
                                                  out.data_ptr(), arena_ld(out), rows, cin, cout, ACT[pre_act], POST_ACT[post_act], dt_code(dtype),
                                                  _stream()), "gm_linear_rows"))
        return out
    if (EDGE_2D_AS_3D and nsp == 2 and x2 is None and packed is None and weight is not None and weight.dim() == 4 and not transposed and not upsample
            and cin <= 4 and k == (1, 3, 3) and s == (1, 1, 1) and dil == (1, 1, 1) and plo == (0, 1, 1) and phi == (0, 1, 1) and pre is None
            and pre_act == "none" and skip is None and force_cfg is None and ksplit is None and cout % vecw == 0 and rows >= DMA_CONV_MIN_VOXELS):
        # conv_in of a 2-D network (C_in <= 4): the taps x inputs ARE the GEMM K of the edge kernel (cfg 12, conv_edge.hip) -- handed over as the depth-1 volume
        # it is, the 3x3 kernel as the centre plane of a 3x3x3 one (zero planes in front of and behind it meet the padding planes only)
        def embed():
            w5 = torch.zeros((weight.shape[0], weight.shape[1], 3, 3, 3), dtype=weight.dtype, device=weight.device)
            w5[:, :, 1] = weight.detach()
            return w5
        got = conv(x.unsqueeze(1), _cached(weight, ("centre_plane_3d",), embed), bias, kernel=3, padding=1, rowvec=rowvec,
                   res=None if res is None else res.unsqueeze(1), post_act=post_act, out=None if out is None else out.unsqueeze(1), want_stats=want_stats)
        y = out if out is not None else got.squeeze(1)
        st = getattr(got, "_gm_cstats", None)
        if st is not None:
            y._gm_cstats = st
        return y
    if (TOKEN_GEMM and SMALL_LINEAR_ROWS < rows <= TOKEN_GEMM_MAX_ROWS and cin <= TOKEN_GEMM_MAX_CIN and 2.0 * rows * cin * cout <= TOKEN_GEMM_MAX_FLOP
            and x2 is None and k == (1, 1, 1) and s == (1, 1, 1) and not upsample and rowvec is None and not want_stats and skip is None
            and force_cfg is None and ksplit is None and cin % vecw == 0 and arena_ld(x) % vecw == 0 and x.data_ptr() % 16 == 0
            and plo == (0, 0, 0) and phi == (0, 0, 0) and opad == (0, 0, 0)):
        out_shape = (*x.shape[:-1], cout)
        if out is None:
            out = torch.empty(out_shape, dtype=dtype, device=x.device)
        elif tuple(out.shape) != out_shape or out.dtype != dtype:
            raise ValueError(f"out has shape {tuple(out.shape)}, expected {out_shape}")
        if res is not None and (tuple(res.shape) != out_shape or res.dtype != dtype):
            raise ValueError(f"residual has shape {tuple(res.shape)} / {res.dtype}, expected {out_shape} / {dtype}")
        b32 = as_f32(bias) if bias is not None else None
        if GN_IN_TOKEN_GEMM and isinstance(pre, GnRecipe) and pre._done is None and not transposed:
            # the GroupNorm is still a recipe (short statistic tables): the wide token GEMM finalises it in its prologue -- no launch for the norm
            l_tok = rows // n
            want_vt = (vt is not None and res is None and post_act == "none" and dtype == torch.bfloat16 and l_tok % 64 == 0)
            if (pre.n == n and sum(pre.cs) == cin and cin <= 384 and cin % (64 // x.element_size()) == 0 and cout >= 32 and l_tok % 64 == 0
                    and cin % pre.groups == 0 and all(st.shape[0] <= GN_IN_CONSUMER_MAX_ROWS for st in pre.stats) and (vt is None or want_vt)):
                g = GmGnTables()
                for i, st in enumerate(pre.stats):
                    g.stats[i], g.S[i], g.C[i] = st.data_ptr(), st.shape[0], pre.cs[i]
                g.gamma, g.beta, g.eps, g.groups = _ptr(pre.gamma), _ptr(pre.beta), pre.eps, pre.groups
                vws, vt_c0, vt_dh = vt if want_vt else (None, 0, 0)
                _timed(f"token_gemm<{str(dtype).split('.')[-1]}>", dict(flops=2.0 * rows * cin * cout, bytes=float(x.element_size() * (rows * (cin + cout) + cin * cout)),
                                                                      shape=f"{rows}x{cin}->{cout} (GroupNorm from statistics{' +V^T image' if want_vt else ''})"),
                       lambda: check(lib().gm_linear_rows_gn(x.data_ptr(), arena_ld(x), C.byref(g), n, l_tok, panel().data_ptr(), _ptr(b32), _ptr(res),
                                                             0 if res is None else arena_ld(res), out.data_ptr(), arena_ld(out), rows, cin, cout, ACT[pre_act],
                                                             POST_ACT[post_act], None if vws is None else vws.data_ptr(), int(vt_c0), int(vt_dh), dt_code(dtype),
                                                             _stream()), "gm_linear_rows_gn"))
                if want_vt:
                    out._gm_vt_packed = True
                return out
        sc = sh = None
        if pre is not None:
            sc, sh = pre
            if sc.dtype != torch.float32 or sh.dtype != torch.float32 or sc.shape != (n, cin) or sh.shape != (n, cin) or sc.stride(-1) != 1 or sh.stride(-1) != 1 \
                    or sc.stride(0) != sh.stride(0):
                raise ValueError("pre = (scale, shift): fp32 [N, Cin] tables with a common row pitch")
        if (vt is not None and res is None and post_act == "none" and dtype == torch.bfloat16 and (rows // n) % 64 == 0
                and (sc is None or (sc.stride(0) % 4 == 0 and sc.data_ptr() % 16 == 0 and sh.data_ptr() % 16 == 0))):
            # vt = (workspace, first V channel, head dim): the stacked q | k | v projection of an attention block also stores the transposed V image
            # of the LDS-DMA attention kernel (one pack launch less per block); the caller finds `_gm_vt_packed` on the result
            vws, vt_c0, vt_dh = vt
            _timed(f"token_gemm<{str(dtype).split('.')[-1]}>", dict(flops=2.0 * rows * cin * cout, bytes=float(x.element_size() * (rows * (cin + cout) + cin * cout)),
                                                                  shape=f"{rows}x{cin}->{cout} (+V^T image)"),
                   lambda: check(lib().gm_linear_rows_affine_vt(x.data_ptr(), arena_ld(x), _ptr(sc), _ptr(sh), 0 if sc is None else sc.stride(0), rows // n,
                                                                panel().data_ptr(), _ptr(b32), out.data_ptr(), arena_ld(out), rows, cin, cout, ACT[pre_act],
                                                                vws.data_ptr(), int(vt_c0), int(vt_dh), dt_code(dtype), _stream()), "gm_linear_rows_affine_vt"))
            out._gm_vt_packed = True
            return out
        if sc is None or (sc.stride(0) % 4 == 0 and sc.data_ptr() % 16 == 0 and sh.data_ptr() % 16 == 0):
            _timed(f"token_gemm<{str(dtype).split('.')[-1]}>", dict(flops=2.0 * rows * cin * cout, bytes=float(x.element_size() * (rows * (cin + cout) + cin * cout)),
                                                                  shape=f"{rows}x{cin}->{cout}"),
                   lambda: check(lib().gm_linear_rows_affine(x.data_ptr(), arena_ld(x), _ptr(sc), _ptr(sh), 0 if sc is None else sc.stride(0), rows // n,
                                                             panel().data_ptr(), _ptr(b32), _ptr(res), 0 if res is None else arena_ld(res), out.data_ptr(),
                                                             arena_ld(out), rows, cin, cout, ACT[pre_act], POST_ACT[post_act], dt_code(dtype), _stream()),
                                 "gm_linear_rows_affine"))
            return out
    if (upsample and x2 is None and SUBPIXEL_UPSAMPLE and allow_subpixel and nsp == 3 and k == (3, 3, 3) and s == (1, 1, 1) and plo == (1, 1, 1) and phi == (1, 1, 1)
            and dil == (1, 1, 1) and pre is None and pre_act == "none" and skip is None and force_cfg is None and weight is not None
            and cin % (64 // x.element_size()) == 0 and cout % vecw == 0 and arena_ld(x) % vecw == 0 and x.data_ptr() % 16 == 0
            and math.prod(src) * n >= DMA_CONV_MIN_VOXELS):
        got = _conv_subpixel(x, weight, bias, rowvec, res, post_act, out, want_stats, n, cin, cout, src)
        if got is not None:
            return got
    if (transposed and TRANSPOSED_S2_SUBPIXEL and x2 is None and nsp == 3 and s == (2, 2, 2) and k in ((3, 3, 3), (4, 4, 4)) and dil == (1, 1, 1)
            and plo[0] == plo[1] == plo[2] and stride2_subpixel_covers(k[0], plo[0]) and pre is None and pre_act == "none" and skip is None and force_cfg is None
            and weight is not None and all((src[i] - 1) * 2 - plo[i] - phi[i] + k[i] + opad[i] == 2 * src[i] for i in range(3))
            and cin % (64 // x.element_size()) == 0 and cout % vecw == 0 and arena_ld(x) % vecw == 0 and x.data_ptr() % 16 == 0
            and math.prod(src) * n >= DMA_CONV_MIN_VOXELS):
        # nn.ConvTranspose3d(stride 2) whose output is exactly twice the input (VQ-VAE up-sampling k = 4 / pad 1, the AutoencoderKL's optional
        # ConvTranspose k = 3 / pad 1 / output_padding 1, the data gradient of the Downsample convolutions): 8 sub-pixel 2x2x2 convolutions,
        # one launch of configuration 17 -- no zero-inserted operand, no gather
        got = _conv_subpixel(x, weight, bias, rowvec, res, post_act, out, want_stats, n, cin, cout, src,
                             packed=packed_stride2_dgrad_weight(weight, dtype, plo[0]),
                             label=f"transposed k{k[0]} s2 as 8 sub-pixel 2x2x2 convolutions")
        if got is not None:
            return got
    d = GmConvDesc()
    d.in_mode, d.fd, d.fh, d.fw = 0, 1, 1, 1
    act_axes = (0,) * (3 - nsp) + (1,) * nsp
    if transposed:
        if upsample:
            raise ValueError("upsample and transposed are exclusive")
        d.in_mode = 2
        d.fd, d.fh, d.fw = s
        if s == (1, 1, 1):
            d.in_mode = 0  # no zero insertion: a plain convolution with the flipped, in/out-swapped panel (every fast kernel applies --
            # this is the data gradient of the stride-1 convolutions)
        out_sp = tuple((src[i] - 1) * s[i] - plo[i] - phi[i] + dil[i] * (k[i] - 1) + opad[i] + 1 for i in range(3))
        conv_pad = tuple(dil[i] * (k[i] - 1) - plo[i] for i in range(3))
        conv_stride = (1, 1, 1)
    else:
        virt = src
        if upsample:
            d.in_mode = 1
            d.fd, d.fh, d.fw = tuple(2 if a else 1 for a in act_axes)
            virt = tuple(src[i] * (2 if act_axes[i] else 1) for i in range(3))
        out_sp = tuple((virt[i] + plo[i] + phi[i] - dil[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(3))
        conv_pad, conv_stride = plo, s
    if min(out_sp) <= 0:
        raise ValueError(f"convolution output would be empty: {out_sp}")
    out_shape = (n, *out_sp[3 - nsp:], cout)
    if out is None:
        out = torch.empty(out_shape, dtype=dtype, device=x.device)
    elif tuple(out.shape) != out_shape or out.dtype != dtype:
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {out_shape}")
    d.x, d.x_ld = x.data_ptr(), arena_ld(x)
    if x2 is not None:
        d.x2, d.x2_ld, d.cin_split = x2.data_ptr(), arena_ld(x2), x.shape[-1]
    packed = panel()
    d.w = packed.data_ptr()
    b32 = as_f32(bias) if bias is not None else None
    d.bias = _ptr(b32)
    recipe = None
    if isinstance(pre, GnRecipe) and pre._done is None and NARROW_N and sum(pre.cs) == cin and pre.n == n and k[-2:] == (3, 3) and s == (1, 1, 1) and not transposed \
            and not upsample and (x2 is None or pre.cs[0] == x.shape[-1]):
        # a GroupNorm whose finalisation has not run: cfg 24 / 25 can do it in their prologue.  The configuration is chosen as if (scale, shift) existed (a
        # readable, aligned stand-in); the tables go into the descriptor once it is 24 / 25, anything else materialises them (below)
        recipe = pre
        d.pre_scale = d.pre_shift = pre.stats[0].data_ptr()
    elif pre is not None:
        sc, sh = pre
        require_device(sc, sh)
        if tuple(sc.shape) != (n, cin) or sc.dtype != torch.float32:
            raise ValueError("pre scale/shift must be fp32 [N, Cin]")
        d.pre_scale, d.pre_shift = sc.data_ptr(), sh.data_ptr()
    else:
        d.pre_scale = d.pre_shift = None
    if rowvec is not None:
        if rowvec.dtype != torch.float32 or rowvec.dim() != 2 or rowvec.shape[1] != cout or rowvec.shape[0] not in (1, n):
            raise ValueError(f"rowvec must be fp32 [1 or N, Cout], got {tuple(rowvec.shape)} {rowvec.dtype}")
        if rowvec.stride(1) != 1:
            raise ValueError("rowvec must have unit stride on its last dim")
        d.rowvec = rowvec.data_ptr()
        d.rowvec_bstride = 0 if rowvec.shape[0] == 1 else rowvec.stride(0)
    else:
        d.rowvec, d.rowvec_bstride = None, 0
    if res is not None:
        if tuple(res.shape) != out_shape or res.dtype != dtype:
            raise ValueError(f"residual has shape {tuple(res.shape)} / {res.dtype}, expected {out_shape} / {dtype}")
        d.res, d.res_ld = res.data_ptr(), arena_ld(res)
    else:
        d.res, d.res_ld = None, 0
    d.y, d.y_ld = out.data_ptr(), arena_ld(out)
    d.skip_w = d.skip_bias = None
    d.skip_x[0] = d.skip_x[1] = None
    skip_keep = None
    if skip is not None:
        parts, sw, sb = skip
        parts = list(parts)
        require_device(sw, sb, *parts)
        if res is not None:
            raise ValueError("skip and res are exclusive (the shortcut IS the residual)")
        if not 1 <= len(parts) <= 2 or sw.shape[0] != cout or sw.shape[1] != sum(t.shape[-1] for t in parts) or math.prod(sw.shape[2:]) != 1:
            raise ValueError("skip = (1 or 2 parts, [Cout, sum(C_part), 1...] weight, bias)")
        for i, t in enumerate(parts):
            if tuple(t.shape[:-1]) != out_shape[:-1] or t.dtype != dtype:
                raise ValueError("skip parts must have the output geometry and dtype")
            d.skip_x[i], d.skip_ld[i], d.skip_cin[i] = t.data_ptr(), arena_ld(t), t.shape[-1]
        skip_keep = (packed_conv_weight(sw, dtype), as_f32(sb) if sb is not None else None)
        d.skip_w, d.skip_bias = skip_keep[0].data_ptr(), _ptr(skip_keep[1])
    d.N, d.Cin, d.Cout = n, cin, cout
    d.Ds, d.Hs, d.Ws = src
    d.Do, d.Ho, d.Wo = out_sp
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = conv_stride
    d.pd, d.ph, d.pw = conv_pad
    d.dd, d.dh, d.dw = dil
    d.pre_act, d.post_act, d.dtype = ACT[pre_act], POST_ACT[post_act], dt_code(dtype)
    d.debug_flags = _CONV_DEBUG_FLAGS
    nvox = math.prod(out_sp)
    # ---- which kernel, and how the prologue is applied ------------------------------------------------------------------------------
    # (1) an LDS-DMA configuration when it covers the geometry: fused shortcut, in-LDS prologue, both parts of a VirtualCat read in place;
    # (2) else the operand is first reduced to ONE activated tensor where needed (two_pass: gn_apply passes, 1x1 shortcut launches), and
    # (3) the register-staged / generic kernels take what is left, with their own fused prologue for small tensors.
    dma_ok = False
    fuse_in_lds = pre is None or DMA_FUSED_PROLOGUE in (True, "always", "1") or (
        DMA_FUSED_PROLOGUE == "auto" and 2.0 * n * nvox * cout * cin * math.prod(k) <= DMA_FUSED_PROLOGUE_MAX_FLOP)
    if force_cfg is not None and force_cfg in DMA_CFGS or (force_cfg is None and DMA_CONV and cout > 16 and nvox * n >= DMA_CONV_MIN_VOXELS
                                                            and fuse_in_lds):
        try:
            _choose_conv_cfg(d, nvox, force_cfg, only=DMA_CFGS)
            dma_ok = True
        except ValueError:
            dma_ok = False
    if NARROW_N and force_cfg is None and ksplit is None and DMA_CONV and fuse_in_lds and nvox * n >= DMA_CONV_MIN_VOXELS:
        bk = 64 // x.element_size()
        if dma_ok and d.cfg == 11 and SPLITK:
            tiles = n * ((out_sp[0] + 3) // 4) * ((out_sp[1] + 3) // 4) * ((out_sp[2] + 15) // 16) * ((cout + 63) // 64)
            deep = cin // bk + (0 if skip is None else sum(t.shape[-1] for t in skip[0]) // bk)
            if tiles < SPLITK_MAX_TILES and min(cin // bk, SPLITK_MAX, SPLITK_TARGET_WGS // tiles) > 1 and deep <= NARROW_N_MAX_CHUNKS:
                try:  # the launch would be split over K: K-complete on 16-channel blocks instead (cfg 24)
                    _choose_conv_cfg(d, nvox, 24, only=DMA_CFGS)
                except ValueError:
                    _choose_conv_cfg(d, nvox, 11, only=DMA_CFGS)
        elif not dma_ok and cout <= 16 and (d.kd == 3 or NARROW_N_2D) and 2.0 * n * nvox * cout * cin * math.prod(k) <= NARROW_HEAD_MAX_FLOP:
            # a narrow output head of a SMALL problem (the latent UNet's 64 -> 4, the 2-D UNet's 32 -> 1): the generic tile kernel otherwise (42 us at 32^3); the
            # C_out = 1 head of a large volume keeps its depth-marching kernel (cfg 20)
            try:
                _choose_conv_cfg(d, nvox, 24 if d.kd == 3 else 25, only=DMA_CFGS)
                dma_ok = True
            except ValueError:
                pass
    def recipe_into_descriptor():
        """The pending GroupNorm as statistic tables in the descriptor; False (descriptor restored) when the kernel does not take that form after all."""
        keep = (d.pre_scale, d.pre_shift)
        d.pre_scale = d.pre_shift = None
        for i, st in enumerate(recipe.stats):
            d.pre_stats[i], d.pre_S[i], d.pre_C[i] = st.data_ptr(), st.shape[0], recipe.cs[i]
        d.pre_gamma, d.pre_beta, d.pre_eps, d.pre_groups = _ptr(recipe.gamma), _ptr(recipe.beta), recipe.eps, recipe.groups
        if lib().gm_conv_lds_bytes(C.byref(d)) > 0:
            return True
        d.pre_stats[0] = d.pre_stats[1] = None  # (a table longer than the short form's bound, more channels than the LDS tables hold)
        d.pre_S[0] = d.pre_S[1] = d.pre_C[0] = d.pre_C[1] = 0
        d.pre_scale, d.pre_shift = keep
        return False

    recipe_pending_split = False
    if recipe is not None:
        if dma_ok and d.cfg in (24, 25):
            if not recipe_into_descriptor():
                recipe = None
        elif dma_ok and d.cfg == 11 and GN_IN_SPLIT_SLICES and (ksplit is not None or SPLITK):
            recipe_pending_split = True  # decided with the split (below): the K slices of conv_sk.hip finalise the norm themselves
        else:
            recipe = None
        if recipe is None:
            sc, sh = pre.materialise()
            d.pre_scale, d.pre_shift = sc.data_ptr(), sh.data_ptr()
    if not dma_ok:
        if force_cfg is not None and force_cfg in DMA_CFGS:
            raise ValueError(f"configuration {force_cfg} does not cover this convolution")
        if x2 is not None:
            return two_pass()  # only the LDS-DMA kernels read a concatenated input in place
        # (round 5: NOT the large 1x1 projections -- the q | k | v projection of C2's mid-block attention, 32 768 x 256 -> 768, is 102 us fused against
        #  62 + 12 two-pass in isolation, and 0.109 against 0.081 + 0.035 inside the forward: nothing, profiles/r05_qkv_projection_c2.txt)
        if pre is not None and force_cfg is None and cout > 16 and math.prod(k) > 1 and not fuse_gn_prologue(x):
            return two_pass()  # a large ResnetBlock convolution: the HBM-bound gm_gn_apply pass + the prologue-free kernel (which still fuses
            # the shortcut) beats both the in-LDS and the register-staged prologue (GN_APPLY_POLICY, DMA_FUSED_PROLOGUE)
        if skip is not None:
            # the shortcut as 1x1 launches over the parts first, added as the residual (`res` is None: the shortcut IS the residual)
            parts, sw, sb = skip
            parts = list(parts)
            acc_t, off = None, 0
            for t in parts:
                c_t = t.shape[-1]
                acc_t = conv(t, sw, sb if acc_t is None else None, kernel=1,
                             packed=packed_conv_weight(sw, dtype, cin_range=(off, off + c_t)), cout=cout, res=acc_t)
                off += c_t
            return conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, dilation=dilation, pad_hi=pad_hi,
                        upsample=upsample, transposed=transposed, output_padding=output_padding, pre=pre, pre_act=pre_act,
                        rowvec=rowvec, res=acc_t, post_act=post_act, out=out, packed=packed, cout=cout, force_cfg=force_cfg,
                        want_stats=want_stats)
        no_cin = (12,) if (weight is None or transposed or tuple(weight.shape[2:]) != (3, 3, 3)) else ()  # (cfg 12 derives its image from the weight)
        _choose_conv_cfg(d, nvox, force_cfg, exclude=(DMA_CFGS + no_cin) if force_cfg is None else ())
    if d.cfg == 12:  # configuration 12 reads its weight fragments from the K-major image
        if weight is None or transposed or tuple(weight.shape[2:]) != (3, 3, 3):
            raise ValueError("configuration 12 needs the original [Cout, Cin, 3, 3, 3] weight (its K-major image is derived from it)")
        cin_keep = packed_cin_weight(weight, dtype)
        d.w = cin_keep.data_ptr()
    kpart = None
    if dma_ok and d.cfg == 11 and (ksplit is not None or SPLITK):
        nchunks = cin // (64 // x.element_size())
        tiles = n * ((out_sp[0] + 3) // 4) * ((out_sp[1] + 3) // 4) * ((out_sp[2] + 15) // 16) * ((cout + 63) // 64)
        ks = ksplit if ksplit is not None else (min(nchunks, SPLITK_MAX, SPLITK_TARGET_WGS // tiles) if tiles < SPLITK_MAX_TILES else 1)
        if ks > 1:
            ks = -(-nchunks // -(-nchunks // ks))  # ceil(nchunks / chunks per slice): 12 chunks over "8" slices = 6 slices of 2, none empty
        if ks > 1:
            if _SK_KERNEL is not None:
                lib().gm_conv_sk_set_enabled(int(_SK_KERNEL))
            d.ksplit = int(ks)
            nbytes = lib().gm_conv_splitk_workspace_bytes(C.byref(d))
            if nbytes > 0:
                stamp_room = (1 << 16) * 4 if (_CONV_DEBUG_FLAGS & 4096) else 0  # tools/sk_timeline.py: 16 int64 stamps per work-group behind the partials
                kpart = torch.zeros(nbytes // 4 + stamp_room, dtype=torch.float32, device=x.device) if stamp_room else \
                    torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)  # stream-ordered scratch: freed on return
                d.kpartial = kpart.data_ptr()
            elif ksplit is not None:
                raise ValueError(f"split-K by {ks} is not available for this convolution")
            else:
                d.ksplit = 0
    if recipe_pending_split:
        if not (d.ksplit > 1 and kpart is not None and recipe_into_descriptor()):
            sc, sh = pre.materialise()
            d.pre_scale, d.pre_shift = sc.data_ptr(), sh.data_ptr()
    if _CONV_TIMELINE_BUFFER is not None and kpart is None:
        d.kpartial = _CONV_TIMELINE_BUFFER.data_ptr()  # ksplit stays 0: the kernel only stamps into it (debug_flags bit 12)
    d.stats = None
    if want_stats:  # the fast kernels fuse the output statistics into their epilogue (else: one stand-alone pass when a consumer asks)
        _attach_conv_stats(d, out, n, cout)
    if _PROFILE is None:
        check(lib().gm_conv_forward(C.byref(d), _stream()), "gm_conv_forward")
    else:
        es = x.element_size()
        taps = k[0] * k[1] * k[2]
        nvo = n * math.prod(out_sp)
        scin = 0 if skip is None else sum(t.shape[-1] for t in skip[0])
        meta = dict(flops=2.0 * nvo * cout * (cin * taps + scin),
                    bytes=float(es * (n * math.prod(src) * cin + nvo * (cout * (2 if res is not None else 1) + scin) + cout * (cin * taps + scin))),
                    shape=f"{cin}->{cout} k{k} s{conv_stride} out{tuple(out_sp)} mode{d.in_mode}")
        _timed(f"conv_igemm<{str(dtype).split('.')[-1]},cfg{d.cfg}{'' if d.ksplit <= 1 else 'k'}>", meta,
               lambda: check(lib().gm_conv_forward(C.byref(d), _stream()), "gm_conv_forward"))
    if d.stats:
        out._gm_cstats = _compact_stats(out._gm_cstats)
    if kpart is not None and (_CONV_DEBUG_FLAGS & 4096):
        global _SK_STAMPS
        _SK_STAMPS = kpart[kpart.numel() - (1 << 18):].view(torch.int64).reshape(-1, 16).clone()
    return out


def linear(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], **kw) -> torch.Tensor:
    """nn.Linear over the last dim of x = (N, L, C) or (rows, C); same fusion hooks as conv."""
    if x.dim() == 3:
        return conv(x, weight, bias, kernel=1, **kw)
    if x.dim() == 2:
        for key in ("res", "out"):
            if kw.get(key) is not None:
                kw[key] = kw[key].unsqueeze(0)
        return conv(x.unsqueeze(0), weight, bias, kernel=1, **kw).squeeze(0)
    raise ValueError("linear expects (rows, C) or (N, L, C)")


# ------------------------------------------------------------------------------------------------------------------------
# backward kernels (wrapped as torch.autograd.Functions in generativemodels_amd/autograd.py)
# ------------------------------------------------------------------------------------------------------------------------
def conv_wgrad(x: torch.Tensor, gy: torch.Tensor, kernel, stride=1, padding=0, out: Optional[torch.Tensor] = None,
               accumulate: bool = False) -> torch.Tensor:
    """Weight gradient of y = conv(x, W) (torch.nn.grad.conv*_weight): x, gy arena tensors (N, *spatial, C); returns fp32
    [Cout, Cin, *kernel].  kernel 1 or 3 (the same on every spatial axis), stride 1 or 2, `padding` = low-side pad (the high side
    is implied by gy's extents)."""
    require_device(x, gy, out)
    nsp = x.dim() - 2
    if nsp < 1 or nsp > 3 or gy.dim() != x.dim() or gy.dtype != x.dtype or gy.shape[0] != x.shape[0]:
        raise ValueError("conv_wgrad expects matching (N, *spatial, C) arena tensors")

    def tup(v):
        v = tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * nsp
        if len(v) != nsp:
            raise ValueError("per-axis argument has the wrong length")
        return v

    k, s_, p_ = tup(kernel), tup(stride), tup(padding)
    if len(set(k)) != 1 or len(set(s_)) != 1:
        raise ValueError("conv_wgrad: kernel and stride must be the same on every axis")
    if k[0] == 4 and s_[0] == 2 and nsp in (2, 3) and all(0 <= v <= 2 for v in p_):
        return _conv_wgrad_k4s2(x, gy, p_, out, accumulate)
    vec = 16 // x.element_size()
    if x.shape[-1] % vec or gy.shape[-1] % vec or arena_ld(x) % vec or arena_ld(gy) % vec or x.data_ptr() % 16 or gy.data_ptr() % 16:
        # ragged channel counts (the 1-channel input / output convolutions): zero-pad the channels to one 16-byte vector
        def padded(t):
            cp = (t.shape[-1] + vec - 1) // vec * vec
            z = torch.zeros((*t.shape[:-1], cp), dtype=t.dtype, device=t.device)
            copy_channels(t, z[..., :t.shape[-1]])
            return z
        if accumulate and out is None:
            raise ValueError("accumulate needs an existing gradient tensor")
        full = conv_wgrad(padded(x), padded(gy), kernel, stride, padding)
        res = full[:gy.shape[-1], :x.shape[-1]].contiguous()
        if out is None:
            return res
        if tuple(out.shape) != tuple(res.shape) or out.dtype != torch.float32:
            raise ValueError("conv_wgrad: out must be a contiguous fp32 [Cout, Cin, *kernel] tensor")
        with torch.no_grad():  # (the two edge convolutions of a network: a few thousand elements)
            out.add_(res) if accumulate else out.copy_(res)
        return out
    cin, cout = x.shape[-1], gy.shape[-1]
    d = GmWgradDesc()
    d.x, d.x_ld, d.gy, d.gy_ld = x.data_ptr(), arena_ld(x), gy.data_ptr(), arena_ld(gy)
    d.N, d.Cin, d.Cout = x.shape[0], cin, cout
    src = (1,) * (3 - nsp) + tuple(x.shape[1:-1])
    dst = (1,) * (3 - nsp) + tuple(gy.shape[1:-1])
    d.Ds, d.Hs, d.Ws = src
    d.Do, d.Ho, d.Wo = dst
    kk = (1,) * (3 - nsp) + k
    if nsp == 1 and k[0] != 1:
        raise ValueError("conv_wgrad: 1-D convolutions are covered for kernel 1 only")
    d.kd, d.kh, d.kw = kk
    d.stride = s_[0]
    d.pd, d.ph, d.pw = (0,) * (3 - nsp) + p_
    d.dtype, d.accumulate = dt_code(x.dtype), int(accumulate)
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing gradient tensor")
        out = torch.empty((cout, cin, *k), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (cout, cin, *k) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("conv_wgrad: out must be a contiguous fp32 [Cout, Cin, *kernel] tensor")
    d.dw = out.data_ptr()
    ws_bytes = lib().gm_conv_wgrad_workspace_bytes(C.byref(d))
    if ws_bytes < 0:
        raise ValueError("conv_wgrad: geometry not covered (kernel 1 or 3, stride 1 or 2, channel counts multiples of a 16-byte vector)")
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
    taps = math.prod(k)
    nvo = gy.shape[0] * math.prod(gy.shape[1:-1])
    _timed(f"conv_wgrad<{str(x.dtype).split('.')[-1]}>", dict(flops=2.0 * nvo * cin * cout * taps, bytes=float(x.element_size() * (x.numel() + gy.numel())),
                                                          shape=f"{cin}->{cout} k{k} s{s_} out{tuple(gy.shape[1:-1])}"),
           lambda: check(lib().gm_conv_wgrad(C.byref(d), _stream()), "gm_conv_wgrad"))
    return out


def stride2_phase_taps(pad: int) -> dict:
    """Kernel 4, stride 2, low padding `pad` (0..2), per axis: {tap parity r: (input phase rho, tap index t of a 3-tap / padding-1 stride-1
    weight gradient over that phase image that holds kernel tap k = r, the one that holds k = r + 2)}.  y[o] = sum_k W[k] x[2 o + k - pad]; with
    k = 2 j + r the input index is 2 (o + j + sigma) + rho, rho = (r - pad) mod 2, sigma = (r - pad - rho) / 2 in {-1, 0}: tap j of phase image
    rho at offset o + j + sigma = tap t = j + sigma + 1 of a 3-tap stride-1 stencil around o.  Pure host logic (tested on CPU)."""
    out = {}
    for r in (0, 1):
        e = r - pad
        rho = e % 2
        sigma = (e - rho) // 2
        assert sigma in (-1, 0)
        out[r] = (rho, sigma + 1, sigma + 2)
    return out


def _conv_wgrad_k4s2(x, gy, pads, out, accumulate):
    """Weight gradient of a k = 4 / stride-2 convolution (the VQ-VAE down-sampling convolutions and, with the operands exchanged, its
    ConvTranspose up-sampling: vqvae.py:127-150,244-261) from the MFMA weight-gradient kernel of 3-tap stride-1 convolutions: per tap-parity
    class r (2^d of them) ONE launch over the phase image x[rho::2] (gm_phase2x) yields the 2^d taps k = 2 j + r of that class as a sub-block of
    its 3^d result (stride2_phase_taps).  27/8 of the minimal multiply-adds on a 600-700 TFLOP/s kernel; fixed-order sums (deterministic)."""
    nsp = x.dim() - 2
    cin, cout = x.shape[-1], gy.shape[-1]
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing gradient tensor")
        res = torch.empty((cout, cin) + (4,) * nsp, dtype=torch.float32, device=x.device)
    else:
        if tuple(out.shape) != (cout, cin) + (4,) * nsp or out.dtype != torch.float32:
            raise ValueError("conv_wgrad: out must be a contiguous fp32 [Cout, Cin, *kernel] tensor")
        res = out
    taps = [stride2_phase_taps(p) for p in pads]
    with torch.no_grad():
        for cls in range(1 << nsp):
            r = [(cls >> (nsp - 1 - a)) & 1 for a in range(nsp)]
            xp = phase2x(x, [taps[a][r[a]][0] for a in range(nsp)])
            g3 = conv_wgrad(xp, gy, 3, 1, 1)  # [Cout, Cin, 3 (x nsp)]
            src = g3[(slice(None), slice(None)) + tuple(slice(taps[a][r[a]][1], taps[a][r[a]][2] + 1) for a in range(nsp))]
            dst = res[(slice(None), slice(None)) + tuple(slice(r[a], 4, 2) for a in range(nsp))]
            # (the 2^d strided sub-blocks of a [Cout, Cin, 4, 4(, 4)] tensor: a scatter of the kernel's outputs into the parameter layout)
            dst.add_(src) if (accumulate and out is not None) else dst.copy_(src)
    return res


def act_backward(y: torch.Tensor, gy: torch.Tensor, act: str) -> torch.Tensor:
    """gy * act'(z) from the activation's OUTPUT y = act(z) for activations whose derivative is a function of the output sign: ReLU
    (y > 0 <=> z > 0).  The fused convolution epilogues store only y; this is the first step of their backward."""
    if act not in ("relu", "leakyrelu"):
        raise NotImplementedError(f"activation '{act}' has no backward kernel (training covers none / relu / leakyrelu epilogues)")
    require_device(y, gy)
    if y.shape != gy.shape or y.dtype != gy.dtype:
        raise ValueError("act_backward: gy must match y")
    y, gy = y.contiguous(), gy.contiguous()
    n, c = y.shape[0], y.shape[-1]
    v = rows_of(y) // max(n, 1)
    one = torch.ones((n, c), dtype=torch.float32, device=y.device)
    zero = torch.zeros((n, c), dtype=torch.float32, device=y.device)
    dx = torch.empty_like(y)
    check(lib().gm_gn_bwd_apply(y.data_ptr(), arena_ld(y), gy.data_ptr(), arena_ld(gy), dx.data_ptr(), arena_ld(dx), one.data_ptr(), zero.data_ptr(), c,
                                one.data_ptr(), zero.data_ptr(), zero.data_ptr(), n, v, c, ACT_BWD[act], dt_code(y.dtype), _stream()), "gm_gn_bwd_apply")
    return dx


def spade_backward(xn: torch.Tensor, g: torch.Tensor, bm: torch.Tensor, gy: torch.Tensor, act: str):
    """Backward of y = act(xn * g + bm) (ops.spade_apply with an identity affine): -> (dxn, dg, dbm), dg / dbm channel slices of one buffer."""
    require_device(xn, g, bm, gy)
    if not (xn.shape == g.shape == bm.shape == gy.shape) or len({xn.dtype, g.dtype, bm.dtype, gy.dtype}) != 1 or arena_ld(g) != arena_ld(bm):
        raise ValueError("spade_backward operands must match in shape / dtype (g and bm share a row pitch)")
    gy = gy.contiguous()
    c = xn.shape[-1]
    dxn = torch.empty(xn.shape, dtype=xn.dtype, device=xn.device)
    dgb = torch.empty((*xn.shape[:-1], 2 * c), dtype=xn.dtype, device=xn.device)
    dg, dbm = dgb[..., :c], dgb[..., c:]
    check(lib().gm_spade_bwd(xn.data_ptr(), arena_ld(xn), g.data_ptr(), bm.data_ptr(), arena_ld(g), gy.data_ptr(), arena_ld(gy), dxn.data_ptr(),
                             arena_ld(dxn), dg.data_ptr(), dbm.data_ptr(), arena_ld(dg), rows_of(xn), c, ACT_BWD[act], dt_code(xn.dtype), _stream()),
          "gm_spade_bwd")
    return dxn, dg, dbm


def bias_grad(gy: torch.Tensor, per_sample: bool = False) -> torch.Tensor:
    """Column sums of an arena tensor over its voxels: fp32 [C] (a bias gradient) or, per sample, [N, C] (the gradient of the
    per-sample row vector a convolution adds in its epilogue)."""
    require_device(gy)
    n, c = gy.shape[0], gy.shape[-1]
    # Fresh statistics, never the tensor's cached table: a gradient tensor can be accumulated IN PLACE by the autograd engine after a first
    # consumer attached its column sums to the Python object (dres = gy hands the same object to two branches) -- measured as 12-24 %
    # errors in exactly the bias gradients behind an identity residual.
    st = _fresh_channel_stats(gy)
    out = torch.empty((n, c) if per_sample else (c,), dtype=torch.float32, device=gy.device)
    check(lib().gm_stats_colsum(st.data_ptr(), st.shape[0], n, c, out.data_ptr(), int(per_sample), _stream()), "gm_stats_colsum")
    return out


def gn_backward(x: torch.Tensor, gy: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, gamma: Optional[torch.Tensor],
                groups: int, eps: float, act: str = "none", want_affine_grads: bool = True):
    """Backward of y = act(GroupNorm(x)) given the forward (scale, shift) fp32 [N, C] tables: -> (dx, dgamma, dbeta) with the
    parameter gradients fp32 [C] (None when not requested)."""
    require_device(x, gy, scale, shift, gamma)
    if gy.shape != x.shape or gy.dtype != x.dtype:
        raise ValueError("gn_backward: gy must match x")
    n, c = x.shape[0], x.shape[-1]
    v = rows_of(x) // max(n, 1)
    if tuple(scale.shape) != (n, c) or tuple(shift.shape) != (n, c) or scale.stride(1) != 1 or scale.stride(0) != shift.stride(0):
        raise ValueError("gn_backward: scale/shift must be matching fp32 [N, C] tables")
    ss_ld = scale.stride(0) if n > 1 else max(scale.stride(0), c)
    fwd = channel_stats(x)
    # one stored fp64 partial per (block, sample, channel): no atomics, nothing to zero, summed in a fixed order by the finalize kernel
    bwd = torch.empty((max(int(lib().gm_gn_bwd_stats_slots(n, v)), 1), n, c, 2), dtype=torch.float64, device=x.device)
    a = ACT[act]
    check(lib().gm_gn_bwd_stats(x.data_ptr(), arena_ld(x), gy.data_ptr(), arena_ld(gy), scale.data_ptr(), shift.data_ptr(), ss_ld, n, v, c, a,
                                bwd.data_ptr(), dt_code(x.dtype), _stream()), "gm_gn_bwd_stats")
    coef = torch.empty((3, n, c), dtype=torch.float32, device=x.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine_grads else None
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine_grads else None
    check(lib().gm_gn_bwd_finalize(fwd.data_ptr(), fwd.shape[0], bwd.data_ptr(), bwd.shape[0], n, c, groups, v, float(eps), _ptr(as_f32(gamma)), coef[0].data_ptr(),
                                   coef[1].data_ptr(), coef[2].data_ptr(), _ptr(dgamma), _ptr(dbeta), _stream()), "gm_gn_bwd_finalize")
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(lib().gm_gn_bwd_apply(x.data_ptr(), arena_ld(x), gy.data_ptr(), arena_ld(gy), dx.data_ptr(), arena_ld(dx), scale.data_ptr(),
                                shift.data_ptr(), ss_ld, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), n, v, c, a,
                                dt_code(x.dtype), _stream()), "gm_gn_bwd_apply")
    return dx, dgamma, dbeta


ATTENTION_BWD_HEAD_DIMS = (16, 32, 64, 128, 256)


def attention_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, go: torch.Tensor, heads: int, scale: float):
    """(dq, dk, dv) of o = softmax(scale q k^T) v per (batch, head) by the fused flash backward (gm_attention_backward): the scores are
    recomputed tile by tile, nothing L x L is stored.  (B, L, heads * dh) operands (channel slices allowed), dh in ATTENTION_BWD_HEAD_DIMS."""
    require_device(q, k, v, o, go)
    b, lq, c = q.shape
    lk = k.shape[1]
    dh = c // heads
    if dh not in ATTENTION_BWD_HEAD_DIMS:
        raise ValueError(f"attention_backward: head dim {dh} not in {ATTENTION_BWD_HEAD_DIMS}")
    if o.shape != q.shape or go.shape != q.shape or k.shape != v.shape or k.shape[2] != c:
        raise ValueError("attention_backward operand shapes are inconsistent")
    dq, dk, dv = torch.empty_like(q.contiguous()), torch.empty_like(k.contiguous()), torch.empty_like(v.contiguous())
    d = GmAttnBwdDesc()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o), ("go", go), ("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(d, name, t.data_ptr())
        setattr(d, name + "_ld", _kv_ld(t))
    for t, l in ((q, lq), (o, lq), (go, lq), (k, lk), (v, lk)):
        if t.shape[0] > 1 and t.stride(0) != l * _kv_ld(t):
            raise ValueError("attention_backward operands must be batch-dense")
    d.B, d.H, d.Lq, d.Lk, d.dh = b, heads, lq, lk, dh
    d.scale, d.dtype = float(scale), dt_code(q.dtype)
    nbytes = lib().gm_attention_backward_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    _timed(f"attention_bwd<{str(q.dtype).split('.')[-1]}>", dict(flops=14.0 * b * heads * lq * lk * dh, bytes=float(5 * q.element_size() * q.numel()),
                                                              shape=f"B{b} H{heads} L{lq}x{lk} d{dh}"),
           lambda: check(lib().gm_attention_backward(C.byref(d), _stream()), "gm_attention_backward"))
    return dq, dk, dv


ATTENTION_BWD_FUSED_HEAD_DIMS = (64, 128, 256)


def attention_backward_fused(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, go: torch.Tensor, heads: int, scale: float,
                             lse: Optional[torch.Tensor] = None, or_none: bool = False):
    """(dq, dk, dv) of o = softmax(scale q k^T) v by the fused bf16 flash backward on the LDS-DMA structure (gm_attention_backward_fused,
    csrc/attention_bwd_dma.hip): scores recomputed per tile on bf16 MFMA, nothing L x L in HBM, deterministic.  bf16 (B, L, heads * dh) operands,
    dh in ATTENTION_BWD_FUSED_HEAD_DIMS; lse: optional fp32 (B, heads, Lq) log-sum-exp of the scaled scores (else one more sweep computes it).
    (reference: torch autograd through diffusion_model_unet.py:407-415)"""
    require_device(q, k, v, o, go)
    b, lq, c = q.shape
    lk = k.shape[1]
    dh = c // heads
    if q.dtype != torch.bfloat16 or dh not in ATTENTION_BWD_FUSED_HEAD_DIMS:
        raise ValueError(f"attention_backward_fused: bf16 operands with a head dim in {ATTENTION_BWD_FUSED_HEAD_DIMS}")
    if o.shape != q.shape or go.shape != q.shape or k.shape != v.shape or k.shape[2] != c:
        raise ValueError("attention_backward operand shapes are inconsistent")
    q, k, v, o, go = (t.contiguous() for t in (q, k, v, o, go))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    d = GmAttnBwdDesc()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o), ("go", go), ("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(d, name, t.data_ptr())
        setattr(d, name + "_ld", _kv_ld(t))
    d.B, d.H, d.Lq, d.Lk, d.dh = b, heads, lq, lk, dh
    d.scale, d.dtype = float(scale), dt_code(q.dtype)
    nbytes = lib().gm_attention_backward_fused_workspace_bytes(C.byref(d))
    if nbytes <= 0:  # the library's own eligibility test (abd_eligible) is the one place the decision is made: callers with another path ask with or_none
        if or_none:
            return None
        raise ValueError("attention_backward_fused: operands not served by the fused bf16 kernels (alignment / size)")
    if lse is not None:
        if lse.dtype != torch.float32 or lse.numel() != b * heads * lq or not lse.is_contiguous():
            raise ValueError("attention_backward_fused: lse must be a contiguous fp32 (B, heads, Lq) tensor")
        require_device(lse)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    _timed("attention_bwd_fused<bfloat16>", dict(flops=(14.0 if lse is not None else 16.0) * b * heads * lq * lk * dh, bytes=float(5 * 2 * q.numel()),
                                                 shape=f"B{b} H{heads} L{lq}x{lk} d{dh}"),
           lambda: check(lib().gm_attention_backward_fused(C.byref(d), lse.data_ptr() if lse is not None else None, _stream()), "gm_attention_backward_fused"))
    return dq, dk, dv


ATTENTION_BWD_BF16_HEAD_DIMS = (32, 64, 128, 256)
ATTENTION_BWD_BF16_MAX_BYTES = 16 << 30  # the P and dS matrices of one call (2 x B x H x Lq x Lk x 2 bytes): 4.3 GB at 32 768 tokens, one head
# (round 5) ... and of one QUERY SLAB: a (sample, head) pair whose score matrices exceed this goes through the score pass in slabs of query rows -- every
# softmax quantity is per query row, dV / dK accumulate over the slabs in fp32 (gm_conv_wgrad's accumulate form), dQ is written slab by slab -- so one
# head of 32 768 tokens needs 0.4 GB of scratch instead of 6.4 GB, at the same FLOPs
ATTENTION_BWD_BF16_SLAB_BYTES = 512 << 20


def attention_backward_bf16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, go: torch.Tensor, heads: int, scale: float):
    """(dq, dk, dv) of o = softmax(scale q k^T) v with every product on the bf16 MFMA path (fp32 accumulation, fp32 softmax state):
    gm_attention_bwd_scores leaves P and dS as bf16 [Lq][Lk] matrices per (sample, head); dV = P^T dO and dK = dS^T Q run on the
    weight-gradient kernel (a contraction over rows: its transposed operand staging and its split over rows exist already), and so does
    dQ = dS K = (dS^T)^T K, a contraction over the keys, from the dS^T image the score pass writes as well.
    bf16 (B, L, heads * dh) operands, dh in ATTENTION_BWD_BF16_HEAD_DIMS.  (reference: torch autograd through diffusion_model_unet.py:407-415)"""
    require_device(q, k, v, o, go)
    b, lq, c = q.shape
    lk = k.shape[1]
    dh = c // heads
    if q.dtype != torch.bfloat16 or dh not in ATTENTION_BWD_BF16_HEAD_DIMS:
        raise ValueError(f"attention_backward_bf16: bf16 operands with a head dim in {ATTENTION_BWD_BF16_HEAD_DIMS}")
    if o.shape != q.shape or go.shape != q.shape or k.shape != v.shape or k.shape[2] != c:
        raise ValueError("attention_backward operand shapes are inconsistent")
    q, k, v, o, go = (t.contiguous() for t in (q, k, v, o, go))
    lkp, lqp = (lk + 63) // 64 * 64, (lq + 63) // 64 * 64
    pair_bytes = (2 * lq * lkp + lkp * lqp) * 2  # P, dS [Lq][Lk] and dS^T [Lk][Lq] of ONE (sample, head) pair
    if pair_bytes > ATTENTION_BWD_BF16_SLAB_BYTES and lq > 64:
        return _attention_backward_bf16_slabs(q, k, v, o, go, heads, scale)
    if pair_bytes > ATTENTION_BWD_BF16_MAX_BYTES:
        raise ValueError("attention_backward_bf16: the score matrices of one (sample, head) pair exceed ATTENTION_BWD_BF16_MAX_BYTES")
    # (sample, head) pairs per score pass: all of them when their matrices fit the scratch bound, else one at a time through ONE set of buffers
    # (round 4: any batch x heads at any length -- 8 heads of 32 768 tokens re-use 6.4 GB instead of asking for 51 GB)
    whole = b * heads * pair_bytes <= ATTENTION_BWD_BF16_MAX_BYTES
    npairs = b * heads if whole else 1
    probs = torch.empty((npairs, lq, lkp), dtype=torch.bfloat16, device=q.device)
    dscores = torch.empty_like(probs)
    dscores_t = torch.empty((npairs, lkp, lqp), dtype=torch.bfloat16, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    es = q.element_size()

    def score_pass(bi0, nb, hi0, nh):
        d = GmAttnBwdDesc()
        for name, t in (("q", q), ("k", k), ("v", v), ("o", o), ("go", go)):
            setattr(d, name, t.data_ptr() + (bi0 * t.shape[1] * t.shape[2] + hi0 * dh) * es)
            setattr(d, name + "_ld", _kv_ld(t))
        d.B, d.H, d.Lq, d.Lk, d.dh = nb, nh, lq, lk, dh
        d.scale, d.dtype = float(scale), dt_code(q.dtype)
        nbytes = lib().gm_attention_bwd_scores_workspace_bytes(C.byref(d))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        _timed("attention_bwd_scores<bfloat16>", dict(flops=6.0 * nb * nh * lq * lk * dh, bytes=float(6 * nb * nh * lq * lkp), shape=f"B{nb} H{nh} L{lq}x{lk} d{dh}"),
               lambda: check(lib().gm_attention_bwd_scores(C.byref(d), probs.data_ptr(), dscores.data_ptr(), lkp, dscores_t.data_ptr(), lqp, _stream()),
                             "gm_attention_bwd_scores"))

    def contractions(bi, hi, i):
        sl = slice(hi * dh, (hi + 1) * dh)
        dvh = conv_wgrad(go[bi:bi + 1, :, sl], probs[i:i + 1, :, :lk], 1, 1, 0)                # [lk, dh, 1] fp32 = P^T dO
        dkh = conv_wgrad(q[bi:bi + 1, :, sl], dscores[i:i + 1, :, :lk], 1, 1, 0)               # [lk, dh, 1] fp32 = dS^T Q
        dqh = conv_wgrad(k[bi:bi + 1, :, sl], dscores_t[i:i + 1, :lk, :lq], 1, 1, 0)           # [lq, dh, 1] fp32 = dS K (rows = keys)
        copy_channels(dqh.reshape(1, lq, dh), dq[bi:bi + 1, :, sl])
        copy_channels(dkh.reshape(1, lk, dh), dk[bi:bi + 1, :, sl])
        copy_channels(dvh.reshape(1, lk, dh), dv[bi:bi + 1, :, sl])

    if whole:
        score_pass(0, b, 0, heads)
        for bi in range(b):
            for hi in range(heads):
                contractions(bi, hi, bi * heads + hi)
    else:
        for bi in range(b):
            for hi in range(heads):
                score_pass(bi, 1, hi, 1)
                contractions(bi, hi, 0)
    return dq, dk, dv


def _attention_backward_bf16_slabs(q, k, v, o, go, heads: int, scale: float):
    """attention_backward_bf16 for (sample, head) pairs whose L x L score matrices exceed ATTENTION_BWD_BF16_SLAB_BYTES: the same kernels over slabs
    of query rows.  The score pass (LSE over all keys, P, dS, dS^T) is row-wise in the queries, so a slab is the call on a row range of q / o / dO;
    dV = sum over slabs P_s^T dO_s and dK = sum dS_s^T Q_s accumulate in fp32 in a fixed slab order (deterministic), dQ_s = dS_s K lands in its rows."""
    b, lq, c = q.shape
    lk = k.shape[1]
    dh = c // heads
    lkp = (lk + 63) // 64 * 64
    bq = max(64, int(ATTENTION_BWD_BF16_SLAB_BYTES // (6 * lkp)) // 64 * 64)  # query rows per slab: 3 matrices x 2 bytes x lkp per row
    bqp = bq
    probs = torch.empty((1, bq, lkp), dtype=torch.bfloat16, device=q.device)
    dscores = torch.empty_like(probs)
    dscores_t = torch.empty((1, lkp, bqp), dtype=torch.bfloat16, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    es = q.element_size()
    for bi in range(b):
        for hi in range(heads):
            sl = slice(hi * dh, (hi + 1) * dh)
            dv32 = torch.empty((lk, dh, 1), dtype=torch.float32, device=q.device)
            dk32 = torch.empty((lk, dh, 1), dtype=torch.float32, device=q.device)
            for si, i0 in enumerate(range(0, lq, bq)):
                rows = min(bq, lq - i0)
                d = GmAttnBwdDesc()
                for name, t in (("q", q), ("o", o), ("go", go)):
                    setattr(d, name, t.data_ptr() + ((bi * lq + i0) * c + hi * dh) * es)
                    setattr(d, name + "_ld", _kv_ld(t))
                for name, t in (("k", k), ("v", v)):
                    setattr(d, name, t.data_ptr() + (bi * lk * c + hi * dh) * es)
                    setattr(d, name + "_ld", _kv_ld(t))
                d.B, d.H, d.Lq, d.Lk, d.dh = 1, 1, rows, lk, dh
                d.scale, d.dtype = float(scale), dt_code(q.dtype)
                nbytes = lib().gm_attention_bwd_scores_workspace_bytes(C.byref(d))
                ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
                d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
                _timed("attention_bwd_scores<bfloat16>", dict(flops=6.0 * rows * lk * dh, bytes=float(6 * rows * lkp), shape=f"slab {rows}x{lk} d{dh}"),
                       lambda d=d: check(lib().gm_attention_bwd_scores(C.byref(d), probs.data_ptr(), dscores.data_ptr(), lkp, dscores_t.data_ptr(), bqp, _stream()),
                                         "gm_attention_bwd_scores"))
                qs, gs = q[bi:bi + 1, i0:i0 + rows, sl], go[bi:bi + 1, i0:i0 + rows, sl]
                conv_wgrad(gs, probs[:, :rows, :lk], 1, 1, 0, out=dv32, accumulate=si > 0)              # dV += P_s^T dO_s
                conv_wgrad(qs, dscores[:, :rows, :lk], 1, 1, 0, out=dk32, accumulate=si > 0)            # dK += dS_s^T Q_s
                dqh = conv_wgrad(k[bi:bi + 1, :, sl], dscores_t[:, :lk, :rows], 1, 1, 0)               # [rows, dh, 1] = dS_s K
                copy_channels(dqh.reshape(1, rows, dh), dq[bi:bi + 1, i0:i0 + rows, sl])
            copy_channels(dk32.reshape(1, lk, dh), dk[bi:bi + 1, :, sl])
            copy_channels(dv32.reshape(1, lk, dh), dv[bi:bi + 1, :, sl])
    return dq, dk, dv


def layernorm_backward(x: torch.Tensor, gy: torch.Tensor, gamma: Optional[torch.Tensor], eps: float, want_param_grads: bool = True):
    """nn.LayerNorm backward over the last dim: -> (dx, dgamma, dbeta) with fp32 [C] parameter gradients (None when not requested)."""
    require_device(x, gy, gamma)
    if gy.shape != x.shape or gy.dtype != x.dtype:
        raise ValueError("layernorm_backward: gy must match x")
    c = x.shape[-1]
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if rows_of(x) == 0:  # nothing is launched over an empty tensor: the partial table would stay uninitialised
        z = torch.zeros(c, dtype=torch.float32, device=x.device) if want_param_grads else None
        return dx, z, None if z is None else z.clone()
    slots = int(lib().gm_layernorm_bwd_slots(rows_of(x)))  # one stored partial per block (no atomics, nothing to zero)
    st = torch.empty((slots, c, 2), dtype=torch.float64, device=x.device) if want_param_grads else None
    check(lib().gm_layernorm_bwd(x.data_ptr(), arena_ld(x), gy.data_ptr(), arena_ld(gy), dx.data_ptr(), arena_ld(dx), _ptr(as_f32(gamma)),
                                 rows_of(x), c, float(eps), _ptr(st), dt_code(x.dtype), _stream()), "gm_layernorm_bwd")
    if not want_param_grads:
        return dx, None, None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    flat = st.view(-1)
    check(lib().gm_stats_colsum(flat.data_ptr(), slots, 1, c, dgamma.data_ptr(), 0, _stream()), "gm_stats_colsum")
    check(lib().gm_stats_colsum(flat[1:].data_ptr(), slots, 1, c, dbeta.data_ptr(), 0, _stream()), "gm_stats_colsum")
    return dx, dgamma, dbeta


def geglu_backward(x: torch.Tensor, gy: torch.Tensor) -> torch.Tensor:
    """Backward of geglu(x) = x[..., :M] * gelu(x[..., M:]): -> dx with the shape of x."""
    require_device(x, gy)
    inner = x.shape[-1] // 2
    if gy.shape != (*x.shape[:-1], inner) or gy.dtype != x.dtype:
        raise ValueError("geglu_backward: gy must be the forward output's shape")
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(lib().gm_geglu_bwd(x.data_ptr(), arena_ld(x), gy.data_ptr(), arena_ld(gy), dx.data_ptr(), arena_ld(dx), rows_of(x), inner,
                             dt_code(x.dtype), _stream()), "gm_geglu_bwd")
    return dx


def softmax_bwd(probs: torch.Tensor, dprobs: torch.Tensor, scale: float) -> torch.Tensor:
    """scale * P * (dP - rowsum(dP * P)) on fp32 (rows, V) matrices: the gradient of the scores scale * Q K^T through the softmax."""
    require_device(probs, dprobs)
    if probs.shape != dprobs.shape or probs.dim() != 2 or probs.dtype != torch.float32 or dprobs.dtype != torch.float32:
        raise ValueError("softmax_bwd expects matching fp32 (rows, V) matrices")
    probs, dprobs = probs.contiguous(), dprobs.contiguous()
    out = torch.empty_like(probs)
    check(lib().gm_softmax_bwd(probs.data_ptr(), dprobs.data_ptr(), out.data_ptr(), probs.shape[0], probs.shape[1], float(scale), _stream()),
          "gm_softmax_bwd")
    return out


# ------------------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------------------
def _kv_ld(t: torch.Tensor) -> int:
    """Row pitch of a (B, L, C) key / value operand whose batch stride is free (the first L rows of a per-sample KV cache)."""
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("attention keys / values must be (B, L, C) with unit channel stride")
    return t.stride(1) if t.shape[1] > 1 else max(t.stride(1), t.shape[2])


def attention_workspace(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, bytes_only: bool = False):
    """The scratch the LDS-DMA attention kernel wants for these operands (its transposed V image comes first), or None when another kernel
    serves the geometry.  An attention block allocates it BEFORE its q | k | v projection so that the projection can store the V image itself
    (conv(..., vt=(workspace, first V channel, head dim))) and passes it on: attention(..., workspace=ws, vt_packed=True)."""
    b, lq, c = q.shape
    d = GmAttnDesc()
    d.q, d.q_ld, d.k, d.k_ld, d.v, d.v_ld = q.data_ptr(), arena_ld(q), k.data_ptr(), _kv_ld(k), v.data_ptr(), _kv_ld(v)
    d.res, d.res_ld, d.o, d.o_ld = None, 0, q.data_ptr(), arena_ld(q)
    d.B, d.H, d.Lq, d.Lk, d.dh, d.scale, d.dtype = b, heads, lq, k.shape[1], c // heads, 1.0, dt_code(q.dtype)
    d.causal, d.k_bs, d.v_bs = 0, 0, 0
    if b > 1 and (k.stride(0) != k.shape[1] * _kv_ld(k) or v.stride(0) != v.shape[1] * _kv_ld(v)):
        return None
    nbytes = lib().gm_attention_workspace_bytes(C.byref(d))
    if bytes_only:
        return int(nbytes)
    return torch.empty(nbytes, dtype=torch.uint8, device=q.device) if nbytes > 0 else None


def attention_writes_lse(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> bool:
    """Whether attention(q, k, v, heads, ..., lse_out=...) is served (the LDS-DMA kernel takes these operands)."""
    return bool(attention_workspace(q, k, v, heads, bytes_only=True))


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float,
              res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, causal: bool = False,
              workspace: Optional[torch.Tensor] = None, vt_packed: bool = False, lse_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scale * Q K^T) V per (batch, head). q: (B, Lq, heads*dh) arena views (channel slices allowed), k/v: (B, Lk, ...);
    k / v may be the first Lk rows of a longer per-sample buffer (a KV cache). causal: query i sees keys j <= i + (Lk - Lq).
    workspace / vt_packed: see attention_workspace.  lse_out: fp32 (B, heads, Lq) tensor that receives log sum_k exp(scale q.k) -- only for
    geometries the LDS-DMA kernel serves (attention_workspace(...) is not None); the training forward keeps it for attention_backward_fused."""
    require_device(q, k, v, res, out)
    b, lq, c = q.shape
    lk = k.shape[1]
    if c % heads != 0 or k.shape[2] != c or v.shape[2] != c or v.shape[1] != lk:
        raise ValueError("attention operand shapes are inconsistent")
    dh = c // heads
    if dh > lib().gm_attention_max_head_dim():
        raise ValueError(f"head dim {dh} exceeds the gfx950 attention kernel limit ({lib().gm_attention_max_head_dim()})")
    if out is None:
        out = torch.empty((b, lq, c), dtype=q.dtype, device=q.device)
    d = GmAttnDesc()
    d.q, d.q_ld = q.data_ptr(), arena_ld(q)
    d.k, d.k_ld = k.data_ptr(), _kv_ld(k)
    d.v, d.v_ld = v.data_ptr(), _kv_ld(v)
    if res is not None:
        if tuple(res.shape) != (b, lq, c) or res.dtype != q.dtype:
            raise ValueError("attention residual shape/dtype mismatch")
        d.res, d.res_ld = res.data_ptr(), arena_ld(res)
    else:
        d.res, d.res_ld = None, 0
    d.o, d.o_ld = out.data_ptr(), arena_ld(out)
    d.B, d.H, d.Lq, d.Lk, d.dh = b, heads, lq, lk, dh
    d.scale, d.dtype = float(scale), dt_code(q.dtype)
    # q / out / res: batch strides must equal L * ld (tokens of one sample are row-dense); k / v may carry their own batch stride
    for t, L in ((q, lq), (out, lq)) + (((res, lq),) if res is not None else ()):
        if t.shape[0] > 1 and t.stride(0) != L * arena_ld(t):
            raise ValueError("attention queries / outputs must be row-dense over (batch, tokens)")
    d.causal = int(bool(causal))
    d.k_bs = k.stride(0) if (b > 1 and k.stride(0) != lk * _kv_ld(k)) else 0
    d.v_bs = v.stride(0) if (b > 1 and v.stride(0) != lk * _kv_ld(v)) else 0
    if causal and lk < lq:
        raise ValueError("causal attention needs at least as many keys as queries")
    d.workspace, d.workspace_bytes = None, 0
    ws_bytes = lib().gm_attention_workspace_bytes(C.byref(d))
    d.vt_packed = 0
    if ws_bytes > 0:  # scratch for the transposed V image of the LDS-DMA kernel; stream-ordered, freed on return
        if workspace is not None and workspace.numel() * workspace.element_size() >= ws_bytes:
            ws = workspace
            d.vt_packed = int(bool(vt_packed))
        else:
            if vt_packed:
                raise ValueError("vt_packed needs the workspace the projection wrote the V image into")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
    elif vt_packed:
        raise ValueError("vt_packed: this geometry is not served by the LDS-DMA attention kernel")
    d.lse = None
    if lse_out is not None:
        require_device(lse_out)
        if ws_bytes <= 0:
            raise ValueError("lse_out: this geometry is not served by the LDS-DMA attention kernel")
        if lse_out.dtype != torch.float32 or lse_out.numel() != b * heads * lq or not lse_out.is_contiguous():
            raise ValueError("lse_out must be a contiguous fp32 (B, heads, Lq) tensor")
        d.lse = lse_out.data_ptr()
    # the split-KV merge kernel can store per-channel (sum, sum of squares) partials of the output it writes: the GroupNorm of the block that
    # follows an attention block then needs no statistics pass (channel_stats() finds them on the tensor, like a convolution's)
    d.stats = None
    st = None
    slots = lib().gm_attention_stats_slots(C.byref(d)) if ws_bytes > 0 else 0
    if slots > 0:
        st = torch.empty((int(slots), b, c, 2), dtype=torch.float64, device=q.device)
        d.stats = st.data_ptr()
    meta = dict(flops=4.0 * b * lq * lk * c, bytes=float(q.element_size() * b * (2 * lq + 2 * lk) * c), shape=f"B{b} H{heads} Lq{lq} Lk{lk} dh{dh}")
    _timed(f"attention<{str(q.dtype).split('.')[-1]}>", meta,
           lambda: check(lib().gm_attention_forward(C.byref(d), _stream()), "gm_attention_forward"))
    if st is not None:
        out._gm_cstats = st
    return out


# ------------------------------------------------------------------------------------------------------------------------
# misc element-wise
# ------------------------------------------------------------------------------------------------------------------------
def embed_tokens(indices: torch.Tensor, token_weight: torch.Tensor, position_weight: torch.Tensor, pos0: int = 0) -> torch.Tensor:
    """token_weight[indices] + position_weight[pos0 + arange(T)] for (B, T) int64 indices -> (B, T, C)."""
    require_device(indices, token_weight, position_weight)
    if indices.dim() != 2 or indices.dtype != torch.long or token_weight.dtype != position_weight.dtype or token_weight.shape[1] != position_weight.shape[1]:
        raise ValueError("embed_tokens expects (B, T) int64 indices and two embedding tables of one dtype / width")
    b, t = indices.shape
    if pos0 + t > position_weight.shape[0]:
        raise ValueError("sequence is longer than the position embedding table")
    indices = indices.contiguous()
    tw, pw = token_weight.detach().contiguous(), position_weight.detach().contiguous()
    out = torch.empty((b, t, tw.shape[1]), dtype=tw.dtype, device=indices.device)
    check(lib().gm_embed_tokens(indices.data_ptr(), tw.data_ptr(), pw.data_ptr(), out.data_ptr(), b, t, tw.shape[1], int(pos0), tw.shape[0],
                                pw.shape[0], dt_code(tw.dtype), _stream()), "gm_embed_tokens")
    return out


def sample_probs(logits: torch.Tensor, temperature: float, top_k: Optional[int], bos_index: int) -> torch.Tensor:
    """The transformer sampling head on (rows, V) logits -> fp32 probabilities (temperature, top-k crop, softmax, BOS zeroed)."""
    require_device(logits)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("sample_probs expects (rows, V) logits with unit stride over V")
    rows, v = logits.shape
    probs = torch.empty((rows, v), dtype=torch.float32, device=logits.device)
    check(lib().gm_sample_probs(logits.data_ptr(), logits.stride(0), probs.data_ptr(), rows, v, float(temperature),
                                0 if top_k is None else int(min(top_k, v)), int(bos_index), dt_code(logits.dtype), _stream()), "gm_sample_probs")
    return probs


def sample_index(probs: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """One categorical draw per row of fp32 (rows, V) probabilities (need not be normalised) -> int64 (rows, 1).  The uniform numbers
    come from torch's device generator; unlike torch.multinomial nothing is read back to the host."""
    require_device(probs)
    if probs.dim() != 2 or probs.dtype != torch.float32 or not probs.is_contiguous():
        raise ValueError("sample_index expects contiguous fp32 (rows, V) probabilities")
    u = torch.rand(probs.shape[0], device=probs.device, generator=generator)
    out = torch.empty((probs.shape[0], 1), dtype=torch.long, device=probs.device)
    check(lib().gm_sample_index(probs.data_ptr(), probs.shape[0], probs.shape[1], u.data_ptr(), out.data_ptr(), _stream()), "gm_sample_index")
    return out


def decode_advance(pos_dev: torch.Tensor, tokens: torch.Tensor, idx: torch.Tensor, seq: torch.Tensor) -> None:
    """seq[:, pos + 1] = idx; tokens[:] = idx; pos += 1, all on the device (the sampling loop's bookkeeping, graph-replayable)."""
    require_device(pos_dev, tokens, idx, seq)
    if pos_dev.dtype != torch.int32 or any(t.dtype != torch.long for t in (tokens, idx, seq)) or not seq.is_contiguous():
        raise TypeError("pos_dev int32; tokens, idx, seq int64 (seq contiguous)")
    check(lib().gm_decode_advance(pos_dev.data_ptr(), tokens.data_ptr(), idx.data_ptr(), seq.data_ptr(), seq.shape[0], seq.shape[1], _stream()),
          "gm_decode_advance")


def token_log_prob(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """log(softmax(logits)[target]) for (rows, V) logits and (rows,) int64 targets -> fp32 (rows,)."""
    require_device(logits, target)
    if logits.dim() != 2 or logits.stride(1) != 1 or target.dtype != torch.long or target.numel() != logits.shape[0]:
        raise ValueError("token_log_prob expects (rows, V) logits and (rows,) int64 targets")
    target = target.contiguous()
    out = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    check(lib().gm_token_log_prob(logits.data_ptr(), logits.stride(0), target.data_ptr(), out.data_ptr(), logits.shape[0], logits.shape[1],
                                  dt_code(logits.dtype), _stream()), "gm_token_log_prob")
    return out


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0, dtype=torch.float32) -> torch.Tensor:
    require_device(timesteps)
    t = timesteps.to(torch.float32).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=dtype, device=t.device)
    check(lib().gm_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, float(max_period), dt_code(dtype), _stream()),
          "gm_timestep_embedding")
    return out


class NoiseBits:
    """The Gaussian noise of a bf16 step as the CPU generator's byte draws (device uint8 [n]) + the pair table (host_noise.py): sched_step expands them itself."""

    def __init__(self, bits: torch.Tensor, table: torch.Tensor, shape) -> None:
        self.bits, self.table, self.shape = bits, table, tuple(shape)

    def materialise(self) -> torch.Tensor:
        return normal_bf16_from_bits(self.bits, self.table).reshape(self.shape)


def sched_step(sample: torch.Tensor, model_output: torch.Tensor, params: GmStepParams, noise=None, want_x0: bool = True):
    """Fused DDIM / DDPM step on logical NC[D]HW tensors (all contiguous, same dtype).  noise: a tensor, or NoiseBits (bf16 chains)."""
    if isinstance(noise, NoiseBits):
        if sample.dtype != torch.bfloat16 or noise.bits.numel() != sample.numel() or sample.numel() % 16 != 0 or params.noise_mode == 0:
            noise = noise.materialise()
        else:
            require_device(sample, model_output, noise.bits, noise.table)
            if sample.dtype != model_output.dtype:
                raise TypeError("sample and model_output must share a dtype")
            sample, model_output = sample.contiguous(), model_output.contiguous()
            batch = sample.shape[0]
            inner = sample.numel() // max(batch, 1)
            mo_bs = model_output.numel() // max(batch, 1)
            if mo_bs not in (inner, 2 * inner) or (params.noise_mode in (2, 3) and mo_bs != 2 * inner):
                raise ValueError("model_output shape does not match the sample / the variance type")
            prev = torch.empty_like(sample)
            x0 = torch.empty_like(sample) if want_x0 else None
            nb = sample.element_size() * sample.numel()
            _timed("sched_step", dict(flops=0.0, bytes=float(nb * (3 + (x0 is not None)) + sample.numel()), shape=str(tuple(sample.shape))),
                   lambda: check(lib().gm_sched_step_noise_bits(sample.data_ptr(), model_output.data_ptr(), noise.bits.data_ptr(), noise.table.data_ptr(), prev.data_ptr(),
                                                                _ptr(x0), batch, inner, mo_bs, C.byref(params), _stream()), "gm_sched_step_noise_bits"))
            return prev, x0
    require_device(sample, model_output, noise)
    if sample.dtype != model_output.dtype:
        raise TypeError("sample and model_output must share a dtype")
    sample = sample.contiguous()
    model_output = model_output.contiguous()
    batch = sample.shape[0]
    inner = sample.numel() // max(batch, 1)
    mo_bs = model_output.numel() // max(batch, 1)
    if mo_bs not in (inner, 2 * inner):
        raise ValueError("model_output shape does not match the sample")
    if params.noise_mode in (2, 3) and mo_bs != 2 * inner:
        raise ValueError("learned variance needs 2*C model output channels")
    if noise is not None:
        noise = noise.contiguous()
        if noise.numel() != sample.numel() or noise.dtype != sample.dtype:
            raise ValueError("noise must match the sample")
    prev = torch.empty_like(sample)
    x0 = torch.empty_like(sample) if want_x0 else None
    nb = sample.element_size() * sample.numel()
    _timed("sched_step", dict(flops=0.0, bytes=float(nb * (3 + (x0 is not None) + (noise is not None))), shape=str(tuple(sample.shape))),
           lambda: check(lib().gm_sched_step(sample.data_ptr(), model_output.data_ptr(), _ptr(noise), prev.data_ptr(), _ptr(x0), batch,
                                             inner, mo_bs, dt_code(sample.dtype), C.byref(params), _stream()), "gm_sched_step"))
    return prev, x0


def likelihood_workspace_elems(batch: int, inner: int) -> int:
    """fp64 elements of likelihood_term's workspace: one partial per (sample, block); needs no initialisation."""
    return int(lib().gm_likelihood_workspace_elems(int(batch), int(inner)))


def likelihood_term(x0: torch.Tensor, xt: torch.Tensor, model_output: torch.Tensor, params: GmKlParams, total: torch.Tensor,
                    workspace: torch.Tensor, want_map: bool = False) -> Optional[torch.Tensor]:
    """One term of get_likelihood's bound: adds mean_over_elements(KL or decoder NLL) to total[n]; returns the map if asked."""
    require_device(x0, xt, model_output, total, workspace)
    if x0.shape != xt.shape or x0.dtype != xt.dtype or x0.dtype != model_output.dtype:
        raise ValueError("inputs, noised inputs and model output must share dtype; inputs and noised inputs share shape")
    if total.dtype != torch.float32 or workspace.dtype != torch.float64:
        raise TypeError("total must be fp32 and workspace fp64")
    if workspace.numel() < likelihood_workspace_elems(x0.shape[0], x0.numel() // max(x0.shape[0], 1)):
        raise ValueError("workspace too small (ops.likelihood_workspace_elems)")
    x0, xt, model_output = x0.contiguous(), xt.contiguous(), model_output.contiguous()
    batch = x0.shape[0]
    inner = x0.numel() // max(batch, 1)
    mo_bs = model_output.numel() // max(batch, 1)
    if mo_bs < inner:
        raise ValueError("model_output is smaller than the inputs")
    kl = torch.empty_like(x0) if want_map else None
    _timed("likelihood_term", dict(flops=0.0, bytes=float(x0.element_size() * x0.numel() * (3 + want_map)), shape=str(tuple(x0.shape))),
           lambda: check(lib().gm_likelihood_term(x0.data_ptr(), xt.data_ptr(), model_output.data_ptr(), _ptr(kl), total.data_ptr(),
                                                  workspace.data_ptr(), batch, inner, mo_bs, dt_code(x0.dtype), C.byref(params),
                                                  _stream()), "gm_likelihood_term"))
    return kl


def lincomb(terms: Sequence[Optional[torch.Tensor]], coefs: Sequence[float], post_mul: float = 1.0, post_div: float = 1.0) -> torch.Tensor:
    """post_mul * (c0*x0 + c1*x1 + ...) / post_div, left to right, each op rounded in fp32 (PNDM's history combinations,
    reference pndm.py:186-195,241-250).  A None term is skipped (the reference's `0 + tensor`)."""
    live = [t for t in terms if t is not None]
    require_device(*live)
    if not live or len(terms) > 4 or len(terms) != len(coefs):
        raise ValueError("lincomb takes 1..4 (tensor, coefficient) pairs, at least one tensor")
    ref = live[0]
    for t in live:
        if t.shape != ref.shape or t.dtype != ref.dtype:
            raise ValueError("lincomb terms must share shape and dtype")
    keep = [None if t is None else t.contiguous() for t in terms]
    out = torch.empty_like(ref, memory_format=torch.contiguous_format)
    ptrs = (C.c_void_p * len(keep))(*[None if t is None else t.data_ptr() for t in keep])
    cs = (C.c_float * len(keep))(*[float(torch.tensor(c, dtype=torch.float32)) for c in coefs])
    _timed("lincomb", dict(flops=0.0, bytes=float(ref.element_size() * ref.numel() * (len(live) + 1)), shape=str(tuple(ref.shape))),
           lambda: check(lib().gm_lincomb(ptrs, cs, len(keep), float(torch.tensor(post_mul, dtype=torch.float32)),
                                          float(torch.tensor(post_div, dtype=torch.float32)), out.data_ptr(), ref.numel(),
                                          dt_code(ref.dtype), _stream()), "gm_lincomb"))
    return out


def axpby_rows(x: torch.Tensor, y: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out[n] = a[n] * x[n] + b[n] * y[n] with fp32 per-sample coefficients on the device."""
    require_device(x, y, a, b)
    if x.shape != y.shape or x.dtype != y.dtype:
        raise ValueError("axpby_rows operands must match")
    x, y = x.contiguous(), y.contiguous()
    a = a.to(torch.float32).contiguous()
    b = b.to(torch.float32).contiguous()
    batch = x.shape[0]
    if a.numel() != batch or b.numel() != batch:
        raise ValueError("one coefficient per batch row expected")
    out = torch.empty_like(x)
    check(lib().gm_axpby_rows(x.data_ptr(), y.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), batch,
                              x.numel() // max(batch, 1), dt_code(x.dtype), _stream()), "gm_axpby_rows")
    return out


def aekl_sample(mu: Optional[torch.Tensor], logvar: torch.Tensor, eps: Optional[torch.Tensor] = None, want_sigma: bool = True):
    """sigma = exp(clamp(logvar, -30, 20) / 2); z = mu + eps * sigma (if eps is given). Any matching dense layout."""
    require_device(mu, logvar, eps)
    logvar = logvar.contiguous()
    sigma = torch.empty_like(logvar) if want_sigma else None
    z = None
    if eps is not None:
        mu, eps = mu.contiguous(), eps.contiguous()
        z = torch.empty_like(logvar)
    check(lib().gm_aekl_sample(_ptr(mu), logvar.data_ptr(), _ptr(eps), _ptr(sigma), _ptr(z), logvar.numel(), dt_code(logvar.dtype),
                               _stream()), "gm_aekl_sample")
    return sigma, z


def addcmul(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """a + b * c element-wise (same shape / dtype, any matching dense layout)."""
    require_device(a, b, c)
    if not (a.shape == b.shape == c.shape) or not (a.dtype == b.dtype == c.dtype):
        raise ValueError("addcmul operands must match in shape and dtype")
    a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
    out = torch.empty_like(a)
    check(lib().gm_addcmul(a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), a.numel(), dt_code(a.dtype), _stream()),
          "gm_addcmul")
    return out


def scale(x: torch.Tensor, s: float, divide: bool = False) -> torch.Tensor:
    """x * s or x / s element-wise."""
    require_device(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().gm_scale(x.data_ptr(), out.data_ptr(), float(s), int(divide), x.numel(), dt_code(x.dtype), _stream()), "gm_scale")
    return out


def activation(x: torch.Tensor, act: str, gy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x) (gy None) or gy * act'(x) element-wise for an activation of POST_ACT: the stand-alone form of a convolution's epilogue activation
    (training forward of layers with a dropout between convolution and activation; derivatives that need the pre-activation)."""
    require_device(x, gy)
    x = x.contiguous()
    if gy is not None:
        if gy.shape != x.shape or gy.dtype != x.dtype:
            raise ValueError("activation: gy must match x")
        gy = gy.contiguous()
    out = torch.empty_like(x)
    check(lib().gm_activation(x.data_ptr(), _ptr(gy), out.data_ptr(), POST_ACT[act], 0 if gy is None else 1, x.numel(), dt_code(x.dtype), _stream()),
          "gm_activation")
    return out


def concat_dim1(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(parts, dim=1) for contiguous NC[D]HW tensors (inferers/inferer.py:72,127 "concat" conditioning)."""
    require_device(*parts)
    parts = [p.contiguous() for p in parts]
    n = parts[0].shape[0]
    inner = [p.numel() // max(n, 1) for p in parts]
    c = sum(p.shape[1] for p in parts)
    out = torch.empty((n, c, *parts[0].shape[2:]), dtype=parts[0].dtype, device=parts[0].device)
    tot, off = sum(inner), 0
    flat = out.reshape(n, tot) if n else out
    for p, k in zip(parts, inner):
        if tuple(p.shape[2:]) != tuple(parts[0].shape[2:]) or p.shape[0] != n:
            raise ValueError("concat_dim1 operands must agree outside dim 1")
        if n and k:
            check(lib().gm_copy_channels(p.data_ptr(), k, dt_code(p.dtype), flat.data_ptr() + off * flat.element_size(), tot,
                                         dt_code(out.dtype), n, k, _stream()), "gm_copy_channels")
        off += k
    return out


def vq_argmin(x: torch.Tensor, embedding: torch.Tensor) -> torch.Tensor:
    """x: arena (N, *spatial, D) -> int64 indices (N, *spatial)."""
    require_device(x, embedding)
    emb = as_f32(embedding)
    idx = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    check(lib().gm_vq_argmin(x.data_ptr(), arena_ld(x), emb.data_ptr(), idx.data_ptr(), rows_of(x), emb.shape[0], emb.shape[1],
                             dt_code(x.dtype), _stream()), "gm_vq_argmin")
    return idx


def vq_gather(indices: torch.Tensor, embedding: torch.Tensor, dtype: torch.dtype, x: Optional[torch.Tensor] = None):
    """indices (N, *spatial) -> arena (N, *spatial, D); with x also returns mean((q - x)^2) as a 0-dim fp32 tensor."""
    require_device(indices, embedding, x)
    emb = as_f32(embedding)
    indices = indices.contiguous()
    out = torch.empty((*indices.shape, emb.shape[1]), dtype=dtype, device=indices.device)
    err = ws = None
    if x is not None:
        err = torch.empty((), dtype=torch.float32, device=indices.device)
        ws = torch.empty(lib().gm_vq_gather_workspace_bytes(), dtype=torch.uint8, device=indices.device)
    check(lib().gm_vq_gather(indices.data_ptr(), emb.data_ptr(), out.data_ptr(), arena_ld(out), _ptr(x), 0 if x is None else arena_ld(x),
                             _ptr(err), _ptr(ws), indices.numel(), emb.shape[0], emb.shape[1], dt_code(dtype), _stream()),
          "gm_vq_gather")
    return (out, err) if x is not None else out
