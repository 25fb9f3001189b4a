"""Differentiable forms of the fused MI355X ops (the backward half a training step needs --
reference: tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270, generative/engines/trainer.py:258-270, where the
backward pass is torch autograd over nn.Conv3d / nn.GroupNorm / nn.SiLU).

Every function here works on arena tensors (N, *spatial, C) and runs native kernels in both directions:
  conv            forward: the fused convolution (bias, per-sample row vector, residual in the epilogue)
                  dx: the transposed convolution of gy with the same weight (the forward LDS-DMA kernel at stride 1)
                  dW: gm_conv_wgrad (MFMA, split-K, deterministic); db / d(row vector): column sums of gy
  group_norm_act  forward: per-channel statistics -> (scale, shift) -> one apply pass (+ SiLU)
                  backward: gm_gn_bwd_stats / _finalize / _apply (dx, dgamma, dbeta)
  upsample_conv   nearest 2x folded into the convolution; backward: dgrad on the fine grid, 2x sum-pool; dW against the upsampled input
  attention       forward: the flash-attention kernel; backward: the fused flash backward gm_attention_backward (head dims 16 .. 256: scores
                  recomputed per tile, nothing L x L stored); other head dims per (sample, head) in fp32 -- scores, softmax, dV = P^T dO,
                  dP = dO V^T, dS (gm_softmax_bwd), dQ = dS K, dK = dS^T Q -- on the GEMM / weight-gradient kernels (up to 8192 tokens)
  add / cat       residual add and channel concatenation
  to_arena / from_arena   the NC[D]HW <-> N[D]HWC permutations
There is no eager fallback: a CPU tensor raises in the first native call."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

__all__ = ["cast", "scale", "spade_modulate", "conv", "conv_transpose", "conv_dilated", "conv_transpose_dilated", "sigma_from_log_var", "linear", "group_norm_act", "layer_norm", "geglu", "resample2x", "embedding", "silu", "upsample_conv", "attention", "add", "cat", "to_arena", "from_arena"]


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * n


def _weight_grad(weight: torch.Tensor, shape, run):
    """The weight gradient of one layer: `run(out, accumulate)` launches gm_conv_wgrad (fp32 [Cout, Cin, *k] result).  When the parameter's
    `.grad` is a GradientReducer bucket view (fp32 master parameters: parallel.direct_grad_hook), the kernel's final reduction ADDS INTO IT
    and autograd gets None (no `add_` launch for this parameter; the reducer learns of the gradient from the engine's post-accumulate hook, which
    fires once after every use of the weight); otherwise the fresh tensor goes back to autograd in the parameter's dtype."""
    from .parallel import direct_grad_hook

    if direct_grad_hook(weight):
        run(weight.grad.view(shape), True)  # readiness is signalled by the engine's post-accumulate hook, after ALL uses of the weight
        return None
    return run(None, False).reshape(weight.shape).to(weight.dtype)


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rowvec, res, kernel, stride, padding, pad_hi, post_act="none"):
        # want_stats: the fast kernels leave the per-channel statistics of y on the tensor, so the GroupNorm that follows needs no pass
        y = ops.conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, pad_hi=pad_hi, rowvec=rowvec, res=res,
                     want_stats=x.dim() >= 4 and post_act == "none", post_act=post_act)
        ctx.post_act = post_act
        if post_act != "none":  # the epilogue activation is differentiated from its OUTPUT (ReLU: y > 0 <=> z > 0): y is all the backward needs
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.geom = (kernel, stride, padding, pad_hi)
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.row_shape = None if rowvec is None else tuple(rowvec.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors  # (read ONCE: under torch.utils.checkpoint every access re-triggers the unpack hooks)
        x, weight = saved[:2]
        kernel, stride, padding, pad_hi = ctx.geom
        gy = gy.contiguous()
        if ctx.post_act != "none":
            gy = ops.act_backward(saved[2], gy, ctx.post_act)
        nsp = x.dim() - 2
        k, s, p = _tup(kernel, nsp), _tup(stride, nsp), _tup(padding, nsp)
        phi = _tup(pad_hi, nsp) if pad_hi is not None else p
        dx = dw = db = drow = dres = None
        if ctx.needs_input_grad[0]:
            opad = tuple(x.shape[1 + i] - ((gy.shape[1 + i] - 1) * s[i] - p[i] - phi[i] + k[i]) for i in range(nsp))
            if any(o < 0 or o >= s[i] for i, o in enumerate(opad)):
                raise ValueError(f"convolution geometry is not invertible: output_padding {opad}")
            dx = None
            if nsp == 3 and k == (3, 3, 3) and s == (2, 2, 2) and p[0] == p[1] == p[2] and phi == (1, 1, 1):
                # stride 2 (the Downsample convolutions): one sub-pixel launch on gy instead of a convolution over its zero-inserted image
                dx = ops.conv_stride2_dgrad(gy, weight, x.shape[1:4], p[0])
            if dx is None:
                dx = ops.conv(gy, weight, None, kernel=k, stride=s, padding=p, pad_hi=phi, transposed=True, output_padding=opad)
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(weight, (weight.shape[0], weight.shape[1], *k), lambda out, acc: ops.conv_wgrad(x, gy, k, s, p, out=out, accumulate=acc))
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        if ctx.needs_input_grad[3]:
            drow = ops.bias_grad(gy, per_sample=True) if ctx.row_shape[0] != 1 else ops.bias_grad(gy)[None]
        if ctx.needs_input_grad[4]:
            dres = gy
        return dx, dw, db, drow, dres, None, None, None, None, None


def conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, kernel, stride=1, padding=0, pad_hi=None,
         rowvec: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, post_act: str = "none") -> torch.Tensor:
    """y = post_act(conv(x, weight) + bias + rowvec[n] + res) over an arena tensor; differentiable in x, weight, bias, rowvec (fp32 [N or 1,
    Cout]) and res.  post_act: "none" or "relu" (fused into the epilogue; the VQ-VAE's convolution + ReLU layers).  Kernel 1 or 3 at stride 1 / 2,
    and kernel 4 at stride 2 (the VQ-VAE down-sampling convolutions: weight gradient over the phase images of x, ops.conv_wgrad)."""
    return _Conv.apply(x, weight, bias, rowvec, res, kernel, stride, padding, pad_hi, post_act)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear over the last dim of (N, L, C)."""
    if x.dim() != 3:
        raise ValueError("linear expects (N, L, C)")
    return _Conv.apply(x, weight, bias, None, res, 1, 1, 0, None)


class _ConvTranspose(torch.autograd.Function):
    """nn.ConvTranspose{2,3}d (weight [Cin, Cout, *k]): AutoencoderKL's `use_convtranspose` up-sampling (autoencoderkl.py:54-63: k3 s2 p1 op1).
    It is the adjoint of the convolution with the same weight read as [out = Cin, in = Cout]: dx is that convolution applied to gy, and
    dW the weight gradient of that convolution with the roles of input and output gradient exchanged."""

    @staticmethod
    def forward(ctx, x, weight, bias, kernel, stride, padding, output_padding, post_act="none"):
        y = ops.conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, transposed=True, output_padding=output_padding,
                     want_stats=post_act == "none", post_act=post_act)
        ctx.post_act = post_act
        if post_act != "none":
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.geom = (kernel, stride, padding)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors  # (read ONCE: under torch.utils.checkpoint every access re-triggers the unpack hooks)
        x, weight = saved[:2]
        kernel, stride, padding = ctx.geom
        gy = gy.contiguous()
        if ctx.post_act != "none":
            gy = ops.act_backward(saved[2], gy, ctx.post_act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv(gy, weight, None, kernel=kernel, stride=stride, padding=padding)  # [Cin, Cout, *k] read as a Conv weight
            if dx.shape != x.shape:
                raise ValueError(f"transposed-convolution geometry is not invertible: {tuple(dx.shape)} vs {tuple(x.shape)}")
        if ctx.needs_input_grad[1]:
            kt = _tup(kernel, x.dim() - 2)
            dw = _weight_grad(weight, (weight.shape[0], weight.shape[1], *kt),
                              lambda out, acc: ops.conv_wgrad(gy, x, kernel, stride, padding, out=out, accumulate=acc))
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        return dx, dw, db, None, None, None, None, None


def conv_transpose(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, kernel, stride, padding,
                   output_padding=0, post_act: str = "none") -> torch.Tensor:
    """post_act(nn.ConvTransposeNd) over an arena tensor (weight [Cin, Cout, *k]); differentiable in x, weight, bias (kernel 3 at stride 1 or 2,
    kernel 4 at stride 2: the VQ-VAE up-sampling); post_act "none" or "relu"."""
    return _ConvTranspose.apply(x, weight, bias, kernel, stride, padding, output_padding, post_act)


def _tap_samples(t: torch.Tensor, out_sp, k, s, p, d):
    """For every kernel tap the arena tensor `t` ([N, *spatial, C]) sampled at the positions a dilated / strided convolution reads for its output
    grid `out_sp`: X_tap[n, v, c] = t[n, s v + d tap - p, c], zero outside.  Pure data movement (one zero-padded copy + one strided copy per tap, by
    torch): the rare dilated layers have no gather kernel of their own -- every multiply-add of their gradients still runs in libgmamd (the 1x1
    weight-gradient kernel on these samples)."""
    nsp = len(out_sp)
    lo = [p[i] for i in range(nsp)]
    hi = [max(0, s[i] * (out_sp[i] - 1) + d[i] * (k[i] - 1) - p[i] - (t.shape[1 + i] - 1)) for i in range(nsp)]
    pad = []
    for i in reversed(range(nsp)):
        pad += [lo[i], hi[i]]
    tp = torch.nn.functional.pad(t, [0, 0] + pad)  # (channels last: the first pair pads C by nothing)
    import itertools
    for tap in itertools.product(*[range(k[i]) for i in range(nsp)]):
        idx = [slice(None)] + [slice(d[i] * tap[i], d[i] * tap[i] + s[i] * (out_sp[i] - 1) + 1, s[i]) for i in range(nsp)] + [slice(None)]
        yield tap, tp[tuple(idx)].contiguous()


class _ConvDilated(torch.autograd.Function):
    """post_act(nn.ConvNd) with dilation > 1 (reference: the VQ-VAE's down-sampling Convolution(dilation=downsample_parameters[i][2]),
    nets/vqvae.py:127-150).  Forward: the generic kernel.  dx: the transposed convolution with the same weight, stride, padding and dilation.
    dW: per tap a 1x1 weight gradient (gm_conv_wgrad) of gy against the input sampled where that tap read it."""

    @staticmethod
    def forward(ctx, x, weight, bias, kernel, stride, padding, dilation, post_act="none"):
        y = ops.conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, dilation=dilation, post_act=post_act)
        ctx.post_act = post_act
        if post_act != "none":
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.geom = (kernel, stride, padding, dilation)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        x, weight = saved[:2]
        kernel, stride, padding, dilation = ctx.geom
        gy = gy.contiguous()
        if ctx.post_act != "none":
            gy = ops.act_backward(saved[2], gy, ctx.post_act)
        nsp = x.dim() - 2
        k, s, p, d = _tup(kernel, nsp), _tup(stride, nsp), _tup(padding, nsp), _tup(dilation, nsp)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            opad = tuple(x.shape[1 + i] - ((gy.shape[1 + i] - 1) * s[i] - 2 * p[i] + d[i] * (k[i] - 1) + 1) for i in range(nsp))
            dx = ops.conv(gy, weight, None, kernel=kernel, stride=stride, padding=padding, dilation=dilation, transposed=True, output_padding=opad)
        if ctx.needs_input_grad[1]:
            dw32 = torch.empty((weight.shape[0], weight.shape[1], *k), dtype=torch.float32, device=x.device)
            for tap, xs in _tap_samples(x, tuple(gy.shape[1:-1]), k, s, p, d):
                dw32[(slice(None), slice(None)) + tap] = ops.conv_wgrad(xs, gy, 1, 1, 0).reshape(weight.shape[0], weight.shape[1])
            dw = dw32.to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        return dx, dw, db, None, None, None, None, None


class _ConvTransposeDilated(torch.autograd.Function):
    """post_act(nn.ConvTransposeNd) with dilation > 1 (the VQ-VAE's up-sampling Convolution(is_transposed=True, dilation=...), nets/vqvae.py:244-261): the
    adjoint of _ConvDilated with the same weight read as [out = Cin, in = Cout]."""

    @staticmethod
    def forward(ctx, x, weight, bias, kernel, stride, padding, output_padding, dilation, post_act="none"):
        y = ops.conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, dilation=dilation, transposed=True, output_padding=output_padding,
                     post_act=post_act)
        ctx.post_act = post_act
        if post_act != "none":
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.geom = (kernel, stride, padding, dilation)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        x, weight = saved[:2]
        kernel, stride, padding, dilation = ctx.geom
        gy = gy.contiguous()
        if ctx.post_act != "none":
            gy = ops.act_backward(saved[2], gy, ctx.post_act)
        nsp = x.dim() - 2
        k, s, p, d = _tup(kernel, nsp), _tup(stride, nsp), _tup(padding, nsp), _tup(dilation, nsp)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv(gy, weight, None, kernel=kernel, stride=stride, padding=padding, dilation=dilation)  # [Cin, Cout, *k] read as a Conv weight
            if dx.shape != x.shape:
                raise ValueError(f"transposed-convolution geometry is not invertible: {tuple(dx.shape)} vs {tuple(x.shape)}")
        if ctx.needs_input_grad[1]:
            dw32 = torch.empty((weight.shape[0], weight.shape[1], *k), dtype=torch.float32, device=x.device)
            for tap, gs in _tap_samples(gy, tuple(x.shape[1:-1]), k, s, p, d):  # gy sampled where the adjoint convolution reads it for every input voxel
                dw32[(slice(None), slice(None)) + tap] = ops.conv_wgrad(gs, x, 1, 1, 0).reshape(weight.shape[0], weight.shape[1])
            dw = dw32.to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        return dx, dw, db, None, None, None, None, None, None


def conv_dilated(x, weight, bias=None, *, kernel, stride=1, padding=0, dilation=1, post_act: str = "none") -> torch.Tensor:
    """post_act(conv(x, weight, dilation) + bias) over an arena tensor, differentiable in x, weight, bias; post_act "none" or "relu"."""
    return _ConvDilated.apply(x, weight, bias, kernel, stride, padding, dilation, post_act)


def conv_transpose_dilated(x, weight, bias=None, *, kernel, stride, padding, output_padding=0, dilation=1, post_act: str = "none") -> torch.Tensor:
    """post_act(nn.ConvTransposeNd with dilation) over an arena tensor (weight [Cin, Cout, *k]), differentiable in x, weight, bias."""
    return _ConvTransposeDilated.apply(x, weight, bias, kernel, stride, padding, output_padding, dilation, post_act)


class _SigmaFromLogVar(torch.autograd.Function):
    """z_sigma = exp(clamp(z_log_var, -30, 20) / 2) (autoencoderkl.py:733-734) on the latent; d sigma / d log_var = sigma / 2 inside the clamp."""

    @staticmethod
    def forward(ctx, log_var):
        sigma, _ = ops.aekl_sample(None, log_var)
        ctx.save_for_backward(log_var, sigma)
        return sigma

    @staticmethod
    def backward(ctx, g):
        log_var, sigma = ctx.saved_tensors
        inside = ((log_var > -30.0) & (log_var < 20.0)).to(g.dtype)
        return g * sigma * inside * 0.5  # latent-sized (a few thousand elements): left to torch, like _Embedding's scatter


def sigma_from_log_var(log_var: torch.Tensor) -> torch.Tensor:
    return _SigmaFromLogVar.apply(log_var)


class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        scale, shift = ops.gn_scale_shift_composed(x, groups, eps, gamma, beta)
        y = ops.gn_apply(x, scale, shift, act)
        ctx.save_for_backward(x, gamma, scale, shift, ops.channel_stats(x))
        ctx.cfg = (groups, eps, act)
        ctx.dtypes = (None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, scale, shift, stats = ctx.saved_tensors
        groups, eps, act = ctx.cfg
        try:
            x._gm_cstats = stats  # the forward statistics: gn_backward does not re-read x for them
        except Exception:  # pragma: no cover
            pass
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dgamma, dbeta = ops.gn_backward(x, gy.contiguous(), scale, shift, gamma, groups, eps, act, want_affine_grads=want)
        if not ctx.needs_input_grad[1]:
            dgamma = None
        elif dgamma is not None:
            dgamma = dgamma.to(ctx.dtypes[0])
        if not ctx.needs_input_grad[2]:
            dbeta = None
        elif dbeta is not None:
            dbeta = dbeta.to(ctx.dtypes[1])
        return dx, dgamma, dbeta, None, None, None


def group_norm_act(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], groups: int, eps: float,
                   act: str = "none") -> torch.Tensor:
    """act(GroupNorm(x)) over an arena tensor, act in {"none", "silu"}; differentiable in x, gamma, beta."""
    return _GroupNormAct.apply(x, gamma, beta, groups, eps, act)


class _ToArena(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.to_channels_last(x)

    @staticmethod
    def backward(ctx, g):
        return ops.to_channels_first(g.contiguous())


class _FromArena(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.to_channels_first(x)

    @staticmethod
    def backward(ctx, g):
        return ops.to_channels_last(g.contiguous())


def to_arena(x: torch.Tensor) -> torch.Tensor:
    """NC[D]HW -> N[D]HWC, differentiable."""
    return _ToArena.apply(x)


def from_arena(x: torch.Tensor) -> torch.Tensor:
    """N[D]HWC -> NC[D]HW, differentiable."""
    return _FromArena.apply(x)


class _UpsampleConv(torch.autograd.Function):
    """Nearest 2x upsampling folded into a 3^d stride-1 convolution (reference Upsample, diffusion_model_unet.py:534-586)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        # the folded up-sampling path: the sub-pixel variant pre-sums its weights per parameter version, i.e. on every training step
        y = ops.conv(x, weight, bias, kernel=3, stride=1, padding=1, upsample=True, allow_subpixel=False, want_stats=True)
        ctx.save_for_backward(x, weight)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        dx = dw = db = None
        nsp = x.dim() - 2
        if ctx.needs_input_grad[0]:
            du = ops.conv(gy, weight, None, kernel=3, stride=1, padding=1, transposed=True)  # gradient on the upsampled grid
            dx = ops.scale(ops.resample2x(du, "down"), float(2 ** nsp))                        # sum over each 2^d cell
        if ctx.needs_input_grad[1]:
            xu = ops.resample2x(x, "up")
            dw = _weight_grad(weight, tuple(weight.shape), lambda out, acc: ops.conv_wgrad(xu, gy, 3, 1, 1, out=out, accumulate=acc))
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        return dx, dw, db


def upsample_conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _UpsampleConv.apply(x, weight, bias)


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.activation(x, act)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.activation(x, ctx.act, gy.contiguous()), None


def activation(x: torch.Tensor, act: str) -> torch.Tensor:
    """Stand-alone activation (any of ops.POST_ACT) with its backward from the pre-activation (gm_activation)."""
    return x if act == "none" else _Activation.apply(x, act)


ATTENTION_BWD_MAX_TOKENS = 8192
ATTENTION_BWD_BF16_MIN_TOKENS = int(__import__("os").environ.get("GM_ATTN_BWD_BF16_MIN_TOKENS", "512"))  # (bench switch: a huge value disables the path)


# (round 5) bf16 operands, head dim 64 / 128 / 256: the fused LDS-DMA flash backward (ops.attention_backward_fused) from this many tokens on.
# Measured on MI355X (tools/attn_bwd_fused_ab.py, profiles/r05_attn_bwd_fused_ab.txt): 32 768 x 256 3.97 ms against 17-20 composed, 4 096 x 256 0.14
# against 0.33, 2 x 8 heads x 256 x 64 0.048 against 0.092 (profiles/r05_attn_bwd_fused_policy.txt: every measured shape from 256 tokens on).  (bench switch: a huge value restores the round-4 policy)
ATTENTION_BWD_FUSED_MIN_TOKENS = int(__import__("os").environ.get("GM_ATTN_BWD_FUSED_MIN_TOKENS", "256"))


def _fused_backward_serves(q, k, heads):
    dh = q.shape[2] // heads
    return (q.dtype == torch.bfloat16 and dh in ops.ATTENTION_BWD_FUSED_HEAD_DIMS and q.shape[2] == heads * dh
            and max(q.shape[1], k.shape[1]) >= ATTENTION_BWD_FUSED_MIN_TOKENS)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        lse = None
        if _fused_backward_serves(q, k, heads) and ops.attention_writes_lse(q, k, v, heads):
            lse = torch.empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32, device=q.device)  # the forward kernel's log-sum-exp: one sweep less in backward
        o = ops.attention(q, k, v, heads, scale, lse_out=lse)
        ctx.save_for_backward(q, k, v, o, *(() if lse is None else (lse,)))
        ctx.cfg = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, *rest = ctx.saved_tensors
        heads, scale = ctx.cfg
        dh = q.shape[2] // heads
        go = go.contiguous()
        if _fused_backward_serves(q, k, heads):
            grads = ops.attention_backward_fused(q, k, v, o, go, heads, scale, lse=rest[0] if rest else None, or_none=True)
            if grads is not None:  # (None: the library declined -- more than 65 535 (sample, head) pairs, very wide rows: the round-4 path serves those)
                return (*grads, None, None)
        if dh in ops.ATTENTION_BWD_HEAD_DIMS:
            return (*_attention_backward(q, k, v, o, go, heads, scale), None, None)
        # any other head dim (<= 256: the forward's bound): zero-pad every head to the next width the kernels are built for.  Zero channels add
        # nothing to q k^T, and v's zero channels give o / dO zero channels: the products are unchanged, the padded gradient channels are dropped.
        dhp = min(d for d in ops.ATTENTION_BWD_HEAD_DIMS if d >= dh)

        def pad(t):
            out = torch.zeros((t.shape[0], t.shape[1], heads * dhp), dtype=t.dtype, device=t.device)
            for hi in range(heads):
                ops.copy_channels(t[:, :, hi * dh:(hi + 1) * dh], out[:, :, hi * dhp:hi * dhp + dh])
            return out

        def unpad(t):
            out = torch.empty((t.shape[0], t.shape[1], heads * dh), dtype=t.dtype, device=t.device)
            for hi in range(heads):
                ops.copy_channels(t[:, :, hi * dhp:hi * dhp + dh], out[:, :, hi * dh:(hi + 1) * dh])
            return out

        dq, dk, dv = _attention_backward(pad(q), pad(k), pad(v), pad(o), pad(go), heads, scale)
        return unpad(dq), unpad(dk), unpad(dv), None, None


def _attention_backward(q, k, v, o, go, heads, scale):
    """(dq, dk, dv) for a head dim in ops.ATTENTION_BWD_HEAD_DIMS: which kernels, measured on MI355X (tools/cmp_attention_backward.py,
    profiles/r03_attention_backward_paths.txt).  Every branch handles any sequence length: nothing raises for size."""
    b, lq, c = q.shape
    lk = k.shape[1]
    dh = c // heads
    long_seq = max(lq, lk) > ATTENTION_BWD_MAX_TOKENS
    # bf16 operands on the bf16-MFMA score pass + the weight-gradient kernel (ops.attention_backward_bf16: every product at the bf16 MFMA rate;
    # P, dS, dS^T of the (sample, head) pairs in flight live in HBM, one pair at a time when all of them would not fit): one or two pairs of
    # >= 512 tokens (C4: one head x 4096 tokens at 16^3, one x 512 at 8^3) -- and ANY number of pairs above 8 192 tokens, where the fused fp32
    # kernels below run at 45 TFLOP/s (84.6 ms per head at 32 768 x 256 against 16-17 ms here)
    if (q.dtype == torch.bfloat16 and dh in ops.ATTENTION_BWD_BF16_HEAD_DIMS and max(lq, lk) >= ATTENTION_BWD_BF16_MIN_TOKENS
            and (b * heads <= 2 or long_seq)):  # (any size: pairs beyond ops.ATTENTION_BWD_BF16_SLAB_BYTES of score matrices go through in query slabs)
        return ops.attention_backward_bf16(q, k, v, o, go, heads, scale)
    # fp32 (or bf16 below the bounds above): the fused kernels own 64 rows per work-group, so ONE head of a few thousand tokens leaves most CUs
    # idle (L = 4096, d = 128: 1.9 ms fused vs 0.86 ms composed) while many (sample, head) pairs favour them (2 x 4 heads of 1024 tokens: 0.21 vs
    # 2.5 ms); the composed path materialises fp32 L x L matrices, so long sequences always take the fused flash kernels
    if long_seq or not (b * heads <= 2 and max(lq, lk) >= 2048):
        return ops.attention_backward(q, k, v, o, go, heads, scale)  # fused flash backward: scores recomputed tile by tile, any sequence length
    f32 = torch.float32
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)

    def head(t, bi, hi, rows):  # fp32 contiguous [1, rows, dh] copy of one (sample, head) slice
        out = torch.empty((1, rows, dh), dtype=f32, device=t.device)
        ops.copy_channels(t[bi:bi + 1, :, hi * dh:(hi + 1) * dh], out)
        return out

    for bi in range(b):
        for hi in range(heads):
            qf, kf, vf, gf = head(q, bi, hi, lq), head(k, bi, hi, lk), head(v, bi, hi, lk), head(go, bi, hi, lq)
            s_ = ops.conv(qf, kf[0], None, kernel=1)                                   # [1, lq, lk] = Q K^T
            p_ = ops.sample_probs(s_[0], 1.0 / scale, None, -1)                        # softmax(scale * S), fp32
            dvh = ops.conv_wgrad(gf, p_[None], 1, 1, 0)                                # [lk, dh, 1] = P^T dO
            dp = ops.conv(gf, vf[0], None, kernel=1)                                   # [1, lq, lk] = dO V^T
            ds = ops.softmax_bwd(p_, dp[0], scale)                                     # [lq, lk]
            dqh = ops.conv(ds[None], kf[0], None, kernel=1, transposed=True)           # [1, lq, dh] = dS K
            dkh = ops.conv_wgrad(qf, ds[None], 1, 1, 0)                                # [lk, dh, 1] = dS^T Q
            sl = slice(hi * dh, (hi + 1) * dh)
            ops.copy_channels(dqh, dq[bi:bi + 1, :, sl])
            ops.copy_channels(dkh.reshape(1, lk, dh), dk[bi:bi + 1, :, sl])
            ops.copy_channels(dvh.reshape(1, lk, dh), dv[bi:bi + 1, :, sl])
    return dq, dk, dv


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """softmax(scale Q K^T) V per (sample, head) over (B, L, heads * dh) operands; differentiable in q, k, v."""
    return _Attention.apply(q, k, v, heads, scale)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ones = torch.ones(a.shape[0], dtype=torch.float32, device=a.device)
        return ops.axpby_rows(a, b, ones, ones).reshape(a.shape)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _Add.apply(a, b)


class _Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.split = a.shape[-1]
        return ops.concat_channels([a, b])

    @staticmethod
    def backward(ctx, g):
        ca = ctx.split
        ga = torch.empty((*g.shape[:-1], ca), dtype=g.dtype, device=g.device)
        gb = torch.empty((*g.shape[:-1], g.shape[-1] - ca), dtype=g.dtype, device=g.device)
        ops.copy_channels(g[..., :ca], ga)
        ops.copy_channels(g[..., ca:], gb)
        return ga, gb


def cat(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Channel concatenation of two arena tensors (the decoder's skip connections, diffusion_model_unet.py:1232,1340,1461)."""
    return _Cat.apply(a, b)


class _SiLU(torch.autograd.Function):
    """SiLU of a small (N, L, C) tensor (the timestep-embedding MLP) on the GroupNorm apply / backward kernels with an identity affine."""

    @staticmethod
    def _tables(x):
        c = x.shape[-1]
        one = torch.ones((x.shape[0], c), dtype=torch.float32, device=x.device)
        return one, torch.zeros_like(one)

    @staticmethod
    def forward(ctx, x):
        one, zero = _SiLU._tables(x)
        ctx.save_for_backward(x)
        return ops.gn_apply(x.contiguous(), one, zero, "silu")

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        x, gy = x.contiguous(), gy.contiguous()
        one, zero = _SiLU._tables(x)
        n, c = x.shape[0], x.shape[-1]
        v = ops.rows_of(x) // max(n, 1)
        dx = torch.empty_like(x)
        ops.check(ops.lib().gm_gn_bwd_apply(x.data_ptr(), ops.arena_ld(x), gy.data_ptr(), ops.arena_ld(gy), dx.data_ptr(), ops.arena_ld(dx),
                                            one.data_ptr(), zero.data_ptr(), c, one.data_ptr(), zero.data_ptr(), zero.data_ptr(), n, v, c, 1,
                                            ops.dt_code(x.dtype), ops._stream()), "gm_gn_bwd_apply")
        return dx


def silu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(x) over an (N, L, C) tensor; differentiable."""
    return _SiLU.apply(x)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.dtypes = (None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, gy):
        x, gamma = ctx.saved_tensors
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dg, db = ops.layernorm_backward(x, gy.contiguous(), gamma, ctx.eps, want_param_grads=want)
        dg = dg.to(ctx.dtypes[0]) if (dg is not None and ctx.needs_input_grad[1]) else None
        db = db.to(ctx.dtypes[1]) if (db is not None and ctx.needs_input_grad[2]) else None
        return dx, dg, db, None


def layer_norm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm over the last dim of (N, L, C); differentiable in x, gamma, beta."""
    return _LayerNorm.apply(x, gamma, beta, eps)


class _GEGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.geglu(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.geglu_backward(x, gy.contiguous())


def geglu(x: torch.Tensor) -> torch.Tensor:
    """x[..., :M] * gelu(x[..., M:]) (MONAI MLPBlock act="GEGLU"); differentiable."""
    return _GEGLU.apply(x)


class _Resample2x(torch.autograd.Function):
    """Nearest 2x up-sampling / 2x average pooling of an arena tensor (the resblock_updown ResnetBlocks, diffusion_model_unet.py:674-682)."""

    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        ctx.cells = float(2 ** (x.dim() - 2))
        return ops.resample2x(x, mode)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        if ctx.mode == "up":   # every input voxel fed 2^d outputs: sum = 2^d x their average
            return ops.scale(ops.resample2x(gy, "down"), ctx.cells), None
        return ops.scale(ops.resample2x(gy, "up"), ctx.cells, True), None  # average pooling: each input gets gy / 2^d


def resample2x(x: torch.Tensor, mode: str) -> torch.Tensor:
    return _Resample2x.apply(x, mode)


class _Embedding(torch.autograd.Function):
    """nn.Embedding lookup of a handful of class labels (diffusion_model_unet.py:1897-1902); the weight gradient is a scatter-add of the
    N gradient rows into a [num_classes, C] table -- left to torch (N rows)."""

    @staticmethod
    def forward(ctx, labels, weight, dtype):
        ctx.save_for_backward(labels)
        ctx.shape, ctx.dtype = weight.shape, weight.dtype
        return ops.vq_gather(labels, weight, dtype or weight.dtype)

    @staticmethod
    def backward(ctx, g):
        (labels,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device).index_add_(0, labels, g.float())
        return None, dw.to(ctx.dtype), None


def embedding(labels: torch.Tensor, weight: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """weight[labels] in `dtype` (default: the table's); differentiable in the table."""
    return _Embedding.apply(labels, weight, dtype)


class _SpadeModulate(torch.autograd.Function):
    """y = act(xn * g + bm): the SPADE modulation of a parameter-free-normalised tensor (blocks/spade_norm.py:79-96); differentiable in all three."""

    @staticmethod
    def forward(ctx, xn, g, bm, act):
        n, c = xn.shape[0], xn.shape[-1]
        one = torch.ones((n, c), dtype=torch.float32, device=xn.device)
        ctx.save_for_backward(xn, g, bm)
        ctx.act = act
        return ops.spade_apply(xn, one, torch.zeros_like(one), g, bm, act)

    @staticmethod
    def backward(ctx, gy):
        xn, g, bm = ctx.saved_tensors
        return (*ops.spade_backward(xn, g, bm, gy, ctx.act), None)


def spade_modulate(xn: torch.Tensor, g: torch.Tensor, bm: torch.Tensor, act: str = "none") -> torch.Tensor:
    return _SpadeModulate.apply(xn, g, bm, act)


class _Scale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return ops.scale(x, s, False).reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        return ops.scale(g.contiguous(), ctx.s, False).reshape(g.shape), None


def scale(x: torch.Tensor, s: float) -> torch.Tensor:
    """x * s (a Python scalar); differentiable in x."""
    return x if s == 1.0 else _Scale.apply(x, float(s))


class _Cast(torch.autograd.Function):
    """dtype conversion of an activation (the entry of a network inside an ops.autocast region); the gradient is cast back."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.cast(g.contiguous(), ctx.src), None


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return x if x.dtype == dtype else _Cast.apply(x, dtype)
