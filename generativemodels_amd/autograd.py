"""Differentiable forms of the fused MI355X ops (the backward half a training step needs --
reference: tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270, generative/engines/trainer.py:258-270, where the
backward pass is torch autograd over nn.Conv3d / nn.GroupNorm / nn.SiLU).

Every function here works on arena tensors (N, *spatial, C) and runs native kernels in both directions:
  conv            forward: the fused convolution (bias, per-sample row vector, residual in the epilogue)
                  dx: the transposed convolution of gy with the same weight (the forward LDS-DMA kernel at stride 1)
                  dW: gm_conv_wgrad (MFMA, split-K, deterministic); db / d(row vector): column sums of gy
  group_norm_act  forward: per-channel statistics -> (scale, shift) -> one apply pass (+ SiLU)
                  backward: gm_gn_bwd_stats / _finalize / _apply (dx, dgamma, dbeta)
  to_arena / from_arena   the NC[D]HW <-> N[D]HWC permutations
There is no eager fallback: a CPU tensor raises in the first native call."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

__all__ = ["conv", "linear", "group_norm_act", "to_arena", "from_arena"]


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * n


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rowvec, res, kernel, stride, padding, pad_hi):
        y = ops.conv(x, weight, bias, kernel=kernel, stride=stride, padding=padding, pad_hi=pad_hi, rowvec=rowvec, res=res)
        ctx.save_for_backward(x, weight)
        ctx.geom = (kernel, stride, padding, pad_hi)
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.row_shape = None if rowvec is None else tuple(rowvec.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        kernel, stride, padding, pad_hi = ctx.geom
        gy = gy.contiguous()
        nsp = x.dim() - 2
        k, s, p = _tup(kernel, nsp), _tup(stride, nsp), _tup(padding, nsp)
        phi = _tup(pad_hi, nsp) if pad_hi is not None else p
        dx = dw = db = drow = dres = None
        if ctx.needs_input_grad[0]:
            opad = tuple(x.shape[1 + i] - ((gy.shape[1 + i] - 1) * s[i] - p[i] - phi[i] + k[i]) for i in range(nsp))
            if any(o < 0 or o >= s[i] for i, o in enumerate(opad)):
                raise ValueError(f"convolution geometry is not invertible: output_padding {opad}")
            dx = ops.conv(gy, weight, None, kernel=k, stride=s, padding=p, pad_hi=phi, transposed=True, output_padding=opad)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(x, gy, k, s, p).reshape(weight.shape).to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = ops.bias_grad(gy).to(ctx.bias_dtype)
        if ctx.needs_input_grad[3]:
            drow = ops.bias_grad(gy, per_sample=True) if ctx.row_shape[0] != 1 else ops.bias_grad(gy)[None]
        if ctx.needs_input_grad[4]:
            dres = gy
        return dx, dw, db, drow, dres, None, None, None, None


def conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, kernel, stride=1, padding=0, pad_hi=None,
         rowvec: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = conv(x, weight) + bias + rowvec[n] + res over an arena tensor; differentiable in x, weight, bias, rowvec (fp32 [N or 1,
    Cout]) and res."""
    return _Conv.apply(x, weight, bias, rowvec, res, kernel, stride, padding, pad_hi)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear over the last dim of (N, L, C)."""
    if x.dim() != 3:
        raise ValueError("linear expects (N, L, C)")
    return _Conv.apply(x, weight, bias, None, res, 1, 1, 0, None)


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
