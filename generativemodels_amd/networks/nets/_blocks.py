"""Parameter containers + arena-level forward helpers shared by DiffusionModelUNet, AutoencoderKL and VQVAE.

The modules hold parameters under the *reference's state_dict names* (e.g. `conv1.conv.weight`, `norm1.weight`,
`to_q.weight`, `proj_attn.weight`) so reference checkpoints load unchanged; their forward passes never call a torch
compute op -- they enqueue the fused HIP kernels of libgmamd.so on N[D]HWC arena tensors (generativemodels_amd.ops)."""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from ... import ops

_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {1: nn.ConvTranspose1d, 2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}


def ensure_tuple_rep(v, n: int) -> tuple:
    """Scalar -> n-tuple; a length-n sequence -> tuple (MONAI `ensure_tuple_rep` semantics used by the reference ctors)."""
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError(f"Sequence must have length {n}, got {len(v)}.")
        return tuple(v)
    return (v,) * n


_warned_eval_grad = False


def wants_grad(module: nn.Module, x: torch.Tensor) -> bool:
    """Which forward a network runs.  The reference's forward is differentiable whenever autograd records; here that costs a different
    (activation-saving) kernel sequence, so the differentiable path is taken when gradients are enabled AND
      * the input requires grad -- whatever the mode and even through frozen parameters: a pixel-space loss through a frozen decoder, latent
        optimisation, a frozen UNet under ControlNet training all need d(output)/d(input) (dropping it silently would lose that loss term), or
      * the module is in train() mode and has a trainable parameter.
    eval() / torch.no_grad() / an input without grad through frozen parameters run the fused inference path.  The one case where this differs
    observably from the reference -- eval() mode, gradients enabled, trainable parameters, input without grad: the output has no grad_fn --
    warns once instead of detaching silently."""
    if not torch.is_grad_enabled():
        return False
    if x.requires_grad:
        return True
    trainable = any(p.requires_grad for p in module.parameters())
    if module.training:
        return trainable
    global _warned_eval_grad
    if trainable and not _warned_eval_grad:
        _warned_eval_grad = True
        import warnings

        warnings.warn(f"{type(module).__name__}: forward in eval() mode with gradients enabled runs the fused inference path and returns a tensor "
                      "without grad_fn (the differentiable forward is taken in train() mode, or when the input requires grad); wrap inference "
                      "in torch.no_grad() to silence this", stacklevel=3)
    return False


def run_stage(fn, x: torch.Tensor, checkpointing: bool) -> torch.Tensor:
    """A training-forward stage, optionally under activation checkpointing exactly as the reference does it
    (`torch.utils.checkpoint.checkpoint(stage, x, use_reentrant=False)`, autoencoderkl.py:726-729,780-783, vqvae.py:418-431): the stage's
    autograd.Functions save nothing across the step; its forward is re-run (the same kernels, the same bits) when backward reaches it."""
    if checkpointing:
        import torch.utils.checkpoint as cp

        return cp.checkpoint(fn, x, use_reentrant=False)
    return fn(x)


def zero_module(module: nn.Module) -> nn.Module:
    for p in module.parameters():
        p.detach().zero_()
    return module


class ConvP(nn.Module):
    """Holder of one convolution's parameters under the child name `conv` (the layout MONAI's Convolution(conv_only=True)
    produces: reference diffusion_model_unet.py:1748-1756). `nn.ConvNd` is used purely as a parameter container with the
    reference's default initialisation; its forward is never called."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size: int = 3, strides: int = 1,
                 padding: Optional[int] = None, dilation: int = 1, transposed: bool = False, output_padding: Optional[int] = None,
                 pad_hi: Optional[int] = None) -> None:
        super().__init__()
        self.spatial_dims = spatial_dims
        self.kernel_size, self.strides, self.dilation = kernel_size, strides, dilation
        self.padding = (kernel_size - 1) // 2 * dilation if padding is None else padding
        self.pad_hi = pad_hi
        self.transposed = transposed
        self.output_padding = (strides - 1 if output_padding is None else output_padding) if transposed else 0
        if transposed:
            self.conv = _CONVT[spatial_dims](in_channels, out_channels, kernel_size, stride=strides, padding=self.padding,
                                             output_padding=self.output_padding, dilation=dilation)
        else:
            self.conv = _CONV[spatial_dims](in_channels, out_channels, kernel_size, stride=strides, padding=self.padding,
                                            dilation=dilation)

    @property
    def out_channels(self) -> int:
        return self.conv.out_channels

    def run(self, x: torch.Tensor, **fusion) -> torch.Tensor:
        return ops.conv(x, self.conv.weight, self.conv.bias, kernel=self.kernel_size, stride=self.strides, padding=self.padding,
                        dilation=self.dilation, pad_hi=self.pad_hi, transposed=self.transposed,
                        output_padding=self.output_padding, **fusion)

    def forward(self, x):  # pragma: no cover - the arena path goes through run()
        raise RuntimeError("ConvP is driven through the fused arena path (run), not called as a torch module")


def gn_prologue(norm: nn.GroupNorm, x):
    """(scale, shift) of a GroupNorm over an arena tensor (or a VirtualCat of two), ready for a consumer's prologue. Built
    from per-channel statistics: free when the producing convolution fused them into its epilogue."""
    return ops.gn_scale_shift_composed(x, norm.num_groups, norm.eps, norm.weight, norm.bias)


def lin(x: torch.Tensor, layer: nn.Linear, **fusion) -> torch.Tensor:
    return ops.linear(x, layer.weight, layer.bias, **fusion)


def tokens(x: torch.Tensor) -> torch.Tensor:
    """(N, *spatial, C) arena -> (N, L, C) view: NDHWC makes the reference's reshape+permute (:430-433) free."""
    return x.reshape(x.shape[0], -1, x.shape[-1])


class AttentionBlock(nn.Module):
    """Spatial self-attention: GroupNorm -> q,k,v Linear(+bias) -> softmax(QK^T/sqrt(d)) V -> + x.

    Reference: diffusion_model_unet.py:345-458 and its twin autoencoderkl.py:196-312.  `proj_attn` is constructed (and kept
    in the state_dict) but never applied by the reference forward, so it is not applied here either.
    Fusion: GN statistics (1 pass) -> one GEMM for the stacked q|k|v projection with the GN affine as its prologue ->
    flash attention with the residual add as its epilogue."""

    def __init__(self, spatial_dims: int, num_channels: int, num_head_channels: Optional[int] = None, norm_num_groups: int = 32,
                 norm_eps: float = 1e-6) -> None:
        super().__init__()
        self.spatial_dims = spatial_dims
        self.num_channels = num_channels
        if num_head_channels is not None and num_channels % num_head_channels != 0:
            raise ValueError("num_channels must be divisible by num_head_channels")
        self.num_heads = num_channels // num_head_channels if num_head_channels is not None else 1
        self.scale = 1 / math.sqrt(num_channels / self.num_heads)
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=num_channels, eps=norm_eps, affine=True)
        self.to_q = nn.Linear(num_channels, num_channels)
        self.to_k = nn.Linear(num_channels, num_channels)
        self.to_v = nn.Linear(num_channels, num_channels)
        self.proj_attn = nn.Linear(num_channels, num_channels)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        c = self.num_channels
        pre = gn_prologue(self.norm, x)
        xt = tokens(x)
        wq = ops.packed_cat_weight([self.to_q.weight, self.to_k.weight, self.to_v.weight], x.dtype)
        bq = ops.cat_f32([self.to_q.bias, self.to_k.bias, self.to_v.bias], [c, c, c], x.device)
        # the projection also stores the transposed V image of the LDS-DMA attention kernel when it runs as the small-row GEMM (one launch less)
        qkv = torch.empty((*xt.shape[:-1], 3 * c), dtype=xt.dtype, device=xt.device)
        q, k, v = qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:3 * c]
        ws = ops.attention_workspace(q, k, v, self.num_heads) if (xt.dtype == torch.bfloat16 and xt.shape[1] % 64 == 0) else None
        qkv = ops.conv(xt, None, bq, kernel=1, pre=pre, packed=wq, cout=3 * c, out=qkv, vt=None if ws is None else (ws, 2 * c, c // self.num_heads))
        packed = bool(getattr(qkv, "_gm_vt_packed", False))
        o = ops.attention(q, k, v, self.num_heads, self.scale, res=xt, workspace=ws, vt_packed=packed)
        y = o.reshape(x.shape)
        st = getattr(o, "_gm_cstats", None)
        if st is not None:  # per-channel statistics of the block's output, written by the attention merge kernel: the next GroupNorm reads them
            y._gm_cstats = st
        return y

    def run_train(self, x: torch.Tensor) -> torch.Tensor:
        """The same block with gradients (generativemodels_amd.autograd): GroupNorm, three projections, attention, residual."""
        from ... import autograd as A

        n = self.norm
        t = tokens(A.group_norm_act(x, n.weight, n.bias, n.num_groups, n.eps, "none"))
        q = A.linear(t, self.to_q.weight, self.to_q.bias)
        k = A.linear(t, self.to_k.weight, self.to_k.bias)
        v = A.linear(t, self.to_v.weight, self.to_v.bias)
        o = A.attention(q, k, v, self.num_heads, self.scale)
        return A.add(o, tokens(x)).reshape(x.shape)


class ResnetBlock(nn.Module):
    """GN -> SiLU -> conv3 (+ timestep row) -> GN -> SiLU -> conv3 -> + skip(x).

    Covers the UNet block (reference diffusion_model_unet.py:589-696; `temb_channels` given) and the AutoencoderKL block
    (autoencoderkl.py:125-193; no timestep path, shortcut named `nin_shortcut`).  Two fused conv launches -- GroupNorm-apply + SiLU in
    their prologues (applied in LDS to the staged patch), bias / timestep row / residual and the NEXT GroupNorm's statistics in their
    epilogues, the 1x1 shortcut inside the second one -- plus two microsecond statistic folds replace the reference's 2 GN + 2 SiLU +
    2-3 conv + 2 add launches: the "fused 3-D ResnetBlock kernel pair" of SURVEY.md 8(a) a5."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: Optional[int] = None, temb_channels: Optional[int] = None,
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, up: bool = False, down: bool = False,
                 shortcut_name: str = "skip_connection", zero_conv2: bool = True) -> None:
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.up, self.down = up, down
        self.shortcut_name = shortcut_name
        self.norm1 = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=norm_eps, affine=True)
        self.conv1 = ConvP(spatial_dims, in_channels, out_channels, 3, 1, 1)
        if temb_channels is not None:
            self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(num_groups=norm_num_groups, num_channels=out_channels, eps=norm_eps, affine=True)
        self.conv2 = ConvP(spatial_dims, out_channels, out_channels, 3, 1, 1)
        if zero_conv2:
            zero_module(self.conv2)
        if in_channels != out_channels:
            setattr(self, shortcut_name, ConvP(spatial_dims, in_channels, out_channels, 1, 1, 0))
        else:
            setattr(self, shortcut_name, nn.Identity())

    def run(self, x, temb_row: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: arena tensor, or an ops.VirtualCat([h, skip]) (decoder half: the concatenation is never materialised)."""
        cat = isinstance(x, ops.VirtualCat)
        if cat and (self.up or self.down):
            x, cat = x.materialise(), False
        pre1 = gn_prologue(self.norm1, x)
        if self.up or self.down:
            # resblock_updown variant: BOTH branches are resampled after norm1+SiLU (diffusion_model_unet.py:674-682);
            # SiLU does not commute with average pooling, so the activated tensor is materialised for this rare path.
            h = ops.gn_apply(x, pre1[0], pre1[1], "silu")
            mode = "up" if self.up else "down"
            x = ops.resample2x(x, mode)
            h = ops.resample2x(h, mode)
            h = self.conv1.run(h, rowvec=temb_row, want_stats=True)
        else:
            # GroupNorm-apply + SiLU ride in conv1's prologue; ops.conv places them: in LDS inside the LDS-DMA kernel (which also reads the
            # two halves of a virtual concat in place), in the register-staged kernels' patch staging, or as one gm_gn_apply pass per part
            h = self.conv1.run(x, pre=pre1, pre_act="silu", rowvec=temb_row, want_stats=True)
        pre2 = gn_prologue(self.norm2, h)
        shortcut = getattr(self, self.shortcut_name)
        fusion = dict(want_stats=True)
        if not isinstance(shortcut, ConvP):
            fusion["res"] = x  # identity (never a VirtualCat: the decoder resnets always change width)
        else:
            # the 1x1 shortcut over x -- or over the two halves of the virtual concat -- rides along in conv2 as extra K chunks
            fusion["skip"] = (list(x.parts) if cat else [x], shortcut.conv.weight, shortcut.conv.bias)
        return self.conv2.run(h, pre=pre2, pre_act="silu", **fusion)

    def run_train(self, x: torch.Tensor, temb: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same block with gradients: x an arena tensor, temb the [N, temb_channels] timestep embedding.
        Native kernels in both directions (generativemodels_amd.autograd); the GroupNorm-apply passes are materialised because their
        outputs are what the weight-gradient kernel contracts against."""
        from ... import autograd as A

        n1, n2 = self.norm1, self.norm2
        h = A.group_norm_act(x, n1.weight, n1.bias, n1.num_groups, n1.eps, "silu")
        if self.up or self.down:  # resblock_updown: BOTH branches are resampled after norm1 + SiLU (diffusion_model_unet.py:674-682)
            mode = "up" if self.up else "down"
            x, h = A.resample2x(x, mode), A.resample2x(h, mode)
        row = None
        if temb is not None:
            # [N, temb] -> [N, Cout]: a handful of rows through the small-row GEMM
            row = A.linear(A.silu(temb[None]), self.time_emb_proj.weight, self.time_emb_proj.bias)[0].float()
        c1, c2 = self.conv1, self.conv2
        h = A.conv(h, c1.conv.weight, c1.conv.bias, kernel=c1.kernel_size, stride=1, padding=c1.padding, rowvec=row)
        h = A.group_norm_act(h, n2.weight, n2.bias, n2.num_groups, n2.eps, "silu")
        shortcut = getattr(self, self.shortcut_name)
        xs = A.conv(x, shortcut.conv.weight, shortcut.conv.bias, kernel=1) if isinstance(shortcut, ConvP) else x
        return A.conv(h, c2.conv.weight, c2.conv.bias, kernel=c2.kernel_size, stride=1, padding=c2.padding, res=xs)


class SPADEResnetBlock(ResnetBlock):
    """ResnetBlock whose two GroupNorms are SPADE-modulated by a segmentation map: the decoder blocks of SPADEDiffusionModelUNet
    (reference spade_diffusion_model_unet.py:72-200; GroupNorm with affine parameters and `norm_eps` as the parameter-free norm) and,
    with `temb_channels=None`, the SPADEResBlock of SPADEAutoencoderKL (spade_autoencoderkl.py:42-134; affine-free GroupNorm with the
    default eps 1e-5, shortcut named `nin_shortcut`).  One fused SPADE pass (norm x modulation x SiLU) replaces each GroupNorm-apply."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: Optional[int], temb_channels: Optional[int], label_nc: int,
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, spade_intermediate_channels: int = 128,
                 shortcut_name: str = "skip_connection", zero_conv2: bool = True, affine: bool = True) -> None:
        super().__init__(spatial_dims, in_channels, out_channels, temb_channels, norm_num_groups, norm_eps, shortcut_name=shortcut_name,
                         zero_conv2=zero_conv2)
        from ..blocks.spade_norm import SPADE

        params = {"num_groups": norm_num_groups, "eps": norm_eps, "affine": True} if affine else {"num_groups": norm_num_groups, "affine": False}
        self.norm1 = SPADE(label_nc, in_channels, 3, spatial_dims, spade_intermediate_channels, "GROUP", params)
        self.norm2 = SPADE(label_nc, self.out_channels, 3, spatial_dims, spade_intermediate_channels, "GROUP", params)

    def run(self, x, temb_row: Optional[torch.Tensor] = None, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
        if seg is None:
            raise ValueError("SPADEResnetBlock needs the segmentation map")
        cat = isinstance(x, ops.VirtualCat)
        h = self.conv1.run(self.norm1.run(x, seg, "silu"), rowvec=temb_row, want_stats=True)
        shortcut = getattr(self, self.shortcut_name)
        fusion = {}
        if not isinstance(shortcut, ConvP):
            fusion["res"] = x
        else:
            fusion["skip"] = (list(x.parts) if cat else [x], shortcut.conv.weight, shortcut.conv.bias)
        return self.conv2.run(self.norm2.run(h, seg, "silu"), want_stats=True, **fusion)

    def run_train(self, x: torch.Tensor, temb: Optional[torch.Tensor] = None, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same block with gradients: SPADE.run_train in place of the two GroupNorm + SiLU steps of ResnetBlock.run_train (reference: torch
        autograd through spade_diffusion_model_unet.py:173-200 / spade_autoencoderkl.py:105-134)."""
        from ... import autograd as A

        if seg is None:
            raise ValueError("SPADEResnetBlock needs the segmentation map")
        h = self.norm1.run_train(x, seg, "silu")
        row = None
        if temb is not None:
            row = A.linear(A.silu(temb[None]), self.time_emb_proj.weight, self.time_emb_proj.bias)[0].float()
        c1, c2 = self.conv1, self.conv2
        h = A.conv(h, c1.conv.weight, c1.conv.bias, kernel=c1.kernel_size, stride=1, padding=c1.padding, rowvec=row)
        h = self.norm2.run_train(h, seg, "silu")
        shortcut = getattr(self, self.shortcut_name)
        xs = A.conv(x, shortcut.conv.weight, shortcut.conv.bias, kernel=1) if isinstance(shortcut, ConvP) else x
        return A.conv(h, c2.conv.weight, c2.conv.bias, kernel=c2.kernel_size, stride=1, padding=c2.padding, res=xs)
