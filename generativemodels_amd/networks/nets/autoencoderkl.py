"""AutoencoderKL for MI355X: constructor, state_dict names (`encoder.blocks.k.*`, `decoder.blocks.k.*`, `quant_conv_mu`, ...)
and encode / decode / sampling / forward contract of the reference's generative/networks/nets/autoencoderkl.py:600-799,
executed with the same fused HIP kernels as the diffusion UNet (N[D]HWC arena; GroupNorm+SiLU folded into the convolutions,
nearest-2x folded into the up-sampling convolution, asymmetric-pad strided convolution for down-sampling)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from ... import ops
from ._blocks import AttentionBlock, ConvP, ResnetBlock, SPADEResnetBlock, ensure_tuple_rep, gn_prologue, run_stage, wants_grad

__all__ = ["AutoencoderKL"]


class _Down(nn.Module):
    """3^d stride-2 convolution with one zero voxel padded on the HIGH side of every axis only
    (reference autoencoderkl.py:96-122); parameters live under `conv.conv.*`."""

    def __init__(self, spatial_dims: int, channels: int) -> None:
        super().__init__()
        self.conv = ConvP(spatial_dims, channels, channels, 3, 2, 0, pad_hi=1)

    def run(self, x):
        return self.conv.run(x, want_stats=True)  # a ResBlock's GroupNorm follows: statistics ride in the epilogue


class _Up(nn.Module):
    """Nearest 2x + conv3 (folded) or ConvTranspose(k3, s2, p1, op1) (reference autoencoderkl.py:41-93)."""

    def __init__(self, spatial_dims: int, channels: int, use_convtranspose: bool) -> None:
        super().__init__()
        self.use_convtranspose = use_convtranspose
        if use_convtranspose:
            self.conv = ConvP(spatial_dims, channels, channels, 3, 2, 1, transposed=True)
        else:
            self.conv = ConvP(spatial_dims, channels, channels, 3, 1, 1)

    def run(self, x):
        return self.conv.run(x, want_stats=True) if self.use_convtranspose else self.conv.run(x, upsample=True, want_stats=True)


def _res(spatial_dims, cin, cout, groups, eps):
    return ResnetBlock(spatial_dims, cin, cout, None, groups, eps, shortcut_name="nin_shortcut", zero_conv2=False)


def _run_blocks(blocks: nn.ModuleList, h: torch.Tensor, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
    """seg: arena segmentation handed to the SPADE blocks of a SPADEDecoder (spade_autoencoderkl.py:283-289)."""
    pre = None
    for blk in blocks:
        if isinstance(blk, nn.GroupNorm):
            pre = gn_prologue(blk, h)  # consumed by the next convolution's prologue (no SiLU: autoencoderkl.py:433-446)
        elif isinstance(blk, ConvP):
            h = blk.run(h, pre=pre, want_stats=True)
            pre = None
        elif isinstance(blk, SPADEResnetBlock):
            h = blk.run(h, None, seg)
        else:
            h = blk.run(h)
    return h


def _run_blocks_train(blocks: nn.ModuleList, h: torch.Tensor, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same block list with gradients (generativemodels_amd.autograd: native kernels in both directions) -- what the reference's
    autoencoder training loops differentiate through torch autograd (tutorials/generative/3d_autoencoderkl, engines/trainer.py:258-270)."""
    from ... import autograd as A

    for blk in blocks:
        if isinstance(blk, nn.GroupNorm):
            h = A.group_norm_act(h, blk.weight, blk.bias, blk.num_groups, blk.eps, "none")  # no SiLU before the last conv (:433-446)
        elif isinstance(blk, ConvP):
            h = A.conv(h, blk.conv.weight, blk.conv.bias, kernel=blk.kernel_size, stride=1, padding=blk.padding)
        elif isinstance(blk, SPADEResnetBlock):
            h = blk.run_train(h, None, seg)  # SPADE decoder (spade_autoencoderkl.py:283-289): seg = the arena segmentation
        elif isinstance(blk, _Down):
            c = blk.conv
            h = A.conv(h, c.conv.weight, c.conv.bias, kernel=3, stride=2, padding=0, pad_hi=1)
        elif isinstance(blk, _Up):
            c = blk.conv
            if blk.use_convtranspose:
                h = A.conv_transpose(h, c.conv.weight, c.conv.bias, kernel=3, stride=2, padding=1, output_padding=1)
            else:
                h = A.upsample_conv(h, c.conv.weight, c.conv.bias)
        else:
            h = blk.run_train(h)  # ResnetBlock (no timestep path), AttentionBlock
    return h


class Encoder(nn.Module):
    """conv_in, per level [ResBlock (+Attention)] x r and an asymmetric-pad stride-2 conv, optional non-local
    (ResBlock, Attention, ResBlock), GroupNorm, conv to the latent width (reference autoencoderkl.py:315-453)."""

    def __init__(self, spatial_dims, in_channels, num_channels, out_channels, num_res_blocks, norm_num_groups, norm_eps,
                 attention_levels, with_nonlocal_attn=True) -> None:
        super().__init__()
        g, eps = norm_num_groups, norm_eps
        blocks: list[nn.Module] = [ConvP(spatial_dims, in_channels, num_channels[0], 3, 1, 1)]
        out_c = num_channels[0]
        for i in range(len(num_channels)):
            in_c, out_c = out_c, num_channels[i]
            for _ in range(num_res_blocks[i]):
                blocks.append(_res(spatial_dims, in_c, out_c, g, eps))
                in_c = out_c
                if attention_levels[i]:
                    blocks.append(AttentionBlock(spatial_dims, in_c, None, g, eps))
            if i != len(num_channels) - 1:
                blocks.append(_Down(spatial_dims, in_c))
        if with_nonlocal_attn:
            c = num_channels[-1]
            blocks += [_res(spatial_dims, c, c, g, eps), AttentionBlock(spatial_dims, c, None, g, eps), _res(spatial_dims, c, c, g, eps)]
        blocks.append(nn.GroupNorm(num_groups=g, num_channels=num_channels[-1], eps=eps, affine=True))
        blocks.append(ConvP(spatial_dims, num_channels[-1], out_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)

    def run(self, x):
        return _run_blocks(self.blocks, x)

    def run_train(self, x):
        return _run_blocks_train(self.blocks, x)


class Decoder(nn.Module):
    """Mirror of the encoder (reference autoencoderkl.py:455-597)."""

    def __init__(self, spatial_dims, num_channels, in_channels, out_channels, num_res_blocks, norm_num_groups, norm_eps,
                 attention_levels, with_nonlocal_attn=True, use_convtranspose=False, spade: Optional[tuple] = None) -> None:
        """spade = (label_nc, spade_intermediate_channels): the residual blocks are SPADE-modulated (reference SPADEDecoder,
        spade_autoencoderkl.py:137-289: affine-free GroupNorm with nn.GroupNorm's default eps as the parameter-free norm)."""
        super().__init__()
        g, eps = norm_num_groups, norm_eps
        if spade is not None:
            make_res = lambda sd, cin, cout, g_, eps_: SPADEResnetBlock(sd, cin, cout, None, spade[0], g_, eps_, spade[1],  # noqa: E731
                                                                        shortcut_name="nin_shortcut", zero_conv2=False, affine=False)
        else:
            make_res = _res
        rc = list(reversed(num_channels))
        rr = list(reversed(num_res_blocks))
        ra = list(reversed(attention_levels))
        blocks: list[nn.Module] = [ConvP(spatial_dims, in_channels, rc[0], 3, 1, 1)]
        if with_nonlocal_attn:
            blocks += [make_res(spatial_dims, rc[0], rc[0], g, eps), AttentionBlock(spatial_dims, rc[0], None, g, eps),
                       make_res(spatial_dims, rc[0], rc[0], g, eps)]
        out_c = rc[0]
        for i in range(len(rc)):
            in_c, out_c = out_c, rc[i]
            for _ in range(rr[i]):
                blocks.append(make_res(spatial_dims, in_c, out_c, g, eps))
                in_c = out_c
                if ra[i]:
                    blocks.append(AttentionBlock(spatial_dims, in_c, None, g, eps))
            if i != len(rc) - 1:
                blocks.append(_Up(spatial_dims, in_c, use_convtranspose))
        blocks.append(nn.GroupNorm(num_groups=g, num_channels=in_c, eps=eps, affine=True))
        blocks.append(ConvP(spatial_dims, in_c, out_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)

    def run(self, x, seg: Optional[torch.Tensor] = None):
        return _run_blocks(self.blocks, x, seg)

    def run_train(self, x, seg: Optional[torch.Tensor] = None):
        return _run_blocks_train(self.blocks, x, seg)


class AutoencoderKL(nn.Module):
    """Drop-in for generative.networks.nets.AutoencoderKL (same arguments, state_dict keys and methods)."""

    def __init__(self, spatial_dims: int, in_channels: int = 1, out_channels: int = 1, num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
                 num_channels: Sequence[int] = (32, 64, 64, 64), attention_levels: Sequence[bool] = (False, False, True, True),
                 latent_channels: int = 3, norm_num_groups: int = 32, norm_eps: float = 1e-6, with_encoder_nonlocal_attn: bool = True,
                 with_decoder_nonlocal_attn: bool = True, use_flash_attention: bool = False, use_checkpointing: bool = False,
                 use_convtranspose: bool = False) -> None:
        super().__init__()
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("AutoencoderKL expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("AutoencoderKL expects num_channels being same size of attention_levels")
        if isinstance(num_res_blocks, int):
            num_res_blocks = ensure_tuple_rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same length as "
                             "`num_channels`.")
        self.spatial_dims = spatial_dims
        self.encoder = Encoder(spatial_dims, in_channels, num_channels, latent_channels, num_res_blocks, norm_num_groups, norm_eps,
                               attention_levels, with_encoder_nonlocal_attn)
        self.decoder = Decoder(spatial_dims, num_channels, latent_channels, out_channels, num_res_blocks, norm_num_groups, norm_eps,
                               attention_levels, with_decoder_nonlocal_attn, use_convtranspose)
        self.quant_conv_mu = ConvP(spatial_dims, latent_channels, latent_channels, 1, 1, 0)
        self.quant_conv_log_sigma = ConvP(spatial_dims, latent_channels, latent_channels, 1, 1, 0)
        self.post_quant_conv = ConvP(spatial_dims, latent_channels, latent_channels, 1, 1, 0)
        self.latent_channels = latent_channels
        self.use_checkpointing = use_checkpointing  # training path: encoder / decoder stages under torch.utils.checkpoint, as the reference (_blocks.run_stage)

    def _dtype(self):
        return self.post_quant_conv.conv.weight.dtype

    def _check(self, x: torch.Tensor) -> torch.Tensor:
        """-> x in the compute dtype: the parameters' (the input must match), or the active autocast region's (ops.autocast: the input is cast --
        differentiably when the differentiable path is taken)."""
        ops.require_device(x)
        if x.dim() != self.spatial_dims + 2:
            raise ValueError(f"expected a (N, C, *{self.spatial_dims} spatial dims) tensor, got {tuple(x.shape)}")
        if ops.autocast_dtype() is not None and x.dtype != ops.autocast_dtype() and wants_grad(self, x):
            from ... import autograd as A

            return A.cast(x, ops.autocast_dtype())
        return ops.entry_cast(x, self._dtype())

    def encode(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """-> (z_mu, z_sigma), sigma = exp(clamp(log_var, -30, 20) / 2) (reference autoencoderkl.py:718-736)."""
        x = self._check(x)
        if wants_grad(self, x):  # a training step: differentiable (z_mu, z_sigma), native kernels in both directions
            from ... import autograd as A

            h = run_stage(self.encoder.run_train, A.to_arena(x), self.use_checkpointing)
            mu_c, ls_c = self.quant_conv_mu.conv, self.quant_conv_log_sigma.conv
            z_mu = A.from_arena(A.conv(h, mu_c.weight, mu_c.bias, kernel=1))
            return z_mu, A.sigma_from_log_var(A.from_arena(A.conv(h, ls_c.weight, ls_c.bias, kernel=1)))
        with torch.no_grad():
            h = self.encoder.run(ops.to_channels_last(x))
            z_mu = ops.to_channels_first(self.quant_conv_mu.run(h))
            z_log_var = ops.to_channels_first(self.quant_conv_log_sigma.run(h))
            z_sigma, _ = ops.aekl_sample(None, z_log_var)
        return z_mu, z_sigma

    def sampling(self, z_mu: torch.Tensor, z_sigma: torch.Tensor) -> torch.Tensor:
        """z = mu + eps * sigma, eps ~ N(0, I) from the device generator (reference autoencoderkl.py:738-753)."""
        ops.require_device(z_mu, z_sigma)
        eps = torch.randn_like(z_sigma)
        if torch.is_grad_enabled() and (z_mu.requires_grad or z_sigma.requires_grad):
            return z_mu + eps * z_sigma  # the reparameterisation on the (latent-sized) training graph: torch autograd, like the loss
        return ops.addcmul(z_mu, eps, z_sigma)

    def reconstruct(self, x: torch.Tensor) -> torch.Tensor:
        z_mu, _ = self.encode(x)
        return self.decode(z_mu)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """post_quant_conv -> Decoder (reference autoencoderkl.py:769-784)."""
        z = self._check(z)
        if wants_grad(self, z):
            from ... import autograd as A

            pq = self.post_quant_conv.conv
            return A.from_arena(run_stage(self.decoder.run_train, A.conv(A.to_arena(z.contiguous()), pq.weight, pq.bias, kernel=1), self.use_checkpointing))
        with torch.no_grad():
            h = self.post_quant_conv.run(ops.to_channels_last(z))
            return ops.to_channels_first(self.decoder.run(h))

    def forward(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        return self.decode(z), z_mu, z_sigma

    def encode_stage_2_inputs(self, x: torch.Tensor) -> torch.Tensor:
        z_mu, z_sigma = self.encode(x)
        return self.sampling(z_mu, z_sigma)

    def decode_stage_2_outputs(self, z: torch.Tensor) -> torch.Tensor:
        return self.decode(z)
