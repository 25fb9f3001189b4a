"""ControlNet for MI355X: constructor arguments, sub-module / state_dict names and forward contract of the reference's
generative/networks/nets/controlnet.py:47-436 (Zhang & Agrawala 2023).  The network is the DiffusionModelUNet's encoder + mid
block applied to `conv_in(x) + embed(controlnet_cond)`, with every skip tensor and the mid output passed through its own
zero-initialised 1x1 convolution and scaled by `conditioning_scale`; it runs on the same fused HIP kernels as the UNet
(the conditioning embedding's SiLU is the conv epilogue, `conv_in` takes the embedding as its residual)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from ... import ops
from ._blocks import _CONV, ConvP, ResnetBlock, ensure_tuple_rep, wants_grad, zero_module
from .diffusion_model_unet import _MidBlock, _Stage, _TimestepPath, _train_encoder, _train_entry, _train_timestep_embedding

__all__ = ["ControlNet", "ControlNetConditioningEmbedding", "copy_weights_to_controlnet"]


class ControlNetConditioningEmbedding(nn.Module):
    """Conditioning image -> feature map at the resolution of `conv_in(x)`: conv, then per level (conv, stride-2 conv), SiLU after
    each, zero-initialised output conv (reference controlnet.py:47-114)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, num_channels: Sequence[int] = (16, 32, 96, 256)) -> None:
        super().__init__()
        self.conv_in = ConvP(spatial_dims, in_channels, num_channels[0], 3, 1, 1)
        self.blocks = nn.ModuleList()
        for i in range(len(num_channels) - 1):
            self.blocks.append(ConvP(spatial_dims, num_channels[i], num_channels[i], 3, 1, 1))
            self.blocks.append(ConvP(spatial_dims, num_channels[i], num_channels[i + 1], 3, 2, 1))
        self.conv_out = zero_module(ConvP(spatial_dims, num_channels[-1], out_channels, 3, 1, 1))

    def run(self, cond: torch.Tensor) -> torch.Tensor:
        e = self.conv_in.run(cond, post_act="silu")
        for blk in self.blocks:
            e = blk.run(e, post_act="silu")
        return self.conv_out.run(e)

    def run_train(self, cond: torch.Tensor) -> torch.Tensor:
        """The same stack with gradients (arena in, arena out): convolution + SiLU as separate differentiable ops."""
        from ... import autograd as A

        def cv(blk, t):
            c = blk.conv
            return A.conv(t, c.weight, c.bias, kernel=3, stride=blk.strides, padding=1)

        e = A.silu(cv(self.conv_in, cond))
        for blk in self.blocks:
            e = A.silu(cv(blk, e))
        return cv(self.conv_out, e)

    def forward(self, conditioning: torch.Tensor) -> torch.Tensor:
        ops.require_device(conditioning)
        if wants_grad(self, conditioning):
            from ... import autograd as A

            return A.from_arena(self.run_train(A.to_arena(conditioning)))
        with torch.no_grad():
            return ops.to_channels_first(self.run(ops.to_channels_last(conditioning)))


def copy_weights_to_controlnet(controlnet: nn.Module, diffusion_model: nn.Module, verbose: bool = True) -> None:
    """Initialise the ControlNet's encoder from a diffusion model's weights (reference controlnet.py:123-146)."""
    output = controlnet.load_state_dict(diffusion_model.state_dict(), strict=False)
    if verbose:
        dm_keys = [k for k, _ in diffusion_model.named_parameters() if k not in output.unexpected_keys]
        print(f"Copied weights from {len(dm_keys)} keys of the diffusion model into the ControlNet:"
              f"\n{'; '.join(dm_keys)}\nControlNet missing keys: {len(output.missing_keys)}:"
              f"\n{'; '.join(output.missing_keys)}\nDiffusion model incompatible keys: {len(output.unexpected_keys)}:"
              f"\n{'; '.join(output.unexpected_keys)}")


class ControlNet(_TimestepPath, nn.Module):
    """Drop-in for generative.networks.nets.ControlNet (same arguments, state_dict keys and forward)."""

    def __init__(self, spatial_dims: int, in_channels: int, num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
                 num_channels: Sequence[int] = (32, 64, 64, 64), attention_levels: Sequence[bool] = (False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels: int | Sequence[int] = 8, with_conditioning: bool = False, transformer_num_layers: int = 1,
                 cross_attention_dim: int | None = None, num_class_embeds: int | None = None, upcast_attention: bool = False,
                 use_flash_attention: bool = False, conditioning_embedding_in_channels: int = 1,
                 conditioning_embedding_num_channels: Sequence[int] | None = (16, 32, 96, 256)) -> None:
        super().__init__()
        if with_conditioning is True and cross_attention_dim is None:
            raise ValueError("ControlNet expects dimension of the cross-attention conditioning (cross_attention_dim) "
                             "when using with_conditioning.")
        if cross_attention_dim is not None and with_conditioning is False:
            raise ValueError("ControlNet expects with_conditioning=True when specifying the cross_attention_dim.")
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("ControlNet expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("ControlNet expects num_channels being same size of attention_levels")
        if isinstance(num_head_channels, int):
            num_head_channels = ensure_tuple_rep(num_head_channels, len(attention_levels))
        if len(num_head_channels) != len(attention_levels):
            raise ValueError("num_head_channels should have the same length as attention_levels. For the i levels without "
                             "attention, i.e. `attention_level[i]=False`, the num_head_channels[i] will be ignored.")
        if isinstance(num_res_blocks, int):
            num_res_blocks = ensure_tuple_rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same length as "
                             "`num_channels`.")
        self.spatial_dims = spatial_dims
        self.in_channels = in_channels
        self.block_out_channels = tuple(num_channels)
        self.num_res_blocks = tuple(num_res_blocks)
        self.attention_levels = tuple(attention_levels)
        self.num_head_channels = tuple(num_head_channels)
        self.with_conditioning = with_conditioning
        self.num_class_embeds = num_class_embeds
        nlev = len(num_channels)
        ted = num_channels[0] * 4
        g, eps = norm_num_groups, norm_eps
        common = dict(cond=with_conditioning, nlayers=transformer_num_layers, cross_dim=cross_attention_dim, upcast=upcast_attention,
                      dropout=0.0)

        self.conv_in = ConvP(spatial_dims, in_channels, num_channels[0], 3, 1, 1)
        self.time_embed = nn.Sequential(nn.Linear(num_channels[0], ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_class_embeds is not None:
            self.class_embedding = nn.Embedding(num_class_embeds, ted)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(spatial_dims, conditioning_embedding_in_channels,
                                                                         num_channels[0], conditioning_embedding_num_channels)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList()
        out_c = num_channels[0]
        # the reference registers the FIRST zero conv as the bare nn.ConvNd (keys `controlnet_down_blocks.0.weight`), every later
        # one as the Convolution wrapper (`controlnet_down_blocks.k.conv.weight`): controlnet.py:277-287 vs :320-345
        self.controlnet_down_blocks.append(zero_module(_CONV[spatial_dims](out_c, out_c, 1)))
        for i in range(nlev):
            in_c, out_c = out_c, num_channels[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(num_res_blocks[i])]
            self.down_blocks.append(_Stage(spatial_dims, io, ted, g, eps, attention_levels[i], heads_ch=num_head_channels[i],
                                           resampler=None if i == nlev - 1 else "downsampler", resblock_updown=resblock_updown,
                                           out_channels=out_c, **common))
            for _ in range(num_res_blocks[i] + (0 if i == nlev - 1 else 1)):
                self.controlnet_down_blocks.append(zero_module(ConvP(spatial_dims, out_c, out_c, 1, 1, 0)))
        self.middle_block = _MidBlock(spatial_dims, num_channels[-1], ted, g, eps, with_conditioning, num_head_channels[-1],
                                      transformer_num_layers, cross_attention_dim, upcast_attention, 0.0)
        self.controlnet_mid_block = zero_module(ConvP(spatial_dims, out_c, out_c, 1, 1, 0))

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, controlnet_cond: torch.Tensor, conditioning_scale: float = 1.0,
                context: torch.Tensor | None = None, class_labels: torch.Tensor | None = None):
        """-> (tuple of down-block residuals, mid-block residual), NC[D]HW, ready for DiffusionModelUNet.forward's
        `down_block_additional_residuals` / `mid_block_additional_residual` (reference controlnet.py:367-436)."""
        if timesteps.ndim != 1:
            raise ValueError("Timesteps should be a 1d-array")
        if context is not None and self.with_conditioning is False:
            raise ValueError("model should have with_conditioning = True if context is provided")
        ops.require_device(x, controlnet_cond)
        if wants_grad(self, x) or (torch.is_grad_enabled() and controlnet_cond.requires_grad):
            return self.forward_train(x, timesteps, controlnet_cond, conditioning_scale, context, class_labels)
        x = ops.entry_cast(x, self.conv_in.conv.weight.dtype, "input")
        controlnet_cond = ops.entry_cast(controlnet_cond, self.conv_in.conv.weight.dtype, "conditioning image")
        dtype = x.dtype
        with torch.no_grad():
            rows = self._temb_rows(timesteps.to(x.device), class_labels)
            if context is not None:
                ops.require_device(context)
                context = ops.cast(context.contiguous(), dtype)
            temb = lambda blk: rows[id(blk)]
            emb = self.controlnet_cond_embedding.run(ops.to_channels_last(controlnet_cond))
            h = self.conv_in.run(ops.to_channels_last(x), res=emb, want_stats=True)
            skips = [h]
            for st in self.down_blocks:
                for j, rb in enumerate(st.resnets):
                    h = rb.run(h, temb(rb))
                    h = st.attend(j, h, context)
                    skips.append(h)
                if st.resampler_name == "downsampler":
                    ds = st.downsampler
                    h = ds.run(h, temb(ds)) if isinstance(ds, ResnetBlock) else ds.run(h)
                    skips.append(h)
            mb = self.middle_block
            h = mb.resnet_1.run(h, temb(mb.resnet_1))
            h = mb.attention.run(h, context) if mb.cond else mb.attention.run(h)
            h = mb.resnet_2.run(h, temb(mb.resnet_2))

            def zero_conv(blk, t):
                conv = blk.conv if isinstance(blk, ConvP) else blk
                y = ops.conv(t, conv.weight, conv.bias, kernel=1)
                return y if conditioning_scale == 1.0 else ops.scale(y, float(conditioning_scale), False)

            down = tuple(ops.to_channels_first(zero_conv(blk, s)) for s, blk in zip(skips, self.controlnet_down_blocks))
            mid = ops.to_channels_first(zero_conv(self.controlnet_mid_block, h))
            return down, mid

    def forward_train(self, x: torch.Tensor, timesteps: torch.Tensor, controlnet_cond: torch.Tensor, conditioning_scale: float = 1.0,
                      context: torch.Tensor | None = None, class_labels: torch.Tensor | None = None):
        """`forward` with gradients (reference: torch autograd through controlnet.py:367-436 -- the ControlNet tutorials train this network
        against a frozen DiffusionModelUNet): the conditioning embedding, the UNet encoder half it shares with DiffusionModelUNet
        (`_train_encoder`) and the zero convolutions, native kernels in both directions (generativemodels_amd.autograd).  Returns the
        differentiable (down-block residuals, mid-block residual) that DiffusionModelUNet.forward takes."""
        from ... import autograd as A

        if context is not None and self.with_conditioning is False:
            raise ValueError("model should have with_conditioning = True if context is provided")
        x, dtype = _train_entry(self, x)
        controlnet_cond, _ = _train_entry(self, controlnet_cond, "conditioning image")
        emb = _train_timestep_embedding(self, timesteps, class_labels, dtype, x.device, x.shape[0])
        if context is not None:
            ops.require_device(context)
            context = ops.cast(context.contiguous(), dtype)
        ce = self.controlnet_cond_embedding.run_train(A.to_arena(controlnet_cond))
        ci = self.conv_in.conv
        h = A.conv(A.to_arena(x), ci.weight, ci.bias, kernel=3, stride=1, padding=1, res=ce)
        skips, h = _train_encoder(self, h, emb, context)

        def zero_conv(blk, t):
            conv = blk.conv if isinstance(blk, ConvP) else blk
            return A.from_arena(A.scale(A.conv(t, conv.weight, conv.bias, kernel=1), float(conditioning_scale)))

        down = tuple(zero_conv(blk, s) for s, blk in zip(skips, self.controlnet_down_blocks))
        return down, zero_conv(self.controlnet_mid_block, h)

    def supports_training(self) -> bool:
        return True
