"""SPADEDiffusionModelUNet for MI355X: constructor arguments, sub-module / state_dict names and forward contract of the reference's
generative/networks/nets/spade_diffusion_model_unet.py:612-912 -- the DiffusionModelUNet whose decoder ResnetBlocks are SPADE-modulated
by a semantic segmentation (encoder, mid block, attention, resamplers and output head are the plain UNet's and run on the same fused
HIP kernels; the SPADE layers add one fused modulation pass each, their gamma / beta maps cached per segmentation)."""
from __future__ import annotations

from typing import Sequence

import torch

from .diffusion_model_unet import DiffusionModelUNet

__all__ = ["SPADEDiffusionModelUNet"]


class SPADEDiffusionModelUNet(DiffusionModelUNet):
    """Drop-in for generative.networks.nets.SPADEDiffusionModelUNet (same arguments, state_dict keys and forward)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, label_nc: int,
                 num_res_blocks: Sequence[int] | int = (2, 2, 2, 2), num_channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True), norm_num_groups: int = 32, norm_eps: float = 1e-6,
                 resblock_updown: bool = False, num_head_channels: int | Sequence[int] = 8, with_conditioning: bool = False,
                 transformer_num_layers: int = 1, cross_attention_dim: int | None = None, num_class_embeds: int | None = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False, spade_intermediate_channels: int = 128) -> None:
        torch.nn.Module.__init__(self)
        self._spade = (label_nc, spade_intermediate_channels)
        self.label_nc = label_nc
        try:
            super().__init__(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels, num_res_blocks=num_res_blocks,
                             num_channels=num_channels, attention_levels=attention_levels, norm_num_groups=norm_num_groups,
                             norm_eps=norm_eps, resblock_updown=resblock_updown, num_head_channels=num_head_channels,
                             with_conditioning=with_conditioning, transformer_num_layers=transformer_num_layers,
                             cross_attention_dim=cross_attention_dim, num_class_embeds=num_class_embeds,
                             upcast_attention=upcast_attention, use_flash_attention=use_flash_attention)
        except ValueError as e:  # the reference's messages name the SPADE class
            raise ValueError(str(e).replace("DiffusionModelUNet", "SPADEDiffusionModelUNet")) from None

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, seg: torch.Tensor, context: torch.Tensor | None = None,
                class_labels: torch.Tensor | None = None, down_block_additional_residuals: tuple[torch.Tensor] | None = None,
                mid_block_additional_residual: torch.Tensor | None = None) -> torch.Tensor:
        """x: (N, C, *spatial); seg: (N, label_nc, *spatial) segmentation (any resolution: each SPADE layer resizes it, nearest)."""
        if seg is None:
            raise ValueError("SPADEDiffusionModelUNet needs the segmentation map `seg`")
        if seg.shape[1] != self.label_nc:
            raise ValueError(f"seg has {seg.shape[1]} channels, the network was built for label_nc = {self.label_nc}")
        residuals = list(down_block_additional_residuals or ()) + ([] if mid_block_additional_residual is None else [mid_block_additional_residual])
        if self._wants_grad(x) or (torch.is_grad_enabled() and any(r.requires_grad for r in residuals)):
            return self.forward_train(x, timesteps, context=context, class_labels=class_labels,
                                      down_block_additional_residuals=down_block_additional_residuals,
                                      mid_block_additional_residual=mid_block_additional_residual, seg=seg)
        return self._forward_impl(x, timesteps, context, class_labels, down_block_additional_residuals, mid_block_additional_residual, seg=seg)
