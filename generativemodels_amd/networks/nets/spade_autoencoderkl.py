"""SPADEAutoencoderKL for MI355X: constructor arguments, state_dict names and encode / decode / forward contract of the reference's
generative/networks/nets/spade_autoencoderkl.py:292-484 -- an AutoencoderKL whose decoder residual blocks are SPADE-modulated by a
semantic segmentation; the encoder and the quantisation convolutions are the plain AutoencoderKL's."""
from __future__ import annotations

from typing import Sequence

import torch

from ... import ops
from ._blocks import wants_grad
from .autoencoderkl import AutoencoderKL, Decoder

__all__ = ["SPADEAutoencoderKL"]


class SPADEAutoencoderKL(AutoencoderKL):
    """Drop-in for generative.networks.nets.SPADEAutoencoderKL (same arguments, state_dict keys and methods)."""

    def __init__(self, spatial_dims: int, label_nc: int, in_channels: int = 1, out_channels: int = 1,
                 num_res_blocks: Sequence[int] | int = (2, 2, 2, 2), num_channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True), latent_channels: int = 3, norm_num_groups: int = 32,
                 norm_eps: float = 1e-6, with_encoder_nonlocal_attn: bool = True, with_decoder_nonlocal_attn: bool = True,
                 use_flash_attention: bool = False, spade_intermediate_channels: int = 128) -> None:
        try:
            super().__init__(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels, num_res_blocks=num_res_blocks,
                             num_channels=num_channels, attention_levels=attention_levels, latent_channels=latent_channels,
                             norm_num_groups=norm_num_groups, norm_eps=norm_eps, with_encoder_nonlocal_attn=with_encoder_nonlocal_attn,
                             with_decoder_nonlocal_attn=with_decoder_nonlocal_attn, use_flash_attention=use_flash_attention)
        except ValueError as e:
            raise ValueError(str(e).replace("AutoencoderKL", "SPADEAutoencoderKL")) from None
        nrb = (num_res_blocks,) * len(num_channels) if isinstance(num_res_blocks, int) else tuple(num_res_blocks)
        self.label_nc = label_nc
        self.decoder = Decoder(spatial_dims, num_channels, latent_channels, out_channels, nrb, norm_num_groups, norm_eps, attention_levels,
                               with_decoder_nonlocal_attn, False, spade=(label_nc, spade_intermediate_channels))

    def _seg(self, seg: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        ops.require_device(seg)
        if seg.shape[1] != self.label_nc or seg.shape[0] != like.shape[0] or seg.dim() != like.dim():
            raise ValueError(f"seg must be (N, {self.label_nc}, *spatial) with the batch size of the input")
        key = (seg.data_ptr(), seg._version, tuple(seg.shape), seg.dtype, like.dtype)
        cached = getattr(self, "_seg_arena", None)
        if cached is None or cached[0] != key:  # kept per segmentation tensor: the SPADE layers cache their maps on the arena copy
            cached = (key, ops.to_channels_last(ops.cast(seg.contiguous(), like.dtype)), seg)
            self._seg_arena = cached
        return cached[1]

    def decode(self, z: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
        """post_quant_conv -> SPADEDecoder (reference spade_autoencoderkl.py:457-469)."""
        z = self._check(z)
        if wants_grad(self, z):  # a training step: differentiable decode, the SPADE map convolutions train with the decoder
            from ... import autograd as A

            pq = self.post_quant_conv.conv
            seg_a = self._seg(seg.detach(), z)
            return A.from_arena(self.decoder.run_train(A.conv(A.to_arena(z.contiguous()), pq.weight, pq.bias, kernel=1), seg_a))
        with torch.no_grad():
            h = self.post_quant_conv.run(ops.to_channels_last(z))
            return ops.to_channels_first(self.decoder.run(h, self._seg(seg, z)))

    def reconstruct(self, x: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
        z_mu, _ = self.encode(x)
        return self.decode(z_mu, seg)

    def forward(self, x: torch.Tensor, seg: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        return self.decode(z, seg), z_mu, z_sigma

    def decode_stage_2_outputs(self, z: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
        return self.decode(z, seg)
