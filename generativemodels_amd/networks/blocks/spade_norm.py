"""SPADE normalisation for MI355X: constructor arguments, sub-module / state_dict names and forward contract of the reference's
generative/networks/blocks/spade_norm.py:20-96 (Park et al. 2019).

    out = param_free_norm(x) * (1 + gamma(seg)) + beta(seg),      gamma = mlp_gamma(actv), beta = mlp_beta(actv),
    actv = LeakyReLU(mlp_shared(nearest_resize(seg, x.spatial)))

As in the reference (MONAI `Convolution(act=None)` keeps its default `norm="INSTANCE"`), the gamma and beta maps are instance-normalised
(affine-free, eps 1e-5) convolution outputs.  MI355X mapping: the two map convolutions run as ONE launch (stacked output channels, the
per-channel statistics of the result fused into its epilogue), the instance norm and the `1 +` are folded into one GroupNorm-apply pass,
and -- because the maps depend on the segmentation and the layer only, not on the timestep -- the finished (1 + gamma, beta) maps are
cached per segmentation tensor: a sampling chain computes them once and every step costs one fused `gm_spade_apply` pass (parameter-
free norm x modulation x SiLU) per SPADE layer."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import ops

__all__ = ["SPADE"]


class SPADE(nn.Module):
    """Drop-in for generative.networks.blocks.spade_norm.SPADE (norm: "INSTANCE" or "GROUP" with `norm_params`)."""

    def __init__(self, label_nc: int, norm_nc: int, kernel_size: int = 3, spatial_dims: int = 2, hidden_channels: int = 64,
                 norm: str | tuple = "INSTANCE", norm_params: dict | None = None) -> None:
        super().__init__()
        norm_params = dict(norm_params or {})
        if isinstance(norm, (tuple, list)):
            norm, norm_params = norm[0], dict(norm[1])
        kind = str(norm).lower()
        self.param_free_norm = nn.Sequential()  # MONAI ADN(ordering="N"): the norm layer is the child "N"
        if kind == "group":
            layer = nn.GroupNorm(num_channels=norm_nc, **norm_params)
            self.groups, self.eps = layer.num_groups, layer.eps
        elif kind == "instance":
            layer = [nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d][spatial_dims - 1](norm_nc, **norm_params)
            if layer.affine or layer.track_running_stats:
                raise NotImplementedError("SPADE: the instance norm is used parameter-free, without running statistics")
            self.groups, self.eps = norm_nc, layer.eps
        else:
            raise NotImplementedError(f"SPADE: base normalisation {norm!r} (the reference networks use GROUP; INSTANCE is the default)")
        self.param_free_norm.add_module("N", layer)
        self.norm_nc, self.kernel_size, self.spatial_dims = norm_nc, kernel_size, spatial_dims
        from ..nets._blocks import ConvP  # deferred: networks.nets imports this package

        pad = kernel_size // 2
        self.mlp_shared = ConvP(spatial_dims, label_nc, hidden_channels, kernel_size, 1, pad)
        self.mlp_gamma = ConvP(spatial_dims, hidden_channels, norm_nc, kernel_size, 1, pad)
        self.mlp_beta = ConvP(spatial_dims, hidden_channels, norm_nc, kernel_size, 1, pad)
        self._cache: Optional[tuple] = None

    # ---- (1 + gamma, beta) maps of a segmentation at one resolution: timestep independent, cached --------------------------------
    def maps(self, seg: torch.Tensor, size, dtype) -> tuple[torch.Tensor, torch.Tensor]:
        """seg: arena tensor (N, *spatial, label_nc).  -> (G, Bm) arena tensors (N, *size, norm_nc), channel slices of one buffer."""
        ws = (self.mlp_shared.conv.weight, self.mlp_shared.conv.bias, self.mlp_gamma.conv.weight, self.mlp_gamma.conv.bias,
              self.mlp_beta.conv.weight, self.mlp_beta.conv.bias)
        key = (seg.data_ptr(), seg._version, tuple(seg.shape), tuple(size), dtype, tuple((w.data_ptr(), w._version) for w in ws))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        c, k, pad = self.norm_nc, self.kernel_size, self.kernel_size // 2
        segr = seg if tuple(seg.shape[1:-1]) == tuple(size) else ops.nearest_resize(seg, size)
        actv = self.mlp_shared.run(segr, post_act="leakyrelu")
        w = ops.packed_cat_weight([self.mlp_gamma.conv.weight, self.mlp_beta.conv.weight], dtype)
        b = ops.cat_f32([self.mlp_gamma.conv.bias, self.mlp_beta.conv.bias], [c, c], seg.device)
        raw = ops.conv(actv, None, b, kernel=k, stride=1, padding=pad, packed=w, cout=2 * c, want_stats=True)
        # InstanceNorm of both maps (eps 1e-5, nn.InstanceNormNd default) with the `1 +` of the modulation folded into gamma's shift
        plus_one = torch.cat([torch.ones(c, dtype=torch.float32, device=seg.device), torch.zeros(c, dtype=torch.float32, device=seg.device)])
        scale, shift = ops.gn_scale_shift_composed(raw, 2 * c, 1e-5, None, plus_one)
        gb = ops.gn_apply(raw, scale, shift, "none")
        out = (gb[..., :c], gb[..., c:])
        self._cache = (key, out, seg)  # the segmentation is kept alive so its data_ptr cannot be recycled under the key
        return out

    def run(self, x, seg: torch.Tensor, act: str = "none") -> torch.Tensor:
        """x: arena tensor or ops.VirtualCat of two; seg: arena segmentation (any resolution); -> act(SPADE(x)) materialised."""
        parts = x.parts if isinstance(x, ops.VirtualCat) else [x]
        n = self.param_free_norm.N
        gamma, beta = (n.weight, n.bias) if getattr(n, "affine", False) else (None, None)
        scale, shift = ops.gn_scale_shift_composed(x, self.groups, self.eps, gamma, beta)
        g, bm = self.maps(seg, tuple(parts[0].shape[1:-1]), parts[0].dtype)
        out = torch.empty((*parts[0].shape[:-1], self.norm_nc), dtype=parts[0].dtype, device=parts[0].device)
        off = 0
        for p in parts:
            c = p.shape[-1]
            ops.spade_apply(p, scale[:, off:off + c], shift[:, off:off + c], g[..., off:off + c], bm[..., off:off + c], act,
                            out=out[..., off:off + c])
            off += c
        return out

    def run_train(self, x: torch.Tensor, seg: torch.Tensor, act: str = "none") -> torch.Tensor:
        """`run` with gradients (generativemodels_amd.autograd; reference: torch autograd through spade_norm.py:79-96): x an arena tensor (a
        concatenation is materialised by the caller), seg the arena segmentation.  The (1 + gamma, beta) maps are recomputed every call -- their
        convolutions are being trained -- : nearest resize -> mlp_shared + LeakyReLU -> mlp_gamma / mlp_beta -> InstanceNorm (+ 1 on gamma), then
        the parameter-free GroupNorm of x and the modulation, every step a differentiable native op."""
        from ... import autograd as A

        c, k, pad = self.norm_nc, self.kernel_size, self.kernel_size // 2
        size = tuple(x.shape[1:-1])
        segr = seg if tuple(seg.shape[1:-1]) == size else ops.nearest_resize(seg, size)  # (the segmentation itself takes no gradient)
        sh_, ga, be = self.mlp_shared.conv, self.mlp_gamma.conv, self.mlp_beta.conv
        actv = A.conv(segr, sh_.weight, sh_.bias, kernel=k, stride=1, padding=pad, post_act="leakyrelu")
        zero = torch.zeros(c, dtype=torch.float32, device=x.device)
        # InstanceNorm (affine-free, eps 1e-5) of each map = GroupNorm with one channel per group; the `1 +` of the modulation is gamma's shift
        g = A.group_norm_act(A.conv(actv, ga.weight, ga.bias, kernel=k, stride=1, padding=pad), None, zero + 1.0, c, 1e-5, "none")
        bm = A.group_norm_act(A.conv(actv, be.weight, be.bias, kernel=k, stride=1, padding=pad), None, zero, c, 1e-5, "none")
        n = self.param_free_norm.N
        gamma, beta = (n.weight, n.bias) if getattr(n, "affine", False) else (None, None)
        xn = A.group_norm_act(x, gamma, beta, self.groups, self.eps, "none")
        return A.spade_modulate(xn, g, bm, act)

    def forward(self, x: torch.Tensor, segmap: torch.Tensor) -> torch.Tensor:
        """NC[D]HW in / out, like the reference module."""
        ops.require_device(x, segmap)
        with torch.no_grad():
            seg = ops.to_channels_last(ops.cast(segmap.contiguous(), x.dtype))
            return ops.to_channels_first(self.run(ops.to_channels_last(x), seg))
