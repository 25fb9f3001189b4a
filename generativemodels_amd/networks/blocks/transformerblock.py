"""Pre-norm transformer block: x + attn(LN(x)); [x + cross_attn(LN(x), context)]; x + MLP(LN(x)) -- parameter names and forward
contract of the reference's generative/networks/blocks/transformerblock.py:20-92 (MLP = MONAI MLPBlock: Linear, GELU, Linear).
Residual adds are GEMM epilogues, GELU is the first MLP GEMM's epilogue."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from .selfattention import SABlock


class MLPBlock(nn.Module):
    """monai.networks.blocks.MLPBlock(hidden, mlp_dim, dropout, act="GELU") parameter layout: linear1, linear2."""

    def __init__(self, hidden_size: int, mlp_dim: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        self.linear1 = nn.Linear(hidden_size, mlp_dim or hidden_size)
        self.linear2 = nn.Linear(mlp_dim or hidden_size, hidden_size)

    def run(self, x_norm: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        a = ops.linear(x_norm, self.linear1.weight, self.linear1.bias, post_act="gelu")
        return ops.linear(a, self.linear2.weight, self.linear2.bias, res=residual)


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size: int, mlp_dim: int, num_heads: int, dropout_rate: float = 0.0, qkv_bias: bool = False,
                 causal: bool = False, sequence_length: int | None = None, with_cross_attention: bool = False,
                 use_flash_attention: bool = False) -> None:
        super().__init__()
        self.with_cross_attention = with_cross_attention
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, causal, sequence_length)
        self.norm2 = None
        self.cross_attn = None
        if with_cross_attention:
            self.norm2 = nn.LayerNorm(hidden_size)
            self.cross_attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, causal=False, with_cross_attention=True)
        self.norm3 = nn.LayerNorm(hidden_size)
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)

    @staticmethod
    def _ln(norm: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
        return ops.layernorm(x, norm.weight, norm.bias, norm.eps)

    def run(self, x: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.attn.run(self._ln(self.norm1, x), x)
        if self.with_cross_attention:
            x = self.cross_attn.run(self._ln(self.norm2, x), x, context)
        return self.mlp.run(self._ln(self.norm3, x), x)

    def run_step(self, x: torch.Tensor, cache: dict, pos: int, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.attn.run_step(self._ln(self.norm1, x), x, cache, pos)
        if self.with_cross_attention:
            x = self.cross_attn.run(self._ln(self.norm2, x), x, context)
        return self.mlp.run(self._ln(self.norm3, x), x)

    def forward(self, x: torch.Tensor, context: torch.Tensor | None = None) -> torch.Tensor:
        ops.require_device(x, context)
        with torch.no_grad():
            return self.run(x.contiguous(), None if context is None else context.contiguous())
