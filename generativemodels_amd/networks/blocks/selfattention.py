"""Self / cross attention block of the autoregressive transformer: constructor, parameter names and forward contract of the
reference's generative/networks/blocks/selfattention.py:28-148 (q/k/v Linear, scaled scores, optional causal mask, softmax,
out_proj), run as one stacked projection GEMM + the flash-attention kernel (causal flag instead of an L x L mask) + one GEMM with
the residual in its epilogue.  `run_step` is the incremental form over a KV cache used by VQVAETransformerInferer.sample."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from ... import ops


class SABlock(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, dropout_rate: float = 0.0, qkv_bias: bool = False, causal: bool = False,
                 sequence_length: int | None = None, with_cross_attention: bool = False, use_flash_attention: bool = False) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        if causal and sequence_length is None:
            raise ValueError("sequence_length is necessary for causal attention.")
        self.hidden_size, self.num_heads = hidden_size, num_heads
        self.head_dim = hidden_size // num_heads
        self.scale = 1.0 / math.sqrt(self.head_dim)
        self.causal, self.sequence_length = causal, sequence_length
        self.with_cross_attention = with_cross_attention
        self.dropout_rate = dropout_rate  # inference path: dropout is the identity
        self.to_q = nn.Linear(hidden_size, hidden_size, bias=qkv_bias)
        self.to_k = nn.Linear(hidden_size, hidden_size, bias=qkv_bias)
        self.to_v = nn.Linear(hidden_size, hidden_size, bias=qkv_bias)
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        # The reference keeps the causal mask as a persistent (1, 1, L, L) buffer (selfattention.py:90-94): 64 MB per block at
        # L = 4096.  The kernel masks by index instead; the key is emitted on state_dict() and dropped on load so that checkpoints
        # stay interchangeable with the reference in both directions.
        if causal:
            self._register_state_dict_hook(SABlock._emit_mask)
            self._register_load_state_dict_pre_hook(SABlock._drop_mask)

    @staticmethod
    def _emit_mask(module, state_dict, prefix, local_metadata):
        n = module.sequence_length
        state_dict[prefix + "causal_mask"] = torch.tril(torch.ones(n, n)).view(1, 1, n, n)

    @staticmethod
    def _drop_mask(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "causal_mask", None)

    def _bias3(self, device):
        if self.to_q.bias is None:
            return None
        c = self.hidden_size
        return ops.cat_f32([self.to_q.bias, self.to_k.bias, self.to_v.bias], [c, c, c], device)

    def run(self, x_norm: torch.Tensor, residual: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        """residual + out_proj(attention(x_norm [, context])) for (B, T, C) arena tensors."""
        c = self.hidden_size
        if context is None:
            w = ops.packed_cat_weight([self.to_q.weight, self.to_k.weight, self.to_v.weight], x_norm.dtype)
            qkv = ops.conv(x_norm, None, self._bias3(x_norm.device), kernel=1, packed=w, cout=3 * c)
            q, k, v = qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:3 * c]
        else:
            q = ops.linear(x_norm, self.to_q.weight, self.to_q.bias)
            wkv = ops.packed_cat_weight([self.to_k.weight, self.to_v.weight], x_norm.dtype)
            bkv = None if self.to_k.bias is None else ops.cat_f32([self.to_k.bias, self.to_v.bias], [c, c], x_norm.device)
            kv = ops.conv(context, None, bkv, kernel=1, packed=wkv, cout=2 * c)
            k, v = kv[..., 0:c], kv[..., c:2 * c]
        y = ops.attention(q, k, v, self.num_heads, self.scale, causal=self.causal and context is None)
        return ops.linear(y, self.out_proj.weight, self.out_proj.bias, res=residual)

    def run_step(self, x_norm: torch.Tensor, residual: torch.Tensor, cache: dict, pos: int) -> torch.Tensor:
        """One new token per sequence: x_norm (B, 1, C); its key / value rows are appended to cache["k"], cache["v"]
        ((B, max_len, C) buffers) at `pos`, attention runs over rows 0..pos."""
        c = self.hidden_size
        b = x_norm.shape[0]
        w = ops.packed_cat_weight([self.to_q.weight, self.to_k.weight, self.to_v.weight], x_norm.dtype)
        qkv = ops.conv(x_norm, None, self._bias3(x_norm.device), kernel=1, packed=w, cout=3 * c)
        for i in range(b):
            ops.copy_channels(qkv[i, :, c:2 * c], cache["k"][i, pos:pos + 1])
            ops.copy_channels(qkv[i, :, 2 * c:3 * c], cache["v"][i, pos:pos + 1])
        y = ops.attention(qkv[..., 0:c], cache["k"][:, :pos + 1], cache["v"][:, :pos + 1], self.num_heads, self.scale)
        return ops.linear(y, self.out_proj.weight, self.out_proj.bias, res=residual)

    def forward(self, x: torch.Tensor, context: torch.Tensor | None = None) -> torch.Tensor:
        ops.require_device(x, context)
        with torch.no_grad():
            zero = torch.zeros_like(x)
            return self.run(x.contiguous(), zero, None if context is None else context.contiguous())
