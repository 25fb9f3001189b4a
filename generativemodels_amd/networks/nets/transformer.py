"""Decoder-only (GPT-style) transformer over VQ-VAE token sequences: constructor, state_dict keys and forward contract of the
reference's generative/networks/nets/transformer.py:20-106.

Beyond the reference's full-sequence `forward`, the module offers an incremental decoding API over a KV cache
(`new_cache` / `step`): the reference's sampling loop re-runs the whole prefix for every new token (O(L^3) attention work,
≈ 390-580 TFLOP for 4096 tokens); with the cache a token costs one row through every GEMM and one
1 x t attention per block."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from ..blocks import TransformerBlock

__all__ = ["DecoderOnlyTransformer", "AbsolutePositionalEmbedding"]


class AbsolutePositionalEmbedding(nn.Module):
    """Learned absolute position embedding (reference transformer.py:20-40)."""

    def __init__(self, max_seq_len: int, embedding_dim: int) -> None:
        super().__init__()
        self.max_seq_len, self.embedding_dim = max_seq_len, embedding_dim
        self.embedding = nn.Embedding(max_seq_len, embedding_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - the fused path uses ops.embed_tokens
        raise RuntimeError("AbsolutePositionalEmbedding is applied by DecoderOnlyTransformer's fused embedding kernel")


class DecoderOnlyTransformer(nn.Module):
    def __init__(self, num_tokens: int, max_seq_len: int, attn_layers_dim: int, attn_layers_depth: int, attn_layers_heads: int,
                 with_cross_attention: bool = False, embedding_dropout_rate: float = 0.0, use_flash_attention: bool = False) -> None:
        super().__init__()
        self.num_tokens, self.max_seq_len = num_tokens, max_seq_len
        self.attn_layers_dim, self.attn_layers_depth, self.attn_layers_heads = attn_layers_dim, attn_layers_depth, attn_layers_heads
        self.with_cross_attention = with_cross_attention
        self.token_embeddings = nn.Embedding(num_tokens, attn_layers_dim)
        self.position_embeddings = AbsolutePositionalEmbedding(max_seq_len=max_seq_len, embedding_dim=attn_layers_dim)
        self.embedding_dropout = nn.Dropout(embedding_dropout_rate)  # identity at inference
        self.blocks = nn.ModuleList([
            TransformerBlock(hidden_size=attn_layers_dim, mlp_dim=attn_layers_dim * 4, num_heads=attn_layers_heads, dropout_rate=0.0,
                             qkv_bias=False, causal=True, sequence_length=max_seq_len, with_cross_attention=with_cross_attention)
            for _ in range(attn_layers_depth)])
        self.to_logits = nn.Linear(attn_layers_dim, num_tokens)

    def _check(self, x: torch.Tensor, context) -> None:
        ops.require_device(x, context)
        if x.dim() != 2 or x.dtype != torch.long:
            raise ValueError("expected (B, T) int64 token indices")
        if x.shape[1] > self.max_seq_len:
            raise ValueError(f"sequence of {x.shape[1]} tokens exceeds max_seq_len {self.max_seq_len}")
        if context is not None and not self.with_cross_attention:
            raise ValueError("context given but the model was built without cross attention")

    def _context(self, context):
        return None if context is None else ops.cast(context.contiguous(), self.to_logits.weight.dtype)

    def forward(self, x: torch.Tensor, context: torch.Tensor | None = None) -> torch.Tensor:
        """(B, T) token indices -> (B, T, num_tokens) logits (reference transformer.py:98-106)."""
        self._check(x, context)
        with torch.no_grad():
            h = ops.embed_tokens(x, self.token_embeddings.weight, self.position_embeddings.embedding.weight, 0)
            ctx = self._context(context)
            for blk in self.blocks:
                h = blk.run(h, ctx)
            return ops.linear(h, self.to_logits.weight, self.to_logits.bias)

    # ---- incremental decoding ------------------------------------------------------------------------------------------------------
    def new_cache(self, batch: int, device) -> list:
        dt = self.to_logits.weight.dtype
        shape = (batch, self.max_seq_len, self.attn_layers_dim)
        return [dict(k=torch.empty(shape, dtype=dt, device=device), v=torch.empty(shape, dtype=dt, device=device)) for _ in self.blocks]

    def step(self, tokens: torch.Tensor, pos: int, cache: list, context: torch.Tensor | None = None) -> torch.Tensor:
        """Logits (B, num_tokens) for the token at position `pos` given `tokens` (B, 1) = that token and a cache holding positions
        0..pos-1; equal to forward(prefix)[:, -1] (pinned by tests)."""
        self._check(tokens, context)
        if tokens.shape[1] != 1 or not 0 <= pos < self.max_seq_len:
            raise ValueError("step takes one token per sequence at a position inside the context window")
        with torch.no_grad():
            h = ops.embed_tokens(tokens, self.token_embeddings.weight, self.position_embeddings.embedding.weight, pos)
            ctx = self._context(context)
            for blk, c in zip(self.blocks, cache):
                h = blk.run_step(h, c, pos, ctx)
            return ops.linear(h, self.to_logits.weight, self.to_logits.bias)[:, 0]
