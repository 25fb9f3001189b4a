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

import ctypes as C

from ... import ops
from ..._native import GmDecodeBlock, GmDecodeDesc, check, lib
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
        self.native_step = True  # False: issue the decode step op by op from Python (same kernels; kept for tests)

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

    def _native_table(self, cache: list):
        """Block-parameter table of gm_transformer_decode_step for this cache (built once per cache; keeps every tensor it points to)."""
        tab = cache[0].get("_native")
        if tab is not None:
            return tab
        dt = self.to_logits.weight.dtype
        dev = cache[0]["k"].device
        c = self.attn_layers_dim
        keep = []

        def f32(p):
            t = ops.as_f32(p) if p is not None else None
            keep.append(t)
            return None if t is None else t.data_ptr()

        def packed(w):
            t = ops.packed_conv_weight(w.reshape(w.shape[0], w.shape[1], 1), dt)
            keep.append(t)
            return t.data_ptr()

        blocks = (GmDecodeBlock * len(self.blocks))()
        for blk, cb, ent in zip(self.blocks, blocks, cache):
            a = blk.attn
            wqkv = ops.packed_cat_weight([a.to_q.weight, a.to_k.weight, a.to_v.weight], dt)
            bqkv = a._bias3(dev)
            keep.extend([wqkv, bqkv])
            cb.ln1_g, cb.ln1_b = f32(blk.norm1.weight), f32(blk.norm1.bias)
            cb.w_qkv, cb.b_qkv = wqkv.data_ptr(), None if bqkv is None else bqkv.data_ptr()
            cb.w_o, cb.b_o = packed(a.out_proj.weight), f32(a.out_proj.bias)
            cb.ln3_g, cb.ln3_b = f32(blk.norm3.weight), f32(blk.norm3.bias)
            cb.w_1, cb.b_1 = packed(blk.mlp.linear1.weight), f32(blk.mlp.linear1.bias)
            cb.w_2, cb.b_2 = packed(blk.mlp.linear2.weight), f32(blk.mlp.linear2.bias)
            cb.k_cache, cb.v_cache = ent["k"].data_ptr(), ent["v"].data_ptr()
        b = cache[0]["k"].shape[0]
        m = self.blocks[0].mlp.linear1.weight.shape[0]
        d = GmDecodeDesc()
        d.B, d.C, d.M, d.heads, d.depth = b, c, m, self.attn_layers_heads, len(self.blocks)
        d.max_len, d.num_tokens, d.dtype, d.ln_eps = self.max_seq_len, self.num_tokens, ops.dt_code(dt), float(self.blocks[0].norm1.eps)
        tok, pos = self.token_embeddings.weight.detach().contiguous(), self.position_embeddings.embedding.weight.detach().contiguous()
        d.tok_emb, d.pos_emb = tok.data_ptr(), pos.data_ptr()
        d.blocks = blocks
        d.w_logits, d.b_logits = packed(self.to_logits.weight), f32(self.to_logits.bias)
        nbytes = lib().gm_decode_scratch_bytes(b, c, m, d.dtype)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d.scratch, d.scratch_bytes = scratch.data_ptr(), nbytes
        keep.extend([tok, pos, scratch, blocks])
        tab = dict(desc=d, keep=keep)
        cache[0]["_native"] = tab
        return tab

    def step_from_device_state(self, tokens: torch.Tensor, pos_dev: torch.Tensor, cache: list, logits: torch.Tensor) -> None:
        """The native decode step with the position read from `pos_dev` (int32, 1 element, on the device) at run time and the logits
        written into the caller's (B, num_tokens) buffer: every argument is a fixed device address, so the call can be captured into
        a HIP graph and replayed once per token (VQVAETransformerInferer.sample).  No cross attention."""
        if self.with_cross_attention:
            raise ValueError("the graph-replayable step covers models without cross attention")
        ops.require_device(tokens, pos_dev, logits)
        if tokens.dtype != torch.long or pos_dev.dtype != torch.int32 or logits.dtype != self.to_logits.weight.dtype or not logits.is_contiguous():
            raise TypeError("tokens int64, pos_dev int32, logits in the model dtype (contiguous)")
        d = self._native_table(cache)["desc"]
        d.tokens, d.logits, d.pos, d.pos_dev = tokens.data_ptr(), logits.data_ptr(), 0, pos_dev.data_ptr()
        try:
            check(lib().gm_transformer_decode_step(C.byref(d), ops._stream()), "gm_transformer_decode_step")
        finally:
            d.pos_dev = None

    def step(self, tokens: torch.Tensor, pos: int, cache: list, context: torch.Tensor | None = None) -> torch.Tensor:
        """Logits (B, num_tokens) for the token at position `pos` given `tokens` (B, 1) = that token and a cache holding positions
        0..pos-1; equal to forward(prefix)[:, -1] (pinned by tests)."""
        self._check(tokens, context)
        if tokens.shape[1] != 1 or not 0 <= pos < self.max_seq_len:
            raise ValueError("step takes one token per sequence at a position inside the context window")
        with torch.no_grad():
            if not self.with_cross_attention and self.native_step:
                # the whole token step (38-62 launches) is enqueued by one native call: 20 us of interpreter work per launch made
                # the per-op path below as slow as recomputing the prefix (tools/bench_c5.py)
                d = self._native_table(cache)["desc"]
                tokens = tokens.contiguous()
                logits = torch.empty((tokens.shape[0], self.num_tokens), dtype=self.to_logits.weight.dtype, device=tokens.device)
                d.tokens, d.logits, d.pos = tokens.data_ptr(), logits.data_ptr(), int(pos)
                check(lib().gm_transformer_decode_step(C.byref(d), ops._stream()), "gm_transformer_decode_step")
                return logits
            h = ops.embed_tokens(tokens, self.token_embeddings.weight, self.position_embeddings.embedding.weight, pos)
            ctx = self._context(context)
            for blk, c in zip(self.blocks, cache):
                h = blk.run_step(h, c, pos, ctx)
            return ops.linear(h, self.to_logits.weight, self.to_logits.bias)[:, 0]
