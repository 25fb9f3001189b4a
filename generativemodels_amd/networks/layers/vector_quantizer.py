"""Vector quantiser: constructor / buffers / method contract of the reference's
generative/networks/layers/vector_quantizer.py:20-228 (`EMAQuantizer`, `VectorQuantizer`).

Nearest-code search and the codebook lookup run in HIP (gm_vq_argmin / gm_vq_gather) directly on NC[D]HW inputs (converted
once to the N[D]HWC arena, which makes the reference's permute+flatten free).  Distances are fp32 for every storage dtype
(vector_quantizer.py:102-103).  In train() mode the forward also performs the EMA codebook update (vector_quantizer.py:166-180):
per-code token counts and vector sums in ONE flat buffer (gm_vq_ema_stats) -- exchanged between data-parallel ranks by ONE all-reduce
where the reference issues two (:155-157) -- then the decayed update + Laplace smoothing + embedding rewrite (gm_vq_ema_update); the
one-hot matrix of the reference is never materialised.  The straight-through estimator and the commitment loss are differentiable in
the input (an autograd.Function with a latent-sized backward)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.nn as nn

from ... import ops

__all__ = ["EMAQuantizer", "VectorQuantizer"]


class EMAQuantizer(nn.Module):
    def __init__(self, spatial_dims: int, num_embeddings: int, embedding_dim: int, commitment_cost: float = 0.25,
                 decay: float = 0.99, epsilon: float = 1e-5, embedding_init: str = "normal", ddp_sync: bool = True):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise ValueError(f"EMAQuantizer only supports 4D and 5D tensor inputs but received spatial dims {spatial_dims}.")
        self.spatial_dims = spatial_dims
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        if embedding_init == "kaiming_uniform":
            nn.init.kaiming_uniform_(self.embedding.weight.data, mode="fan_in", nonlinearity="linear")
        self.embedding.weight.requires_grad = False
        self.commitment_cost = commitment_cost
        self.register_buffer("ema_cluster_size", torch.zeros(num_embeddings))
        self.register_buffer("ema_w", self.embedding.weight.data.clone())
        self.decay, self.epsilon, self.ddp_sync = decay, epsilon, ddp_sync
        self.flatten_permutation: Sequence[int] = [0] + list(range(2, spatial_dims + 2)) + [1]
        self.quantization_permutation: Sequence[int] = [0, spatial_dims + 1] + list(range(1, spatial_dims + 1))

    # arena-level entry points used by VQVAE ------------------------------------------------------------------------------
    def indices_of(self, z_arena: torch.Tensor) -> torch.Tensor:
        return ops.vq_argmin(z_arena, self.embedding.weight)

    def lookup(self, indices: torch.Tensor, dtype: torch.dtype, x_arena=None):
        return ops.vq_gather(indices, self.embedding.weight, dtype, x_arena)

    # reference-shaped API (NC[D]HW tensors) --------------------------------------------------------------------------------
    def quantize(self, inputs: torch.Tensor):
        """-> (flat_input [tokens, D], None, encoding_indices (N, *spatial)); the one-hot matrix of the reference
        (vector_quantizer.py:117) is never materialised in eval mode, hence None."""
        ops.require_device(inputs)
        za = ops.to_channels_last(inputs)
        return za.reshape(-1, self.embedding_dim), None, self.indices_of(za)

    def embed(self, embedding_indices: torch.Tensor) -> torch.Tensor:
        ops.require_device(embedding_indices)
        with torch.no_grad():
            return ops.to_channels_first(self.lookup(embedding_indices, self.embedding.weight.dtype))

    @torch.no_grad()
    def ema_update(self, z_arena: torch.Tensor, indices: torch.Tensor) -> None:
        """The training-mode codebook update (vector_quantizer.py:166-180) from the arena view of the inputs and their code indices."""
        k, d = self.num_embeddings, self.embedding_dim
        stats = torch.empty(k + k * d, dtype=torch.float32, device=z_arena.device)
        idx = indices.reshape(-1).contiguous()
        work = torch.empty(int(ops.lib().gm_vq_ema_stats_workspace_elems(idx.numel(), k, d)), dtype=torch.float32, device=z_arena.device)
        ops.check(ops.lib().gm_vq_ema_stats(z_arena.data_ptr(), ops.arena_ld(z_arena), idx.data_ptr(), idx.numel(), k, d, stats.data_ptr(),
                                            work.data_ptr(), ops.dt_code(z_arena.dtype), ops._stream()), "gm_vq_ema_stats")
        if self.ddp_sync and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)  # counts and vector sums together: one exchange
        # the kernels update fp32 state in place; a module cast to bf16 keeps fp32 master copies of its three tables for the update
        bufs = [self.ema_cluster_size, self.ema_w, self.embedding.weight]
        f32 = [b if b.dtype == torch.float32 else b.float() for b in bufs]
        ops.check(ops.lib().gm_vq_ema_update(stats.data_ptr(), f32[0].data_ptr(), f32[1].data_ptr(), f32[2].data_ptr(), k, d, float(self.decay),
                                             float(self.epsilon), ops._stream()), "gm_vq_ema_update")
        for b, f in zip(bufs, f32):
            if f is not b:
                b.copy_(f)
        ops.invalidate_param_cache(self.embedding.weight)

    def forward(self, inputs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (quantized, commitment loss, indices) (vector_quantizer.py:161-188).  eval(): no autograd, the straight-through expression
        x + (q - x).detach() is returned as q itself (identical up to one rounding).  train(): the codes are looked up BEFORE the EMA
        update rewrites the embedding (as in the reference, :162-163 precede :166-180), the update runs, and (quantized, loss) are
        differentiable in the input: d quantized / d x = identity, d loss / d x = 2 * commitment_cost * (x - q) / numel."""
        ops.require_device(inputs)
        if self.training:
            with torch.no_grad():
                za = ops.to_channels_last(inputs.detach())
                idx = self.indices_of(za)
                q, mse = self.lookup(idx, inputs.dtype, za)
                self.ema_update(za, idx)
                q = ops.to_channels_first(q)
                loss = self.commitment_cost * mse.to(inputs.dtype)
            if torch.is_grad_enabled() and inputs.requires_grad:
                return (*_StraightThrough.apply(inputs, q, loss, float(self.commitment_cost)), idx)
            return q, loss, idx
        with torch.no_grad():
            za = ops.to_channels_last(inputs)
            idx = self.indices_of(za)
            q, mse = self.lookup(idx, inputs.dtype, za)
            loss = self.commitment_cost * mse.to(inputs.dtype)
            return ops.to_channels_first(q), loss, idx


class _StraightThrough(torch.autograd.Function):
    """(quantized, loss) as functions of the input x: quantized = x + (q - x).detach(), loss = cc * mse(q.detach(), x)."""

    @staticmethod
    def forward(ctx, x, q, loss, cc):
        ctx.save_for_backward(x, q)
        ctx.cc = cc
        return q.clone(), loss.clone()

    @staticmethod
    def backward(ctx, gq, gloss):
        x, q = ctx.saved_tensors
        dx = gq
        if gloss is not None:  # latent-sized element-wise expression: left to torch, like autograd._Embedding
            dx = dx + gloss * (2.0 * ctx.cc / x.numel()) * (x - q)
        return dx, None, None, None


class VectorQuantizer(nn.Module):
    """AMP-isolation wrapper of the reference (vector_quantizer.py:191-228): forward -> (loss, quantized)."""

    def __init__(self, quantizer: nn.Module = None):
        super().__init__()
        self.quantizer = quantizer
        self.perplexity = torch.rand(1)

    def forward(self, inputs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        quantized, loss, idx = self.quantizer(inputs)
        k = self.quantizer.num_embeddings
        probs = torch.bincount(idx.reshape(-1), minlength=k).float().div(idx.numel())  # code-usage statistic, not on the data path
        self.perplexity = torch.exp(-torch.sum(probs * torch.log(probs + 1e-10)))
        return loss, quantized

    def embed(self, embedding_indices: torch.Tensor) -> torch.Tensor:
        return self.quantizer.embed(embedding_indices=embedding_indices)

    def quantize(self, encodings: torch.Tensor) -> torch.Tensor:
        _, _, idx = self.quantizer(encodings)
        return idx
