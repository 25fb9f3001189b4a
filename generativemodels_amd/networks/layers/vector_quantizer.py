"""Vector quantiser (inference): constructor / buffers / method contract of the reference's
generative/networks/layers/vector_quantizer.py:20-228 (`EMAQuantizer`, `VectorQuantizer`).

Nearest-code search and the codebook lookup run in HIP (gm_vq_argmin / gm_vq_gather) directly on NC[D]HW inputs (converted
once to the N[D]HWC arena, which makes the reference's permute+flatten free).  Distances are fp32 for every storage dtype
(vector_quantizer.py:102-103).  The EMA codebook update (training only, vector_quantizer.py:140-180) is not part of the
sampling path and raises."""
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

    def forward(self, inputs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (quantized, commitment loss, indices).  Eval semantics of vector_quantizer.py:161-188: the straight-through
        expression x + (q - x).detach() is returned as q itself (identical up to one rounding, no autograd here)."""
        if self.training:
            raise RuntimeError("the EMA codebook update (training) is outside the MI355X sampling path; call .eval()")
        ops.require_device(inputs)
        with torch.no_grad():
            za = ops.to_channels_last(inputs)
            idx = self.indices_of(za)
            q, mse = self.lookup(idx, inputs.dtype, za)
            loss = self.commitment_cost * mse.to(inputs.dtype)
            return ops.to_channels_first(q), loss, idx


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
