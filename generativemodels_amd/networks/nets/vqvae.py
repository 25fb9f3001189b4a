"""VQVAE for MI355X: constructor, state_dict names and encode / quantize / decode / index_quantize / decode_samples /
forward contract of the reference's generative/networks/nets/vqvae.py:274-455, on the fused HIP convolution kernel
(strided conv / gather-form transposed conv with the activation and the residual add + ReLU in the epilogue)."""
from __future__ import annotations

from collections.abc import Sequence

import torch
import torch.nn as nn

from ... import ops
from ..layers.vector_quantizer import EMAQuantizer, VectorQuantizer
from ._blocks import ConvP, ensure_tuple_rep, run_stage, wants_grad

__all__ = ["VQVAE"]


class Act:
    """The spellings of monai.networks.layers.Act this file's signatures use (MONAI's factory attribute returns the registered NAME;
    names are matched case-insensitively here, as MONAI's get_act_layer does)."""

    RELU = "relu"
    LEAKYRELU = "leakyrelu"
    TANH = "tanh"
    SIGMOID = "sigmoid"


def _act_name(act) -> str:
    """MONAI `Act[...]` spec (name or (name, kwargs)) -> epilogue activation of the HIP convolution."""
    if act is None:
        return "none"
    name = act[0] if isinstance(act, (tuple, list)) else act
    if isinstance(act, (tuple, list)) and len(act) > 1 and act[1]:
        raise ValueError(f"activation arguments {act[1]} are not supported by the fused HIP epilogue")
    key = str(name).lower()
    if key not in ops.POST_ACT or key == "none":
        raise ValueError(f"activation '{name}' is not supported by the fused HIP epilogue (supported: "
                         f"{sorted(k for k in ops.POST_ACT if k != 'none')})")
    return key


class _ConvAct(nn.Module):
    """MONAI Convolution with adn_ordering="DA" (dropout, activation; the default instance norm is constructed by MONAI but
    never inserted): parameters under `conv.*` (reference vqvae.py:127-163,220-260)."""

    def __init__(self, spatial_dims, cin, cout, kernel, stride, padding, dilation=1, act="none", transposed=False, output_padding=0, dropout=0.0):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.dropout = float(dropout or 0.0)
        self.kernel, self.stride, self.padding, self.dilation = kernel, stride, padding, dilation
        self.transposed, self.output_padding, self.act = transposed, output_padding, act
        holder = ConvP(spatial_dims, cin, cout, kernel, stride, padding, dilation, transposed, output_padding)
        self.conv = holder.conv  # flat `conv.weight` naming, as MONAI's Convolution registers it

    def run(self, x, **fusion):
        return ops.conv(x, self.conv.weight, self.conv.bias, kernel=self.kernel, stride=self.stride, padding=self.padding,
                        dilation=self.dilation, transposed=self.transposed, output_padding=self.output_padding,
                        post_act=fusion.pop("post_act", self.act), **fusion)

    def run_train(self, x):
        """The same layer with gradients (generativemodels_amd.autograd): convolution / transposed convolution with the activation in its epilogue."""
        from ... import autograd as A

        fused = _fused_epilogue_trains(self.act, self.dropout, self.training)
        post = self.act if fused else "none"
        if self.dilation != 1:  # (round 5) dilated resampling convolutions: gradients composed per tap from the 1x1 weight-gradient kernel
            if self.transposed:
                z = A.conv_transpose_dilated(x, self.conv.weight, self.conv.bias, kernel=self.kernel, stride=self.stride, padding=self.padding,
                                             output_padding=self.output_padding, dilation=self.dilation, post_act=post)
            else:
                z = A.conv_dilated(x, self.conv.weight, self.conv.bias, kernel=self.kernel, stride=self.stride, padding=self.padding,
                                   dilation=self.dilation, post_act=post)
            return z if fused else _dropout_act(z, self.dropout, self.training, self.act)
        if self.transposed:
            z = A.conv_transpose(x, self.conv.weight, self.conv.bias, kernel=self.kernel, stride=self.stride, padding=self.padding,
                                 output_padding=self.output_padding, post_act=post)
        else:
            z = A.conv(x, self.conv.weight, self.conv.bias, kernel=self.kernel, stride=self.stride, padding=self.padding, post_act=post)
        return z if fused else _dropout_act(z, self.dropout, self.training, self.act)


def _fused_epilogue_trains(act: str, dropout: float, training: bool) -> bool:
    """The activation can ride in the convolution's epilogue during training when nothing sits between the two (MONAI's ADN ordering "DA" puts
    the dropout there: vqvae.py:61-80,127-150) and its derivative is a function of the output's sign (the epilogue keeps only the output)."""
    return act in ("none", "relu") and not (training and dropout > 0.0)


def _dropout_act(z, dropout: float, training: bool, act: str):
    """Dropout (torch's op: the draw must be torch's) then the activation with its backward from the pre-activation (gm_activation)."""
    from ... import autograd as A

    if training and dropout > 0.0:
        z = torch.nn.functional.dropout(z, dropout, True)
    return A.activation(z, act)


class VQVAEResidualUnit(nn.Module):
    """relu(x + conv2(act(conv1(x)))) (reference vqvae.py:27-80): two launches, add + ReLU in the second epilogue."""

    def __init__(self, spatial_dims, num_channels, num_res_channels, act="relu", dropout=0.0) -> None:
        super().__init__()
        self.act = act
        self.dropout = float(dropout or 0.0)
        self.conv1 = ConvP(spatial_dims, num_channels, num_res_channels, 3, 1, 1)
        self.conv2 = ConvP(spatial_dims, num_res_channels, num_channels, 3, 1, 1)

    def run(self, x):
        return self.conv2.run(self.conv1.run(x, post_act=self.act), res=x, post_act="relu")

    def run_train(self, x):
        from ... import autograd as A

        c1, c2 = self.conv1.conv, self.conv2.conv
        fused = _fused_epilogue_trains(self.act, self.dropout, self.training)
        h = A.conv(x, c1.weight, c1.bias, kernel=3, stride=1, padding=1, post_act=self.act if fused else "none")
        if not fused:
            h = _dropout_act(h, self.dropout, self.training, self.act)
        return A.conv(h, c2.weight, c2.bias, kernel=3, stride=1, padding=1, res=x, post_act="relu")


class Encoder(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, num_channels, num_res_layers, num_res_channels,
                 downsample_parameters, dropout, act) -> None:
        super().__init__()
        blocks: list[nn.Module] = []
        for i, c in enumerate(num_channels):
            s, k, d, p = downsample_parameters[i]
            blocks.append(_ConvAct(spatial_dims, in_channels if i == 0 else num_channels[i - 1], c, k, s, p, d, act, dropout=0.0 if i == 0 else dropout))
            blocks += [VQVAEResidualUnit(spatial_dims, c, num_res_channels[i], act, dropout) for _ in range(num_res_layers)]
        blocks.append(_ConvAct(spatial_dims, num_channels[-1], out_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)

    def run(self, x):
        for b in self.blocks:
            x = b.run(x)
        return x

    def run_train(self, x):
        for b in self.blocks:
            x = b.run_train(x)
        return x


class Decoder(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, num_channels, num_res_layers, num_res_channels,
                 upsample_parameters, dropout, act, output_act) -> None:
        super().__init__()
        rc, rr = list(reversed(num_channels)), list(reversed(num_res_channels))
        blocks: list[nn.Module] = [_ConvAct(spatial_dims, in_channels, rc[0], 3, 1, 1)]
        n = len(rc)
        for i in range(n):
            blocks += [VQVAEResidualUnit(spatial_dims, rc[i], rr[i], act, dropout) for _ in range(num_res_layers)]
            s, k, d, p, op = upsample_parameters[i]
            last = i == n - 1
            blocks.append(_ConvAct(spatial_dims, rc[i], out_channels if last else rc[i + 1], k, s, p, d,
                                   (output_act or "none") if last else act, transposed=True, output_padding=op, dropout=0.0 if last else dropout))
        self.blocks = nn.ModuleList(blocks)

    def run(self, x):
        for b in self.blocks:
            x = b.run(x)
        return x

    def run_train(self, x):
        for b in self.blocks:
            x = b.run_train(x)
        return x


class VQVAE(nn.Module):
    """Drop-in for generative.networks.nets.VQVAE (same arguments, state_dict keys and methods).  In train() mode with gradients enabled
    `encode` / `decode` / `forward` are differentiable (native kernels in both directions; the quantiser performs the EMA codebook update and
    passes the gradient straight through, vector_quantizer.py:161-188): the VQ-VAE training loop of the reference's tutorials
    (engines/trainer.py:258-270) runs unchanged -- any activation of the fused epilogue (ReLU, LeakyReLU, tanh, sigmoid, SiLU, GELU) and dropout > 0
    (torch's dropout op between convolution and activation, MONAI's "DA" ordering); not covered: dilated resampling convolutions."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, num_channels: Sequence[int] | int = (96, 96, 192),
                 num_res_layers: int = 3, num_res_channels: Sequence[int] | int = (96, 96, 192),
                 downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1), (2, 4, 1, 1)),
                 upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings: int = 32,
                 embedding_dim: int = 64, embedding_init: str = "normal", commitment_cost: float = 0.25, decay: float = 0.5,
                 epsilon: float = 1e-5, dropout: float = 0.0, act: tuple | str | None = Act.RELU, output_act: tuple | str | None = None, ddp_sync: bool = True,
                 use_checkpointing: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.spatial_dims = in_channels, out_channels, spatial_dims
        self.num_channels = num_channels
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.use_checkpointing = use_checkpointing
        if isinstance(num_res_channels, int):
            num_res_channels = ensure_tuple_rep(num_res_channels, len(num_channels))
        if len(num_res_channels) != len(num_channels):
            raise ValueError("`num_res_channels` should be a single integer or a tuple of integers with the same length as "
                             "`num_channels`.")
        if not all(isinstance(v, (int, Sequence)) for v in downsample_parameters):
            raise ValueError("`downsample_parameters` should be a single tuple of integer or a tuple of tuples.")
        if not all(isinstance(v, (int, Sequence)) for v in upsample_parameters):
            raise ValueError("`upsample_parameters` should be a single tuple of integer or a tuple of tuples.")
        if all(isinstance(v, int) for v in upsample_parameters):
            upsample_parameters = (upsample_parameters,) * len(num_channels)
        if all(isinstance(v, int) for v in downsample_parameters):
            downsample_parameters = (downsample_parameters,) * len(num_channels)
        for p in downsample_parameters:
            if len(p) != 4:
                raise ValueError("`downsample_parameters` should be a tuple of tuples with 4 integers.")
        for p in upsample_parameters:
            if len(p) != 5:
                raise ValueError("`upsample_parameters` should be a tuple of tuples with 5 integers.")
        if len(downsample_parameters) != len(num_channels):
            raise ValueError("`downsample_parameters` should be a tuple of tuples with the same length as `num_channels`.")
        if len(upsample_parameters) != len(num_channels):
            raise ValueError("`upsample_parameters` should be a tuple of tuples with the same length as `num_channels`.")
        self.num_res_layers, self.num_res_channels = num_res_layers, num_res_channels
        a = _act_name(act)
        oa = _act_name(output_act) if output_act else None
        self.encoder = Encoder(spatial_dims, in_channels, embedding_dim, num_channels, num_res_layers, num_res_channels,
                               downsample_parameters, dropout, a)
        self.decoder = Decoder(spatial_dims, embedding_dim, out_channels, num_channels, num_res_layers, num_res_channels,
                               upsample_parameters, dropout, a, oa)
        self.quantizer = VectorQuantizer(quantizer=EMAQuantizer(spatial_dims, num_embeddings, embedding_dim, commitment_cost,
                                                                decay, epsilon, embedding_init, ddp_sync))
        self.dropout = float(dropout)

    def _dtype(self) -> torch.dtype:
        return self.encoder.blocks[0].conv.weight.dtype

    def _check(self, x: torch.Tensor) -> torch.Tensor:
        """-> x in the compute dtype: the parameters' (the input must match), or the active ops.autocast region's (the input is cast)."""
        ops.require_device(x)
        return ops.entry_cast(x, self._dtype())

    def _train_entry(self, x: torch.Tensor) -> torch.Tensor:
        from ... import autograd as A

        ops.require_device(x)
        dt = ops.compute_dtype(self._dtype())
        if x.dtype != dt:
            if ops.autocast_dtype() is None:
                raise TypeError(f"input dtype {x.dtype} does not match the model dtype {self._dtype()}")
            x = A.cast(x, dt)
        return x

    def encode(self, images: torch.Tensor) -> torch.Tensor:
        if wants_grad(self, images):
            from ... import autograd as A

            return A.from_arena(run_stage(self.encoder.run_train, A.to_arena(self._train_entry(images)), self.use_checkpointing))
        images = self._check(images)
        with torch.no_grad():
            return ops.to_channels_first(self.encoder.run(ops.to_channels_last(images)))

    def quantize(self, encodings: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        x_loss, x = self.quantizer(encodings)
        return x, x_loss

    def decode(self, quantizations: torch.Tensor) -> torch.Tensor:
        if wants_grad(self, quantizations):
            from ... import autograd as A

            return A.from_arena(run_stage(self.decoder.run_train, A.to_arena(self._train_entry(quantizations).contiguous()), self.use_checkpointing))
        quantizations = self._check(quantizations)
        with torch.no_grad():
            return ops.to_channels_first(self.decoder.run(ops.to_channels_last(quantizations)))

    def index_quantize(self, images: torch.Tensor) -> torch.Tensor:
        images = self._check(images)
        with torch.no_grad():  # encoder output stays in the arena: no layout round trip before the code search
            return self.quantizer.quantizer.indices_of(self.encoder.run(ops.to_channels_last(images)))

    def decode_samples(self, embedding_indices: torch.Tensor) -> torch.Tensor:
        ops.require_device(embedding_indices)
        with torch.no_grad():
            q = self.quantizer.quantizer.lookup(embedding_indices, ops.compute_dtype(self._dtype()))
            return ops.to_channels_first(self.decoder.run(q))

    def forward(self, images: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        quantizations, quantization_losses = self.quantize(self.encode(images))
        return self.decode(quantizations), quantization_losses

    def encode_stage_2_inputs(self, x: torch.Tensor, quantized: bool = True) -> torch.Tensor:
        z = self.encode(x)
        e, _ = self.quantize(z)
        return e if quantized else z

    def decode_stage_2_outputs(self, z: torch.Tensor) -> torch.Tensor:
        e, _ = self.quantize(z)
        return self.decode(e)
