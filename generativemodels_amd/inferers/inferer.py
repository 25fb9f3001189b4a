"""Inferers: the `__call__` (training forward) and `sample` (reverse sampling) contracts of the reference's
generative/inferers/inferer.py:31-143 (DiffusionInferer) and :324-487 (LatentDiffusionInferer).

MI355X-specific execution of `sample`:
  * the timestep values live on the device for the whole chain (one upload instead of one H2D copy per step,
    reference inferer.py:129,133), the host loop only computes the fp32 step scalars;
  * per step: DiffusionModelUNet forward (fused HIP kernels) + ONE fused scheduler-step kernel;
  * optional HIP-graph replay of the UNet forward (`use_hip_graph=True`): the ~200 kernel launches of one forward are
    captured once on the first step and replayed for the remaining steps."""
from __future__ import annotations

import os

import functools
import math
from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import ops
from .._native import GmKlParams
from ..networks.nets import VQVAE, DecoderOnlyTransformer, DiffusionModelUNet, SPADEAutoencoderKL, SPADEDiffusionModelUNet

try:  # progress bar is optional, like in the reference (inferer.py:28)
    from tqdm import tqdm

    has_tqdm = True
except Exception:  # pragma: no cover
    has_tqdm = False


class Inferer:
    """Minimal stand-in for monai.inferers.Inferer (an ABC with `__call__`)."""

    def __call__(self, *args, **kwargs):  # pragma: no cover
        raise NotImplementedError


def _bind_seg(diffusion_model, seg):
    """SPADE networks take the segmentation as a forward argument (reference inferer.py:68-70, 121-123, 193-195: functools.partial)."""
    if isinstance(diffusion_model, SPADEDiffusionModelUNet):
        return functools.partial(diffusion_model, seg=seg)
    return diffusion_model


def _check_mode(mode: str) -> None:
    if mode not in ["crossattn", "concat"]:
        raise NotImplementedError(f"{mode} condition is not supported")


class _GraphedUNet:
    """HIP-graph replay of `model(x, t, context)` for fixed shapes; falls back to nothing -- errors propagate."""

    def __init__(self, model: DiffusionModelUNet, x: torch.Tensor, t: torch.Tensor, context: Optional[torch.Tensor],
                 row: Optional[torch.Tensor] = None):
        # (the model is NOT kept: the cache holds it weakly, so a dropped model releases its captured graph and the graph's private memory pool)
        # row: this step's row of `model.time_rows_table()` -- the captured forward then reads its timestep rows from a static buffer the loop
        # fills per step (one copy) instead of running the embedding + MLP + stacked projection inside the graph (4 launches)
        self.x = x.clone()
        self.t = t.clone()
        self.row = None if row is None else row.clone()
        self.context = None if context is None else context.clone()

        def run():
            if self.row is not None:
                model._time_rows_row = self.row  # consumed by this forward
            return model(self.x, self.t, self.context)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up: packs weights, sets kernel attributes, sizes the allocator
            run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: HIP calls of OTHER host threads (DataLoader pin-memory, the RCCL watchdog, a GradientReducer side stream
        # during in-training validation sampling) do not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = run()

    def __call__(self, x: torch.Tensor, t: torch.Tensor, row: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.x.copy_(x)
        if self.row is not None:
            self.row.copy_(row)
        else:
            self.t.copy_(t)
        self.graph.replay()
        return self.out

    def set_context(self, context: Optional[torch.Tensor]) -> None:
        """Once per sample() call: a cached graph serves later calls, whose conditioning is copied into the captured buffer."""
        if self.context is not None and context is not None:
            self.context.copy_(context)

    @staticmethod
    def signature(model) -> tuple:
        """What a captured forward depends on besides its inputs: the eval / train mode, the autocast compute dtype, and every parameter's and
        buffer's identity, version, STORAGE, dtype and device -- the packed MFMA panels a capture baked in are derived per parameter version (an
        optimizer step between two sample() calls must re-capture), and writes through `.data` (`p.data = ema_p.data`, `module.to()`, `.half()`)
        change the storage without bumping `_version`: the captured kernels would read the old, possibly freed, storage.  Mirrors the `ver` tuple
        of `ops._cached`.  (`p.data.copy_()` in place keeps all of these: call `inferer.clear_graph_cache()` after such a write.)"""
        tensors = list(model.parameters()) + list(model.buffers())
        return (bool(model.training), ops.autocast_dtype()) + tuple((id(p), p._version, p.data_ptr(), p.dtype, p.device) for p in tensors)


class DiffusionInferer(Inferer):
    """Drop-in for generative.inferers.DiffusionInferer. `diffusion_model` may be any callable `(x, timesteps=, context=)`."""

    GRAPH_AUTO_MAX_ELEMENTS = 1 << 19  # use_hip_graph=None: replay a HIP graph when the model input is at most this many elements ...
    GRAPH_AUTO_MIN_STEPS = 20          # ... and the chain is long enough to repay the capture (a warm-up forward, a capture forward, instantiation)

    def __init__(self, scheduler: nn.Module, use_hip_graph: bool | None = None) -> None:
        """use_hip_graph: True / False, or None = decide per call: small problems are host-launch-bound and replaying the forward from
        a HIP graph is 1.4-3x faster (C1b, tools/bench_c1b.py), large ones are GPU-bound and eager launches are faster (C2)."""
        self.scheduler = scheduler
        self.use_hip_graph = use_hip_graph

    def __call__(self, inputs: torch.Tensor, diffusion_model: Callable[..., torch.Tensor], noise: torch.Tensor,
                 timesteps: torch.Tensor, condition: torch.Tensor | None = None, mode: str = "crossattn",
                 seg: torch.Tensor | None = None) -> torch.Tensor:
        """Training forward: noise the inputs at `timesteps` (fused kernel) and predict (reference inferer.py:44-81)."""
        _check_mode(mode)
        noisy = self.scheduler.add_noise(original_samples=inputs, noise=noise, timesteps=timesteps)
        if mode == "concat":
            noisy = ops.concat_dim1([noisy, condition])
            condition = None
        diffusion_model = _bind_seg(diffusion_model, seg)
        if torch.is_grad_enabled() and getattr(diffusion_model, "supports_training", lambda: False)() \
                and any(p.requires_grad for p in diffusion_model.parameters()):
            # a training step (ddpm_training_ddp.py:249-270): the differentiable forward, native kernels in both directions
            return diffusion_model.forward_train(noisy, timesteps, context=condition)
        return diffusion_model(x=noisy, timesteps=timesteps, context=condition)

    BATCHED_TIME_ROWS = os.environ.get("GM_BATCHED_TIME_ROWS", "1") != "0"  # (bench switch: "0" = the embedding + MLP + stacked projection once per step)
    GRAPH_CACHE_SIZE = 2  # captured forwards kept per inferer (a capture costs two eager forwards: ~10 ms for the C3 latent UNet, 7 % of a sample)

    def _cached_graph(self, model, x: torch.Tensor, t: torch.Tensor, ctx: Optional[torch.Tensor], row: Optional[torch.Tensor] = None) -> "_GraphedUNet":
        """The captured forward of `model` for these shapes, reused across sample() calls while the model's parameters are unchanged."""
        import weakref

        cache = self.__dict__.setdefault("_graph_cache", [])
        key = (tuple(x.shape), x.dtype, x.device, tuple(t.shape), None if ctx is None else (tuple(ctx.shape), ctx.dtype), row is not None)
        sig = _GraphedUNet.signature(model)
        for ent in cache:
            if ent[0]() is model and ent[1] == key and ent[2] == sig:
                return ent[3]
        cache[:] = [ent for ent in cache if ent[0]() is not None and not (ent[0]() is model and ent[1] == key)][-(self.GRAPH_CACHE_SIZE - 1):]
        g = _GraphedUNet(model, x, t, ctx, row)
        cache.append((weakref.ref(model), key, sig, g))
        return g

    def clear_graph_cache(self) -> None:
        """Drop every captured forward this inferer keeps (and with them the graphs' private memory pools): after an in-place parameter write the
        signature cannot see (`p.data.copy_(...)`), or to give the activation memory back."""
        self.__dict__.pop("_graph_cache", None)

    @torch.no_grad()
    def sample(self, input_noise: torch.Tensor, diffusion_model: Callable[..., torch.Tensor],
               scheduler: Callable[..., torch.Tensor] | None = None, save_intermediates: bool | None = False,
               intermediate_steps: int | None = 100, conditioning: torch.Tensor | None = None, mode: str = "crossattn",
               verbose: bool = True, seg: torch.Tensor | None = None):
        """Reverse chain over `scheduler.timesteps` (reference inferer.py:83-143). Returns the final image, or
        (image, intermediates) with an intermediate kept whenever `t % intermediate_steps == 0`."""
        _check_mode(mode)
        if not scheduler:
            scheduler = self.scheduler
        ops.require_device(input_noise)
        image = input_noise
        steps = [int(t) for t in torch.as_tensor(scheduler.timesteps).cpu().tolist()]
        t_dev = torch.tensor(steps, dtype=torch.float32).to(input_noise.device)  # whole chain, one upload
        it = tqdm(range(len(steps))) if (verbose and has_tqdm) else range(len(steps))
        graphable = isinstance(diffusion_model, DiffusionModelUNet) and not isinstance(diffusion_model, SPADEDiffusionModelUNet)
        # the timestep rows of the whole chain from one batched pass (4 launches per chain instead of 4 per step); row i goes to step i
        unet = diffusion_model if isinstance(diffusion_model, DiffusionModelUNet) else None
        table = unet.time_rows_table(t_dev) if (unet is not None and len(steps) > 1 and self.BATCHED_TIME_ROWS) else None
        diffusion_model = _bind_seg(diffusion_model, seg)
        try:
            return self._sample_loop(it, steps, t_dev, table, unet, graphable, image, diffusion_model, scheduler, conditioning, mode,
                                     save_intermediates, intermediate_steps)
        finally:
            if unet is not None:
                unet.__dict__.pop("_time_rows_row", None)  # (a forward that raised before consuming its row)

    def _sample_loop(self, it, steps, t_dev, table, unet, graphable, image, diffusion_model, scheduler, conditioning, mode, save_intermediates,
                     intermediate_steps):
        graphed = None
        intermediates = []
        for i in it:
            t, tt = steps[i], t_dev[i:i + 1]
            row = None if table is None else table[i:i + 1]
            if mode == "concat":
                model_input, ctx = ops.concat_dim1([image, conditioning]), None
            else:
                model_input, ctx = image, conditioning
            use_graph = self.use_hip_graph if self.use_hip_graph is not None else (
                model_input.numel() <= self.GRAPH_AUTO_MAX_ELEMENTS and len(steps) >= self.GRAPH_AUTO_MIN_STEPS)
            if use_graph and graphable:
                if graphed is None:
                    graphed = self._cached_graph(diffusion_model, model_input, tt, ctx, row)
                    graphed.set_context(ctx)
                model_output = graphed(model_input, tt, row)
                if getattr(scheduler, "keeps_model_outputs", False):
                    model_output = model_output.clone()  # the graph's static output buffer is overwritten by the next replay
            else:
                if row is not None:
                    unet._time_rows_row = row  # consumed by this forward
                model_output = diffusion_model(model_input, timesteps=tt, context=ctx)
            image, _ = scheduler.step(model_output, t, image)
            if save_intermediates and t % intermediate_steps == 0:
                intermediates.append(image)
        return (image, intermediates) if save_intermediates else image

    @torch.no_grad()
    def get_likelihood(self, inputs: torch.Tensor, diffusion_model: Callable[..., torch.Tensor],
                       scheduler: Callable[..., torch.Tensor] | None = None, save_intermediates: bool | None = False,
                       conditioning: torch.Tensor | None = None, mode: str = "crossattn",
                       original_input_range: tuple | None = (0, 255), scaled_input_range: tuple | None = (0, 1),
                       verbose: bool = True, seg: torch.Tensor | None = None, _noise: torch.Tensor | None = None):
        """Variational bound on -log p(x) per sample (reference inferer.py:145-256): for every t of the scheduler, noise the
        inputs to t with ONE fixed noise draw, predict, and add KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) -- the discretised
        decoder NLL at t = 0 -- averaged over elements.  Returns total_kl (N,) fp32, plus the per-step maps (CPU tensors, as
        the reference) when `save_intermediates`.  Per step: add_noise kernel + UNet forward + ONE fused kernel
        (gm_likelihood_term) instead of ~40 element-wise launches.  `_noise` (not in the reference) lets tests fix the draw."""
        if not scheduler:
            scheduler = self.scheduler
        if scheduler._get_name() != "DDPMScheduler":
            raise NotImplementedError(f"Likelihood computation is only compatible with DDPMScheduler, you are using {scheduler._get_name()}")
        _check_mode(mode)
        ops.require_device(inputs)
        diffusion_model = _bind_seg(diffusion_model, seg)
        steps = [int(t) for t in torch.as_tensor(scheduler.timesteps).cpu().tolist()]
        it = tqdm(steps) if (verbose and has_tqdm) else steps
        noise = torch.randn_like(inputs) if _noise is None else _noise
        n = inputs.shape[0]
        total_kl = torch.zeros(n, dtype=torch.float32, device=inputs.device)
        workspace = torch.empty(ops.likelihood_workspace_elems(n, inputs.numel() // max(n, 1)), dtype=torch.float64, device=inputs.device)
        acp, betas, alphas = (scheduler._host_table(k) for k in ("alphas_cumprod", "betas", "alphas"))
        one = scheduler.one.detach().to("cpu", torch.float32)
        bin_width = (scaled_input_range[1] - scaled_input_range[0]) / (original_input_range[1] - original_input_range[0])
        intermediates = []
        for t in it:
            timesteps = torch.full((n,), t, dtype=torch.long, device=inputs.device)
            noisy = self.scheduler.add_noise(original_samples=inputs, noise=noise, timesteps=timesteps)
            if mode == "concat":
                model_output = diffusion_model(ops.concat_dim1([noisy, conditioning]), timesteps=timesteps, context=None)
            else:
                model_output = diffusion_model(x=noisy, timesteps=timesteps, context=conditioning)
            if model_output.shape[1] == inputs.shape[1] * 2 and scheduler.variance_type in ["learned", "learned_range"]:
                # the reference evaluates `if predicted_variance` on the whole tensor here (inferer.py:240) and raises
                raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous "
                                   "(get_likelihood does not support learned variances, as in the reference)")
            a_t = acp[t]
            a_prev = acp[t - 1] if t > 0 else one
            b_t, b_prev = 1 - a_t, 1 - a_prev
            p = GmKlParams()
            p.pred_type = {"epsilon": 0, "sample": 1, "v_prediction": 2}[str(scheduler.prediction_type)]
            p.c_sa, p.c_sb = float(a_t**0.5), float(b_t**0.5)
            p.clip = int(bool(scheduler.clip_sample))
            p.k0 = float((a_prev**0.5 * betas[t]) / b_t)
            p.k1 = float(alphas[t] ** 0.5 * b_prev / b_t)
            p.m0 = float(a_prev.sqrt() * betas[t] / (1 - a_t))
            p.m1 = float(alphas[t].sqrt() * (1 - a_prev) / (1 - a_t))
            log_post = torch.log(scheduler._get_variance(timestep=t, predicted_variance=None))
            log_pred = log_post
            if t == 0:
                p.t0, p.e, p.half_bin = 1, float(torch.exp(-(0.5 * log_pred))), float(torch.tensor(bin_width / 2, dtype=torch.float32))
            else:
                p.t0 = 0
                p.s = float(-1.0 + log_pred - log_post + torch.exp(log_post - log_pred))
                p.e = float(torch.exp(-log_pred))
            kl = ops.likelihood_term(inputs, noisy, model_output, p, total_kl, workspace, want_map=bool(save_intermediates))
            if save_intermediates:
                intermediates.append(kl.cpu())
        return (total_kl, intermediates) if save_intermediates else total_kl


def _center_crop(x: torch.Tensor, roi) -> torch.Tensor:
    """MONAI CenterSpatialCrop on a batch: roi entries <= 0 or >= size keep the axis (inferer.py:353,465)."""
    sl = [slice(None), slice(None)]
    for size, r in zip(x.shape[2:], roi):
        if r <= 0 or r >= size:
            sl.append(slice(None))
        else:
            start = size // 2 - r // 2
            sl.append(slice(start, start + r))
    return x[tuple(sl)].contiguous()


def _spatial_pad(x: torch.Tensor, size) -> torch.Tensor:
    """MONAI SpatialPad (symmetric, odd remainder on the high side) on a batch (inferer.py:352,389)."""
    tgt = [max(int(s), int(cur)) for s, cur in zip(size, x.shape[2:])]
    if tgt == list(x.shape[2:]):
        return x
    out = torch.zeros((x.shape[0], x.shape[1], *tgt), dtype=x.dtype, device=x.device)
    sl = [slice(None), slice(None)]
    for cur, t in zip(x.shape[2:], tgt):
        lo = (t - cur) // 2
        sl.append(slice(lo, lo + cur))
    out[tuple(sl)].copy_(x)
    return out


class LatentDiffusionInferer(DiffusionInferer):
    """Drop-in for generative.inferers.LatentDiffusionInferer: stage-1 autoencoder (AutoencoderKL / VQVAE) around the
    diffusion chain, latent scaling, optional latent pad / crop."""

    def __init__(self, scheduler: nn.Module, scale_factor: float = 1.0, ldm_latent_shape: list | None = None,
                 autoencoder_latent_shape: list | None = None, use_hip_graph: bool | None = None) -> None:
        super().__init__(scheduler=scheduler, use_hip_graph=use_hip_graph)
        self.scale_factor = scale_factor
        if (ldm_latent_shape is None) ^ (autoencoder_latent_shape is None):
            raise ValueError("If ldm_latent_shape is None, autoencoder_latent_shape must be None and vice versa.")
        self.ldm_latent_shape = ldm_latent_shape
        self.autoencoder_latent_shape = autoencoder_latent_shape

    def _scaled(self, x: torch.Tensor, divide: bool) -> torch.Tensor:
        return x if self.scale_factor == 1.0 else ops.scale(x, self.scale_factor, divide)

    def __call__(self, inputs: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor],
                 diffusion_model: Callable[..., torch.Tensor], noise: torch.Tensor, timesteps: torch.Tensor,
                 condition: torch.Tensor | None = None, mode: str = "crossattn", seg: torch.Tensor | None = None,
                 quantized: bool = True) -> torch.Tensor:
        with torch.no_grad():
            if isinstance(autoencoder_model, VQVAE):
                latent = autoencoder_model.encode_stage_2_inputs(inputs, quantized=quantized)
            else:
                latent = autoencoder_model.encode_stage_2_inputs(inputs)
            latent = self._scaled(latent, divide=False)
        if self.ldm_latent_shape is not None:
            latent = _spatial_pad(latent, self.ldm_latent_shape)
        return super().__call__(inputs=latent, diffusion_model=diffusion_model, noise=noise, timesteps=timesteps,
                                condition=condition, mode=mode, seg=seg)

    @torch.no_grad()
    def sample(self, input_noise: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor],
               diffusion_model: Callable[..., torch.Tensor], scheduler: Callable[..., torch.Tensor] | None = None,
               save_intermediates: bool | None = False, intermediate_steps: int | None = 100,
               conditioning: torch.Tensor | None = None, mode: str = "crossattn", verbose: bool = True,
               seg: torch.Tensor | None = None):
        outputs = super().sample(input_noise=input_noise, diffusion_model=diffusion_model, scheduler=scheduler,
                                 save_intermediates=save_intermediates, intermediate_steps=intermediate_steps,
                                 conditioning=conditioning, mode=mode, verbose=verbose, seg=seg)
        latent, latent_intermediates = outputs if save_intermediates else (outputs, [])
        if self.autoencoder_latent_shape is not None:
            latent = _center_crop(latent, self.autoencoder_latent_shape)
            latent_intermediates = [_center_crop(l, self.autoencoder_latent_shape) for l in latent_intermediates]
        decode = autoencoder_model.decode_stage_2_outputs
        if isinstance(autoencoder_model, SPADEAutoencoderKL):  # reference inferer.py:461-462
            decode = functools.partial(autoencoder_model.decode_stage_2_outputs, seg=seg)
        image = decode(self._scaled(latent, divide=True))
        if save_intermediates:
            return image, [decode(self._scaled(l, divide=True)) for l in latent_intermediates]
        return image

    @torch.no_grad()
    def get_likelihood(self, inputs: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor],
                       diffusion_model: Callable[..., torch.Tensor], scheduler: Callable[..., torch.Tensor] | None = None,
                       save_intermediates: bool | None = False, conditioning: torch.Tensor | None = None, mode: str = "crossattn",
                       original_input_range: tuple | None = (0, 255), scaled_input_range: tuple | None = (0, 1),
                       verbose: bool = True, resample_latent_likelihoods: bool = False,
                       resample_interpolation_mode: str = "nearest", seg: torch.Tensor | None = None, quantized: bool = True,
                       _noise: torch.Tensor | None = None):
        """Likelihood bound of the *latent* of `inputs` (reference inferer.py:489-562). The intermediate KL maps are CPU
        tensors (as in the reference, which moves them with `.cpu()`), so the optional resampling to the image grid is the
        reference's own host-side `nn.Upsample`."""
        if resample_latent_likelihoods and resample_interpolation_mode not in ("nearest", "bilinear", "trilinear"):
            raise ValueError(
                f"resample_interpolation mode should be either nearest, bilinear, or trilinear, got {resample_interpolation_mode}")
        if isinstance(autoencoder_model, VQVAE):
            latents = autoencoder_model.encode_stage_2_inputs(inputs, quantized=quantized)
        else:
            latents = autoencoder_model.encode_stage_2_inputs(inputs)
        latents = self._scaled(latents, divide=False)
        if self.ldm_latent_shape is not None:
            latents = _spatial_pad(latents, self.ldm_latent_shape)
        outputs = super().get_likelihood(inputs=latents, diffusion_model=diffusion_model, scheduler=scheduler,
                                         save_intermediates=save_intermediates, conditioning=conditioning, mode=mode,
                                         verbose=verbose, seg=seg, _noise=_noise)
        if save_intermediates and resample_latent_likelihoods:
            resizer = nn.Upsample(size=inputs.shape[2:], mode=resample_interpolation_mode)
            outputs = (outputs[0], [resizer(x) for x in outputs[1]])
        return outputs


class _Controlled:
    """`diffusion_model(x, timesteps, context)` with the ControlNet evaluated first and its residuals injected -- the body the
    reference repeats in every ControlNet inferer method (inferer.py:610-626, 676-694, 778-793)."""

    def __init__(self, diffusion_model, controlnet, cn_cond, seg=None) -> None:
        # a SPADE network takes the segmentation as a forward argument: bound here like the reference's functools.partial (inferer.py:606-608)
        self.diffusion_model, self.controlnet, self.cn_cond = _bind_seg(diffusion_model, seg), controlnet, cn_cond

    def supports_training(self) -> bool:
        # DiffusionInferer.__call__ then simply calls this object; the two networks decide for themselves (`_blocks.wants_grad`): a ControlNet
        # in train() mode returns differentiable residuals, and residuals that require grad put the (usually frozen) UNet on its
        # differentiable forward -- the ControlNet training step of the reference's tutorials
        return False

    def parameters(self):
        return iter(())

    def __call__(self, x, timesteps, context=None):
        down, mid = self.controlnet(x=x, timesteps=timesteps, controlnet_cond=self.cn_cond, context=context)
        return self.diffusion_model(x=x, timesteps=timesteps, context=context, down_block_additional_residuals=down,
                                    mid_block_additional_residual=mid)


class ControlNetDiffusionInferer(DiffusionInferer):
    """Drop-in for generative.inferers.ControlNetDiffusionInferer (inferer.py:565-868): DiffusionInferer with a ControlNet
    evaluated on every model call."""

    def __init__(self, scheduler: nn.Module) -> None:
        super().__init__(scheduler)

    def __call__(self, inputs: torch.Tensor, diffusion_model: Callable[..., torch.Tensor], controlnet: Callable[..., torch.Tensor],
                 noise: torch.Tensor, timesteps: torch.Tensor, cn_cond: torch.Tensor, condition: torch.Tensor | None = None,
                 mode: str = "crossattn", seg: torch.Tensor | None = None) -> torch.Tensor:
        return super().__call__(inputs=inputs, diffusion_model=_Controlled(diffusion_model, controlnet, cn_cond, seg), noise=noise,
                                timesteps=timesteps, condition=condition, mode=mode)

    @torch.no_grad()
    def sample(self, input_noise: torch.Tensor, diffusion_model: Callable[..., torch.Tensor], controlnet: Callable[..., torch.Tensor],
               cn_cond: torch.Tensor, scheduler: Callable[..., torch.Tensor] | None = None, save_intermediates: bool | None = False,
               intermediate_steps: int | None = 100, conditioning: torch.Tensor | None = None, mode: str = "crossattn",
               verbose: bool = True, seg: torch.Tensor | None = None):
        return super().sample(input_noise=input_noise, diffusion_model=_Controlled(diffusion_model, controlnet, cn_cond, seg),
                              scheduler=scheduler, save_intermediates=save_intermediates, intermediate_steps=intermediate_steps,
                              conditioning=conditioning, mode=mode, verbose=verbose)

    @torch.no_grad()
    def get_likelihood(self, inputs: torch.Tensor, diffusion_model: Callable[..., torch.Tensor], controlnet: Callable[..., torch.Tensor],
                       cn_cond: torch.Tensor, scheduler: Callable[..., torch.Tensor] | None = None,
                       save_intermediates: bool | None = False, conditioning: torch.Tensor | None = None, mode: str = "crossattn",
                       original_input_range: tuple | None = (0, 255), scaled_input_range: tuple | None = (0, 1), verbose: bool = True,
                       seg: torch.Tensor | None = None, _noise: torch.Tensor | None = None):
        return super().get_likelihood(inputs=inputs, diffusion_model=_Controlled(diffusion_model, controlnet, cn_cond, seg), scheduler=scheduler,
                                      save_intermediates=save_intermediates, conditioning=conditioning, mode=mode,
                                      original_input_range=original_input_range, scaled_input_range=scaled_input_range,
                                      verbose=verbose, _noise=_noise)


def _resize_like(cn_cond: torch.Tensor, spatial) -> torch.Tensor:
    """F.interpolate(cn_cond, size) with the default nearest mode (inferer.py:926-927): the conditioning image on the latent grid."""
    if tuple(cn_cond.shape[2:]) == tuple(spatial):
        return cn_cond
    ops.require_device(cn_cond)
    return ops.to_channels_first(ops.nearest_resize(ops.to_channels_last(cn_cond), tuple(spatial)))


class ControlNetLatentDiffusionInferer(LatentDiffusionInferer):
    """Drop-in for generative.inferers.ControlNetLatentDiffusionInferer (inferer.py:871-1123)."""

    def __call__(self, inputs: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor], diffusion_model: Callable[..., torch.Tensor],
                 controlnet: Callable[..., torch.Tensor], noise: torch.Tensor, timesteps: torch.Tensor, cn_cond: torch.Tensor,
                 condition: torch.Tensor | None = None, mode: str = "crossattn", seg: torch.Tensor | None = None,
                 quantized: bool = True) -> torch.Tensor:
        model = _Controlled(diffusion_model, controlnet, _resize_like(cn_cond, noise.shape[2:]), seg)  # noise has the latent's shape
        return super().__call__(inputs=inputs, autoencoder_model=autoencoder_model, diffusion_model=model, noise=noise,
                                timesteps=timesteps, condition=condition, mode=mode, quantized=quantized)

    @torch.no_grad()
    def sample(self, input_noise: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor],
               diffusion_model: Callable[..., torch.Tensor], controlnet: Callable[..., torch.Tensor], cn_cond: torch.Tensor,
               scheduler: Callable[..., torch.Tensor] | None = None, save_intermediates: bool | None = False,
               intermediate_steps: int | None = 100, conditioning: torch.Tensor | None = None, mode: str = "crossattn",
               verbose: bool = True, seg: torch.Tensor | None = None):
        model = _Controlled(diffusion_model, controlnet, _resize_like(cn_cond, input_noise.shape[2:]), seg)
        return super().sample(input_noise=input_noise, autoencoder_model=autoencoder_model, diffusion_model=model, scheduler=scheduler,
                              save_intermediates=save_intermediates, intermediate_steps=intermediate_steps, conditioning=conditioning,
                              mode=mode, verbose=verbose)

    @torch.no_grad()
    def get_likelihood(self, inputs: torch.Tensor, autoencoder_model: Callable[..., torch.Tensor],
                       diffusion_model: Callable[..., torch.Tensor], controlnet: Callable[..., torch.Tensor], cn_cond: torch.Tensor,
                       scheduler: Callable[..., torch.Tensor] | None = None, save_intermediates: bool | None = False,
                       conditioning: torch.Tensor | None = None, mode: str = "crossattn", original_input_range: tuple | None = (0, 255),
                       scaled_input_range: tuple | None = (0, 1), verbose: bool = True, resample_latent_likelihoods: bool = False,
                       resample_interpolation_mode: str = "nearest", seg: torch.Tensor | None = None, quantized: bool = True,
                       _noise: torch.Tensor | None = None):
        # the latent grid is only known after encoding; resize lazily on the first model call
        inferer = self

        class _Lazy(_Controlled):
            def __call__(self, x, timesteps, context=None):
                self.cn_cond = _resize_like(self.cn_cond, x.shape[2:])
                return super().__call__(x, timesteps, context)

        return LatentDiffusionInferer.get_likelihood(inferer, inputs=inputs, autoencoder_model=autoencoder_model,
                                                     diffusion_model=_Lazy(diffusion_model, controlnet, cn_cond, seg), scheduler=scheduler,
                                                     save_intermediates=save_intermediates, conditioning=conditioning, mode=mode,
                                                     original_input_range=original_input_range, scaled_input_range=scaled_input_range,
                                                     verbose=verbose, resample_latent_likelihoods=resample_latent_likelihoods,
                                                     resample_interpolation_mode=resample_interpolation_mode, quantized=quantized, _noise=_noise)


class VQVAETransformerInferer(Inferer):
    """Drop-in for generative.inferers.VQVAETransformerInferer (inferer.py:1126-1330): training forward, autoregressive sampling
    and per-token likelihoods of a VQ-VAE + decoder-only transformer pair.

    `sample` decodes incrementally over a KV cache when the transformer is this package's DecoderOnlyTransformer: the reference
    re-runs the whole prefix for every new token; here a token is one row through each GEMM + one 1 x t attention per block, and
    the sampling head (temperature, top-k, softmax, BOS mask) is one fused kernel.  When the prefix outgrows `max_seq_len` the
    reference's window slides and the absolute positions restart, so those (few) steps recompute the cropped window like the
    reference does.  The categorical draw is an inverse-CDF kernel fed by torch's device generator (same distribution as the reference's
    torch.multinomial, without its per-call device -> host validation read)."""

    def __init__(self, use_hip_graph: bool = False) -> None:
        """use_hip_graph: replay each decode iteration (token step + sampling head + draw + bookkeeping) from one HIP graph with the
        position kept on the device, when the transformer is this package's DecoderOnlyTransformer without cross attention.  Off by
        default: measured the same as eager launches (round 2: 0.75 vs 0.74 ms per token at 62 launches; round 3: 0.32 vs 0.31 at 50) -- the token is
        bound by the ~4.5 us every dependent launch costs on the GPU side, graph node or not; a replay only takes the launches off the host."""
        self.use_hip_graph = use_hip_graph

    @staticmethod
    def _sample_graphed(latent_seq: torch.Tensor, seq_len: int, tr, temperature, top_k, bos):
        """Draws as many of the `seq_len` tokens as fit in the context window with ONE graph replay per token.  Device state: the
        position (int32), the token fed next, the logits of the last step and the growing sequence.  -> (sequence, tokens drawn)."""
        dev = latent_seq.device
        b, n0 = latent_seq.shape
        # a draw appended at position n0 + i is followed (inside the graph) by feeding it at that position: valid while n0 + i < max_seq_len
        n_graph = max(0, min(seq_len, tr.max_seq_len - n0))
        if n_graph < 3:
            return latent_seq, 0
        cache = tr.new_cache(b, dev)
        for pz in range(n0):  # prefill: positions 0 .. n0-1
            logits = tr.step(latent_seq[:, pz:pz + 1].contiguous(), pz, cache)
        logits = logits.contiguous()
        seq = torch.zeros((b, n0 + n_graph), dtype=torch.long, device=dev)
        seq[:, :n0] = latent_seq
        pos_dev = torch.full((1,), n0 - 1, dtype=torch.int32, device=dev)
        tokens = torch.zeros((b, 1), dtype=torch.long, device=dev)

        def body():
            probs = ops.sample_probs(logits, temperature, top_k, bos)
            idx = ops.sample_index(probs)
            ops.decode_advance(pos_dev, tokens, idx, seq)
            tr.step_from_device_state(tokens, pos_dev, cache, logits)

        body()  # draw 0 eagerly: first-use initialisation of every kernel happens outside the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body()  # captured, not executed: draw 1 happens at the first replay
        for _ in range(n_graph - 1):
            graph.replay()
        return seq, n_graph

    @staticmethod
    def _sequence(latent: torch.Tensor, ordering) -> torch.Tensor:
        lat = latent.reshape(latent.shape[0], -1)
        order = torch.as_tensor(ordering.get_sequence_ordering().copy(), device=lat.device)
        return lat[:, order]

    def __call__(self, inputs: torch.Tensor, vqvae_model, transformer_model, ordering, condition: torch.Tensor | None = None,
                 return_latent: bool = False):
        """Next-token logits for the (BOS-shifted) latent sequence of `inputs` (reference inferer.py:1134-1181)."""
        with torch.no_grad():
            latent = vqvae_model.index_quantize(inputs)
        latent_spatial_dim = tuple(latent.shape[1:])
        latent = self._sequence(latent, ordering)
        target = latent.clone()
        bos = torch.full((latent.shape[0], 1), vqvae_model.num_embeddings, dtype=latent.dtype, device=latent.device)
        latent = torch.cat([bos, latent], dim=1)[:, :-1].long()
        seq_len, max_seq_len = latent.shape[1], transformer_model.max_seq_len
        start = int(torch.randint(low=0, high=seq_len + 1 - max_seq_len, size=(1,)).item()) if max_seq_len < seq_len else 0
        prediction = transformer_model(x=latent[:, start:start + max_seq_len].contiguous(), context=condition)
        if return_latent:
            return prediction, target[:, start:start + max_seq_len], latent_spatial_dim
        return prediction

    @torch.no_grad()
    def sample(self, latent_spatial_dim, starting_tokens: torch.Tensor, vqvae_model, transformer_model, ordering,
               conditioning: torch.Tensor | None = None, temperature: float = 1.0, top_k: int | None = None, verbose: bool = True):
        """Autoregressive sampling of prod(latent_spatial_dim) tokens after the BOS token(s), decoded to an image
        (reference inferer.py:1183-1245)."""
        ops.require_device(starting_tokens)
        seq_len = math.prod(latent_spatial_dim)
        latent_seq = starting_tokens.long()
        bos = vqvae_model.num_embeddings
        native = (isinstance(transformer_model, DecoderOnlyTransformer) and not transformer_model.with_cross_attention
                  and transformer_model.native_step and conditioning is None and self.use_hip_graph)
        done = 0
        if native and latent_seq.size(1) <= transformer_model.max_seq_len:
            latent_seq, done = self._sample_graphed(latent_seq, seq_len, transformer_model, temperature, top_k, bos)
        it = range(done, seq_len)
        if verbose and has_tqdm:
            it = tqdm(it)
        # after a graphed run the few remaining draws (window full) recompute the cropped window like the reference; re-feeding the
        # whole prefix through a fresh cache would cost one step per prefix token
        cached = isinstance(transformer_model, DecoderOnlyTransformer) and done == 0
        cache, filled = None, 0
        for _ in it:
            n = latent_seq.size(1)
            if cached and n <= transformer_model.max_seq_len:
                if cache is None:
                    cache = transformer_model.new_cache(latent_seq.shape[0], latent_seq.device)
                while filled < n:  # feeds the whole prefix on the first iteration, one new token afterwards
                    logits = transformer_model.step(latent_seq[:, filled:filled + 1].contiguous(), filled, cache, conditioning)
                    filled += 1
            else:
                idx_cond = latent_seq if n <= transformer_model.max_seq_len else latent_seq[:, -transformer_model.max_seq_len:]
                logits = transformer_model(x=idx_cond.contiguous(), context=conditioning)[:, -1, :]
            probs = ops.sample_probs(logits, temperature, top_k, bos)
            idx_next = ops.sample_index(probs)  # inverse-CDF draw on the device; torch.multinomial would drain the pipeline per token
            latent_seq = torch.cat((latent_seq, idx_next), dim=1)
        latent_seq = latent_seq[:, 1:]
        revert = torch.as_tensor(ordering.get_revert_sequence_ordering().copy(), device=latent_seq.device)
        latent = latent_seq[:, revert].reshape((starting_tokens.shape[0],) + tuple(latent_spatial_dim))
        return vqvae_model.decode_samples(latent)

    @torch.no_grad()
    def get_likelihood(self, inputs: torch.Tensor, vqvae_model, transformer_model, ordering, condition: torch.Tensor | None = None,
                       resample_latent_likelihoods: bool = False, resample_interpolation_mode: str = "nearest", verbose: bool = False):
        """log p(token | predecessors) for every latent token, on the latent grid (reference inferer.py:1247-1330)."""
        if resample_latent_likelihoods and resample_interpolation_mode not in ("nearest", "bilinear", "trilinear"):
            raise ValueError(
                f"resample_interpolation mode should be either nearest, bilinear, or trilinear, got {resample_interpolation_mode}")
        latent = vqvae_model.index_quantize(inputs)
        latent_spatial_dim = tuple(latent.shape[1:])
        latent = self._sequence(latent, ordering)
        seq_len = math.prod(latent_spatial_dim)
        bos = torch.full((latent.shape[0], 1), vqvae_model.num_embeddings, dtype=latent.dtype, device=latent.device)
        latent = torch.cat([bos, latent], dim=1).long()
        msl = transformer_model.max_seq_len
        target = latent[:, 1:]
        logits = transformer_model(x=latent[:, :msl].contiguous(), context=condition)
        t0 = logits.shape[1] if target.shape[1] >= logits.shape[1] else target.shape[1]
        lp = ops.token_log_prob(logits[:, :t0].reshape(-1, logits.shape[-1]), target[:, :t0].reshape(-1)).reshape(latent.shape[0], t0)
        if lp.shape[1] < target.shape[1]:
            it = tqdm(range(msl, seq_len)) if (verbose and has_tqdm) else range(msl, seq_len)
            cols = [lp]
            for i in it:
                lg = transformer_model(x=latent[:, i + 1 - msl:i + 1].contiguous(), context=condition)[:, -1, :]
                cols.append(ops.token_log_prob(lg, target[:, i]).unsqueeze(1))
            lp = torch.cat(cols, dim=1)
        revert = torch.as_tensor(ordering.get_revert_sequence_ordering().copy(), device=lp.device)
        out = lp[:, revert].reshape((inputs.shape[0],) + latent_spatial_dim)
        if resample_latent_likelihoods:
            # like the reference, on whatever device the values live: nn.Upsample is host-framework resampling of a result map
            out = nn.Upsample(size=inputs.shape[2:], mode=resample_interpolation_mode)(out[:, None, ...].cpu()).to(out.device)
        return out
