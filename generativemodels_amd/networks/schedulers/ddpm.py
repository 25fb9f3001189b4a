"""DDPM scheduler: constructor / `set_timesteps` / `step` contract of the reference's
generative/networks/schedulers/ddpm.py:67-252, with the whole reverse step fused into one HIP kernel (gm_sched_step)."""
from __future__ import annotations

import os

import torch

from ... import host_noise, ops
from ..._native import GmStepParams
from .scheduler import Scheduler, inference_timesteps, x0_prediction_code


class DDPMVarianceType:
    FIXED_SMALL = "fixed_small"
    FIXED_LARGE = "fixed_large"
    LEARNED = "learned"
    LEARNED_RANGE = "learned_range"
    _ALL = (FIXED_SMALL, FIXED_LARGE, LEARNED, LEARNED_RANGE)


class DDPMPredictionType:
    EPSILON = "epsilon"
    SAMPLE = "sample"
    V_PREDICTION = "v_prediction"
    _ALL = (EPSILON, SAMPLE, V_PREDICTION)


class DDPMScheduler(Scheduler):
    """Ho et al. 2020 ancestral sampler. Arguments as the reference (ddpm.py:84-94).

    `fp32_noise_draw` (class / instance attribute, not in the reference; default False): how the noise of a REDUCED-PRECISION chain is drawn.  The reference
    calls `torch.randn(shape, dtype=model_output.dtype)` on the CPU generator (ddpm.py:244-248); for bf16 that is torch's serial scalar bf16 fill --
    0.9 ms per 16 x 1 x 64 x 64 draw on the MI355X box's host, more than the whole replayed bf16 forward of the BASELINE configs[0] UNet (0.48 ms,
    profiles/r06_c1b_2d_ddpm.json).  The DEFAULT (False) now reproduces exactly those bf16 values from the generator's byte draws and a device-side table
    (host_noise.py: 0.2 ms of host time, same stream, same bits).  True: a bf16 / fp16 chain draws fp32 values from the same generator (torch's vectorised fill)
    and rounds them on the device -- the same distribution and seed dependence, NOT the reference's bf16 values (its bf16 fill is another algorithm over the
    same generator state); kept for fp16 chains and as the A/B of round 6.  fp32 chains are untouched: same seed, same chain as the reference."""

    fp32_noise_draw = os.environ.get("GM_DDPM_FP32_NOISE_DRAW", "0") == "1"

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta",
                 variance_type: str = DDPMVarianceType.FIXED_SMALL, clip_sample: bool = True,
                 prediction_type: str = DDPMPredictionType.EPSILON, clip_sample_min: int = -1, clip_sample_max: int = 1,
                 **schedule_args) -> None:
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if variance_type not in DDPMVarianceType._ALL:
            raise ValueError("Argument `variance_type` must be a member of `DDPMVarianceType`")
        if prediction_type not in DDPMPredictionType._ALL:
            raise ValueError("Argument `prediction_type` must be a member of `DDPMPredictionType`")
        if clip_sample_min >= clip_sample_max:
            raise ValueError("clip_sample_min must be < clip_sample_max")
        self.clip_sample = clip_sample
        self.variance_type = variance_type
        self.prediction_type = prediction_type
        self.clip_sample_values = [clip_sample_min, clip_sample_max]

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.num_train_timesteps`:"
                f" {self.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        self.timesteps = inference_timesteps(self.num_train_timesteps, num_inference_steps).to(device)

    # posterior statistics as fp32 0-dim CPU tensors, same expressions as the reference (ddpm.py:133-189)
    def _posterior_scalars(self, t: int):
        ac, betas, alphas = self._host_table("alphas_cumprod"), self._host_table("betas"), self._host_table("alphas")
        a_t = ac[t]
        a_prev = ac[t - 1] if t > 0 else self.one
        return a_t, a_prev, betas[t], alphas[t]

    def _get_mean(self, timestep: int, x_0: torch.Tensor, x_t: torch.Tensor) -> torch.Tensor:
        """Posterior mean q(x_{t-1} | x_t, x_0) (reference ddpm.py:133-156), computed by the fused step kernel."""
        t = self._as_int(timestep)
        a_t, a_prev, beta_t, alpha_t = self._posterior_scalars(t)
        p = GmStepParams()
        p.mode, p.pred_type, p.clip, p.noise_mode = 1, 1, 0, 0
        p.k0 = self._f(a_prev.sqrt() * beta_t / (1 - a_t))
        p.k1 = self._f(alpha_t.sqrt() * (1 - a_prev) / (1 - a_t))
        prev, _ = ops.sched_step(x_t, x_0, p, want_x0=False)
        return prev

    def _get_variance(self, timestep: int, predicted_variance: torch.Tensor | None = None) -> torch.Tensor:
        """Posterior variance (reference ddpm.py:158-189); scalar cases return a 0-dim fp32 CPU tensor."""
        t = self._as_int(timestep)
        a_t, a_prev, beta_t, _ = self._posterior_scalars(t)
        variance = (1 - a_prev) / (1 - a_t) * beta_t
        if self.variance_type == DDPMVarianceType.FIXED_SMALL:
            variance = torch.clamp(variance, min=1e-20)
        elif self.variance_type == DDPMVarianceType.FIXED_LARGE:
            variance = beta_t
        elif self.variance_type == DDPMVarianceType.LEARNED:
            return predicted_variance
        elif self.variance_type == DDPMVarianceType.LEARNED_RANGE:
            frac = (predicted_variance + 1) / 2
            variance = frac * beta_t.to(frac.device) + (1 - frac) * variance.to(frac.device)
        return variance

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor,
             generator: torch.Generator | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """One reverse step; returns (x_{t-1}, predicted x_0). The Gaussian noise is drawn from the *CPU* generator and
        copied to the device exactly as the reference does (ddpm.py:244-248), so seeded chains see the same stream."""
        ops.require_device(model_output, sample)
        t = self._as_int(timestep)
        learned = model_output.shape[1] == sample.shape[1] * 2 and self.variance_type in ("learned", "learned_range")
        a_t, a_prev, beta_t, alpha_t = self._posterior_scalars(t)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        p = GmStepParams()
        p.mode, p.pred_type = 1, x0_prediction_code(self.prediction_type)
        p.c_sa, p.c_sb = self._f(a_t**0.5), self._f(b_t**0.5)
        p.clip = int(bool(self.clip_sample))
        p.clip_lo, p.clip_hi = float(self.clip_sample_values[0]), float(self.clip_sample_values[1])
        p.k0 = self._f((a_prev**0.5 * beta_t) / b_t)
        p.k1 = self._f(alpha_t**0.5 * b_prev / b_t)
        noise = None
        p.noise_mode = 0
        if t > 0:
            shape = list(model_output.shape)
            if learned:
                shape[1] //= 2
            if self.fp32_noise_draw and model_output.dtype in (torch.bfloat16, torch.float16):
                noise = ops.cast(host_noise.randn(shape, torch.float32, generator, model_output.device), model_output.dtype)
            else:  # (bf16: the same values from the generator's byte draws + a device table lookup, host_noise.py)
                noise = host_noise.randn(shape, model_output.dtype, generator, model_output.device, as_bits=True)
            variance = (1 - a_prev) / (1 - a_t) * beta_t
            if self.variance_type == DDPMVarianceType.FIXED_SMALL:
                p.noise_mode, p.c_noise = 1, self._f(torch.clamp(variance, min=1e-20) ** 0.5)
            elif self.variance_type == DDPMVarianceType.FIXED_LARGE:
                p.noise_mode, p.c_noise = 1, self._f(beta_t**0.5)
            elif not learned:
                raise ValueError("learned variance types need a model output with 2*C channels")
            elif self.variance_type == DDPMVarianceType.LEARNED:
                p.noise_mode = 2
            else:
                p.noise_mode, p.min_log, p.max_log = 3, self._f(variance), self._f(beta_t)
        if model_output.shape[1] == sample.shape[1] * 2 and not learned:
            raise ValueError("model_output has 2*C channels but variance_type is not learned")
        return ops.sched_step(sample, model_output, p, noise=noise)
