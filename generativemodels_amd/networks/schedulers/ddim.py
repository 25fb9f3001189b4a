"""DDIM scheduler: constructor / `set_timesteps` / `step` / `reversed_step` contract of the reference's
generative/networks/schedulers/ddim.py:55-301, with the reverse step fused into one HIP kernel (gm_sched_step)."""
from __future__ import annotations

import torch

from ... import host_noise, ops
from ..._native import GmStepParams
from .scheduler import Scheduler, inference_timesteps, x0_prediction_code


class DDIMPredictionType:
    EPSILON = "epsilon"
    SAMPLE = "sample"
    V_PREDICTION = "v_prediction"
    _ALL = (EPSILON, SAMPLE, V_PREDICTION)


class DDIMScheduler(Scheduler):
    """Song et al. 2020 implicit sampler. Arguments as the reference (ddim.py:79-90)."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = DDIMPredictionType.EPSILON,
                 clip_sample_min: int = -1, clip_sample_max: int = 1, **schedule_args) -> None:
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if prediction_type not in DDIMPredictionType._ALL:
            raise ValueError("Argument `prediction_type` must be a member of DDIMPredictionType")
        if clip_sample_min >= clip_sample_max:
            raise ValueError("clip_sample_min must be < clip_sample_max")
        self.prediction_type = prediction_type
        # abar of the step "before t = 0": 1 (no noise left) or abar_0 (ddim.py:101-105)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        # abar of the step "after t = T-1" used by reversed_step: 0 (pure noise) or abar_{T-1} (ddim.py:107-109)
        self.first_alpha_cumprod = torch.tensor(0.0) if set_alpha_to_one else self.alphas_cumprod[-1]
        self.init_noise_sigma = 1.0
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.clip_sample = clip_sample
        self.clip_sample_values = [clip_sample_min, clip_sample_max]
        self.steps_offset = steps_offset
        self.set_timesteps(self.num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.num_train_timesteps`:"
                f" {self.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        ts = inference_timesteps(self.num_train_timesteps, num_inference_steps).to(device)
        self.timesteps = ts + self.steps_offset

    def _abar(self, t: int) -> torch.Tensor:
        return self._host_table("alphas_cumprod")[t] if t >= 0 else self.final_alpha_cumprod.detach().to("cpu", torch.float32)

    def _get_variance(self, timestep: int, prev_timestep: int) -> torch.Tensor:
        a_t, a_prev = self._abar(self._as_int(timestep)), self._abar(self._as_int(prev_timestep))
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0,
             generator: torch.Generator | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """x_t -> x_{t - T/n}; returns (previous sample, predicted x_0) (reference ddim.py:156-237)."""
        ops.require_device(model_output, sample)
        t = self._as_int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t, a_prev = self._abar(t), self._abar(prev_t)
        b_t = 1 - a_t
        variance = self._get_variance(t, prev_t)
        std_dev_t = eta * variance**0.5
        p = GmStepParams()
        p.mode, p.pred_type = 0, x0_prediction_code(self.prediction_type)
        p.c_sa, p.c_sb = self._f(a_t**0.5), self._f(b_t**0.5)
        p.clip = int(bool(self.clip_sample))
        p.clip_lo, p.clip_hi = float(self.clip_sample_values[0]), float(self.clip_sample_values[1])
        p.c_prev = self._f(a_prev**0.5)
        p.c_dir = self._f((1 - a_prev - std_dev_t**2) ** 0.5)
        noise = None
        p.noise_mode = 0
        if eta > 0:
            # CPU-generator draw + H2D copy, like the reference (ddim.py:229-235): identical random stream (bf16: from the generator's byte draws, host_noise.py)
            noise = host_noise.randn(model_output.shape, model_output.dtype, generator, model_output.device, as_bits=True)
            p.noise_mode, p.c_noise = 1, self._f(variance**0.5 * eta)
        return ops.sched_step(sample, model_output, p, noise=noise)

    def reversed_step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Deterministic DDIM inversion x_t -> x_{t + T/n} (reference ddim.py:239-301): the same fused kernel with the
        "previous" cumulative alpha taken at the *next* timestep."""
        ops.require_device(model_output, sample)
        t = self._as_int(timestep)
        nxt = t + self.num_train_timesteps // self.num_inference_steps
        a_t = self._abar(t)
        a_next = (self._host_table("alphas_cumprod")[nxt] if nxt < len(self.alphas_cumprod)
                  else self.first_alpha_cumprod.detach().to("cpu", torch.float32))
        b_t = 1 - a_t
        p = GmStepParams()
        p.mode, p.pred_type = 0, x0_prediction_code(self.prediction_type)
        p.c_sa, p.c_sb = self._f(a_t**0.5), self._f(b_t**0.5)
        p.clip = int(bool(self.clip_sample))
        p.clip_lo, p.clip_hi = float(self.clip_sample_values[0]), float(self.clip_sample_values[1])
        p.c_prev = self._f(a_next**0.5)
        p.c_dir = self._f((1 - a_next) ** 0.5)
        p.noise_mode = 0
        return ops.sched_step(sample, model_output, p)
