"""Noise-schedule registry and scheduler base class.

Mirrors the public contract of the reference's generative/networks/schedulers/scheduler.py:40-200 (NoiseSchedules names and
keyword arguments, table attributes `betas / alphas / alphas_cumprod / one / timesteps`, `add_noise`, `get_velocity`).
The tables are built on the host with fp32 torch-CPU expressions -- the same ones the reference uses, so they are bit-equal --
and stay plain CPU attributes (scheduler.py:155-167); the per-sample mixing itself runs in one fused HIP kernel
(gm_axpby_rows) instead of the reference's 3-4 element-wise launches.

Deliberate deviation: the reference *mutates* `self.alphas_cumprod` to the sample's device and dtype inside add_noise
(scheduler.py:182,193).  Here the fp32 CPU table is left untouched; the dtype-rounded coefficients the reference would
obtain are reproduced from a per-(device, dtype) cached copy."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...utils import ComponentStore

NoiseSchedules = ComponentStore("NoiseSchedules", "Functions to generate noise schedules")


@NoiseSchedules.add_def("linear_beta", "Linear beta schedule")
def _linear_beta(num_train_timesteps: int, beta_start: float = 1e-4, beta_end: float = 2e-2):
    """betas linearly spaced in [beta_start, beta_end]."""
    return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)


@NoiseSchedules.add_def("scaled_linear_beta", "Scaled linear beta schedule")
def _scaled_linear_beta(num_train_timesteps: int, beta_start: float = 1e-4, beta_end: float = 2e-2):
    """sqrt(beta) linearly spaced, then squared (the latent-diffusion schedule)."""
    return torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2


@NoiseSchedules.add_def("sigmoid_beta", "Sigmoid beta schedule")
def _sigmoid_beta(num_train_timesteps: int, beta_start: float = 1e-4, beta_end: float = 2e-2, sig_range: float = 6):
    """betas follow a sigmoid ramp over [-sig_range, sig_range]."""
    ramp = torch.linspace(-sig_range, sig_range, num_train_timesteps)
    return torch.sigmoid(ramp) * (beta_end - beta_start) + beta_start


@NoiseSchedules.add_def("cosine", "Cosine schedule")
def _cosine_beta(num_train_timesteps: int, s: float = 8e-3):
    """Nichol & Dhariwal cosine schedule; returns the (betas, alphas, alphas_cumprod) triple."""
    grid = torch.linspace(0, num_train_timesteps, num_train_timesteps + 1)
    cum = torch.cos(((grid / num_train_timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    cum /= cum[0].item()
    alphas = torch.clip(cum[1:] / cum[:-1], 0.0001, 0.9999)
    return 1.0 - alphas, alphas, cum[:-1]


class _StrEnum(str):
    pass


def _members(cls):
    return [v for k, v in vars(cls).items() if not k.startswith("_") and isinstance(v, str)]


class Scheduler(nn.Module):
    """Base scheduler: noise tables + the forward-process mixing ops."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", **schedule_args) -> None:
        super().__init__()
        schedule_args["num_train_timesteps"] = num_train_timesteps
        made = NoiseSchedules[schedule](**schedule_args)
        if isinstance(made, tuple):
            self.betas, self.alphas, self.alphas_cumprod = made
        else:
            self.betas = made
            self.alphas = 1.0 - self.betas
            self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self._mix_tables: dict = {}

    # -- helpers ----------------------------------------------------------------------------------------------------------
    def _host_table(self, name: str) -> torch.Tensor:
        return getattr(self, name).detach().to("cpu", torch.float32)

    def _mix_coefficients(self, timesteps: torch.Tensor, device, dtype):
        """(sqrt(abar_t), sqrt(1 - abar_t)) per sample as fp32 device vectors, rounded through `dtype` exactly like the
        reference's in-dtype table lookup (scheduler.py:182-186)."""
        ac = self._host_table("alphas_cumprod")
        key = (str(device), dtype, ac.data_ptr(), ac._version)
        tabs = self._mix_tables.get(key)
        if tabs is None:
            acd = ac.to(dtype)
            sa = (acd**0.5).to(torch.float32)
            sb = ((1 - acd) ** 0.5).to(torch.float32)
            tabs = (sa.to(device), sb.to(device))
            self._mix_tables = {key: tabs}
        idx = timesteps.to(device=device, dtype=torch.long)
        return tabs[0][idx], tabs[1][idx]

    # -- forward process --------------------------------------------------------------------------------------------------
    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(abar_t) * x0 + sqrt(1 - abar_t) * eps, one timestep per sample (reference scheduler.py:169-189)."""
        ops.require_device(original_samples, noise)
        original_samples, noise = _promote(original_samples, noise)
        a, b = self._mix_coefficients(timesteps, original_samples.device, original_samples.dtype)
        return ops.axpby_rows(original_samples, noise, a, b)

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(abar_t) * eps - sqrt(1 - abar_t) * x (reference scheduler.py:191-200)."""
        ops.require_device(sample, noise)
        sample, noise = _promote(sample, noise)
        a, b = self._mix_coefficients(timesteps, sample.device, sample.dtype)
        return ops.axpby_rows(noise, sample, a, -b)

    @staticmethod
    def _as_int(timestep) -> int:
        return int(timestep.item()) if torch.is_tensor(timestep) else int(timestep)

    @staticmethod
    def _f(x) -> float:
        return float(x.item()) if torch.is_tensor(x) else float(x)


def _promote(a: torch.Tensor, b: torch.Tensor):
    """Operands of a mixing op in their promoted dtype, as torch's type promotion gives the reference (a bf16 latent of an autocast'ed
    auto-encoder mixed with fp32 noise is an fp32 result: scheduler.py:186-189)."""
    if a.dtype == b.dtype:
        return a, b
    dt = torch.promote_types(a.dtype, b.dtype)
    return ops.cast(a, dt), ops.cast(b, dt)


def x0_prediction_code(prediction_type: str) -> int:
    return {"epsilon": 0, "sample": 1, "v_prediction": 2}[str(prediction_type)]


def inference_timesteps(num_train_timesteps: int, num_inference_steps: int) -> torch.Tensor:
    """Evenly strided, descending int64 timesteps (reference ddim.py:139-143 / ddpm.py:127-131)."""
    import numpy as np

    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
    return torch.from_numpy(ts)
