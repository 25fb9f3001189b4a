"""PNDM scheduler (pseudo numerical methods: 4 Runge-Kutta warm-up rounds, then a 4th-order linear multi-step method).

Public contract of the reference's generative/networks/schedulers/pndm.py:55-317: constructor arguments, the `prk_timesteps` /
`plms_timesteps` / `timesteps` tables (50 requested steps -> 59 model evaluations, 100 -> 109), the `counter` / `ets` /
`cur_sample` / `cur_model_output` running state, and `step -> (prev_sample, None)`.

MI355X mapping: the history combinations ((55 e1 - 59 e2 + 37 e3 - 9 e4)/24, ...) are ONE launch of gm_lincomb and the transfer
formula (9) of the paper is gm_sched_step mode 2 -- two launches per step instead of the reference's 8-14 element-wise ones; both
kernels keep the reference's operation order with every op rounded, so an fp32 step is bit-identical to the reference on CPU.
The model outputs kept in `ets` are the tensors the UNet returned (no copies)."""
from __future__ import annotations

import numpy as np
import torch

from ... import ops
from ..._native import GmStepParams
from .scheduler import Scheduler


class PNDMPredictionType:
    EPSILON = "epsilon"
    V_PREDICTION = "v_prediction"
    _ALL = (EPSILON, V_PREDICTION)


class PNDMScheduler(Scheduler):
    """Liu et al. 2022, F-PNDM. Arguments as the reference (pndm.py:80-89)."""

    keeps_model_outputs = True  # `ets` aliases past model outputs: callers must not hand in a buffer they overwrite

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", skip_prk_steps: bool = False,
                 set_alpha_to_one: bool = False, prediction_type: str = PNDMPredictionType.EPSILON, steps_offset: int = 0,
                 **schedule_args) -> None:
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if prediction_type not in PNDMPredictionType._ALL:
            raise ValueError("Argument `prediction_type` must be a member of PNDMPredictionType")
        self.prediction_type = prediction_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4  # only F-PNDM (Runge-Kutta warm-up) exists in the reference (pndm.py:102-105)
        self.skip_prk_steps = skip_prk_steps
        self.steps_offset = steps_offset
        self.cur_model_output = None  # the reference's integer 0 (pndm.py:111): "nothing accumulated yet"
        self.counter = 0
        self.cur_sample = None
        self.ets: list = []
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        """Timestep tables of pndm.py:119-163: each of the last four strided timesteps is visited 4x by the Runge-Kutta rounds
        (at t and t - ratio/2), the remaining ones once by the multi-step method."""
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.num_train_timesteps`:"
                f" {self.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // self.num_inference_steps
        self._timesteps = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + self.steps_offset
        if self.skip_prk_steps:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = self._timesteps[::-1]
        else:
            tail = np.array(self._timesteps[-self.pndm_order:]).repeat(2)
            tail = tail + np.tile(np.array([0, self.num_train_timesteps // num_inference_steps // 2]), self.pndm_order)
            self.prk_timesteps = (tail[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = self._timesteps[:-3][::-1].copy()
        ts = np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.num_inference_steps = len(self.timesteps)  # model evaluations, not requested steps (pndm.py:158-159)
        self.ets = []
        self.counter = 0

    # -- reverse process ---------------------------------------------------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor):
        """One model evaluation's worth of progress; Runge-Kutta while `counter` is inside the prk table, multi-step after
        (pndm.py:165-186). Returns (previous sample, None)."""
        ops.require_device(model_output, sample)
        if self.counter < len(self.prk_timesteps) and not self.skip_prk_steps:
            return self.step_prk(model_output=model_output, timestep=timestep, sample=sample), None
        return self.step_plms(model_output=model_output, timestep=timestep, sample=sample), None

    def step_prk(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        """Classical RK4 over one strided interval, 4 model evaluations (pndm.py:188-229)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t = self._as_int(timestep)
        diff_to_prev = 0 if self.counter % 2 else self.num_train_timesteps // self.num_inference_steps // 2
        prev_timestep = t - diff_to_prev
        t = int(self.prk_timesteps[self.counter // 4 * 4])
        phase = self.counter % 4
        if phase == 0:
            self.cur_model_output = ops.lincomb([self.cur_model_output, model_output], [1.0, 1 / 6])
            self.ets.append(model_output)
            self.cur_sample = sample
        elif phase in (1, 2):
            self.cur_model_output = ops.lincomb([self.cur_model_output, model_output], [1.0, 1 / 3])
        else:
            model_output = ops.lincomb([self.cur_model_output, model_output], [1.0, 1 / 6])
            self.cur_model_output = None
        cur_sample = self.cur_sample if self.cur_sample is not None else sample
        prev_sample = self._get_prev_sample(cur_sample, t, prev_timestep, model_output)
        self.counter += 1
        return prev_sample

    def step_plms(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        """Adams-Bashforth over the stored model outputs, one evaluation per step (pndm.py:231-274)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if not self.skip_prk_steps and len(self.ets) < 3:
            raise ValueError(f"{self.__class__} can only be run AFTER scheduler has been run in 'prk' mode for at least 12 iterations ")
        t = self._as_int(timestep)
        ratio = self.num_train_timesteps // self.num_inference_steps
        prev_timestep = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = t
            t = t + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = ops.lincomb([model_output, e[-1]], [1.0, 1.0], post_div=2.0)
            sample = self.cur_sample
            self.cur_sample = None
        elif len(e) == 2:
            model_output = ops.lincomb([e[-1], e[-2]], [3.0, -1.0], post_div=2.0)
        elif len(e) == 3:
            model_output = ops.lincomb([e[-1], e[-2], e[-3]], [23.0, -16.0, 5.0], post_div=12.0)
        else:
            model_output = ops.lincomb([e[-1], e[-2], e[-3], e[-4]], [55.0, -59.0, 37.0, -9.0], post_mul=1 / 24)
        prev_sample = self._get_prev_sample(sample, t, prev_timestep, model_output)
        self.counter += 1
        return prev_sample

    def _get_prev_sample(self, sample: torch.Tensor, timestep: int, prev_timestep: int, model_output: torch.Tensor) -> torch.Tensor:
        """Formula (9) of the paper with x_t added to both sides (pndm.py:276-316); scalars on the host in fp32 torch-CPU
        arithmetic exactly as the reference evaluates them, tensor work in the fused step kernel (mode 2)."""
        ac = self._host_table("alphas_cumprod")
        a_t = ac[timestep]
        a_prev = ac[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod.detach().to("cpu", torch.float32)
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        p = GmStepParams()
        p.mode = 2
        p.pred_type = 2 if self.prediction_type == PNDMPredictionType.V_PREDICTION else 0
        p.c_sa, p.c_sb = self._f(a_t**0.5), self._f(b_t**0.5)
        p.k0 = self._f((a_prev / a_t) ** 0.5)
        p.k1 = self._f(a_prev - a_t)
        p.c_prev = self._f(a_t * b_prev**0.5 + (a_t * b_t * a_prev) ** 0.5)
        p.clip, p.noise_mode = 0, 0
        prev, _ = ops.sched_step(sample, model_output, p, want_x0=False)
        return prev
