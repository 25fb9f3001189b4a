from .ddim import DDIMPredictionType, DDIMScheduler
from .ddpm import DDPMPredictionType, DDPMScheduler, DDPMVarianceType
from .scheduler import NoiseSchedules, Scheduler

__all__ = ["DDIMScheduler", "DDIMPredictionType", "DDPMScheduler", "DDPMPredictionType", "DDPMVarianceType",
           "NoiseSchedules", "Scheduler"]
