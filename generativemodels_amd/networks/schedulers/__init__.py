from .ddim import DDIMPredictionType, DDIMScheduler
from .ddpm import DDPMPredictionType, DDPMScheduler, DDPMVarianceType
from .pndm import PNDMPredictionType, PNDMScheduler
from .scheduler import NoiseSchedules, Scheduler

__all__ = ["DDIMScheduler", "DDIMPredictionType", "DDPMScheduler", "DDPMPredictionType", "DDPMVarianceType",
           "PNDMScheduler", "PNDMPredictionType", "NoiseSchedules", "Scheduler"]
