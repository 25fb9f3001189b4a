from .autoencoderkl import AutoencoderKL
from .controlnet import ControlNet, ControlNetConditioningEmbedding, copy_weights_to_controlnet
from .diffusion_model_unet import DiffusionModelUNet
from .spade_autoencoderkl import SPADEAutoencoderKL
from .spade_diffusion_model_unet import SPADEDiffusionModelUNet
from .transformer import DecoderOnlyTransformer
from .vqvae import VQVAE

__all__ = ["AutoencoderKL", "ControlNet", "ControlNetConditioningEmbedding", "DecoderOnlyTransformer", "DiffusionModelUNet",
           "SPADEAutoencoderKL", "SPADEDiffusionModelUNet", "VQVAE", "copy_weights_to_controlnet"]
