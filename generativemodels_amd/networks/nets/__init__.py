from .autoencoderkl import AutoencoderKL
from .controlnet import ControlNet, ControlNetConditioningEmbedding, copy_weights_to_controlnet
from .diffusion_model_unet import DiffusionModelUNet
from .transformer import DecoderOnlyTransformer
from .vqvae import VQVAE

__all__ = ["AutoencoderKL", "ControlNet", "ControlNetConditioningEmbedding", "DecoderOnlyTransformer", "DiffusionModelUNet", "VQVAE",
           "copy_weights_to_controlnet"]
