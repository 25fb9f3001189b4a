from .autoencoderkl import AutoencoderKL
from .diffusion_model_unet import DiffusionModelUNet
from .vqvae import VQVAE

__all__ = ["AutoencoderKL", "DiffusionModelUNet", "VQVAE"]
