"""generativemodels_amd -- MI355X-native (gfx950) implementation of the diffusion sampling path of
Project-MONAI/GenerativeModels: DiffusionModelUNet, AutoencoderKL, VQVAE, the DDPM/DDIM schedulers and the
(Latent)DiffusionInferer, behind the reference's own Python API and state_dict layout, executed by the hand-written HIP
kernels of libgmamd.so (include/gm_amd.h).

Sub-modules mirror the reference package layout (`networks.nets`, `networks.schedulers`, `networks.layers`, `inferers`,
`utils`); `install_as_generative()` aliases them as `generative.*` for true drop-in use."""
from __future__ import annotations

import importlib
import sys

__version__ = "0.1.0"


def __getattr__(name):
    # `generativemodels_amd.autocast(torch.bfloat16)`: fp32 master parameters, bf16 compute (ops.autocast); resolved lazily so that importing
    # the package does not import torch
    if name in ("autocast", "autocast_dtype"):
        from . import ops

        return getattr(ops, name)
    if name == "GraphedForwardBackward":  # HIP-graph replay of a training step's forward + backward (graphs.py)
        from .graphs import GraphedForwardBackward

        return GraphedForwardBackward
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

_MIRRORED = ["", ".networks", ".networks.nets", ".networks.blocks", ".networks.blocks.spade_norm", ".networks.schedulers", ".networks.layers",
             ".inferers", ".utils"]


def install_as_generative(force: bool = False) -> None:
    """Register this package's sub-modules under the reference's import paths (`generative.networks.nets`, ...).
    Refuses to shadow an already imported `generative` package unless `force`."""
    if "generative" in sys.modules and not force:
        mod = sys.modules["generative"]
        if getattr(mod, "__name__", "") != __name__ and not getattr(mod, "__file__", "").startswith(__path__[0]):
            raise RuntimeError("a different `generative` package is already imported; pass force=True to replace it")
    for suffix in _MIRRORED:
        sys.modules["generative" + suffix] = importlib.import_module(__name__ + suffix)
