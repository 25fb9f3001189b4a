from .inferer import (ControlNetDiffusionInferer, ControlNetLatentDiffusionInferer, DiffusionInferer, LatentDiffusionInferer)

__all__ = ["DiffusionInferer", "LatentDiffusionInferer", "ControlNetDiffusionInferer", "ControlNetLatentDiffusionInferer"]
