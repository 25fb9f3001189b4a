from .inferer import DiffusionInferer, LatentDiffusionInferer

__all__ = ["DiffusionInferer", "LatentDiffusionInferer"]
