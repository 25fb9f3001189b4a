from .inferer import (ControlNetDiffusionInferer, ControlNetLatentDiffusionInferer, DiffusionInferer, LatentDiffusionInferer,
                      VQVAETransformerInferer)

__all__ = ["DiffusionInferer", "LatentDiffusionInferer", "ControlNetDiffusionInferer", "ControlNetLatentDiffusionInferer",
           "VQVAETransformerInferer"]
