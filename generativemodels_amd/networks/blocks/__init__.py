from .selfattention import SABlock
from .spade_norm import SPADE
from .transformerblock import MLPBlock, TransformerBlock

__all__ = ["SABlock", "TransformerBlock", "MLPBlock", "SPADE"]
