from .selfattention import SABlock
from .transformerblock import MLPBlock, TransformerBlock

__all__ = ["SABlock", "TransformerBlock", "MLPBlock"]
