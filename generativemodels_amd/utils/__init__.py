from .component_store import ComponentStore
from .misc import unsqueeze_left, unsqueeze_right
from .ordering import Ordering, OrderingTransformations, OrderingType

__all__ = ["ComponentStore", "unsqueeze_left", "unsqueeze_right", "Ordering", "OrderingType", "OrderingTransformations"]
