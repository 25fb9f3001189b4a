from .component_store import ComponentStore
from .misc import unsqueeze_left, unsqueeze_right

__all__ = ["ComponentStore", "unsqueeze_left", "unsqueeze_right"]
