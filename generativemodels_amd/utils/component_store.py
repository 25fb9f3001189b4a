"""Named-component registry (API of the reference's generative/utils/component_store.py:27-117): the plugin point through
which users register their own noise schedules (`NoiseSchedules.add_def`)."""
from __future__ import annotations

import inspect
import keyword
import textwrap
from typing import Any, Callable, Iterator, NamedTuple, Tuple


class _Entry(NamedTuple):
    description: str
    value: Any


class ComponentStore:
    """A dict of (description, object) pairs keyed by identifier-like names, with decorator registration."""

    def __init__(self, name: str, description: str) -> None:
        self.components: dict[str, _Entry] = {}
        self.name = name
        self.description = description

    def add(self, name: str, desc: str, value: Any) -> Any:
        if not (name.isidentifier() and not keyword.iskeyword(name)):
            raise ValueError("Name of component must be valid Python identifier")
        self.components[name] = _Entry(desc, value)
        return value

    def add_def(self, name: str, desc: str) -> Callable:
        return lambda fn: self.add(name, desc, fn)

    def __contains__(self, name: str) -> bool:
        return name in self.components

    def __len__(self) -> int:
        return len(self.components)

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return ((k, e.value) for k, e in self.components.items())

    def __getitem__(self, name: str) -> Any:
        try:
            return self.components[name].value
        except KeyError:
            raise ValueError(f"Component '{name}' not found") from None

    def __getattr__(self, name: str) -> Any:
        comps = self.__dict__.get("components", {})
        if name in comps:
            return comps[name].value
        raise AttributeError(name)

    def __str__(self) -> str:
        lines = [f"Component Store '{self.name}': {self.description}", "Available components:"]
        for k, e in self.components.items():
            doc = inspect.getdoc(e.value) if getattr(e.value, "__doc__", None) else None
            lines.append(f"* {k}:" + ("" if doc else f" {e.description}"))
            if doc:
                lines.append(textwrap.indent(doc, "    "))
        return "\n".join(lines)
