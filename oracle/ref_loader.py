"""TEST INFRASTRUCTURE ONLY. Loads the UNMODIFIED reference as top-level `generative` (through oracle/monai_stub.py) from /root/reference --
i.e. in the build container only.  The reference is Python: it does not travel to the GPU box in any form (no source, no bytecode); what travels
are the vectors it produced here (tests/golden/*.pt, each with the committed script that made it).  Returns None where the tree is absent."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("GM_REFERENCE_ROOT", "/root/reference")


def reference_root():
    """-> the directory that holds the `generative` package, or None."""
    return REFERENCE_ROOT if os.path.isdir(os.path.join(REFERENCE_ROOT, "generative")) else None


def reference_available() -> bool:
    return reference_root() is not None


def load_reference():
    """-> module `generative` (the reference), or None when no reference tree is present (the GPU box)."""
    root = reference_root()
    if root is None:
        return None
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import monai_stub  # noqa: F401  (registers monai.* in sys.modules)

    if root not in sys.path:
        sys.path.insert(0, root)
    import generative  # noqa: F401
    import generative.inferers  # noqa: F401
    import generative.networks.nets  # noqa: F401
    import generative.networks.schedulers  # noqa: F401
    return generative
