"""TEST INFRASTRUCTURE ONLY. Loads the UNMODIFIED reference from /root/reference as top-level `generative` (through
oracle/monai_stub.py). Only usable in the build container; returns None elsewhere (the GPU box has no /root/reference)."""
import os
import sys

REFERENCE_ROOT = os.environ.get("GM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "generative"))


def load_reference():
    """-> module `generative` (the reference), or None when the reference tree is absent."""
    if not reference_available():
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import monai_stub  # noqa: F401  (registers monai.* in sys.modules)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import generative  # noqa: F401
    import generative.inferers  # noqa: F401
    import generative.networks.nets  # noqa: F401
    import generative.networks.schedulers  # noqa: F401

    return generative
