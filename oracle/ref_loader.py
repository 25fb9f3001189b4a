"""TEST INFRASTRUCTURE ONLY. Loads the UNMODIFIED reference as top-level `generative` (through oracle/monai_stub.py):
from /root/reference where that tree exists (the build container), else from `oracle/_ref/` -- the same modules byte-compiled by
`oracle/make_ref.py` (build outputs, git-ignored, shipped to the GPU box like the built `.so`).  Returns None when neither is there."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("GM_REFERENCE_ROOT", "/root/reference")
COMPILED_ROOT = os.path.join(HERE, "_ref")


def reference_root():
    """-> the directory that holds the `generative` package (source tree or compiled `_ref`), or None."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "generative")):
        return REFERENCE_ROOT
    if os.path.exists(os.path.join(COMPILED_ROOT, "generative", "__init__.pyc")):
        return COMPILED_ROOT
    return None


def reference_available() -> bool:
    return reference_root() is not None


def load_reference():
    """-> module `generative` (the reference), or None when no reference tree (source or compiled) is usable."""
    root = reference_root()
    if root is None:
        return None
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import monai_stub  # noqa: F401  (registers monai.* in sys.modules)

    if root not in sys.path:
        sys.path.insert(0, root)
    try:
        import generative  # noqa: F401
        import generative.inferers  # noqa: F401
        import generative.networks.nets  # noqa: F401
        import generative.networks.schedulers  # noqa: F401
    except Exception as ex:  # a compiled tree from another interpreter (bad magic number): report absent, the bench says "port"
        if root is COMPILED_ROOT:
            print(f"[ref_loader] oracle/_ref is unusable here ({type(ex).__name__}: {ex}); falling back to the restatement", file=sys.stderr)
            for k in [k for k in sys.modules if k == "generative" or k.startswith("generative.")]:
                del sys.modules[k]
            sys.path.remove(root)
            return None
        raise
    return generative
