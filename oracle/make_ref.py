"""TEST INFRASTRUCTURE ONLY -- the recipe that builds `oracle/_ref/`: the UNMODIFIED reference, byte-compiled from the sources where they lie.

`python oracle/make_ref.py` (also called by `__graft_entry__.build()` when /root/reference exists) compiles every module of
`/root/reference/generative` with `py_compile` into sourceless `oracle/_ref/generative/**/*.pyc` -- build OUTPUTS only, no reference source is
copied anywhere.  `oracle/_ref/` is git-ignored (it never enters the history) but not gpurun-ignored, so it travels to the GPU box like the built
`.so` files; there `oracle/ref_loader.py` falls back to it, and `bench.py`'s `cpu_baseline` leg times the reference's own
`DiffusionModelUNet.forward` + `DDIMScheduler.step` on the host cores (`"kind": "reference"`).  Nothing in the product package reads it.
The interpreter on the GPU box is the one of this image (same magic number); a stale or foreign `_ref` fails to import and the loader reports
"absent" (the bench then says `"kind": "port"`)."""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.environ.get("GM_REFERENCE_ROOT", "/root/reference")
OUT_ROOT = os.path.join(HERE, "_ref")


def build_ref(verbose: bool = False) -> str | None:
    src_pkg = os.path.join(SRC_ROOT, "generative")
    if not os.path.isdir(src_pkg):
        return None  # (the GPU box: nothing to build from, the shipped _ref is used as it is)
    out_pkg = os.path.join(OUT_ROOT, "generative")
    stamp = os.path.join(OUT_ROOT, "STAMP")
    newest = max(os.path.getmtime(os.path.join(d, f)) for d, _, fs in os.walk(src_pkg) for f in fs if f.endswith(".py"))
    tag = f"{sys.version_info[0]}.{sys.version_info[1]} {newest:.0f}"
    if os.path.exists(stamp) and open(stamp).read() == tag:
        return out_pkg
    shutil.rmtree(OUT_ROOT, ignore_errors=True)
    n = 0
    for d, _, fs in os.walk(src_pkg):
        rel = os.path.relpath(d, src_pkg)
        for f in fs:
            if f.endswith(".py"):
                dst = os.path.normpath(os.path.join(out_pkg, rel, f + "c"))  # sourceless layout: x.pyc next to where x.py would be
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                py_compile.compile(os.path.join(d, f), cfile=dst, dfile=f"<reference>/generative/{rel}/{f}", doraise=True, optimize=0)
                n += 1
    with open(stamp, "w") as fh:
        fh.write(tag)
    if verbose:
        print(f"oracle/_ref: {n} reference modules byte-compiled from {src_pkg}")
    return out_pkg


if __name__ == "__main__":
    build_ref(verbose=True)
