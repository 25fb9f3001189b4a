"""Builds the C-ABI HIP library (generativemodels_amd/lib/libgmamd.so) for gfx950 with hipcc. In-tree, no torch headers.

`python -m generativemodels_amd._build` or `__graft_entry__.build()`; cross-compiles without a GPU."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libgmamd.so")
OBJDIR = os.path.join(HERE, "build")
# "file.hip#k": the file compiled with -DGM_DMA_PART=k into its own object (conv_dma.hip: 26 kernel instantiations, 4.5 minutes as one
# translation unit -- five parts build in parallel)
SOURCES = ["capi.cpp", "elementwise.hip", "groupnorm.hip", "conv.hip", "conv_fast.hip", "conv_dma.hip#1", "conv_dma.hip#0", "conv_dma.hip#2", "conv_dma.hip#3",
           "conv_dma.hip#4", "conv_sk.hip", "conv_sn.hip", "conv_edge.hip", "attention.hip", "attention_dma.hip", "attention_bwd.hip", "attention_bwd_dma.hip", "transformer_ops.hip", "decode_step.hip", "small_ops.hip", "backward.hip", "vq.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def hip_runtime_library() -> str:
    """The HIP runtime the library is linked against. It MUST be the one the host framework uses: kernels are enqueued on the
    caller's hipStream_t, which only means something inside the same runtime instance. PyTorch-ROCm wheels bundle their own
    libamdhip64.so (without a SONAME), so linking the system copy would silently put a second runtime -- with its own queues,
    unordered against torch's -- into the process. Override with GM_HIP_RUNTIME=/path/to/libamdhip64.so for other hosts."""
    env = os.environ.get("GM_HIP_RUNTIME")
    if env:
        return env
    try:
        import torch

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    except Exception:  # pragma: no cover
        pass
    return "/opt/rocm/lib/libamdhip64.so"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file extra flags: the scheduler / mixing kernels must reproduce the reference's fp32 op order bit for bit, so fused
# multiply-add contraction is off for that translation unit (HIP's __fmul_rn/__fadd_rn are plain operators and DO contract)
EXTRA_FLAGS = {"elementwise.hip": ["-ffp-contract=off"]}


# bench-only variants: extra defines, their own object directory and library name (the product library is never built with them).  The table
# lives in _build_variants.py, which is NOT a dependency of the objects: adding an experiment does not rebuild the product library.
from ._build_variants import VARIANTS  # noqa: E402
# translation units whose LDS-DMA inline assembly does not survive the sanitizer's instrumentation (its "s" operands stop being provably uniform):
# compiled WITHOUT -fsanitize in the asan variant; their global accesses are the DMA requests themselves, which no sanitizer sees anyway
ASAN_PLAIN = {"conv_dma.hip", "conv_sk.hip", "conv_sn.hip", "attention_dma.hip", "attention_bwd_dma.hip", "conv_edge.hip"}
_variant = None


def _compile(src: str) -> str:
    part = None
    if "#" in src:
        src, part = src.split("#")
    obj = os.path.join(OBJDIR if _variant is None else OBJDIR + "_" + _variant, os.path.splitext(src)[0] + ("" if part is None else f"_p{part}") + ".o")
    srcp = os.path.join(CSRC, src)
    deps = [srcp, os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    if _stale(obj, deps):
        extra = os.environ.get("GM_EXTRA_HIPCC_FLAGS", "").split() + (VARIANTS[_variant] if _variant else []) + ([] if part is None else [f"-DGM_DMA_PART={part}"])
        if _variant == "asan" and src in ASAN_PLAIN:  # (still built for gfx950:xnack+ so that the library loads as one code object family)
            extra = [f for f in extra if not f.startswith("-fsanitize") and f != "-shared-libsan"]
        flags = [f if not f.startswith("--offload-arch=") or _variant != "asan" else f"--offload-arch={ARCH}:xnack+" for f in FLAGS]
        cmd = [_hipcc(), *flags, *EXTRA_FLAGS.get(src, []), *extra, "-x", "hip", "-c", srcp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_native(force: bool = False, verbose: bool = False, variant: str | None = None) -> str:
    """variant: None = the product library lib/libgmamd.so; "timeline" / "ablate" = bench-only builds lib/libgmamd_<variant>.so
    (loaded through GM_NATIVE_LIB by the tools that need them)."""
    global _variant
    _variant = variant
    objdir = OBJDIR if variant is None else OBJDIR + "_" + variant
    libpath = LIBPATH if variant is None else os.path.join(LIBDIR, f"libgmamd_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    try:
        with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            objs = list(ex.map(_compile, SOURCES))
    finally:
        _variant = None
    return _link(objs, libpath, force, verbose, asan=variant == "asan")


def _link(objs, LIBPATH, force, verbose, asan=False) -> str:
    if force or _stale(LIBPATH, objs):
        rt = hip_runtime_library()
        linker = shutil.which("g++") or shutil.which("c++") or _hipcc()
        cmd = [linker, "-shared", "-fPIC", *objs, "-o", LIBPATH, rt, f"-Wl,-rpath,{os.path.dirname(rt)}", "-Wl,--enable-new-dtags"]
        if asan:  # the sanitizer runtime comes from LD_PRELOAD in the process that loads this library: leave its symbols undefined
            cmd = ["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", *objs, "-o", LIBPATH, rt,
                   f"-Wl,-rpath,{os.path.dirname(rt)}", "-Wl,--enable-new-dtags"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", LIBPATH, os.path.getsize(LIBPATH), "bytes")
    return LIBPATH


if __name__ == "__main__":
    build_native(force="--force" in sys.argv, verbose=True, variant=sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None)
