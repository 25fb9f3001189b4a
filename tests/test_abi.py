"""CPU (-m "not gpu"): the C-ABI shared library builds, loads and exports exactly the symbols include/gm_amd.h declares
(no compute call is made -- there is no GPU here)."""
import ctypes
import os
import re

from generativemodels_amd import _build, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gm_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = _build.build_native()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/gm_amd.h but not exported by libgmamd.so"


def test_ctypes_prototypes_cover_the_header():
    assert sorted(_native.PROTOTYPES.keys()) == _header_functions()
    lib = _native.lib()
    assert lib.gm_abi_version() == 1


def test_struct_layouts_match_the_header_field_order():
    src = open(os.path.join(ROOT, "include", "gm_amd.h")).read()
    for cname, cls in (("GmStepParams", _native.GmStepParams), ("GmConvDesc", _native.GmConvDesc), ("GmAttnDesc", _native.GmAttnDesc),
                       ("GmGnTables", _native.GmGnTables)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            names = stmt.replace("*", " ").split(",")
            first = names[0].split()
            fields.append(first[-1])
            fields += [n.strip() for n in names[1:]]
        # `name[2]` in the header is a 2-element ctypes array field `name`
        arrays = {f.split("[")[0]: int(f.split("[")[1].rstrip("]")) for f in fields if "[" in f}
        fields = [f.split("[")[0] for f in fields]
        assert fields == [f[0] for f in cls._fields_], cname
        for fname, ftype in cls._fields_:
            if fname in arrays:
                assert getattr(ftype, "_length_", None) == arrays[fname], (cname, fname)


def test_pure_host_entry_points_run_without_a_gpu():
    lib = _native.lib()
    bm, bn = ctypes.c_int(), ctypes.c_int()
    assert lib.gm_conv_cfg_tile(0, ctypes.byref(bm), ctypes.byref(bn)) == 0 and (bm.value, bn.value) == (256, 64)
    assert lib.gm_packed_conv_weight_elems(64, 64, 3, 3, 3, 1) == 2 * 27 * 64 * 32
    assert lib.gm_gn_workspace_bytes(1, 128 ** 3, 64, 32, 1) > 0
    assert lib.gm_attention_max_head_dim() == 256
    for policy in (-1, 0, 16, 0):  # grid policy of the LDS-DMA convolutions: process-wide host state, no device call; leave the default (0)
        lib.gm_conv_dma_set_persistent(policy)
    lib.gm_attention_dma_set_variant(0, 0)
