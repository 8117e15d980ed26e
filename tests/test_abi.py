"""CPU-side checks of the C ABI: the library loads and exports every symbol include/mst_engine.h
declares (no compute calls -- those need a GPU)."""
import os
import re

import mst_amd  # noqa: F401
from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mst_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mst_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mst_amd import _native
    lib = _native.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.SIGNATURES) == names          # binding and header agree
    assert lib.mst_version() >= 1


def test_struct_layouts_match_header():
    import ctypes as C
    from mst_amd import _native
    assert C.sizeof(_native.MstConfig) == 10 * 4
    # 9 int32 + float + (4 bytes pad) + uint64 + 6 pointers
    assert C.sizeof(_native.MstLoopArgs) == 9 * 4 + 4 + 8 + 6 * 8
    assert _native.MstLoopArgs.seed.offset == 40
    assert _native.MstLoopArgs.scale_dev.offset == 48


def test_bad_arguments_are_reported_without_a_gpu():
    import ctypes as C
    from mst_amd import _native
    lib = _native.lib()
    h = C.c_void_p()
    cfg = _native.MstConfig(263, 196, 4, 256, 4, 1024, 8, 512, 5000, 0)   # latent_dim 256: unsupported
    assert lib.mst_engine_create(C.byref(cfg), C.byref(h)) != 0
    assert b"512" in lib.mst_last_error()
    cfg = _native.MstConfig(263, 500, 4, 512, 4, 1024, 8, 512, 5000, 0)   # too many frames
    assert lib.mst_engine_create(C.byref(cfg), C.byref(h)) != 0
    assert b"max_frames" in lib.mst_last_error()
