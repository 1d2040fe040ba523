"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mvskit_engine.h declares,
and refuses to run without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mvskit_amd import build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_engine()
    return engine.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mvskit_engine.h")).read()
    declared = set(re.findall(r"\b(mvs_[a-z_]+)\s*\(", hdr))
    declared -= {"mvs_engine"}  # the opaque struct name
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_record_layouts_match_header():
    assert engine.PATCH_DTYPE.itemsize == 128
    assert engine.PATCH_DTYPE.fields["images"][1] == 64 and engine.PATCH_DTYPE.fields["vimages"][1] == 96
    assert C.sizeof(engine.Counters) == 80
    assert C.sizeof(engine.Config) == 96


def test_default_config_follows_option_defaults(lib):
    c = engine.Config()
    lib.mvs_default_config(C.byref(c))
    # Option::Option, pmmvps/option.cpp:19-33
    assert (c.level, c.csize, c.wsize, c.minImageNum) == (1, 2, 7, 3)
    assert abs(c.nccThreshold - 0.7) < 1e-7 and abs(c.maxAngleThreshold - np.float32(10 * np.pi / 180)) < 1e-7
    assert abs(c.quadThreshold - 2.5) < 1e-7 and c.max_propag == 2


def test_create_rejects_bad_config_and_missing_gpu(lib):
    c = engine.Config()
    lib.mvs_default_config(C.byref(c))
    h = C.c_void_p()
    c.nviews = 0
    assert lib.mvs_engine_create(C.byref(c), C.byref(h)) == -1  # MVS_ERR_ARG
    c.nviews = 3
    c.wsize = 9
    assert lib.mvs_engine_create(C.byref(c), C.byref(h)) == -1
    if lib.mvs_device_count() == 0:
        c.wsize = 7
        assert lib.mvs_engine_create(C.byref(c), C.byref(h)) == -5  # MVS_ERR_NO_DEVICE: no CPU fallback
        assert b"no HIP device" in lib.mvs_last_error()
        with pytest.raises(engine.EngineError):
            engine.Engine(3)


def test_cap32_library_exports_the_same_abi():
    """libmvskit_engine_cap32.so = the same sources built with -DMVS_LISTCAP=32 (view lists of up to 32 entries)."""
    build.build_engine(cap32=True)
    lib32 = engine.load_library(cap32=True)
    for name in engine.EXPORTS:
        assert hasattr(lib32, name), name
    assert lib32.mvs_list_cap() == 32 and engine.load_library().mvs_list_cap() == 16


def test_struct_sizes_of_the_binding():
    # mvs_timing: 3 floats, int32, float, (pad), 2 int64; mvs_filter_stats: 6 floats + 10 int64
    assert C.sizeof(engine.Timing) == 40
    assert C.sizeof(engine.FilterStats) == 24 + 10 * 8


def test_loopback_transport_exports_what_the_engine_binds():
    """tests/loopback_ccl (the shared-memory stand-in for RCCL that lets several engine ranks share the one GPU of a test box)
    must export exactly the eight entry points mvs_engine.cpp binds through MVS_CCL_LIBRARY."""
    import subprocess

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback_ccl")
    subprocess.check_call(["make", "-C", d, "-s"])
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(d, "_build", "libloopback_ccl.so")], text=True)
    have = {line.split()[-1] for line in out.splitlines() if " T " in line}
    want = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"}
    assert want <= have
    src = open(os.path.join(os.path.dirname(d), "..", "mvskit_amd", "csrc", "mvs_engine.cpp")).read()
    for name in want:
        assert f'"{name}"' in src, name
