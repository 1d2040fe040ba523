"""Builds the engine's shared library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmvskit_engine.so")
LIB32_PATH = os.path.join(LIB_DIR, "libmvskit_engine_cap32.so")  # the same sources with 32-view lists (-DMVS_LISTCAP=32)
LIB64_PATH = os.path.join(LIB_DIR, "libmvskit_engine_cap64.so")  # 64-view lists and 192-byte records (-DMVS_LISTCAP=64 -DMVS_MAX_IMAGES=64)
ENGINE_LIBS = {16: LIB_PATH, 32: LIB32_PATH, 64: LIB64_PATH}
# TEST build of the 16-view engine with the fault-injection hooks compiled in (-DMVS_FAULT_INJECTION: MVS_FAULT_PASS / MVS_FAULT_FILTER in the
# environment make one rank fail at a chosen point).  tests/test_gpu_dist.py loads it; the product libraries never read those variables.
FAULT_LIB_PATH = os.path.join(LIB_DIR, "libmvskit_engine_faultinj.so")
CAP_FLAGS = {16: [], 32: ["-DMVS_LISTCAP=32"], 64: ["-DMVS_LISTCAP=64", "-DMVS_MAX_IMAGES=64"]}
SOURCES = ["mvs_kernels.hip", "mvs_engine.cpp"]
DEPS = SOURCES + ["mvs_device.cuh", "mvs_check.cuh", "mvs_types.h", "mvs_kernels.h", os.path.join(ROOT, "include", "mvskit_engine.h")]

# -ffp-contract=off: the explicit fmaf chains in the source are the only fused operations (DESIGN.md,
# "engine arithmetic"), which is what lets the CPU oracle reproduce the results bit for bit.
# -fno-slp-vectorize: the SLP vectoriser pairs the scalar fp32 chains of the sampling loops into v_pk_* operations and
# pays for it with register shuffles (v_mov) and DPP moves that no longer fold into the adds: 18.9 vs 17.6 M patches/s.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wno-implicit-const-int-float-conversion"]


def needs_build(path: str = LIB_PATH) -> bool:
    if not os.path.exists(path):
        return True
    if os.environ.get("GRAFT_REPO_ROOT"):  # on a GPU box the snapshot's prebuilt library is used as is
        return False
    t = os.path.getmtime(path)
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(p) > t:
            return True
    return False


def build_engine(force: bool = False, verbose: bool = False, cap32: bool = False, cap: int = 0, fault_injection: bool = False) -> str:
    """cap: 16 (default), 32 or 64 views per m_images / m_vimages list (cap32=True is cap=32).
    fault_injection: the 16-view TEST build with -DMVS_FAULT_INJECTION (FAULT_LIB_PATH)."""
    cap = cap or (32 if cap32 else 16)
    out = FAULT_LIB_PATH if fault_injection else ENGINE_LIBS[cap]
    if not force and not needs_build(out):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + CAP_FLAGS[cap] + (["-DMVS_FAULT_INJECTION"] if fault_injection else []) + os.environ.get("MVS_EXTRA_FLAGS", "").split()
    cmd += ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libmvskit_host.so")
HOST_LIB32_PATH = os.path.join(LIB_DIR, "libmvskit_host_cap32.so")
HOST_LIB64_PATH = os.path.join(LIB_DIR, "libmvskit_host_cap64.so")
HOST_LIBS = {16: HOST_LIB_PATH, 32: HOST_LIB32_PATH, 64: HOST_LIB64_PATH}


def build_host(force: bool = False, verbose: bool = False, cap32: bool = False, cap: int = 0) -> str:
    """The host-side mirror of the reference classes (C++, g++), linked against the engine's C ABI -- one host library per
    engine build: libmvskit_host.so -> libmvskit_engine.so (view lists of 16), libmvskit_host_cap32.so -> the 32-view build,
    libmvskit_host_cap64.so -> the 64-view build (192-byte records: -DMVS_MAX_IMAGES=64)."""
    cap = cap or (32 if cap32 else 16)
    out = HOST_LIBS[cap]
    eng = ENGINE_LIBS[cap]
    srcs = [os.path.join(HOST_DIR, "pmmvps_host.cpp"), os.path.join(HOST_DIR, "jpeg_decode.cpp"), os.path.join(HOST_DIR, "ply_read.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, "pmmvps_host.hpp"), os.path.join(HOST_DIR, "jpeg_decode.hpp"), os.path.join(HOST_DIR, "ply_read.hpp"),
                   os.path.join(ROOT, "include", "mvskit_engine.h"), eng]
    if not force and os.path.exists(out) and os.environ.get("GRAFT_REPO_ROOT"):
        return out
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + (["-DMVS_MAX_IMAGES=64"] if cap == 64 else [])
    cmd += ["-I", os.path.join(ROOT, "include"), *srcs, "-o", out, "-L", LIB_DIR, "-l" + os.path.basename(eng)[3:-3], "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    for c in (16, 32, 64):
        print(build_engine(force=True, verbose=True, cap=c))
        print(build_host(force=True, verbose=True, cap=c))
    print(build_engine(force=True, verbose=True, fault_injection=True))
