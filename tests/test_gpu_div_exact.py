"""rcp_rn / div_rn of mvs_device.cuh -- v_rcp_f32 + one Newton step in fma; Markstein's quotient correction on top -- stand in for
IEEE-754 division in the hot geometry (projection, ray normalisation, units, 1 / msd, robust INCC, isNeighbor): 3 and 6 instructions
where the compiler's expansion takes 11.  They must return the very bits `1.0f / x` and `a / b` return, or the engine would leave
the oracle's arithmetic.  tools/microbench/div_exact.hip checks that on the device: the reciprocal EXHAUSTIVELY -- every float with
2^-120 <= |x| < 2^121, 4.04e9 inputs -- and the quotient on 1.18e10 pairs (pseudo-random and hard mantissa patterns)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rcp_and_div_return_the_ieee_quotient(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "div_exact")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-Wno-unused-result", "-o", exe,
                           os.path.join(ROOT, "tools", "microbench", "div_exact.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert "4043309056 inputs" in r.stdout and ", 0 differ from 1.0f / x" in r.stdout


@pytest.mark.gpu
def test_run_lookup_by_marks_is_the_search(tmp_path):
    """findNeighbors' row walk (mvs_check.cuh, MK): the run an id belongs to comes from marks and four DPP running maxima instead of
    a binary search per id.  tools/microbench/run_lookup.hip runs both on 65536 waves of 64 runs of random lengths (empty runs, runs
    longer than the 256-position window) and counts the positions that differ."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "run_lookup")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-Wno-unused-result", "-o", exe,
                           os.path.join(ROOT, "tools", "microbench", "run_lookup.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "PASS" in r.stdout and " 0 positions differ" in r.stdout, r.stdout + r.stderr
