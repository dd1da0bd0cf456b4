"""exp() and log() of the product (classpro_amd/csrc/cp_libm.h) against the host's libm, bit for bit.

The reference's floating-point results are whatever glibc's exp/log return; the decision path compares such values
where they are equal in mathematics (log(px*py) against log(px)+log(py), classify_unrel), so the product carries glibc
2.35's own routines to the device.  CPU: the header compiled for the host (tests/libm_check.cpp) on 6e7 arguments per
run (4e8 were run once: scripts/README.md).  GPU: the device's results (cp_math_eval) against the host libm's on the
same argument families, plus bessi and logp_skellam against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, build_if_changed

SRC = os.path.join(ROOT, "tests", "libm_check.cpp")
DEPS = [SRC] + [os.path.join(ROOT, "classpro_amd", "csrc", f) for f in ("cp_libm.h", "cp_libm_tables.h", "cp_types.h")]


def test_host_build_matches_libm():
    exe = os.path.join(ROOT, "tests", "_libm_check")
    build_if_changed(exe, ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, SRC, "-lpthread", "-lm"], DEPS)
    r = subprocess.run([exe, "8", "8"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok 64000000"), r.stdout + r.stderr


def test_tables_are_what_the_generator_writes(tmp_path):
    """cp_libm_tables.h is generated from this image's libm-2.35.a; where the archive is present, regenerate and compare."""
    if not os.path.exists("/lib/x86_64-linux-gnu/libm-2.35.a"):
        pytest.skip("no static libm here")
    cur = open(os.path.join(ROOT, "classpro_amd", "csrc", "cp_libm_tables.h")).read()
    out = str(tmp_path / "tables.h")                    # (never into the source tree: the test must not touch what it checks)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "tools", "gen_libm_tables.py"), out], stdout=subprocess.DEVNULL)
    assert open(out).read() == cur


def _checker():
    so = os.path.join(ROOT, "tests", "_libm_check.so")
    build_if_changed(so, ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DLIBM_CHECK_NO_MAIN",
                          "-o", so, SRC, "-lm"], DEPS)
    return C.CDLL(so)


@pytest.mark.gpu
def test_device_exp_log_are_the_hosts(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from classpro_amd.api import math_eval
    H = _checker()
    n = 4_000_000
    for fn in (0, 1):
        for fam in range(4):
            x = np.zeros(n, np.float64)
            H.cp_libm_args(fn, fam, C.c_uint64(1000 + 17 * fam + fn), x.ctypes.data_as(C.c_void_p), C.c_long(n))
            want = np.zeros(n, np.float64)
            H.cp_libm_eval(fn, x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_long(n))
            got = math_eval(fn, x)
            bad = (got.view(np.uint64) != want.view(np.uint64)) & ~(np.isnan(got) & np.isnan(want))
            assert not bad.any(), (fn, fam, x[bad][:3], got[bad][:3], want[bad][:3])
    x = np.abs(np.random.default_rng(1).standard_normal(1_000_000)) * 1e3
    assert np.array_equal(math_eval(2, x), np.sqrt(x))                 # correctly rounded on both sides


@pytest.mark.gpu
def test_device_bessi_and_skellam_are_the_oracles(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from classpro_amd.api import math_eval
    from oracle.oracle import Oracle
    O = Oracle()
    rng = np.random.default_rng(2)
    n = 20000
    k = rng.integers(0, 200, n).astype(np.float64)
    lam = np.concatenate([rng.random(n // 2) * 60, rng.random(n - n // 2) * 2])
    got_b = math_eval(3, 2 * lam, k)
    got_s = math_eval(4, lam, k)
    for i in range(n):
        assert got_b[i] == O.bessi(int(k[i]), 2 * lam[i]), (k[i], lam[i])
        w = O.logp_skellam(int(k[i]), lam[i])
        assert got_s[i] == w or (np.isnan(got_s[i]) and np.isnan(w)), (k[i], lam[i], got_s[i], w)
