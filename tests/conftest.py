import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# CP_SANITIZE=1 (scripts/sanitize.sh): every host helper the tests build gets AddressSanitizer + UBSan.
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if os.environ.get("CP_SANITIZE") == "1" else []


def build_if_changed(out, cmd, deps):
    """Runs `cmd` (which writes `out`) unless `out` exists and was built from exactly these sources and this command:
    the key is a hash of the CONTENT of `deps` kept beside the output (`<out>.srchash`), not file times -- after a
    fresh checkout, or on a snapshot copied to another box, every mtime is the copy time and says nothing."""
    if SAN and cmd[0] in ("gcc", "g++"):
        cmd = [cmd[0]] + SAN + cmd[1:]
    h = hashlib.sha1(" ".join(cmd).encode())
    for d in sorted(deps):
        h.update(d.encode())
        h.update(open(d, "rb").read())
    key = h.hexdigest()
    side = out + ".srchash"
    if os.path.exists(out) and os.path.exists(side) and open(side).read().strip() == key:
        return out
    subprocess.check_call(cmd)
    with open(side, "w") as f:
        f.write(key + "\n")
    return out


@pytest.fixture(scope="session")
def built():
    """Product library (cross-compiled for gfx950) + oracle."""
    from classpro_amd import build
    build.build()
    from oracle import oracle
    oracle.build()
    return True


@pytest.fixture(scope="session")
def harness(built):
    """tests/host_harness.cpp: the product's scalar device functions compiled for the host (test-only)."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    out = os.path.join(ROOT, "tests", "_host_harness.so")
    csrc = os.path.join(ROOT, "classpro_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    build_if_changed(out, ["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src], deps)
    H = C.CDLL(out)
    H.hh_params_new.restype = C.c_void_p
    H.hh_params_cthres.restype = C.c_void_p
    H.hh_params_logfact.restype = C.c_void_p
    H.hh_bessi.restype = C.c_double
    H.hh_bessi.argtypes = [C.c_int, C.c_double]
    return H


@pytest.fixture(scope="session")
def small_ds():
    from classpro_amd import synth
    return synth.make_dataset(genome_len=120000, cov=40, read_len=8000, seed=5)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def pytest_sessionfinish(session, exitstatus):
    """scripts/bounds_check.sh: with a -DCP_BOUNDS build of the library loaded (cp_bounds.h), report how many accesses of the
    per-read kernels fell outside their read during the whole session -- it must be none -- and fail the run otherwise."""
    rep = os.environ.get("CP_BOUNDS_REPORT")
    if not rep:
        return
    import ctypes as C
    from classpro_amd import _lib
    L = _lib.lib()
    out = (C.c_ulonglong * 4)()
    rc = L.cp_debug_bounds(out)
    with open(rep, "w") as f:
        f.write("cp_debug_bounds rc=%d: %d accesses outside a read (first: kind %d, index %d, length %d)\n"
                % (rc, out[0], out[1], C.c_longlong(out[2]).value, out[3]))
    if rc != 0 or out[0] != 0:
        session.exitstatus = 1
