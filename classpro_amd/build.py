"""Build libclasspro_amd.so in-tree: hipcc cross-compiles gfx950 without a GPU.

    python -m classpro_amd.build [--force]
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libclasspro_amd.so")
SYNTH = os.path.join(_HERE, "libcp_synth.so")  # device-side synthetic read sets: test / bench infrastructure (csrc/synth/)
CLI = os.path.join(_HERE, "ClassPro")          # drop-in command line (csrc/host/classpro_main.cpp)
TOOLS = {"prof2class": "prof2class.cpp", "class2acc": "class2acc.cpp"}   # host-only evaluation tools
# -ffp-contract=off: the decision path compares doubles against thresholds and truncates them to
# ints (class_rel.c:449,483); fused multiply-adds would change those values.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


# CP_SANITIZE=1 (scripts/sanitize.sh): the host-only tools are built with AddressSanitizer + UBSan.  The HIP objects never
# are (GPU sanitizers are not available on this pool); the scalar device functions are covered through
# tests/host_harness.cpp, which compiles the same headers for the host.
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if os.environ.get("CP_SANITIZE") == "1" else []


def _src_key(extra):
    """Hash of the CONTENT of every source the outputs are built from (+ the flags): after a fresh checkout, or on a
    snapshot copied to another box, every mtime is the copy time and says nothing (same rule as tests/conftest.py)."""
    import hashlib
    h = hashlib.sha1(" ".join(extra).encode())
    files = [os.path.join(_HERE, "..", "include", "classpro_amd.h")]
    for root, _d, names in os.walk(CSRC):
        files += [os.path.join(root, f) for f in names]
    for f in sorted(files):
        h.update(os.path.relpath(f, _HERE).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _fresh(outs, side, key, force):
    return (not force and all(os.path.exists(o) for o in outs)
            and os.path.exists(side) and open(side).read().strip() == key)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)


def build(force=False, verbose=False):
    # the HIP objects: library, command line, synthetic-set generator
    outs, side, key = [OUT, CLI, SYNTH], os.path.join(_HERE, ".build.srchash"), _src_key(FLAGS)
    if not _fresh(outs, side, key, force):
        if os.path.exists(side):
            os.remove(side)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc] + FLAGS + [os.path.join(CSRC, "capi.hip"), "-o", OUT], verbose)
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(CSRC, "host", "classpro_main.cpp"), "-o", CLI,
              "-L" + _HERE, "-lclasspro_amd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"], verbose)
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
              os.path.join(CSRC, "synth", "synth_gen.hip"), "-o", SYNTH], verbose)
        with open(side, "w") as f:
            f.write(key + "\n")
    # the host-only evaluation tools
    outs, side, key = [os.path.join(_HERE, t) for t in TOOLS], os.path.join(_HERE, ".tools.srchash"), _src_key(["tools"] + SAN)
    if not _fresh(outs, side, key, force):
        if os.path.exists(side):
            os.remove(side)
        for tool, src in TOOLS.items():
            _run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-ffp-contract=off"] + SAN
                 + [os.path.join(CSRC, "host", src), "-o", os.path.join(_HERE, tool), "-lz"], verbose)
        with open(side, "w") as f:
            f.write(key + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
