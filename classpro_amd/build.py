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


def _newest_src():
    t = 0.0
    for root, _d, files in os.walk(CSRC):
        for f in files:
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    t = max(t, os.path.getmtime(os.path.join(_HERE, "..", "include", "classpro_amd.h")))
    return t


def build(force=False, verbose=False):
    outs = [OUT, CLI, SYNTH] + [os.path.join(_HERE, t) for t in TOOLS]
    if (not force and all(os.path.exists(o) for o in outs)
            and min(os.path.getmtime(o) for o in outs) >= _newest_src()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, "capi.hip"), "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(CSRC, "host", "classpro_main.cpp"), "-o", CLI,
           "-L" + _HERE, "-lclasspro_amd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
           os.path.join(CSRC, "synth", "synth_gen.hip"), "-o", SYNTH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    for tool, src in TOOLS.items():
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-ffp-contract=off",
               os.path.join(CSRC, "host", src), "-o", os.path.join(_HERE, tool), "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
