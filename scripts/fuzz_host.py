#!/usr/bin/env python3
"""CPU soak over the parameter space: the product's scalar code (tests/host_harness.cpp over classpro_amd/csrc/cp_*.h) against the
oracle on adversarial and tail-run reads made for many K, -r and coverage settings.  With CP_SANITIZE=1 and libasan
preloaded (see scripts/sanitize.sh) both sides also run under ASan + UBSan.      python scripts/fuzz_host.py [seeds=3]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
from oracle.oracle import Oracle
from adversarial import adversarial_reads, tail_run_reads
from test_host_logic import run_harness_read
src = os.path.join(ROOT, "tests", "host_harness.cpp")
out = os.path.join(ROOT, "tests", "_host_harness.so")
csrc = os.path.join(ROOT, "classpro_amd", "csrc")
deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
conftest.build_if_changed(out, ["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src], deps)
H = C.CDLL(out)
H.hh_params_new.restype = C.c_void_p
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tot = bad = rej = 0
for Kx in (15, 21, 32, 40, 50, 63):
    for rl in (1000, 20000, 60000):
        for hc, dc in ((5, 10), (15, 30), (50, 100)):
            O = Oracle(Kx, rl, hc, dc)
            P = H.hh_params_new(Kx, rl, hc, dc)
            for seed in range(nseeds):
                a_s, a_p = adversarial_reads(7000 + seed + Kx, n=40, K=Kx)
                t_s, t_p = tail_run_reads(8000 + seed + Kx, n=24, K=Kx)
                for s, p in zip(a_s + t_s, a_p + t_p):
                    try:
                        want = O.classify_read(s, p)
                    except OverflowError:
                        rej += 1
                        continue
                    N, lab, *_ = run_harness_read(H, P, s, p)
                    tot += 1
                    if lab != want:
                        bad += 1
                        print("DIFF K", Kx, "r", rl, "cov", hc, dc, "seed", seed, "rlen", len(s))
            H.hh_params_free(C.c_void_p(P))
    print("K", Kx, "done:", tot, "reads,", rej, "rejected,", bad, "bad", flush=True)
print("TOTAL", tot, "reads", rej, "rejected", bad, "bad")
sys.exit(1 if bad else 0)
