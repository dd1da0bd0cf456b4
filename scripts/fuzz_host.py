#!/usr/bin/env python3
"""CPU soak over the parameter space: the product's scalar code (tests/host_harness.cpp over classpro_amd/csrc/cp_*.h) against the
oracle on adversarial and tail-run reads made for many K, -r and coverage settings -- and, in the build container (where
oracle/_ref holds the GSL-free part of the reference's wall.c), BOTH against the reference's own per-read functions
(Ref.classify_read: context.c -> wall.c:245-1051 -> class_rel.c -> class_unrel.c -> paint); the reads the oracle rejects
must be the reads on which the reference exit(1)s.  With CP_SANITIZE=1 and libasan preloaded (see scripts/sanitize.sh) the
oracle and the harness also run under ASan + UBSan.      python scripts/fuzz_host.py [seeds=3] [first_seed=0]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
from oracle.oracle import Oracle, Ref, ref_wall_available
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
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = bad = rej = refbad = 0
REF = ref_wall_available() and os.environ.get("CP_SANITIZE") != "1"     # (the reference library is not a sanitizer build)
for Kx in (15, 21, 32, 40, 50, 63):
    for rl in (1000, 20000, 60000):
        for hc, dc in ((5, 10), (15, 30), (50, 100)):
            O = Oracle(Kx, rl, hc, dc)
            P = H.hh_params_new(Kx, rl, hc, dc)
            R = Ref(rl, hc, dc).wall_setup_from(O) if REF else None
            for seed in range(first, first + nseeds):
                a_s, a_p = adversarial_reads(7000 + seed + Kx, n=40, K=Kx)
                t_s, t_p = tail_run_reads(8000 + seed + Kx, n=24, K=Kx)
                for s, p in zip(a_s + t_s, a_p + t_p):
                    try:
                        want = O.classify_read(s, p)
                    except OverflowError:
                        rej += 1
                        if R is not None and R.find_wall_exit_status(s, p, Kx) != 1:
                            refbad += 1
                            print("REF does not exit where the oracle rejects: K", Kx, "r", rl, "cov", hc, dc, "seed", seed, "rlen", len(s))
                        continue
                    N, lab, *_ = run_harness_read(H, P, s, p)
                    tot += 1
                    if lab != want:
                        bad += 1
                        print("DIFF K", Kx, "r", rl, "cov", hc, dc, "seed", seed, "rlen", len(s))
                    if R is not None and len(p) >= 12 and R.classify_read(s, p, Kx) != want:     # (tiny reads may abort: checked in a child above)
                        refbad += 1
                        print("REF DIFF K", Kx, "r", rl, "cov", hc, dc, "seed", seed, "rlen", len(s))
            H.hh_params_free(C.c_void_p(P))
    print("K", Kx, "done:", tot, "reads,", rej, "rejected,", bad, "bad,", refbad, "differing from the reference" if REF else "(no reference leg)", flush=True)
print("TOTAL", tot, "reads", rej, "rejected", bad, "bad", refbad, "differing from the reference" if REF else "(no reference leg)")
sys.exit(1 if bad or refbad else 0)
