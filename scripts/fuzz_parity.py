#!/usr/bin/env python3
"""One-off soak: the adversarial read generator of tests/adversarial.py over many seeds, HIP labels vs the oracle
(and the device raising CP_EOVERFLOW on exactly the reads the oracle rejects).  python scripts/fuzz_parity.py [first=100] [count=24]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from classpro_amd.api import Classifier, Batch
from classpro_amd._lib import ClassProError
from oracle.oracle import Oracle
from adversarial import adversarial_reads, tail_run_reads
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
vary = len(sys.argv) > 3 and sys.argv[3] == "params"        # K and -r change with the seed too (K 21-63, -r 1000-60000)
covs = [(20, 40), (19, 38), (30, 60), (45, 90), (12, 25), (60, 120)]
nreads = nbad = nrej = 0
for seed in range(first, first + count):
    Kx = (21, 32, 40, 50, 63)[seed % 5] if vary else 40
    rl = (1000, 20000, 60000)[(seed // 5) % 3] if vary else 20000
    seqs, profs = adversarial_reads(seed, K=Kx)
    ts, tp = tail_run_reads(seed, n=60, K=Kx)               # reads that end in a low-complexity run (hazard 8)
    seqs, profs = seqs + ts, profs + tp
    hc, dc = covs[seed % len(covs)]
    O = Oracle(Kx, rl, hc, dc)
    clf = Classifier(K=Kx, read_len=rl, hcov=hc, dcov=dc)
    keep_s, keep_p, want = [], [], []
    for s_, p_ in zip(seqs, profs):
        try:
            want.append(O.classify_read(s_, p_))
        except OverflowError:
            nrej += 1
            try:
                clf.classify(Batch.from_reads([s_], [p_]))
                print("seed", seed, ": device did not reject a read the oracle rejects"); nbad += 1
            except ClassProError as e:
                if e.code != -5:
                    print("seed", seed, ": wrong error code", e.code); nbad += 1
            continue
        keep_s.append(s_); keep_p.append(p_)
    got = clf.classify(Batch.from_reads(keep_s, keep_p))
    off = 0
    for r, w in enumerate(want):
        if got[off:off + len(w)].tobytes() != w:
            print("seed", seed, "read", r, "differs"); nbad += 1
        off += len(w)
    nreads += len(want)
    clf.close()
    print("seed", seed, "ok so far:", nreads, "reads,", nrej, "rejected,", nbad, "bad", flush=True)
print("TOTAL", nreads, "reads", nrej, "rejected", nbad, "bad")
