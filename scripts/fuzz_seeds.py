#!/usr/bin/env python3
"""One-off soak of the -s seed kernel: random label strings / profiles / odd letters (the generator of
tests/test_gpu_seeds.py::test_seeds_random_labels_odd_letters) over many more cases, vs the oracle.
python scripts/fuzz_seeds.py [cases=400] [rng_seed=2024] [K,K,...=40,21,63]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from classpro_amd.api import Classifier, Batch
from oracle.oracle import Oracle
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
KS = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (40, 21, 63)
letters = np.frombuffer(b"ACGTacgtNnUuRYKM", np.uint8)
bad = tot = 0
for Kx in KS:
    O = Oracle(Kx, 20000, 20, 40)
    clf = Classifier(K=Kx, read_len=20000, hcov=20, dcov=40)
    for chunk in range(ncases // len(KS) // 25):
        cases = []
        for rep_i in range(25):
            plen = int(rng.integers(1, 12000))
            mode = rng.integers(0, 4)
            seq = bytes(letters[rng.choice(len(letters), plen + Kx - 1, p=[.22, .22, .22, .22] + [.01] * 12)])
            runs = rng.integers(1, [300, 40, 8, 2000][mode], plen)
            lab = np.repeat(np.frombuffer(b"EHDR", np.uint8)[rng.choice(4, plen, p=[[.25, .25, .25, .25], [.1, .2, .5, .2], [.4, .1, .1, .4], [.02, .08, .8, .1]][mode])], runs)[:plen]
            prof = np.repeat(rng.integers(1, [70, 1500, 40, 300][mode], plen), rng.integers(1, [12, 4, 30, 6][mode], plen))[:plen].astype(np.uint16)
            cases.append((seq, b"N" * (Kx - 1) + lab.tobytes(), prof))
        b = Batch.from_reads([c[0] for c in cases], [c[2] for c in cases])
        b.labels = torch.from_numpy(np.frombuffer(b"".join(c[1] for c in cases), np.uint8).copy()).to(b.device)
        seeds, reps = clf.find_seeds(b)
        so = b.seq_off_h
        for j, (seq, labs, prof) in enumerate(cases):
            want, wrep = O.find_seeds(seq, labs, prof)
            ok = np.array_equal(seeds[so[j] + Kx - 1:so[j + 1]], want) and np.array_equal(reps[j].reshape(-1, 2), wrep.reshape(-1, 2))
            tot += 1
            if not ok:
                bad += 1
                print("K", Kx, "chunk", chunk, "case", j, "plen", len(prof), "DIFFERS", flush=True)
        if chunk % 4 == 3: print("K", Kx, "ok so far:", tot, "cases,", bad, "bad", flush=True)
    clf.close()
    print("K", Kx, "done:", tot, "cases,", bad, "bad", flush=True)
print("TOTAL", tot, "cases", bad, "bad")
