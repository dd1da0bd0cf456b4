import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from conftest import load_golden
from classpro_amd.api import Classifier, Batch
g = load_golden("seeds.npz")
Kx = 40
idx = [i for i in range(int(g["n"])) if int(g["K%d" % i]) == Kx and len(g["prof%d" % i]) >= 1]
clf = Classifier(K=Kx, read_len=20000, hcov=20, dcov=40)
b = Batch.from_reads([g["seq%d" % i].tobytes() for i in idx], [g["prof%d" % i] for i in idx])
lab = np.concatenate([g["lab%d" % i] for i in idx])
b.labels = torch.from_numpy(lab.copy()).to(b.device)
seeds, reps = clf.find_seeds(b)
so = b.seq_off_h
nb = 0
for j, i in enumerate(idx):
    got = seeds[so[j] + Kx - 1:so[j + 1]]; want = g["sasgn%d" % i]
    d = np.nonzero(got != want)[0]
    if len(d):
        nb += 1
        prof = g["prof%d" % i]; l = g["lab%d" % i][Kx - 1:]
        print("read", i, "plen", len(prof), "ndiff", len(d), "first diffs:", [(int(p), chr(got[p]), chr(want[p]), chr(l[p]), int(prof[p])) for p in d[:12]])
        print("   classes of differing marks: got", sorted(set(chr(got[p]) for p in d)), "want", sorted(set(chr(want[p]) for p in d)))
print("bad reads", nb, "of", len(idx))
