import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from classpro_amd.api import Classifier, Batch
from classpro_amd._lib import lib
g = load_golden("seeds.npz")
K=40
idx = [i for i in range(int(g["n"])) if int(g["K%d" % i]) == K and len(g["prof%d" % i]) >= 1]
target = int(sys.argv[1]) if len(sys.argv) > 1 else 13
L = lib()
L.cp_debug_seed_set(idx.index(target))
clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
b = Batch.from_reads([g["seq%d" % i].tobytes() for i in idx], [g["prof%d" % i] for i in idx])
b.labels = torch.from_numpy(np.concatenate([g["lab%d" % i] for i in idx]).copy()).to(b.device)
clf.find_seeds(b)
out4 = (C.c_int * 4)(); recs = np.zeros(8192*4, np.int32)
L.cp_debug_seed_get(out4, recs.ctypes.data_as(C.c_void_p))
nt, M, n, _ = list(out4)
print("ntake", nt, "M", M, "n", n)
print("takes:", recs[:2*nt].reshape(-1,2).tolist())
