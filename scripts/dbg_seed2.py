import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0,'scripts/proto'); sys.path.insert(0,'scripts')
import numpy as np, torch
from conftest import load_golden
from seed_windows import closed_form_compact
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
out4 = (C.c_int * 4)(); recs = np.zeros((8192, 4), np.int32)
L.cp_debug_seed_get(out4, recs.ctypes.data_as(C.c_void_p))
n, M, b_last, nrep = list(out4)
print("device: n", n, "M", M, "b_last", b_last, "nrep", nrep)
i = target
prof=g["prof%d"%i].astype(int); lab=g["lab%d"%i][K-1:]; sas=g["sasgn%d"%i]; rep=g["rep%d"%i].reshape(-1,2)-(K-1)
plen=len(prof); inrep=np.zeros(plen,bool)
for bb,ee in rep: inrep[bb:ee]=True
state=np.where((sas==ord('H'))|(sas==ord('D')), sas, ord('E'))
valid=[bool(lab[p]!=ord('E') and state[p]==ord('E') and inrep[p]) for p in range(plen)]
starts=[]; run_has=False; prev_run_has=False
for p in range(plen):
    bnd=(p==0) or prof[p]!=prof[p-1]
    if bnd: prev_run_has=run_has; run_has=False
    fv=valid[p] and not run_has
    if (p==0) or fv or (bnd and p>0 and prev_run_has): starts.append(p)
    if valid[p]: run_has=True
endp=plen
if starts and starts[-1]>=plen-1: endp=plen-1; starts=starts[:-1]
cs=[int(prof[p]) if valid[p] else -1 for p in starts]
nw=closed_form_compact(starts,cs,endp,200,True)
ends=starts[1:]+[endp]
V=[k for k in range(len(starts)) if cs[k]>=0]
print("python: valid", len(V), "invalid", len(starts)-len(V), "b_last", starts[-1] if starts else -1, "nrep", len(rep))
bad=0
for v,k in enumerate(V):
    want=(starts[k], ends[k], nw[k], (32767-cs[k])+1)
    got=tuple(int(x) for x in recs[v]) if v<n else None
    if got!=want:
        bad+=1
        if bad<=8: print("  seg",v,"want",want,"got",got)
print("differing records:", bad)
