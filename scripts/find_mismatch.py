"""Locate label mismatches between the HIP path and the oracle on the bench workload and show the
first diverging stage/interval (diagnostic; run on the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpro_amd import synth
from classpro_amd.api import Classifier, Batch, hist_covs, STAGE_WALL, STAGE_REL, STAGE_CLASS_REL, STAGE_CLASS_ALL
from oracle.oracle import Oracle

ds = synth.make_dataset(genome_len=5_000_000, cov=40, read_len=20000, K=40, het=0.001, n_repeats=62, min_len=3000, seed=1)
n = 6000
seqs, profs = ds["seqs"][:n], ds["profiles"][:n]
low, high, il, ih, h = ds["hist"]
hc, dc = hist_covs(h, low, high, il, ih, 0)
seq, so, prof, po = synth.pack_batch(seqs, profs)
clf = Classifier(40, 20000, hc, dc)
b = Batch(seq, so, prof, po)
lab = clf.classify(b)
O = Oracle(40, 20000, hc, dc)
want = O.classify_batch(seq, so, prof, po, nthreads=16)
bad = np.nonzero(lab != want)[0]
print("mismatching positions", len(bad))
reads = sorted(set(int(np.searchsorted(so, p, side="right") - 1) for p in bad))
print("reads", reads)
for stage, name in ((STAGE_WALL, "wall"), (STAGE_REL, "rel"), (STAGE_CLASS_REL, "class_rel"), (STAGE_CLASS_ALL, "class_all")):
    clf.run(b, stage)
    got = clf.intervals(b)
    ra = clf.rel_asgn(b) if stage == STAGE_CLASS_REL else None
    for r in reads:
        s, p = seqs[r], profs[r]
        l, rr = O.seq_context(s)
        iv = O.find_wall(p, l, rr)
        iv2, riv = O.find_rel_intvl(iv, p, l, rr)
        ro, io, fw, bw = O.classify_rel(riv, iv2, len(p))
        io2 = O.classify_unrel(io)
        g_iv, g_riv = got[r]
        if stage == STAGE_WALL:
            same = len(g_iv) == len(iv) and all(np.array_equal(g_iv[f], iv[f]) for f in ("b", "e", "cb", "ce"))
            print(name, r, "N", len(iv), len(g_iv), "int fields equal", same)
            if same:
                for f in ("pe", "peo_b", "peo_e"):
                    d = np.nonzero(g_iv[f] != iv[f])[0]
                    if len(d): print("   ", f, "differs at", d[:5], g_iv[f][d[:3]], iv[f][d[:3]])
        elif stage == STAGE_REL:
            same = len(g_riv) == len(riv) and all(np.array_equal(g_riv[f], riv[f]) for f in ("b", "e", "ccb", "cce"))
            print(name, r, "M", len(riv), len(g_riv), "equal", same)
            if not same:
                print("   is_rel diff at", np.nonzero(g_iv["is_rel"] != iv2["is_rel"])[0][:5])
        elif stage == STAGE_CLASS_REL:
            print(name, r, "fw eq", np.array_equal(ra[r][0], fw), "bw eq", np.array_equal(ra[r][1], bw),
                  "final eq", len(g_riv) == len(ro) and np.array_equal(g_riv["asgn"], ro["asgn"]))
            if len(ra[r][0]) == len(fw):
                d = np.nonzero(ra[r][0] != fw)[0]
                if len(d): print("    fw diff idx", d[:8], "gpu", ra[r][0][d[:8]], "oracle", fw[d[:8]])
                d = np.nonzero(ra[r][1] != bw)[0]
                if len(d): print("    bw diff idx", d[:8], "gpu", ra[r][1][d[:8]], "oracle", bw[d[:8]])
        else:
            d = np.nonzero(g_iv["asgn"] != io2["asgn"])[0] if len(g_iv) == len(io2) else [-1]
            print(name, r, "diff intervals", d[:8], [ (int(io2["b"][k]), int(io2["e"][k]), int(io2["cb"][k]), int(io2["ce"][k]), int(io2["is_rel"][k]), int(g_iv["asgn"][k]), int(io2["asgn"][k])) for k in d[:4] if k >= 0])
