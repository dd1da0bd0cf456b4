"""Diagnostic (GPU box): one read (npz with seq, prof) through the stage API, alone and optionally inside a batch of
adversarial reads, against the oracle: the first stage / interval that differs.
    python scripts/diag_read.py read.npz hcov dcov [read_len]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from classpro_amd.api import Classifier, Batch, STAGE_WALL, STAGE_REL, STAGE_CLASS_REL, STAGE_CLASS_ALL
from oracle.oracle import Oracle

z = np.load(sys.argv[1])
s, p = z["seq"].tobytes(), z["prof"]
hc, dc = int(sys.argv[2]), int(sys.argv[3])
rl = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
O = Oracle(40, rl, hc, dc)
clf = Classifier(40, rl, hc, dc)
l, rr = O.seq_context(s)
iv = O.find_wall(p, l, rr)
iv2, riv = O.find_rel_intvl(iv, p, l, rr)
ro, io, fw, bw = O.classify_rel(riv, iv2, len(p))
io2 = O.classify_unrel(io)
want = O.classify_read(s, p)
b = Batch.from_reads([s], [p])
lab = clf.classify(b).tobytes()
print("labels equal:", lab == want, "N", len(iv), "M", len(riv))
if lab != want:
    d = np.nonzero(np.frombuffer(lab, np.uint8) != np.frombuffer(want, np.uint8))[0]
    print("  positions", d[:10], "...", len(d), "gpu", lab[d[0]:d[0] + 1], "oracle", want[d[0]:d[0] + 1])
for stage, name in ((STAGE_WALL, "wall"), (STAGE_REL, "rel"), (STAGE_CLASS_REL, "class_rel"), (STAGE_CLASS_ALL, "class_all")):
    clf.run(b, stage)
    g_iv, g_riv = clf.intervals(b)[0]
    if stage == STAGE_WALL:
        same = len(g_iv) == len(iv) and all(np.array_equal(g_iv[f], iv[f]) for f in ("b", "e", "cb", "ce"))
        print(name, "N", len(iv), len(g_iv), "int fields equal", same)
        for f in ("pe", "peo_b", "peo_e"):
            if len(g_iv) == len(iv):
                d = np.nonzero(g_iv[f] != iv[f])[0]
                if len(d): print("   ", f, "differs at", d[:5], g_iv[f][d[:3]], iv[f][d[:3]])
    elif stage == STAGE_REL:
        same = len(g_riv) == len(riv) and all(np.array_equal(g_riv[f], riv[f]) for f in ("b", "e", "ccb", "cce"))
        print(name, "M", len(riv), len(g_riv), "equal", same)
        if not same and len(g_iv) == len(iv2):
            print("   is_rel diff at", np.nonzero(g_iv["is_rel"] != iv2["is_rel"])[0][:5])
    elif stage == STAGE_CLASS_REL:
        ra = clf.rel_asgn(b)[0]
        print(name, "fw eq", np.array_equal(ra[0], fw), "bw eq", np.array_equal(ra[1], bw),
              "final eq", len(g_riv) == len(ro) and np.array_equal(g_riv["asgn"], ro["asgn"]))
        print("   gpu fw", ra[0], "\n   ora fw", fw, "\n   gpu bw", ra[1], "\n   ora bw", bw)
        print("   gpu fin", g_riv["asgn"], "\n   ora fin", ro["asgn"])
        print("   rel intervals (b,e,ccb,cce):", [(int(x["b"]), int(x["e"]), int(x["ccb"]), int(x["cce"])) for x in riv])
    else:
        d = np.nonzero(g_iv["asgn"] != io2["asgn"])[0] if len(g_iv) == len(io2) else [-1]
        print(name, "diff intervals", d[:8], [(int(io2["b"][k]), int(io2["e"][k]), int(io2["cb"][k]), int(io2["ce"][k]), int(io2["is_rel"][k]), int(g_iv["asgn"][k]), int(io2["asgn"][k])) for k in d[:6] if k >= 0])
for tb in ("0",):
    os.environ["CLASSPRO_TABLES"] = tb
    c2 = Classifier(40, rl, hc, dc)
    print("without tables: labels equal oracle:", c2.classify(Batch.from_reads([s], [p])).tobytes() == want)
    c2.close()
