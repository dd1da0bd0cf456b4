"""Stage-by-stage diagnostic of the HIP path against the oracle on one synthetic set (run on the GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpro_amd import synth
from classpro_amd.api import Classifier, Batch, STAGE_SCAN, STAGE_WALL, STAGE_REL, STAGE_CLASS_REL, STAGE_CLASS_ALL
from oracle.oracle import Oracle

def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    ds = synth.make_dataset(genome_len=G, cov=40, read_len=10000, seed=5)
    seq, seq_off, prof, prof_off = synth.pack_batch(ds["seqs"], ds["profiles"])
    hc, dc = 20, 40
    O = Oracle(40, 20000, hc, dc)
    clf = Classifier(40, 20000, hc, dc)
    ex = clf.export()
    assert np.array_equal(ex["cthres"], O.cthres()) and np.array_equal(ex["logfact"], O.logfact())
    b = Batch(seq, seq_off, prof, prof_off)
    n = b.nreads
    print("reads", n, "bases", b.total_bases, flush=True)
    # context
    ctx = clf.seq_context(b)
    bad = 0
    for i in range(min(n, 200)):
        l, r = O.seq_context(ds["seqs"][i])
        bad += not (np.array_equal(l, ctx[i][0]) and np.array_equal(r, ctx[i][1]))
    print("ctx mismatching reads", bad, flush=True)
    # scan
    clf.run(b, STAGE_SCAN)
    bm = clf.bitmap(b)
    bits = np.unpackbits(bm.view(np.uint8), bitorder="little")[:b.total_kmers]
    p = prof.astype(np.int64)
    exp = np.zeros(b.total_kmers, np.uint8)
    exp[1:] = ((np.minimum(p[1:], p[:-1]) < ex["cov"][1]) & (np.abs(p[1:] - p[:-1]) >= 3))
    mask = np.ones(b.total_kmers, bool); mask[prof_off[:-1]] = False
    print("scan mismatches", int((bits[mask] != exp[mask]).sum()), "cands", int(exp[mask].sum()), flush=True)
    # oracle stages
    orc = []
    t0 = time.time()
    for s, pr in zip(ds["seqs"], ds["profiles"]):
        l, r = O.seq_context(s)
        iv = O.find_wall(pr, l, r)
        iv2, riv = O.find_rel_intvl(iv, pr, l, r)
        ro, io, fw, bw = O.classify_rel(riv, iv2, len(pr))
        io2 = O.classify_unrel(io)
        orc.append((iv, iv2, riv, ro, io, fw, bw, io2))
    print("oracle stages %.1fs" % (time.time() - t0), flush=True)
    def cmp_iv(a, o, fields):
        if len(a) != len(o): return False
        for f in fields:
            x, y = a[f], o[f]
            if x.dtype.kind == "f":
                if not np.array_equal(x, y): return False
            elif not np.array_equal(x, y): return False
        return True
    for stage, name in ((STAGE_WALL, "wall"), (STAGE_REL, "rel"), (STAGE_CLASS_REL, "class_rel"), (STAGE_CLASS_ALL, "class_all")):
        t0 = time.time()
        clf.run(b, stage); clf.check()
        torch.cuda.synchronize()
        dt = time.time() - t0
        got = clf.intervals(b)
        bad = badf = 0
        first = None
        for i in range(n):
            iv, iv2, riv, ro, io, fw, bw, io2 = orc[i]
            g_iv, g_riv = got[i]
            if stage == STAGE_WALL:
                ok = cmp_iv(g_iv, iv, ("b", "e", "cb", "ce")); okf = ok and cmp_iv(g_iv, iv, ("pe", "peo_b", "peo_e"))
            elif stage == STAGE_REL:
                ok = cmp_iv(g_iv, iv2, ("b", "e", "is_rel")) and cmp_iv(g_riv, riv, ("b", "e", "ccb", "cce")); okf = ok
            elif stage == STAGE_CLASS_REL:
                ok = cmp_iv(g_riv, ro, ("b", "e", "asgn")) and cmp_iv(g_iv, io, ("b", "e", "asgn")); okf = ok
            else:
                ok = cmp_iv(g_iv, io2, ("b", "e", "asgn")); okf = ok
            bad += not ok; badf += not okf
            if not ok and first is None: first = i
        print("stage %-9s %.3fs  reads with int mismatch %d, incl. float fields %d (first %s)" % (name, dt, bad, badf, first), flush=True)
        if stage == STAGE_CLASS_REL:
            ra = clf.rel_asgn(b)
            bf = sum(not np.array_equal(ra[i][0], orc[i][5]) for i in range(n))
            bb = sum(not np.array_equal(ra[i][1], orc[i][6]) for i in range(n))
            print("   fw mismatching reads", bf, "bw", bb, flush=True)
        if first is not None and stage == STAGE_WALL:
            iv = orc[first][0]; g = got[first][0]
            print("   first bad read", first, "N oracle", len(iv), "N gpu", len(g))
            for k in range(min(len(iv), len(g))):
                if (iv[k]["b"], iv[k]["e"]) != (g[k]["b"], g[k]["e"]) or iv[k]["pe"] != g[k]["pe"] or iv[k]["peo_b"] != g[k]["peo_b"] or iv[k]["peo_e"] != g[k]["peo_e"]:
                    print("   k", k, "oracle", iv[k], "gpu", g[k]); break
    t0 = time.time()
    lab = clf.classify(b)
    torch.cuda.synchronize()
    dt = time.time() - t0
    want = O.classify_batch(seq, seq_off, prof, prof_off, nthreads=8)
    nb = int((lab != want).sum())
    print("labels: %d mismatching positions of %d (%.3g), full pipeline %.3fs = %.1f Mbases/s, ws %.1f MB" % (
        nb, len(lab), nb / len(lab), dt, b.total_bases / dt / 1e6, clf.workspace_bytes() / 1e6), flush=True)

if __name__ == "__main__":
    main()
