#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref, built by oracle/Makefile
from /root/reference/src).  Run in the build container (the reference tree does not travel):

    python oracle/gen_golden.py

Vectors (inputs + the reference's outputs; data only, no reference source):
  prims.npz     bessi / logp_poisson / logp_skellam / logp_binom / binom_test_g / logp_trans known answers,
                logfact[0..4095], GLOBAL_COV / DR_RATIO for several (H,D)
  context.npz   calc_seq_context lctx/rctx for crafted + random low-complexity strings
  classify.npz  classify_rel (fw, bw, reconciled) and classify_unrel outputs for interval sets
                (the interval sets are produced by the oracle's find_wall/find_rel_intvl on seeded synthetic reads)
  fastk.npz     process_global_hist (H,D) and Fetch_Profile output for FASTK files written by classpro_amd.fastk
  seeds.npz     find_seeds (seed.c, the -s path): seed labels, repeat-mask intervals and canonical ntHash values
  wall.npz      find_wall + find_rel_intvl (wall.c:570-958, 960-1051; the GSL-free part of wall.c compiled as it
                stands, oracle/Makefile): per read the reference's Intvl[N] and rintvl[M] records, doubles included;
                reads on which the reference's own exit(1) fires ("# E-intvls >= plen") carry status 1 and no records
  labels.npz    whole reads through reference text ONLY (context.c -> wall.c slice -> class_rel.c -> class_unrel.c ->
                paint, ClassPro.c:229-271): label strings and the final interval classes
Both also hold the threshold tables handed to find_wall (calc_init_thres itself needs GSL: DESIGN 3.2).
"""
import os
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != os.path.join(ROOT, "oracle")]
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Ref, INTVL_DTYPE, ref_available, ref_wall_available  # noqa: E402
from classpro_amd import synth, fastk  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def prims():
    rng = np.random.default_rng(101)
    R = Ref(20000, 20, 40)
    n = rng.integers(0, 90, 400); x = np.concatenate([rng.uniform(0, 5, 200), rng.uniform(0, 200, 200)])
    bess = np.array([R.bessi(int(a), float(b)) for a, b in zip(n, x)])
    k = rng.integers(0, 400, 400); lam = rng.integers(1, 120, 400)
    pois = np.array([R.logp_poisson(int(a), int(b)) for a, b in zip(k, lam)])
    sk_k = rng.integers(-60, 60, 400); sk_l = rng.uniform(0.001, 80, 400)
    skel = np.array([R.logp_skellam(int(a), float(b)) for a, b in zip(sk_k, sk_l)])
    bn = rng.integers(1, 300, 400); bk = (rng.random(400) * (bn + 1)).astype(np.int64)
    bp = rng.choice([0.004, 0.01, 0.034, 0.1, 0.802, 0.99], 400)
    binom = np.array([R.logp_binom(int(a), int(b), float(c)) for a, b, c in zip(bk, bn, bp)])
    btest = np.array([R.binom_test_g(int(a), int(b), float(c)) for a, b, c in zip(bk, bn, bp)])
    tb = rng.integers(0, 20000, 400); te = rng.integers(0, 20000, 400)
    tcb = rng.integers(1, 80, 400); tce = rng.integers(1, 80, 400); tcov = rng.integers(1, 80, 400)
    trans = np.array([R.logp_trans(int(a), int(b), int(c), int(d), int(e)) for a, b, c, d, e in zip(tb, te, tcb, tce, tcov)])
    lf = R.logfact()[:4096]
    covs = []
    for h, d in ((20, 40), (19, 38), (30, 60), (12, 25), (50, 99)):
        g, dr = Ref(20000, h, d).globals()
        covs.append(g + [dr])
    np.savez_compressed(os.path.join(OUT, "prims.npz"), bess_n=n, bess_x=x, bess=bess, pois_k=k, pois_l=lam, pois=pois,
                        sk_k=sk_k, sk_l=sk_l, skel=skel, bn=bn, bk=bk, bp=bp, binom=binom, btest=btest,
                        tb=tb, te=te, tcb=tcb, tce=tce, tcov=tcov, trans=trans, logfact=lf,
                        covs=np.array(covs, dtype=np.float64))


def context():
    rng = np.random.default_rng(102)
    R = Ref()
    AL = np.frombuffer(b"ACGT", np.uint8)
    seqs = [b"A", b"AC", b"AAAAAAAAAA", b"ACACACACACAC", b"ACGACGACGACGACG", b"AACACACGGGTATATATTTT",
            b"ACACAGAGAG", b"CAACAC", b"A" * 140, b"AC" * 140 + b"G", b"ACG" * 135 + b"T", b"TTTACGACGACGAAA"]
    for it in range(120):
        L = int(rng.integers(1, 160))
        s = bytes(AL[rng.integers(0, int(rng.integers(1, 5)), size=L)])
        if it % 2:
            u = bytes(AL[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
            s = s[:L // 2] + u * int(rng.integers(2, 30)) + s[L // 2:]
        seqs.append(s)
    off = np.zeros(len(seqs) + 1, np.int64)
    ls, rs = [], []
    for i, s in enumerate(seqs):
        l, r = R.seq_context(s)
        ls.append(l); rs.append(r)
        off[i + 1] = off[i] + len(s)
    np.savez_compressed(os.path.join(OUT, "context.npz"), seq=np.frombuffer(b"".join(seqs), np.uint8), off=off,
                        lctx=np.concatenate(ls), rctx=np.concatenate(rs))


def classify():
    cases = []
    for seed, h, d, kw in ((21, 20, 40, dict(genome_len=120000, cov=40, read_len=9000)),
                           (22, 15, 30, dict(genome_len=80000, cov=30, read_len=7000, het=0.004)),
                           (23, 30, 60, dict(genome_len=60000, cov=60, read_len=12000, err_sub=0.002))):
        ds = synth.make_dataset(seed=seed, **kw)
        O = Oracle(40, 20000, h, d)
        R = Ref(20000, h, d)
        for s, p in list(zip(ds["seqs"], ds["profiles"]))[:14]:
            l, r = O.seq_context(s)
            iv = O.find_wall(p, l, r)
            iv2, riv = O.find_rel_intvl(iv, p, l, r)
            r1, i1 = R.classify(riv, iv2, len(p), stage=1)
            r2, i2 = R.classify(riv, iv2, len(p), stage=2)
            fw = R.classify_rel_dir(riv, len(p), True)[0] if len(riv) else np.zeros(0, np.int8)
            bw = R.classify_rel_dir(riv, len(p), False)[0] if len(riv) else np.zeros(0, np.int8)
            cases.append((h, d, len(p), iv2, riv, r1["asgn"].copy(), i1["asgn"].copy(), i2["asgn"].copy(), fw, bw))
    arrs = {"n": np.array(len(cases))}
    for k, c in enumerate(cases):
        arrs["meta%d" % k] = np.array(c[:3])
        arrs["intvl%d" % k] = c[3].view(np.uint8)
        arrs["rintvl%d" % k] = c[4].view(np.uint8)
        arrs["rel_rasgn%d" % k] = c[5]
        arrs["rel_iasgn%d" % k] = c[6]
        arrs["all_iasgn%d" % k] = c[7]
        arrs["fw%d" % k] = c[8]
        arrs["bw%d" % k] = c[9]
    np.savez_compressed(os.path.join(OUT, "classify.npz"), **arrs)


def fastk_files():
    R = Ref()
    ds = synth.make_dataset(genome_len=60000, cov=40, read_len=5000, seed=31)
    rng = np.random.default_rng(32)
    profs = [p for p in ds["profiles"][:40]]
    # exercise every code form: long runs, big jumps (15-bit deltas), counts >= 128, near 32767
    weird = np.concatenate([np.full(200, 5), [300, 301, 32767, 32700, 1, 1, 1, 129, 128, 127], rng.integers(1, 32767, 50),
                            np.full(70, 4000)]).astype(np.uint16)
    profs.append(weird)
    profs.append(np.array([7], np.uint16))
    with tempfile.TemporaryDirectory() as td:
        fastk.write_fastk(td, "syn", 40, profs, ds["hist"], nparts=2)
        h, d = R.hist_covs(os.path.join(td, "syn"), 0)
        h2, d2 = R.hist_covs(os.path.join(td, "syn"), 37)
        k, got = R.fetch_profiles(os.path.join(td, "syn"))
        _, codes = fastk.read_fastk_codes(td, "syn")
    assert k == 40 and len(got) == len(profs)
    for a, b in zip(got, profs):
        assert np.array_equal(a, b), "FASTK writer/reference reader round trip failed"
    low, high, il, ih, hist = ds["hist"]
    off = np.zeros(len(profs) + 1, np.int64)
    coff = np.zeros(len(profs) + 1, np.int64)
    for i, (p, c) in enumerate(zip(got, codes)):
        off[i + 1] = off[i] + len(p)
        coff[i + 1] = coff[i] + len(c)
    np.savez_compressed(os.path.join(OUT, "fastk.npz"), hist=hist, low=low, high=high, ilow=il, ihigh=ih,
                        covs=np.array([h, d, h2, d2]), cov_opt=np.array(37),
                        prof=np.concatenate(got), prof_off=off,
                        codes=np.frombuffer(b"".join(codes), np.uint8), code_off=coff)


def eval_tools():
    """Outputs of the reference's own prof2class / class2acc (built by oracle/Makefile into oracle/_ref/)
    on the scenario of tests/eval_case.py."""
    import json, subprocess, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import eval_case
    ref = os.path.join(ROOT, "oracle", "_ref")
    out = dict(cases=[])
    for tiny in (True, False):
        with tempfile.TemporaryDirectory() as d:
            eval_case.build_case(d, eval_case.oracle_labels, tiny)
            subprocess.check_call([os.path.join(ref, "prof2class"), os.path.join(d, "truth"), os.path.join(d, "reads.fasta")])
            out["est_sha256_%d" % tiny] = eval_case.sha(os.path.join(d, "est.class"))
            out["truth_sha256_%d" % tiny] = eval_case.sha(os.path.join(d, "truth.class"))
            for t, args in eval_case.ACC_CASES:
                if t != tiny:
                    continue
                a = [x.format(dir=d) for x in args]
                r = subprocess.run([os.path.join(ref, "class2acc")] + a + [os.path.join(d, "est.class"), os.path.join(d, "truth.class")],
                                   capture_output=True, text=True)
                out["cases"].append(dict(tiny=tiny, args=args, returncode=r.returncode, stdout=r.stdout,
                                         stderr=r.stderr.replace(d, "{dir}")))
            if tiny:                                                   # error contract
                r = subprocess.run([os.path.join(ref, "class2acc"), os.path.join(d, "est.class")], capture_output=True, text=True)
                out["usage_stderr"] = r.stderr
                out["usage_returncode"] = r.returncode
    json.dump(out, open(os.path.join(OUT, "eval_tools.json"), "w"), indent=1)


def dazz_db():
    """sha256 of the reference prof2class output on the .db / .dam scenarios of tests/test_dazz_db.py."""
    import json, subprocess, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import eval_case, test_dazz_db as t
    out = {}
    for dam in (False, True):
        with tempfile.TemporaryDirectory() as d:
            t.make_case(d, dam)
            src = os.path.join(d, "reads.dam" if dam else "reads.db")
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "prof2class"), os.path.join(d, "truth"), src])
            out["prof2class_sha256_dam" if dam else "prof2class_sha256_db"] = eval_case.sha(os.path.join(d, "truth.class"))
    json.dump(out, open(os.path.join(OUT, "dazz_db.json"), "w"), indent=1)


def seed_cases():
    """Inputs for the -s path: (seq, label string, profile, K).  Labels of synthetic reads come from the oracle's own
    classification; the crafted ones put every class in odd places (all R, all E, R islands shorter / longer than
    2.5 K, single-position runs, a last position of its own count, N bases)."""
    rng = np.random.default_rng(77)
    AL = np.frombuffer(b"ACGT", np.uint8)
    cases = []
    for seed, (g, cov, rl, het) in enumerate([(120000, 40, 6000, 0.002), (90000, 60, 9000, 0.004)]):
        ds = synth.make_dataset(genome_len=g, cov=cov, read_len=rl, seed=seed + 3, het=het, n_repeats=8)
        Oc = Oracle(40, 20000, cov // 2, cov)
        for s, p in list(zip(ds["seqs"], ds["profiles"]))[:14]:
            cases.append((s, Oc.classify_read(s, p), p, 40))
    for K, plen in ((40, 1), (40, 2), (40, 150), (40, 3000), (40, 5200), (25, 2600), (63, 2400)):
        for mode in range(4):
            seq = bytearray(AL[rng.integers(0, 4, plen + K - 1)].tobytes())
            if mode == 3 and plen > 100:
                seq[50] = ord("N"); seq[plen // 2] = ord("n")
                for q, c in zip(rng.integers(0, plen, 12), b"UuRYKMSWacgt"):     # every row of nthash.h's seedTab
                    seq[int(q)] = c
            runs = rng.integers(1, [400, 60, 12, 900][mode], plen)
            lab = np.repeat(np.frombuffer(b"EHDR", np.uint8)[rng.choice(4, plen, p=[[.1, .2, .5, .2], [.25, .25, .25, .25], [.4, .1, .1, .4], [0, .1, .8, .1]][mode])], runs)[:plen]
            cruns = rng.integers(1, [9, 3, 30, 5][mode], plen)
            prof = np.repeat(rng.integers(1, [60, 300, 40, 2000][mode], plen), cruns)[:plen].astype(np.uint16)
            cases.append((bytes(seq), b"N" * (K - 1) + lab.tobytes(), prof, K))
    plen = 4000
    seq = bytes(AL[rng.integers(0, 4, plen + 39)])
    for lab in (b"R" * plen, b"E" * plen, b"D" * plen, b"H" * 99 + b"R" * 3801 + b"D" * 100, b"D" * 100 + b"R" * 3800 + b"H" * 100):
        cases.append((seq, b"N" * 39 + lab, rng.integers(20, 60, plen).astype(np.uint16), 40))
    return cases


def seeds():
    R = Ref()
    cases = seed_cases()
    out = {"n": np.array(len(cases))}
    for i, (s, lab, p, K) in enumerate(cases):
        sas, rep, hsh = R.find_seeds(s, lab, p, K)
        out["seq%d" % i] = np.frombuffer(s, np.uint8)
        out["lab%d" % i] = np.frombuffer(lab, np.uint8)
        out["prof%d" % i] = p
        out["K%d" % i] = np.array(K)
        out["sasgn%d" % i] = sas
        out["rep%d" % i] = rep
        out["hash%d" % i] = hsh
    np.savez_compressed(os.path.join(OUT, "seeds.npz"), **out)


MODEL_GROWTH = (0.0004, 0.0005, 0.0006)      # tests/test_error_model.py: flatter than the default model, changes wall decisions


def model_path(d):
    """The synthetic HIsim model of the -M parameter set (classpro_amd.synth.write_himodel is deterministic), written to d."""
    p = os.path.join(d, "golden_hifi.model")
    synth.write_himodel(p, growth=MODEL_GROWTH)
    return p


def _param_sets():
    """(K, -r, H, D, model) of the wall / label vectors; model = 1: the error model of -M<model_path(...)> instead of the
    default one (the reference's find_wall takes the model as a parameter; its tables come from the oracle's load + fit)."""
    return [(40, 20000, 20, 40, 0), (40, 20000, 30, 60, 0), (40, 2000, 20, 40, 0), (21, 20000, 20, 40, 0), (25, 25000, 30, 60, 0),
            (63, 20000, 12, 25, 0), (40, 20000, 20, 40, 1)]


def _edge_cases(K, rng):
    """tests/test_gpu_parity.py::test_edge_reads' inputs for any K: shortest legal reads, flat / repeat / error / ramp
    profiles, homopolymer and micro-satellite reads -- several of them end in the reference's exit(1)."""
    AL = np.frombuffer(b"ACGT", np.uint8)
    cases = []
    for plen in (1, 2, 3, K - 1, K, K + 1, 200, 1000):
        rlen = plen + K - 1
        seq = bytes(AL[rng.integers(0, 4, rlen)])
        cases += [(seq, np.full(plen, 40, np.uint16)), (seq, np.full(plen, 500, np.uint16)), (seq, np.full(plen, 1, np.uint16)),
                  (seq, (1 + np.arange(plen) % 90).astype(np.uint16)),
                  (b"A" * rlen, rng.integers(1, 80, plen).astype(np.uint16)),
                  ((b"AC" * rlen)[:rlen], rng.integers(1, 80, plen).astype(np.uint16)),
                  ((b"ACG" * rlen)[:rlen], rng.integers(1, 300, plen).astype(np.uint16))]
    p = np.full(3000, 40, np.uint16)
    p[1000:1000 + K - 1] = 1
    p[2000:2500] = 20
    cases.append((bytes(AL[rng.integers(0, 4, 3000 + K - 1)]), p))
    p = np.full(5000, 32767, np.uint16)
    p[100:150] = 3
    cases.append((bytes(AL[rng.integers(0, 4, 5000 + K - 1)]), p))
    return cases


def wall_label_cases(which):
    """[(set index, seq, profile)] for wall.npz ('wall') / labels.npz ('labels'): disjoint seeds, the same kinds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from adversarial import adversarial_reads, tail_run_reads
    w = which == "wall"
    out = []
    for si, (K, rl, h, d, _m) in enumerate(_param_sets()):
        rng = np.random.default_rng(500 + si + (0 if w else 50))
        ds = synth.make_dataset(genome_len=60000, cov=d, read_len=min(rl, 9000) if w else min(rl, 14000), K=K,
                                seed=(60 if w else 160) + si, het=0.004 if si % 2 else 0.001, n_repeats=6)
        n_syn = (10 if w else 16) if si < 2 else (5 if w else 8)
        out += [(si, s_, p_) for s_, p_ in list(zip(ds["seqs"], ds["profiles"]))[:n_syn]]
        S, P = adversarial_reads((1 if w else 11) + si, n=(36 if w else 30), K=K)
        out += [(si, s_, p_) for s_, p_ in zip(S, P)]
        S, P = tail_run_reads((21 if w else 31) + si, n=(24 if w else 24), K=K)
        out += [(si, s_, p_) for s_, p_ in zip(S, P)]
        if si in (0, 3, 5, 6):
            out += [(si, s_, p_) for s_, p_ in _edge_cases(K, rng)]
        AL = np.frombuffer(b"ACGT", np.uint8)                 # tiny reads with jumping counts: about a fifth of them
        for it in range(30 if si in (0, 3, 5, 6) else 9):         # end in the reference's exit(1) ("# E-intvls >= plen")
            plen = int(rng.integers(1, 12)); rlen = plen + K - 1
            seq = [bytes(AL[rng.integers(0, 4, rlen)]), b"A" * rlen, (b"AC" * rlen)[:rlen]][it % 3]
            out.append((si, seq, rng.choice([1, 2, 5, 20, 40, 80, 300], plen).astype(np.uint16)))
    return out


def _tables(psets):
    """Per parameter set the threshold tables find_wall gets (from the oracle; checked entry by entry with exact integer
    arithmetic in tests/test_first_principles.py) -- stored so the tests can see that the product uses the same ones."""
    td = tempfile.mkdtemp()
    O = [Oracle(K, rl, h, d, model=(model_path(td) if m else None)) for K, rl, h, d, m in psets]
    return O, dict(psets=np.array(psets, np.int32),
                   cthres=np.stack([o.cthres() for o in O]), pe=np.stack([o.pe() for o in O]),
                   lmax=np.stack([o.lmax() for o in O]),
                   cmax=np.array([o.scalars()[2] for o in O], np.int32), hc_erate=np.array([o.scalars()[3] for o in O]))


def _refs(psets, O):
    return [Ref(rl, h, d).wall_setup_from(o) for (K, rl, h, d, m), o in zip(psets, O)]


def wall():
    psets = _param_sets()
    O, arrs = _tables(psets)
    R = _refs(psets, O)
    cases = wall_label_cases("wall")
    seqs, profs, sets, status, ivs, rvs = [], [], [], [], [], []
    for si, s, p in cases:
        K = psets[si][0]
        st = R[si].find_wall_exit_status(s, p, K)
        assert st in (0, 1), st
        if st == 0:
            iv, rv = R[si].find_wall_rel(s, p, K)
        else:
            iv = rv = np.zeros(0, INTVL_DTYPE)
        seqs.append(np.frombuffer(s, np.uint8)); profs.append(p); sets.append(si); status.append(st)
        ivs.append(iv); rvs.append(rv)
    def cat(L, dt):                          # (np.concatenate would repack the padded record dtype)
        out = np.zeros(sum(len(x) for x in L), dt)
        o = 0
        for x in L:
            out[o:o + len(x)] = x
            o += len(x)
        return out
    off = lambda L: np.concatenate([[0], np.cumsum([len(x) for x in L])]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "wall.npz"), set=np.array(sets, np.int32), status=np.array(status, np.int8),
                        seq=cat(seqs, np.uint8), seq_off=off(seqs), prof=cat(profs, np.uint16), prof_off=off(profs),
                        intvl=cat(ivs, INTVL_DTYPE).view(np.uint8), intvl_off=off(ivs),
                        rintvl=cat(rvs, INTVL_DTYPE).view(np.uint8), rintvl_off=off(rvs), **arrs)
    print("wall.npz: %d reads, %d exit(1), %d intervals, %d reliable" % (len(cases), sum(status), sum(map(len, ivs)), sum(map(len, rvs))))


def labels():
    psets = _param_sets()
    O, arrs = _tables(psets)
    R = _refs(psets, O)
    cases = wall_label_cases("labels")
    seqs, profs, sets, status, labs, asg = [], [], [], [], [], []
    for si, s, p in cases:
        K = psets[si][0]
        st = R[si].find_wall_exit_status(s, p, K)
        assert st in (0, 1), st
        if st == 0:
            lab, iv = R[si].classify_read(s, p, K, want_intvl=True)
            assert lab[:K - 1] == b"N" * (K - 1) and len(lab) == len(s)
        else:
            lab, iv = b"", np.zeros(0, INTVL_DTYPE)
        seqs.append(np.frombuffer(s, np.uint8)); profs.append(p); sets.append(si); status.append(st)
        labs.append(np.frombuffer(lab, np.uint8)); asg.append(iv["asgn"].copy())
    cat = lambda L, dt: np.concatenate(L) if len(L) else np.zeros(0, dt)
    off = lambda L: np.concatenate([[0], np.cumsum([len(x) for x in L])]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "labels.npz"), set=np.array(sets, np.int32), status=np.array(status, np.int8),
                        seq=cat(seqs, np.uint8), seq_off=off(seqs), prof=cat(profs, np.uint16), prof_off=off(profs),
                        labels=cat(labs, np.uint8), labels_off=off(labs), asgn=cat(asg, np.int8), asgn_off=off(asg), **arrs)
    print("labels.npz: %d reads, %d exit(1), %d bases labelled" % (len(cases), sum(status), sum(map(len, labs))))


if __name__ == "__main__":
    if not ref_available() and not os.path.exists("/root/reference/src/ClassPro.h"):
        sys.exit("oracle/_ref is not built and /root/reference is absent")
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for f in (prims, context, classify, fastk_files, eval_tools, dazz_db, seeds, wall, labels):
        if not only or f.__name__ in only:
            f()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
