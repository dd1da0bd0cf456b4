"""Parity of the HIP path (through the C ABI) with the oracle on a real MI355X.  `-m gpu`.

Bar: everything is bit-exact -- integer / index / byte outputs and, since round 4, the floating-point outputs of the
stage API (cp_intvl.pe, .peo_b, .peo_e = logs of probabilities) as well: the device runs glibc 2.35's own exp/log
(csrc/cp_libm.h, tests/test_libm.py), so FLOAT_RTOL = 0.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 40
FLOAT_RTOL = 0.0


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.fixture(scope="module")
def ds_a():
    from classpro_amd import synth
    return synth.make_dataset(genome_len=200000, cov=40, read_len=10000, seed=5), 20, 40


@pytest.fixture(scope="module")
def ds_b():
    from classpro_amd import synth
    return synth.make_dataset(genome_len=100000, cov=60, read_len=7000, seed=8, het=0.004, err_sub=0.002), 30, 60


def oracle_stages(O, ds):
    out = []
    for s, p in zip(ds["seqs"], ds["profiles"]):
        l, r = O.seq_context(s)
        iv = O.find_wall(p, l, r)
        iv2, riv = O.find_rel_intvl(iv, p, l, r)
        ro, io, fw, bw = O.classify_rel(riv, iv2, len(p))
        out.append(dict(wall=iv, rel=(iv2, riv), crel=(ro, io, fw, bw), call=O.classify_unrel(io)))
    return out


def close(a, b):
    fin = np.isfinite(b)
    return np.array_equal(np.isfinite(a), fin) and np.array_equal(a[~fin], b[~fin]) and np.allclose(a[fin], b[fin], rtol=FLOAT_RTOL, atol=0)


def test_scan_in_several_launches(torch_dev, ds_a, monkeypatch):
    """cp_run_stages scans a batch in launches of at most 2^31 positions (capi.hip: launch_scan); a launch that continues a
    profile takes the count before its first position from the launch before it.  With the launch size set to 4096 and 12288
    positions (CLASSPRO_SCAN_CHUNK_KMERS) the bitmap of a batch of a few hundred thousand positions -- cut inside reads, at
    counts that are candidates or not -- equals the single launch's, and so do the labels."""
    from classpro_amd.api import Classifier, Batch, STAGE_SCAN
    from classpro_amd import synth
    ds, h, d = ds_a
    seq, so, prof, po = synth.pack_batch(ds["seqs"][:60], ds["profiles"][:60])
    clf = Classifier(K, 20000, h, d)
    b = Batch(seq, so, prof, po)
    clf.run(b, STAGE_SCAN)
    want = clf.bitmap(b).copy()
    lab = clf.classify(b).copy()
    assert b.total_kmers > 5 * 12288
    for chunk in (4096, 12288):
        monkeypatch.setenv("CLASSPRO_SCAN_CHUNK_KMERS", str(chunk))
        clf.run(b, STAGE_SCAN)
        got = clf.bitmap(b)
        nw = b.total_kmers // 64
        assert np.array_equal(got[:nw], want[:nw]), chunk
        assert np.array_equal(np.unpackbits(got.view(np.uint8), bitorder="little")[:b.total_kmers],
                              np.unpackbits(want.view(np.uint8), bitorder="little")[:b.total_kmers])
        assert np.array_equal(clf.classify(b), lab)
    monkeypatch.delenv("CLASSPRO_SCAN_CHUNK_KMERS")
    clf.close()


@pytest.mark.parametrize("which", ["a", "b"])
def test_stage_parity(torch_dev, ds_a, ds_b, which):
    from classpro_amd.api import (Classifier, Batch, STAGE_SCAN, STAGE_WALL, STAGE_REL, STAGE_CLASS_REL, STAGE_CLASS_ALL)
    from classpro_amd import synth
    from oracle.oracle import Oracle
    ds, h, d = ds_a if which == "a" else ds_b
    O = Oracle(K, 20000, h, d)
    clf = Classifier(K, 20000, h, d)
    ex = clf.export()
    cov, dr, cmax, hc = O.scalars()
    assert ex["cov"] == cov and ex["dr_ratio"] == dr and ex["cmax"] == cmax and ex["hc_erate"] == hc
    assert np.array_equal(ex["cthres"], O.cthres()) and np.array_equal(ex["logfact"], O.logfact()) and np.array_equal(ex["pe"], O.pe())
    seq, so, prof, po = synth.pack_batch(ds["seqs"], ds["profiles"])
    b = Batch(seq, so, prof, po)
    want = oracle_stages(O, ds)

    # candidate scan: bit g <=> min(c[g-1],c[g]) < R and |c[g-1]-c[g]| >= 3   (wall.c:592-608)
    clf.run(b, STAGE_SCAN)
    bits = np.unpackbits(clf.bitmap(b).view(np.uint8), bitorder="little")[:b.total_kmers]
    p = prof.astype(np.int64)
    exp = np.zeros(b.total_kmers, np.uint8)
    exp[1:] = (np.minimum(p[1:], p[:-1]) < cov[1]) & (np.abs(p[1:] - p[:-1]) >= 3)
    inner = np.ones(b.total_kmers, bool)
    inner[po[:-1]] = False
    assert np.array_equal(bits[inner], exp[inner])

    clf.run(b, STAGE_WALL)
    for (iv, _), w in zip(clf.intervals(b), want):
        o = w["wall"]
        assert len(iv) == len(o)
        for f in ("b", "e", "cb", "ce"):
            assert np.array_equal(iv[f], o[f]), f
        for f in ("pe", "peo_b", "peo_e"):
            assert close(iv[f], o[f]), f

    clf.run(b, STAGE_REL)
    for (iv, riv), w in zip(clf.intervals(b), want):
        o_iv, o_riv = w["rel"]
        assert np.array_equal(iv["is_rel"], o_iv["is_rel"]) and len(riv) == len(o_riv)
        for f in ("b", "e", "cb", "ce", "ccb", "cce"):
            assert np.array_equal(riv[f], o_riv[f]), f

    clf.run(b, STAGE_CLASS_REL)
    got = clf.intervals(b)
    for (iv, riv), (fw, bw), w in zip(got, clf.rel_asgn(b), want):
        ro, io, ofw, obw = w["crel"]
        assert np.array_equal(fw, ofw) and np.array_equal(bw, obw)
        assert np.array_equal(riv["asgn"], ro["asgn"]) and np.array_equal(iv["asgn"], io["asgn"])

    clf.run(b, STAGE_CLASS_ALL)
    for (iv, _), w in zip(clf.intervals(b), want):
        assert np.array_equal(iv["asgn"], w["call"]["asgn"])

    lab = clf.classify(b)
    ref = O.classify_batch(seq, so, prof, po, nthreads=8)
    assert np.array_equal(lab, ref)                      # byte-identical label strings
    clf.close()


def test_context_golden_on_device(torch_dev):
    """calc_seq_context through the ABI against vectors produced by the reference's own context.c."""
    from conftest import load_golden
    from classpro_amd.api import Classifier, Batch
    g = load_golden("context.npz")
    off = g["off"].astype(np.int64)
    n = len(off) - 1
    po = np.concatenate([[0], np.cumsum(np.maximum(np.diff(off) - (K - 1), 0))]).astype(np.int64)
    clf = Classifier(K, 20000, 20, 40)
    b = Batch(g["seq"], off, np.ones(max(int(po[-1]), 1), np.uint16)[:int(po[-1])], po)
    for i, (l, r) in enumerate(clf.seq_context(b)):
        assert np.array_equal(l, g["lctx"][off[i]:off[i + 1]]) and np.array_equal(r, g["rctx"][off[i]:off[i + 1]])
    clf.close()


def test_edge_reads(torch_dev):
    """Shortest legal reads (rlen == K), flat / all-repeat / all-error / ramp profiles, homopolymer and
    microsatellite reads, an isolated sequencing error, a heterozygous stretch."""
    from classpro_amd.api import Classifier, Batch
    from classpro_amd import synth
    from oracle.oracle import Oracle
    AL = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(11)
    O = Oracle(K, 20000, 20, 40)
    seqs, profs, labs = [], [], []
    cases = []
    for plen in (1, 2, 3, 39, 40, 41, 200, 1000):
        rlen = plen + K - 1
        seq = bytes(AL[rng.integers(0, 4, rlen)])
        cases += [(seq, np.full(plen, 40, np.uint16)), (seq, np.full(plen, 500, np.uint16)), (seq, np.full(plen, 1, np.uint16)),
                  (seq, (1 + np.arange(plen) % 90).astype(np.uint16)),
                  (b"A" * rlen, rng.integers(1, 80, plen).astype(np.uint16)),
                  ((b"AC" * rlen)[:rlen], rng.integers(1, 80, plen).astype(np.uint16)),
                  ((b"ACG" * rlen)[:rlen], rng.integers(1, 300, plen).astype(np.uint16))]
    p = np.full(3000, 40, np.uint16)
    p[1000:1039] = 1
    p[2000:2500] = 20
    cases.append((bytes(AL[rng.integers(0, 4, 3039)]), p))
    p = np.full(5000, 32767, np.uint16)                   # maximum count everywhere
    p[100:150] = 3
    cases.append((bytes(AL[rng.integers(0, 4, 5039)]), p))
    rejected = []
    for s, p in cases:
        try:
            labs.append(O.classify_read(s, p))
        except OverflowError:                             # the reference aborts on this read
            rejected.append((s, p))
            continue
        seqs.append(s)
        profs.append(p)
    assert len(seqs) >= len(cases) - 6
    clf = Classifier(K, 20000, 20, 40)
    # the device raises CP_EOVERFLOW on exactly the reads the oracle rejects ("# E-intvls >= plen", wall.c:783-788):
    # each rejected read alone is flagged, and the accepted ones together (below) are not
    from classpro_amd._lib import ClassProError
    for s, p in rejected:
        with pytest.raises(ClassProError) as ei:
            clf.classify(Batch.from_reads([s], [p]))
        assert ei.value.code == -5
    b = Batch.from_reads(seqs, profs)
    got = clf.classify(b).tobytes()
    assert got == b"".join(labs)
    # empty batch is a no-op
    e = Batch(np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.uint16), np.zeros(1, np.int64))
    assert len(clf.classify(e)) == 0
    clf.close()


def test_scale_properties(torch_dev):
    """~50 Mbases: label alphabet, N-prefix, idempotence, batch-split and read-order invariance, and a
    full comparison with the oracle (bit-exact: 0 mismatching positions)."""
    from classpro_amd.api import Classifier, Batch
    from classpro_amd import synth
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=1_250_000, cov=40, read_len=20000, seed=41, het=0.001, n_repeats=16, min_len=3000)
    seq, so, prof, po = synth.pack_batch(ds["seqs"], ds["profiles"])
    n = len(so) - 1
    clf = Classifier(K, 20000, 19, 38)
    b = Batch(seq, so, prof, po)
    lab = clf.classify(b)
    lab2 = clf.classify(b)
    assert np.array_equal(lab, lab2)                      # idempotent / deterministic
    starts = so[:-1]
    pref = np.concatenate([np.arange(s, s + K - 1) for s in starts])
    assert np.all(lab[pref] == ord("N"))
    body = np.ones(len(lab), bool)
    body[pref] = False
    assert set(np.unique(lab[body]).tolist()) <= set(b"EHDR")
    # split into two batches: same labels (reads are independent; scratch offsets differ)
    h = n // 2
    b1 = Batch(seq[:so[h]], so[:h + 1], prof[:po[h]], po[:h + 1])
    b2 = Batch(seq[so[h]:], so[h:] - so[h], prof[po[h]:], po[h:] - po[h])
    assert np.array_equal(np.concatenate([clf.classify(b1), clf.classify(b2)]), lab)
    # reversed read order
    order = np.arange(n)[::-1]
    rs = [ds["seqs"][i] for i in order]
    rp = [ds["profiles"][i] for i in order]
    lr = clf.classify(Batch.from_reads(rs, rp))
    o2 = np.concatenate([[0], np.cumsum([len(x) for x in rs])])
    for j, i in enumerate(order[:200]):
        assert np.array_equal(lr[o2[j]:o2[j + 1]], lab[so[i]:so[i + 1]])
    # sampled oracle comparison
    m = n                                             # every read of the batch
    want = Oracle(K, 20000, 19, 38).classify_batch(seq[:so[m]], so[:m + 1], prof[:po[m]], po[:m + 1], nthreads=8)
    bad = int((lab[:so[m]] != want).sum())
    assert bad == 0, "%d mismatching positions of %d" % (bad, so[m])
    clf.close()


def test_cli_drop_in(torch_dev, tmp_path):
    """The ClassPro-compatible binary: FASTA(.gz) + FASTK .hist/.prof in, .class out, byte-identical
    to the records the reference writes (ClassPro.c:188,289) with labels from the oracle."""
    import os
    import subprocess
    from classpro_amd import synth, fastk, build
    from classpro_amd.api import hist_covs
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=150000, cov=40, read_len=9000, seed=77)
    seqs, profs, names = list(ds["seqs"]), list(ds["profiles"]), list(ds["names"])
    # a read shorter than K (printed by the host, ClassPro.c:209-226) and kseq's stale-comment quirk
    seqs.insert(5, b"ACGTACGTAC"); profs.insert(5, np.zeros(0, np.uint16)); names.insert(5, "tiny")
    comments = [None] * len(seqs)
    comments[3] = "first comment"
    comments[10] = "second one"
    d = str(tmp_path)
    with open(os.path.join(d, "reads.fasta"), "wb") as f:
        for n, s, c in zip(names, seqs, comments):
            f.write(b">" + n.encode() + ((b" " + c.encode()) if c else b"") + b"\n")
            for o in range(0, len(s), 70):                  # multi-line FASTA
                f.write(s[o:o + 70] + b"\n")
    subprocess.check_call(["gzip", "-k", os.path.join(d, "reads.fasta")])
    os.remove(os.path.join(d, "reads.fasta"))
    fastk.write_fastk(d, "reads", K, profs, ds["hist"], nparts=3)
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    r = subprocess.run([cli, "-v", "-T4", "-P" + d, os.path.join(d, "reads.fasta.gz")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Estimated (H,D) cov   = (%d,%d)" % (hc, dc) in r.stderr
    got = open(os.path.join(d, "reads.class"), "rb").read()
    O = Oracle(K, 20000, hc, dc)
    exp = []
    last = "(null)"
    for n, s, p, c in zip(names, seqs, profs, comments):
        if c:
            last = c
        lab = O.classify_read(s, p) if len(s) >= K else b"N" * len(s)
        exp.append(b"@" + n.encode() + b" " + last.encode() + b"\n" + s + b"\n+\n" + lab + b"\n")
    assert got == b"".join(exp)
    # -c overrides the histogram; error contract: message + exit code 1
    r = subprocess.run([cli, "-c40", os.path.join(d, "reads")], capture_output=True, text=True)
    assert r.returncode == 0
    r = subprocess.run([cli, os.path.join(d, "missing.fa")], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open" in r.stderr


def test_regression_merge_tail(torch_dev):
    """The read that exposed the merge-loop bound (see tests/test_host_logic.py): through the C ABI."""
    from conftest import load_golden
    from classpro_amd.api import Classifier, Batch
    g = load_golden("regress_merge_tail.npz")
    h, d = (int(x) for x in g["cov"])
    clf = Classifier(K, 20000, h, d)
    lab = clf.classify(Batch.from_reads([g["seq"].tobytes()], [np.ascontiguousarray(g["prof"])]))
    assert np.array_equal(lab, g["labels"])
    clf.close()


def test_long_and_noisy_reads_all_size_classes(torch_dev):
    """50 kb reads with elevated het/error rates (N 400-900, M 150-380: the one-read-per-wave rel class
    and the two-reads-per-wave unrel class) and block-noise profiles that push N and M past 1024 (the
    sequential fallback kernels).  Labels and final interval classes must equal the oracle's."""
    from classpro_amd.api import Classifier, Batch, STAGE_CLASS_ALL
    from classpro_amd import synth
    from oracle.oracle import Oracle
    AL = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(5)
    ds = synth.make_dataset(genome_len=400000, cov=40, read_len=50000, seed=9, het=0.004, err_sub=0.002,
                            err_indel=0.002, min_len=30000)
    seqs, profs = list(ds["seqs"][:60]), list(ds["profiles"][:60])
    for blk, lo, hi in ((60, 15, 45), (45, 5, 60), (25, 18, 42), (90, 30, 50), (45, 5, 60)):
        plen = 58000
        vals = rng.integers(lo, hi, plen // blk + 1)
        profs.append(np.repeat(vals, blk)[:plen].astype(np.uint16))
        seqs.append(bytes(AL[rng.integers(0, 4, plen + K - 1)]))
    O = Oracle(K, 20000, 20, 40)
    want, Ns, Ms = [], [], []
    for s, p in zip(seqs, profs):
        lab, iv, M = O.classify_read(s, p, want_intvl=True)
        want.append(lab)
        Ns.append(len(iv))
        Ms.append(M)
    assert max(Ns) > 1024 and max(Ms) > 1024 and sum(256 < n <= 1024 for n in Ns) > 10 and sum(128 < m <= 1024 for m in Ms) > 10
    clf = Classifier(K, 20000, 20, 40)
    b = Batch.from_reads(seqs, profs)
    got = clf.classify(b).tobytes()
    assert got == b"".join(want)
    nc, ni, nr, off = clf.counts(b)
    assert list(ni) == Ns and list(nr) == Ms
    clf.close()


def test_reads_longer_than_65535_kmers(torch_dev, tmp_path):
    """A Dazzler database may hold reads beyond the FASTX limit (ClassPro.c:87,110 size everything by db->maxlen): they
    take the sequential classify kernels (the lane-parallel ones keep 16-bit interval ends).  Library and command line."""
    import os
    import subprocess
    from classpro_amd import synth, fastk, dazz, build
    from classpro_amd.api import Classifier, Batch, hist_covs
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=300000, cov=30, read_len=72000, seed=33, min_len=20000, max_len=110000, het=0.002)
    seqs, profs = list(ds["seqs"]), list(ds["profiles"])
    assert max(len(p) for p in profs) > 66000 and min(len(p) for p in profs) < 60000
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    O = Oracle(K, 20000, hc, dc)
    want = [O.classify_read(s, p) for s, p in zip(seqs, profs)]
    clf = Classifier(K, 20000, hc, dc)
    b = Batch.from_reads(seqs, profs)
    assert clf.classify(b).tobytes() == b"".join(want)
    seeds, reps = clf.find_seeds(b)                          # the seed path too (reads of more than 65535 k-mers)
    so = b.seq_off_h
    for j, (s_, p_, lab) in enumerate(zip(seqs, profs, want)):
        sas, rep = O.find_seeds(s_, lab, p_)
        assert np.array_equal(seeds[so[j] + K - 1:so[j + 1]], sas) and np.array_equal(reps[j].reshape(-1, 2), rep.reshape(-1, 2))
    clf.close()
    d = str(tmp_path)
    n = len(seqs)
    recs = dazz.write_db(d, "long", seqs, [(n, "a.fasta", "m1")])
    heads = dazz.db_headers([(n, "a.fasta", "m1")], recs)
    fastk.write_fastk(d, "long", K, profs, ds["hist"], nparts=1)
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    r = subprocess.run([cli, "-T4", os.path.join(d, "long.db")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = b"".join(hd.encode() + b"\n" + s + b"\n+\n" + lab + b"\n" for hd, s, lab in zip(heads, seqs, want))
    assert open(os.path.join(d, "long.class"), "rb").read() == exp


def test_device_profile_decode(torch_dev, ds_a):
    """cp_decode_profiles (Fetch_Profile on the device, libfastk.c:1467-1534) is bit-exact with the host
    decoder on every read, feeds cp_classify_batch without a host round trip, and reports a code that
    does not expand to rlen-(K-1) counts (the reference's abort, ClassPro.c:234-237)."""
    from classpro_amd.api import Classifier, Batch, encode_profiles
    from classpro_amd._lib import ClassProError
    from classpro_amd.synth import pack_batch
    ds, hc, dc = ds_a
    rng = np.random.default_rng(2)
    profs = [p.copy() for p in ds["profiles"]]
    profs[3][100:140] = rng.integers(300, 32768, 40)          # 15-bit deltas
    profs[5][:] = 17                                           # one value + runs only
    profs[7][:] = rng.integers(0, 32768, len(profs[7]))       # back-to-back 2-byte tokens, second bytes with bit 7 set
    profs[9][:] = np.repeat(rng.integers(0, 32768, len(profs[9]) // 50 + 1), 50)[:len(profs[9])]
    profs[11][:] = np.where(rng.random(len(profs[11])) < 0.5, 40, 200)   # alternating 1-/2-byte deltas
    seq, so, prof, po = pack_batch(ds["seqs"], profs)
    codes, co = encode_profiles(profs)
    clf = Classifier(K=K, read_len=10000, hcov=hc, dcov=dc)
    d = clf.decode_profiles(codes, co, po)
    got = d[:po[-1]].cpu().numpy().view(np.uint16)
    assert np.array_equal(got, prof)
    b = Batch(seq, so, prof, po)
    want = clf.classify(b)
    b.prof = d
    assert np.array_equal(clf.classify(b), want)
    bad = po.copy()
    bad[1:] += 1                                               # read 0 now claims one more count than its code holds
    with pytest.raises(ClassProError):
        clf.decode_profiles(codes, co, bad)
    empty = clf.decode_profiles(np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(1, np.int64))
    assert empty is not None
    clf.close()


def test_error_model_option(torch_dev, tmp_path):
    """-M<model>: thresholds and labels under a HIsim error model (wall.c:55-115) equal the oracle's under
    the same model (fits agree to 1e-12, see tests/test_error_model.py), and differ from the default model's."""
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch
    from oracle.oracle import Oracle
    path = str(tmp_path / "hifi.model")
    synth.write_himodel(path, growth=(0.0004, 0.0005, 0.0006))
    ds = synth.make_dataset(genome_len=100000, cov=40, read_len=8000, seed=12, err_indel=0.002)
    clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40, model=path)
    clf0 = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
    O = Oracle(K, 20000, 20, 40, model=path)
    seqs, profs = [], []
    want = []
    for s, p in zip(ds["seqs"], ds["profiles"]):
        try:
            want.append(O.classify_read(s, p))
        except OverflowError:
            continue
        seqs.append(s); profs.append(p)
    b = Batch.from_reads(seqs, profs)
    got = clf.classify(b)
    assert got.tobytes() == b"".join(want)
    assert clf0.classify(b).tobytes() != got.tobytes()
    np.testing.assert_allclose(clf.export()["pe"], O.model_pe, rtol=1e-12)     # two host-side fits (product: closed form; oracle: its own)
    clf.close(); clf0.close()


@pytest.mark.parametrize("dam", [False, True], ids=["db", "dam"])
def test_cli_on_dazzler_database(torch_dev, tmp_path, dam):
    """ClassPro on a .db / .dam (ClassPro.c:161-180, 290-304; io.c:123-313): .class records with the DB
    headers and oracle labels, plus the .class track (2-bit labels) and the header-only .rep track."""
    import os
    import subprocess
    from classpro_amd import synth, fastk, dazz, build
    from classpro_amd.api import hist_covs
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=120000, cov=40, read_len=8000, seed=78)
    seqs, profs = list(ds["seqs"]), list(ds["profiles"])
    seqs.insert(3, b"ACGTACGTACGG"); profs.insert(3, np.zeros(0, np.uint16))
    n = len(seqs)
    files = [(n // 2, "a.fasta", "m1_prolog"), (n - n // 2, "b.fasta", "m2_prolog")]
    hdr = [">ctg%d part=%d" % (i, i % 3) for i in range(n)] if dam else None
    d = str(tmp_path)
    recs = dazz.write_db(d, "reads", seqs, files, dam=dam, hdr_lines=hdr)
    heads = dazz.db_headers(files, recs, dam=dam, hdr_lines=hdr)
    fastk.write_fastk(d, "reads", K, profs, ds["hist"], nparts=2)
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    r = subprocess.run([cli, "-T2", os.path.join(d, "reads.dam" if dam else "reads.db")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    O = Oracle(K, 20000, hc, dc)
    labels = [O.classify_read(s, p) if len(s) >= K else b"N" * len(s) for s, p in zip(seqs, profs)]
    exp = b"".join(hd.encode() + b"\n" + s + b"\n+\n" + lab + b"\n" for hd, s, lab in zip(heads, seqs, labels))
    assert open(os.path.join(d, "reads.class"), "rb").read() == exp
    nreads, size, offs, raw = dazz.read_class_track(d, "reads")
    assert (nreads, size) == (n, 8) and offs[0] == 0 and offs[-1] == len(raw)
    code = np.zeros(256, np.uint8)
    code[ord("R")], code[ord("H")], code[ord("D")] = 1, 2, 3
    for i, lab in enumerate(labels):
        assert offs[i + 1] - offs[i] == (len(lab) + 3) // 4
        got = dazz.unpack_2bit(raw[offs[i]:offs[i + 1]], len(lab))
        assert np.array_equal(got, code[np.frombuffer(lab, np.uint8)])
    nr, sz, roffs, rraw = dazz.read_class_track(d, "reads", "rep")
    assert (nr, sz, len(roffs), len(rraw)) == (n, 0, 1, 0)
    # -s (ClassPro.c:281-304): same .class text, the .class track carries the seed labels, .rep the mask intervals
    r = subprocess.run([cli, "-s", "-T3", os.path.join(d, "reads")], capture_output=True, text=True,
                       env=dict(os.environ, CLASSPRO_DEVICES="0,0", CLASSPRO_BATCH_KBASES="300"))
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(d, "reads.class"), "rb").read() == exp
    nreads, size, offs, raw = dazz.read_class_track(d, "reads")
    nr, sz, roffs, rraw = dazz.read_class_track(d, "reads", "rep")
    assert (nreads, size, nr, sz) == (n, 8, n, 0) and len(roffs) == n + 1 and roffs[-1] == len(rraw)
    rints = np.frombuffer(rraw, np.int32)
    nseeds = 0
    for i, (s_, p_, lab) in enumerate(zip(seqs, profs, labels)):
        got = dazz.unpack_2bit(raw[offs[i]:offs[i + 1]], len(lab))
        if len(s_) < K:
            assert not got.any() and roffs[i + 1] == roffs[i]
            continue
        sas, rep = O.find_seeds(s_, lab, p_)
        assert np.array_equal(got[K - 1:], code[sas]) and not got[:K - 1].any(), i
        assert np.array_equal(rints[roffs[i] // 4:roffs[i + 1] // 4], rep.reshape(-1)), i
        nseeds += int((sas != ord("E")).sum())
    assert nseeds > n
    # both tracks load through DAZZ_DB's own Open_Track / Load_All_Track_Data (the reference's DB.c, when oracle/_ref travelled)
    import ctypes as C
    refso = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libclasspro_ref.so")
    if os.path.exists(refso):
        L = C.CDLL(refso)
        L.ref_db_track.restype = C.c_longlong
        L.ref_db_track.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_longlong]
        assert L.ref_db_open(os.path.join(d, "reads.dam" if dam else "reads.db").encode()) == (1 if dam else 0)
        alen = (C.c_int * n)()
        data = C.create_string_buffer(len(raw) + len(rraw) + 16)
        assert L.ref_db_track(b"class", alen, data, len(data)) == len(raw) and data.raw[:len(raw)] == raw
        assert L.ref_db_track(b"rep", alen, data, len(data)) == len(rraw) and data.raw[:len(rraw)] == rraw
        assert [alen[i] for i in range(n)] == [int(roffs[i + 1] - roffs[i]) for i in range(n)]
        L.ref_db_close()
    r = subprocess.run([cli, "-s", os.path.join(d, "nothere.fasta")], capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.parametrize("k,read_len,cov", [(25, 6000, 30), (63, 12000, 50), (101, 12000, 50)], ids=["k25", "k63", "k101_no_window"])
def test_other_kmer_lengths(torch_dev, k, read_len, cov):
    """Nothing in the path is specialised to K = 40: labels for other k-mer lengths (the paired-flag window
    and the partner offsets of find_wall scale with K) equal the oracle's."""
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch, hist_covs
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=150000, cov=cov, read_len=read_len, K=k, seed=90 + k, err_indel=0.001)
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    O = Oracle(k, 20000, hc, dc)
    seqs, profs, want = [], [], []
    for s_, p_ in zip(ds["seqs"], ds["profiles"]):
        try:
            want.append(O.classify_read(s_, p_))
        except OverflowError:
            continue
        seqs.append(s_); profs.append(p_)
    clf = Classifier(K=k, read_len=20000, hcov=hc, dcov=dc)
    got = clf.classify(Batch.from_reads(seqs, profs))
    assert got.tobytes() == b"".join(want)
    clf.close()


def test_unpack_bases_on_device(torch_dev):
    """cp_unpack_bases == Load_Read(...,2) of DAZZ_DB (DB.c:1232-1298): 2-bit bases, first base in the top bits,
    lengths that are not multiples of four, a read of length 1."""
    from classpro_amd import dazz
    from classpro_amd.api import Classifier
    rng = np.random.default_rng(3)
    lens = [1, 2, 3, 4, 5, 63, 64, 65, 1000, 4097, 20001]
    seqs = [bytes(b"ACGT"[x] for x in rng.integers(0, 4, n)) for n in lens]
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    packed = [dazz.pack_2bit([code[c] for c in s]) for s in seqs]
    poff = np.concatenate([[0], np.cumsum([len(p) for p in packed])]).astype(np.int64)
    soff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
    out = clf.unpack_bases(np.concatenate(packed), poff, soff)
    assert out[:soff[-1]].cpu().numpy().tobytes() == b"".join(seqs)
    clf.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_adversarial_inputs(torch_dev, seed):
    from classpro_amd.api import Classifier, Batch
    from oracle.oracle import Oracle
    from adversarial import adversarial_reads
    seqs, profs = adversarial_reads(seed)
    hc, dc = [(20, 40), (19, 38), (30, 60), (45, 90)][seed - 1]
    O = Oracle(K, 20000, hc, dc)
    keep_s, keep_p, want, rejected = [], [], [], []
    for s_, p_ in zip(seqs, profs):
        try:
            want.append(O.classify_read(s_, p_))
        except OverflowError:                                             # the reference aborts on this read
            rejected.append((s_, p_))
            continue
        keep_s.append(s_); keep_p.append(p_)
    assert len(want) > 150
    clf = Classifier(K=K, read_len=20000, hcov=hc, dcov=dc)
    from classpro_amd._lib import ClassProError
    for s_, p_ in rejected:                                               # ... and the device flags exactly those reads
        with pytest.raises(ClassProError) as ei:
            clf.classify(Batch.from_reads([s_], [p_]))
        assert ei.value.code == -5
    got = clf.classify(Batch.from_reads(keep_s, keep_p))
    off = 0
    for r, w in enumerate(want):
        assert got[off:off + len(w)].tobytes() == w, "read %d of seed %d" % (r, seed)
        off += len(w)
    clf.close()


def test_find_rel_fused_and_as_a_kernel(torch_dev, ds_a, monkeypatch):
    """find_rel_intvl runs inside k_find_wall on the whole-path call (CLASSPRO_FUSE_REL unset) or as k_find_rel
    (CLASSPRO_FUSE_REL=0): same labels, same interval records byte for byte, both equal to the oracle's labels."""
    from classpro_amd.api import Classifier, Batch, INTVL_DTYPE, STAGE_CLASS_ALL
    from classpro_amd._lib import check
    from oracle.oracle import Oracle
    from adversarial import adversarial_reads, tail_run_reads
    ds, h, d = ds_a
    a_s, a_p = adversarial_reads(12, n=120)
    t_s, t_p = tail_run_reads(13, n=60)
    O = Oracle(K, 20000, h, d)
    seqs, profs = list(ds["seqs"][:150]), list(ds["profiles"][:150])
    for s_, p_ in zip(a_s + t_s, a_p + t_p):
        try:
            O.classify_read(s_, p_)
        except OverflowError:
            continue
        seqs.append(s_); profs.append(p_)
    want = b"".join(O.classify_read(s_, p_) for s_, p_ in zip(seqs, profs))
    out = []
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("CLASSPRO_FUSE_REL", raising=False)
        else:
            monkeypatch.setenv("CLASSPRO_FUSE_REL", env)
        clf = Classifier(K, 20000, h, d)
        b = Batch.from_reads(seqs, profs)
        lab = clf.classify(b).tobytes()
        clf.run(b, STAGE_CLASS_ALL)
        nc, ni, nr, off = clf.counts(b)
        tot = int(off[-1])
        iv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)
        rv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)
        check(clf.L.cp_get_intervals(clf.ws, iv.ctypes.data, rv.ctypes.data, tot))
        idx = np.arange(tot) - off[np.searchsorted(off, np.arange(tot), side="right") - 1]
        out.append((lab, iv[idx < np.repeat(ni, np.diff(off))].tobytes(), rv[idx < np.repeat(nr, np.diff(off))].tobytes(), nr.tobytes()))
        clf.close()
    assert out[0][0] == want
    assert out[0] == out[1]


def test_whole_path_compact_records_equal_full_records(torch_dev, ds_a, monkeypatch):
    """cp_classify_batch hands classify_rel 24-byte records and the label paint 4-byte (end, class) words instead of the
    48-byte records (capi.hip: compact_rel; CLASSPRO_COMPACT_REL=0 keeps the records): labels, the interval records that
    cp_get_intervals returns (final classes patched in on read-back) and the label runs are the same bytes either way --
    on generated reads, adversarial and tail-run ones and reads of the rare size classes (M > 112, N > 256, M / N > 1024)."""
    from classpro_amd.api import Classifier, Batch, expand_label_runs
    from classpro_amd._lib import check
    from classpro_amd import synth
    from oracle.oracle import Oracle, INTVL_DTYPE
    from adversarial import adversarial_reads, tail_run_reads
    ds, h, d = ds_a
    O = Oracle(K, 20000, h, d)
    AL = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(15)
    seqs, profs = list(ds["seqs"][:40]), list(ds["profiles"][:40])
    a_s, a_p = adversarial_reads(31, n=80)
    t_s, t_p = tail_run_reads(32, n=40)
    big = synth.make_dataset(genome_len=300000, cov=40, read_len=50000, seed=9, het=0.004, err_sub=0.002, err_indel=0.002, min_len=30000)
    b_s, b_p = list(big["seqs"][:12]), list(big["profiles"][:12])
    for blk, lo, hi in ((60, 15, 45), (45, 5, 60)):           # N and M beyond 1024: the sequential kernels
        plen = 58000
        b_p.append(np.repeat(rng.integers(lo, hi, plen // blk + 1), blk)[:plen].astype(np.uint16))
        b_s.append(bytes(AL[rng.integers(0, 4, plen + K - 1)]))
    for s_, p_ in zip(a_s + t_s + b_s, a_p + t_p + b_p):
        try:
            O.classify_read(s_, p_)
        except OverflowError:
            continue
        seqs.append(s_); profs.append(p_)
    out = []
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("CLASSPRO_COMPACT_REL", raising=False)
        else:
            monkeypatch.setenv("CLASSPRO_COMPACT_REL", env)
        clf = Classifier(K, 20000, h, d)
        b = Batch.from_reads(seqs, profs)
        lab = clf.classify(b).tobytes()
        nc, ni, nr, off = clf.counts(b)
        tot = int(off[-1])
        iv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)
        rv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)
        check(clf.L.cp_get_intervals(clf.ws, iv.ctypes.data, rv.ctypes.data, tot))
        idx = np.arange(tot) - off[np.searchsorted(off, np.arange(tot), side="right") - 1]
        live = idx < np.repeat(ni, np.diff(off))
        so = b.seq_off_h
        clf.classify(b)
        runs = clf.label_runs(b, rerun=False)               # straight after the whole-path call: from the (end, class) words
        strings = b"".join(expand_label_runs(e_, c_, int(so[r + 1] - so[r]), K) for r, (e_, c_) in enumerate(runs))
        runs2 = clf.label_runs(b)                           # and after a run that stops at CP_STAGE_CLASS_ALL: from the records
        assert strings == b"".join(expand_label_runs(e_, c_, int(so[r + 1] - so[r]), K) for r, (e_, c_) in enumerate(runs2))
        out.append((lab, iv[live].tobytes(), ni.tobytes(), nr.tobytes(), strings))
        if env is None:
            assert not rv.any()                             # no copies of the reliable intervals on the compact path
        assert strings == lab
        clf.close()
    monkeypatch.delenv("CLASSPRO_COMPACT_REL", raising=False)
    assert max(np.frombuffer(out[0][3], np.int32)) > 1024 and max(np.frombuffer(out[0][2], np.int32)) > 1024
    assert out[0] == out[1]
    assert out[0][0] == b"".join(O.classify_read(s_, p_) for s_, p_ in zip(seqs, profs))


def test_second_sweep_rule_on_device(torch_dev, ds_a, monkeypatch):
    """classify_unrel's second sweep re-evaluates only intervals whose inputs changed since the first (kernels.hip,
    k_classify_unrel_grp; the rule itself: tests/test_unrel_memo.py).  CLASSPRO_UNREL_SWEEP2=full evaluates every one, as
    class_unrel.c:267-274 does: the same labels and interval classes, and the oracle's, on generated, adversarial and
    tail-run reads and on reads of the one-read-per-wave class (256 < N <= 1024), with full and with compact records."""
    from classpro_amd.api import Classifier, Batch, STAGE_CLASS_ALL
    from classpro_amd import synth
    from oracle.oracle import Oracle
    from adversarial import adversarial_reads, tail_run_reads
    ds, h, d = ds_a
    O = Oracle(K, 20000, h, d)
    seqs, profs = list(ds["seqs"][:60]), list(ds["profiles"][:60])
    a_s, a_p = adversarial_reads(41, n=160)
    t_s, t_p = tail_run_reads(42, n=60)
    big = synth.make_dataset(genome_len=300000, cov=40, read_len=50000, seed=19, het=0.004, err_sub=0.002, err_indel=0.002, min_len=30000)
    for s_, p_ in zip(a_s + t_s + list(big["seqs"][:10]), a_p + t_p + list(big["profiles"][:10])):
        try:
            O.classify_read(s_, p_)
        except OverflowError:
            continue
        seqs.append(s_); profs.append(p_)
    want = b"".join(O.classify_read(s_, p_) for s_, p_ in zip(seqs, profs))
    out = []
    for sweep2 in (None, "full"):
        for compact in (None, "0"):
            for name, v in (("CLASSPRO_UNREL_SWEEP2", sweep2), ("CLASSPRO_COMPACT_REL", compact)):
                if v is None:
                    monkeypatch.delenv(name, raising=False)
                else:
                    monkeypatch.setenv(name, v)
            clf = Classifier(K, 20000, h, d)
            b = Batch.from_reads(seqs, profs)
            lab = clf.classify(b).tobytes()
            clf.run(b, STAGE_CLASS_ALL)
            clf.check()
            ivs = clf.intervals(b)
            out.append((lab, b"".join(iv["asgn"].tobytes() for iv, _ in ivs)))
            clf.close()
    monkeypatch.delenv("CLASSPRO_UNREL_SWEEP2", raising=False)
    monkeypatch.delenv("CLASSPRO_COMPACT_REL", raising=False)
    assert max(len(iv) for iv, _ in ivs) > 256
    assert all(o == out[0] for o in out[1:])
    assert out[0][0] == want


def test_fuzz_regressions(torch_dev):
    """Reads that the fuzz soak (scripts/fuzz_parity.py) once found different.  fuzz305_34: classify_unrel's argmax meets
    log(px*py) against log(px)+log(py); with ocml's log the device said E where the oracle (glibc) says D."""
    import os
    from classpro_amd.api import Classifier, Batch
    from oracle.oracle import Oracle
    from conftest import GOLDEN
    for name, hc, dc in (("fuzz305_34.npz", 60, 120),):
        z = np.load(os.path.join(GOLDEN, name))
        s_, p_ = z["seq"].tobytes(), z["prof"]
        want = Oracle(K, 20000, hc, dc).classify_read(s_, p_)
        clf = Classifier(K=K, read_len=20000, hcov=hc, dcov=dc)
        assert clf.classify(Batch.from_reads([s_], [p_])).tobytes() == want, name
        clf.close()


def test_skellam_table_is_what_the_kernels_compute(torch_dev, ds_b, monkeypatch):
    """logp_trans comes from a device-built table (cp_types.h: |ce-cb| <= 255, cov*|e-b| below a bound) or, outside
    it, from the recurrence on the spot.  The table is filled by that very code, so a classifier without a table
    (CLASSPRO_TABLES=0), one with a tiny table (most products outside it) and the default one must agree to
    the last bit: labels, interval records (incl. their doubles) and the run on a 200-Mbase device-generated batch."""
    torch = torch_dev
    import ctypes as C
    from classpro_amd.api import Classifier, Batch, INTVL_DTYPE, hist_covs
    from classpro_amd._lib import check
    from classpro_amd.synth_dev import DeviceSynth
    ds, hc, dc = ds_b

    def run(mb, batch_fn, hc_, dc_, rl):
        monkeypatch.delenv("CLASSPRO_SKELLAM_TABLE_MB", raising=False)
        monkeypatch.delenv("CLASSPRO_TABLES", raising=False)
        if mb == 0:
            monkeypatch.setenv("CLASSPRO_TABLES", "0")               # none of the three tables
        elif mb is not None:
            monkeypatch.setenv("CLASSPRO_SKELLAM_TABLE_MB", str(mb))
        clf = Classifier(K=K, read_len=rl, hcov=hc_, dcov=dc_)
        tb = clf.tables()
        assert (tb["skel"] == 0) == (mb == 0) and (tb["uerr"] == 0) == (mb == 0) and (tb["petab"] == 0) == (mb == 0), tb
        if mb == 1:
            assert tb["skel"] == 2 << 20                    # the logp_trans table and the table of its exponentials
        b = batch_fn()
        lab = clf.classify(b).copy()
        nc, ni, nr, off = clf.counts(b)
        tot = int(off[-1])
        iv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)          # raw records: every byte of them is compared
        rv = np.zeros((tot, INTVL_DTYPE.itemsize), np.uint8)
        check(clf.L.cp_get_intervals(clf.ws, iv.ctypes.data, rv.ctypes.data, tot))
        live = np.arange(tot) - off[np.searchsorted(off, np.arange(tot), side="right") - 1] < np.repeat(ni, np.diff(off))
        clf.close()
        return lab, iv[live].tobytes()
    small = lambda: Batch.from_reads(ds["seqs"], ds["profiles"])
    ref = run(0, small, hc, dc, 7000)
    for mb in (1, None):
        got = run(mb, small, hc, dc, 7000)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1], mb
    sy = DeviceSynth(genome_len=5_000_000, cov=40, read_len=20000, seed=4)
    h2, d2 = hist_covs(sy.hist[4], 1, 32767, 0, 0, 0)
    big = lambda: Batch.from_device(sy.reads(0, sy.n_reads))
    ref = run(0, big, h2, d2, 20000)
    got = run(None, big, h2, d2, 20000)
    assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]
