"""Host logic and the product's scalar device functions (compiled for the host by tests/host_harness.cpp)
against the oracle.  No GPU needed."""
import ctypes as C
import numpy as np
import pytest

from conftest import load_golden
from oracle.oracle import Oracle, INTVL_DTYPE
from classpro_amd import fastk, synth

AL = np.frombuffer(b"ACGT", np.uint8)


def hh_ctx(H, s):
    rlen = len(s)
    l = np.zeros((rlen, 3), np.uint8)
    r = np.zeros((rlen, 3), np.uint8)
    H.hh_seq_context(C.c_char_p(s), rlen, l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
    return l, r


def test_closed_form_context_random(harness):
    """cp_ctx.h evaluates contexts in closed form; the oracle runs the reference's sequential pass."""
    O = Oracle()
    rng = np.random.default_rng(1)
    for it in range(4000):
        L = int(rng.integers(1, 300))
        s = bytes(AL[rng.integers(0, int(rng.integers(1, 5)), size=L)])
        k = it % 4
        if k == 1:
            u = bytes(AL[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
            s = s[:L // 3] + u * int(rng.integers(1, 60)) + s[L // 3:]
        elif k == 2:
            parts = []
            for _ in range(int(rng.integers(1, 6))):
                u = bytes(AL[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
                parts += [u * int(rng.integers(1, 12)), bytes(AL[rng.integers(0, 4, size=int(rng.integers(0, 4)))])]
            s = b"".join(parts) or b"A"
        elif k == 3 and it % 40 == 3:          # runs beyond the 127 cap
            u = bytes(AL[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
            s = s[:5] + u * int(rng.integers(100, 300)) + s[5:]
        a, b = O.seq_context(s), hh_ctx(harness, s)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), s


def test_context_words_equal_the_scans(harness):
    """cp_ctx.h, round 5: the contexts from eight bases at a time (cp_lctx3 / cp_rctx3, what the kernels call) against the
    per-base scans and the oracle's sequential pass: random reads, runs of every unit length that end inside, at and beyond
    the eight-base word, runs at both ends of a read, runs beyond the 127 cap, reads shorter than a word."""
    O = Oracle()
    rng = np.random.default_rng(5)
    harness.hh_seq_context3.restype = C.c_longlong
    hits = total = 0
    cases = []
    for ul in (1, 2, 3):
        for rep in list(range(1, 14)) + [40, 130, 300]:
            u = bytes(AL[rng.permutation(4)[:ul]])
            for pre, post in ((0, 0), (0, 9), (9, 0), (3, 3), (20, 20)):
                cases.append(bytes(AL[rng.integers(0, 4, size=pre)]) + u * rep + bytes(AL[rng.integers(0, 4, size=post)]))
    for _ in range(600):
        cases.append(bytes(AL[rng.integers(0, int(rng.integers(2, 5)), size=int(rng.integers(1, 200)))]))
    for s in cases:
        rlen = len(s)
        a = [np.zeros((rlen, 3), np.uint8) for _ in range(4)]
        hits += harness.hh_seq_context3(C.c_char_p(s), rlen, *[x.ctypes.data_as(C.c_void_p) for x in a])
        total += 2 * rlen
        ol, orr = O.seq_context(s)
        assert np.array_equal(a[0], a[2]) and np.array_equal(a[1], a[3]), s
        assert np.array_equal(a[0], ol) and np.array_equal(a[1], orr), s
    assert hits > 0.3 * total          # the words decide most positions of ordinary sequence (and none inside long runs)


def test_closed_form_context_golden(harness):
    g = load_golden("context.npz")
    off = g["off"]
    for i in range(len(off) - 1):
        s = g["seq"][off[i]:off[i + 1]].tobytes()
        l, r = hh_ctx(harness, s)
        assert np.array_equal(l, g["lctx"][off[i]:off[i + 1]]) and np.array_equal(r, g["rctx"][off[i]:off[i + 1]])


def test_host_tables(harness):
    for h, d in ((20, 40), (19, 38), (30, 60), (12, 25)):
        O = Oracle(40, 20000, h, d)
        P = harness.hh_params_new(40, 20000, h, d)
        ct = np.ctypeslib.as_array(C.cast(harness.hh_params_cthres(C.c_void_p(P)), C.POINTER(C.c_uint8)), shape=(3, 21, 256, 2, 2))
        lf = np.ctypeslib.as_array(C.cast(harness.hh_params_logfact(C.c_void_p(P)), C.POINTER(C.c_double)), shape=(32768,))
        assert np.array_equal(ct, O.cthres()) and np.array_equal(lf, O.logfact())
        harness.hh_params_free(C.c_void_p(P))
    assert not harness.hh_params_new(40, 20000, 100, 200)


def test_host_hist_and_decode(harness):
    g = load_golden("fastk.npz")
    h, d = C.c_int(), C.c_int()
    hist = np.ascontiguousarray(g["hist"], np.int64)
    rc = harness.hh_hist_covs(hist.ctypes.data_as(C.c_void_p), int(g["low"]), int(g["high"]), C.c_int64(int(g["ilow"])),
                              C.c_int64(int(g["ihigh"])), 0, C.byref(h), C.byref(d))
    assert rc == 0 and (h.value, d.value) == (int(g["covs"][0]), int(g["covs"][1]))
    po, co = g["prof_off"], g["code_off"]
    out = np.zeros(60000, np.uint16)
    for i in range(len(po) - 1):
        code = np.ascontiguousarray(g["codes"][co[i]:co[i + 1]])
        n = harness.hh_decode_profile(code.ctypes.data_as(C.c_void_p), C.c_int64(len(code)), out.ctypes.data_as(C.c_void_p), 60000)
        assert n == po[i + 1] - po[i] and np.array_equal(out[:n], g["prof"][po[i]:po[i + 1]])
        # truncated buffer: same count, nothing written past the cap (libfastk.c:1492-1495,1522-1525)
        if n > 10:
            out2 = np.full(60000, 9999, np.uint16)
            n2 = harness.hh_decode_profile(code.ctypes.data_as(C.c_void_p), C.c_int64(len(code)), out2.ctypes.data_as(C.c_void_p), 10)
            assert n2 == n and np.all(out2[10:] == 9999)


def test_hist_covs_small_range(harness):
    """A histogram that ends right after its peak (high = 100, peak at 80): the partner-peak window around 160 lies past
    the array (the reference reads past its own buffer there); cells past the end read as zero, so the answer is the
    peak as D and peak/2 as H."""
    h, d = C.c_int(), C.c_int()
    hist = np.zeros(100, np.int64)
    hist[79] = 5000                       # count 80
    hist[78] = hist[80] = 3000
    hist[39] = 2000                       # count 40
    hist[38] = hist[40] = 900
    rc = harness.hh_hist_covs(hist.ctypes.data_as(C.c_void_p), 1, 100, C.c_int64(0), C.c_int64(0), 0, C.byref(h), C.byref(d))
    assert rc == 0 and (h.value, d.value) == (40, 80)
    hist[39] = hist[38] = hist[40] = 0    # no left partner either: the right side is all zero -> H = peak, D = 2 * peak
    rc = harness.hh_hist_covs(hist.ctypes.data_as(C.c_void_p), 1, 100, C.c_int64(0), C.c_int64(0), 0, C.byref(h), C.byref(d))
    assert rc == 0 and (h.value, d.value) == (80, 160)


def test_fastk_encode_roundtrip():
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(1, 400))
        c = rng.integers(1, 60, n)
        if rng.random() < 0.3:
            c[rng.integers(0, n, 3)] = rng.integers(1, 32767, 3)
        c = np.repeat(c, rng.integers(1, 5, n))[:n].astype(np.uint16)
        assert np.array_equal(fastk.decode_profile(fastk.encode_profile(c)), c)


def run_harness_read(H, P, s, p):
    rlen = len(s)
    cap = rlen + 2
    lab = np.zeros(rlen, np.uint8)
    iv = np.zeros(cap, INTVL_DTYPE)
    riv = np.zeros(cap, INTVL_DTYPE)
    fw = np.zeros(cap, np.int8)
    bw = np.zeros(cap, np.int8)
    M = C.c_int()
    sb = np.frombuffer(s, np.uint8)
    N = H.hh_classify_read(C.c_void_p(P), sb.ctypes.data_as(C.c_void_p), rlen, p.ctypes.data_as(C.c_void_p),
                           lab.ctypes.data_as(C.c_void_p), iv.ctypes.data_as(C.c_void_p), cap, C.byref(M),
                           riv.ctypes.data_as(C.c_void_p), fw.ctypes.data_as(C.c_void_p), bw.ctypes.data_as(C.c_void_p))
    return N, lab.tobytes(), iv[:max(N, 0)], riv[:M.value], fw[:M.value], bw[:M.value]


@pytest.mark.parametrize("seed,h,d,kw", [
    (5, 20, 40, dict(genome_len=150000, cov=40, read_len=10000)),
    (7, 25, 50, dict(genome_len=100000, cov=50, read_len=6000, het=0.004)),
    (9, 15, 30, dict(genome_len=100000, cov=30, read_len=15000, het=0.001, err_sub=0.002, err_indel=0.002)),
])
def test_device_functions_vs_oracle(harness, seed, h, d, kw):
    """Every stage of the product's scalar device code (on-demand context, O(1) path trackers, ...)
    must reproduce the oracle bit for bit when compiled for the host (same libm)."""
    ds = synth.make_dataset(seed=seed, **kw)
    O = Oracle(40, 20000, h, d)
    P = harness.hh_params_new(40, 20000, h, d)
    for s, p in zip(ds["seqs"], ds["profiles"]):
        lab_o, iv_o, M_o = O.classify_read(s, p, want_intvl=True)
        l, r = O.seq_context(s)
        ivr, riv_o = O.find_rel_intvl(O.find_wall(p, l, r), p, l, r)
        ro, _io, fw_o, bw_o = O.classify_rel(riv_o, ivr, len(p))
        N, lab, iv, riv, fw, bw = run_harness_read(harness, P, s, p)
        assert N == len(iv_o) and lab == lab_o
        for f in ("b", "e", "cb", "ce", "is_rel", "asgn", "pe", "peo_b", "peo_e"):
            assert np.array_equal(iv[f], iv_o[f]), f
        assert len(riv) == M_o
        for f in ("b", "e", "ccb", "cce", "asgn"):
            assert np.array_equal(riv[f], ro[f]), f
        assert np.array_equal(fw, fw_o) and np.array_equal(bw, bw_o)
    harness.hh_params_free(C.c_void_p(P))


def test_edge_reads(harness):
    """Flat, all-repeat, ramp and single-k-mer profiles; homopolymer and microsatellite reads."""
    O = Oracle(40, 20000, 20, 40)
    P = harness.hh_params_new(40, 20000, 20, 40)
    rng = np.random.default_rng(11)
    cases = []
    for plen in (1, 2, 3, 39, 40, 41, 200):
        rlen = plen + 39
        seq = bytes(AL[rng.integers(0, 4, rlen)])
        cases.append((seq, np.full(plen, 40, np.uint16)))
        cases.append((seq, np.full(plen, 500, np.uint16)))
        cases.append((seq, np.full(plen, 1, np.uint16)))
        cases.append((seq, (1 + np.arange(plen) % 90).astype(np.uint16)))
        cases.append((b"A" * rlen, rng.integers(1, 80, plen).astype(np.uint16)))
        cases.append(((b"AC" * rlen)[:rlen], rng.integers(1, 80, plen).astype(np.uint16)))
    p = np.full(3000, 40, np.uint16)
    p[1000:1039] = 1                              # one clean sequencing error
    p[2000:2500] = 20                             # a heterozygous stretch
    cases.append((bytes(AL[rng.integers(0, 4, 3039)]), p))
    n_ok = 0
    for s, p in cases:
        try:
            lab_o = O.classify_read(s, p)
        except OverflowError:                     # the reference aborts on this read ("# E-intvls >= plen")
            continue
        n_ok += 1
        N, lab, *_ = run_harness_read(harness, P, s, p)
        assert N >= 0 and lab == lab_o
        assert set(lab[39:]) <= set(b"EHDR") and lab[:39] == b"N" * min(39, len(s))
    assert n_ok >= len(cases) - 4
    harness.hh_params_free(C.c_void_p(P))


def test_regression_merge_scans_appended_entries(harness):
    """wall.c:878-909 bounds its merge loops by NS while appending unions, so the scan continues into the
    appended entries; a frozen bound changes which duplicate (b,e) record the later binary search finds.
    Read taken from the bench workload (seed 1, read 4184); expected labels are the oracle's."""
    g = load_golden("regress_merge_tail.npz")
    h, d = (int(x) for x in g["cov"])
    s, p = g["seq"].tobytes(), np.ascontiguousarray(g["prof"])
    O = Oracle(40, 20000, h, d)
    assert O.classify_read(s, p) == g["labels"].tobytes()
    P = harness.hh_params_new(40, 20000, h, d)
    N, lab, iv, riv, fw, bw = run_harness_read(harness, P, s, p)
    assert N == int(g["n_intvl"]) and len(riv) == int(g["n_rel"]) and lab == g["labels"].tobytes()
    harness.hh_params_free(C.c_void_p(P))


def test_device_functions_vs_oracle_long_reads(harness):
    """20 kb reads (more intervals per read, boundary E-intervals at both ends)."""
    ds = synth.make_dataset(genome_len=600000, cov=40, read_len=20000, seed=1, het=0.001, n_repeats=8, min_len=3000)
    O = Oracle(40, 20000, 19, 38)
    P = harness.hh_params_new(40, 20000, 19, 38)
    for s, p in zip(ds["seqs"], ds["profiles"]):
        N, lab, *_ = run_harness_read(harness, P, s, p)
        assert lab == O.classify_read(s, p)
    harness.hh_params_free(C.c_void_p(P))


def test_device_functions_vs_oracle_adversarial(harness):
    """The product's scalar functions (pre / filter / live / replay split of the candidate walk, post-walk,
    classify) compiled for the host against the oracle on adversarial inputs (tests/adversarial.py)."""
    from adversarial import adversarial_reads
    from oracle.oracle import Oracle
    seqs, profs = adversarial_reads(7, n=120)
    O = Oracle(40, 20000, 20, 40)
    P = harness.hh_params_new(40, 20000, 20, 40)
    n = 0
    for s, p in zip(seqs, profs):
        try:
            want = O.classify_read(s, p)
        except OverflowError:
            continue
        N, lab, *_ = run_harness_read(harness, P, s, p)
        assert lab == want
        n += 1
    assert n > 60
    harness.hh_params_free(C.c_void_p(P))


def _with_neighbour(s, p, first_count, K=40):
    """(seq, seq_off, prof, prof_off) of a two-read batch: the read, then a neighbour whose first count is given."""
    nb = b"ACGTTGCA" * 15
    npf = np.full(len(nb) - K + 1, first_count, np.uint16)
    seq = np.frombuffer(s + nb, np.uint8)
    prof = np.concatenate([p, npf])
    return seq, [0, len(s), len(s) + len(nb)], prof, [0, len(p), len(p) + len(npf)]


@pytest.mark.parametrize("read_len", [20000, 2000])
def test_tail_runs_result_is_a_function_of_the_read(harness, read_len):
    """Hazard 8: correct_wall_cnt reads profile[plen] when a low-complexity run reaches the end of the
    read (wall.c:976-978).  That cell is DEFINED as 0: the oracle and the product's scalar code must agree on reads
    built to hit it, and a read's labels and interval records must not change with what follows it in a batch.
    (Under scripts/sanitize.sh the exact-size numpy buffers of the single-read calls make any read past the end of a
    profile a hard error in both implementations.)"""
    from adversarial import tail_run_reads
    seqs, profs = tail_run_reads(3, n=150)
    O = Oracle(40, read_len, 20, 40)
    P = harness.hh_params_new(40, read_len, 20, 40)
    hit = rel_tail = 0
    for s, p in zip(seqs, profs):
        try:
            want, iv, M = O.classify_read(s, p, want_intvl=True)
        except OverflowError:
            continue
        hit += 1
        rel_tail += int(iv[-1]["is_rel"])
        N, lab, hiv, hriv, *_ = run_harness_read(harness, P, s, p)
        assert lab == want and N == len(iv)
        for f in ("b", "e", "cb", "ce", "is_rel"):
            assert np.array_equal(hiv[f], iv[f]), f
        rel = iv[iv["is_rel"] != 0]                   # corrected counts mean something on reliable intervals only
        for f in ("b", "e", "ccb", "cce"):
            assert np.array_equal(hriv[f], rel[f]), f
        for first in (0, 1, 32767):
            got = O.classify_batch(*_with_neighbour(s, p, first))[:len(s)].tobytes()
            assert got == want, "labels depend on the next read's first count (%d)" % first
    assert hit > 100 and rel_tail > 20            # the generator does reach reliable last intervals
    harness.hh_params_free(C.c_void_p(P))


def test_pack_bases_and_unpack_labels_host(built):
    """cp_pack_bases == Compress_Read's layout (gene_core.c:235-254; classpro_amd.dazz.pack_2bit restates it for the
    database writer) for pure upper-case ACGT and refuses anything else; cp_unpack_labels inverts the track payload."""
    from classpro_amd import dazz
    from classpro_amd.api import pack_bases, unpack_labels
    rng = np.random.default_rng(7)
    seqs = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]) for n in (1, 2, 3, 4, 5, 63, 64, 65, 1000, 4099)]
    packed, off = pack_bases(seqs)
    code = np.zeros(256, np.uint8)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    for i, s in enumerate(seqs):
        want = dazz.pack_2bit(code[np.frombuffer(s, np.uint8)])
        assert np.array_equal(packed[off[i]:off[i + 1]], want), i
    for bad in (b"ACGN", b"acgt", b"ACGTACGU", b"AC-T", b"ACG\x00"):
        assert pack_bases([seqs[5], bad]) is None
    assert pack_bases([b""])[1].tolist() == [0, 0]
    # the 32-bases-per-step path (reads of >= 64 bases): every length mod 32, a foreign letter at every place of a group,
    # and the threaded batch call against the read-by-read one
    from classpro_amd._lib import lib
    L = lib()
    long_reads = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 64 + n)]) for n in range(70)] + [b"", b"A"]
    pk, off = pack_bases(long_reads)
    for i, s_ in enumerate(long_reads):
        assert np.array_equal(pk[off[i]:off[i + 1]], dazz.pack_2bit(code[np.frombuffer(s_, np.uint8)])), i
    for pos in range(0, 96):
        for ch in (b"N", b"a", b"U", b"\x00", b"\xc1"):
            bad = bytearray(long_reads[40]); bad[pos] = ch[0]
            assert pack_bases([bytes(bad)]) is None, (pos, ch)
    seq = np.frombuffer(b"".join(long_reads), np.uint8)
    so = np.concatenate([[0], np.cumsum([len(x) for x in long_reads])]).astype(np.int64)
    for nt in (1, 3, 8, 200):
        out = np.zeros(int(off[-1]) + 1, np.uint8)
        assert L.cp_pack_bases_batch(seq.ctypes.data, so.ctypes.data, len(long_reads), out.ctypes.data, off.ctypes.data, nt) == 1
        assert np.array_equal(out[:off[-1]], pk[:off[-1]]), nt
    seq2 = seq.copy(); seq2[so[33] + 5] = ord("N")
    assert L.cp_pack_bases_batch(seq2.ctypes.data, so.ctypes.data, len(long_reads), out.ctypes.data, off.ctypes.data, 4) == 0
    # labels: K-1 'N', then E/R/H/D from the 2-bit codes (0,1,2,3), whatever sits in the N part and the padding
    K = 7
    for n in (1, 5, 6, 7, 8, 9, 100, 1001):
        lab = np.frombuffer(b"ERHD", np.uint8)[rng.integers(0, 4, n)].copy()
        lab[:min(K - 1, n)] = ord("N")
        codes = np.zeros(n, np.uint8)
        for ch, v in ((b"R", 1), (b"H", 2), (b"D", 3)):
            codes[lab == ch[0]] = v
        pk = dazz.pack_2bit(codes)
        got = unpack_labels(pk, [0], [n], K)
        assert got.tobytes() == lab.tobytes(), n


def test_expand_label_runs_host(built):
    """cp_expand_label_runs: K-1 'N', then the runs; refuses runs that are out of order, beyond the read or do not cover it."""
    from classpro_amd.api import expand_label_runs
    from classpro_amd._lib import ClassProError
    K = 5
    assert expand_label_runs([7, 9, 12], np.frombuffer(b"DEH", np.uint8), 12, K) == b"NNNNDDDEEHHH"
    assert expand_label_runs([], [], 3, K) == b"NNN"                       # a read shorter than K
    assert expand_label_runs([5], np.frombuffer(b"R", np.uint8), 5, K) == b"NNNNR"
    for ends, rlen in (([7, 6, 12], 12), ([7, 13], 12), ([7, 9], 12)):
        with pytest.raises(ClassProError):
            expand_label_runs(ends, np.frombuffer(b"DEH", np.uint8)[:len(ends)], rlen, K)


@pytest.mark.parametrize("Kx", [21, 25, 63])
def test_device_functions_other_k(harness, Kx):
    """The path at k-mer lengths other than 40 (FASTK's -k; K = 25 / 63 are the ends of the reference's seed tables): the
    product's scalar code against the oracle on generated, adversarial and tail-run reads made for that K."""
    from adversarial import adversarial_reads, tail_run_reads
    ds = synth.make_dataset(genome_len=60000, cov=30, read_len=5000, K=Kx, seed=40 + Kx)
    a_s, a_p = adversarial_reads(50 + Kx, n=60, K=Kx)
    t_s, t_p = tail_run_reads(60 + Kx, n=40, K=Kx)
    O = Oracle(Kx, 20000, 15, 30)
    P = harness.hh_params_new(Kx, 20000, 15, 30)
    n = 0
    for s, p in zip(list(ds["seqs"]) + a_s + t_s, list(ds["profiles"]) + a_p + t_p):
        if len(p) != len(s) - Kx + 1:
            continue
        try:
            want = O.classify_read(s, p)
        except OverflowError:
            continue
        N, lab, *_ = run_harness_read(harness, P, s, p)
        assert lab == want
        n += 1
    assert n > 100
    harness.hh_params_free(C.c_void_p(P))
