"""A read's result is a function of the read alone; a workspace may grow between calls.  `-m gpu`.

1. Hazard 8: correct_wall_cnt reads profile[plen] when a low-complexity run reaches the end of the read
   (wall.c:976-978).  Defined as 0.  Reads built to hit it (tests/adversarial.py: tail_run_reads) are classified alone,
   last in a batch whose buffers are followed by poison, first in a batch whose buffers are preceded by poison, and
   between neighbours whose first / last counts are 0, 1 and 32767: identical bytes every time, equal to the oracle
   (labels, interval ends, reliable flags, corrected counts).  The same is asked of the general adversarial reads.
2. ADVICE r3 (high): one Classifier on a small batch, then on a batch large enough to outgrow the prefix-sum state
   array (the launch tag used to start over and the host took the previous batch's totals), with a -s call in between.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K = 40
POISON_COUNT = 0x7fff


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def poisoned_batch(torch, seqs, profs, front=0, back=4096):
    """A Batch whose profile and base buffers are views into larger ones filled with poison (counts 32767, bases 'A'
    or 'T') before and after: what lies around a batch in memory must not matter.  `front` keeps 16-byte alignment."""
    from classpro_amd.api import Batch
    b = Batch.from_reads(seqs, profs)
    assert front % 16 == 0
    nk, nb = b.total_kmers, b.total_bases
    big_p = torch.full((front + max(nk, 1) + back,), POISON_COUNT, dtype=torch.int16, device=b.device)
    big_p[front:front + nk] = b.prof[:nk]
    big_s = torch.full((front + max(nb, 1) + back,), ord("T") if front else ord("A"), dtype=torch.uint8, device=b.device)
    big_s[front:front + nb] = b.seq[:nb]
    b._keep = (big_p, big_s)
    b.prof, b.seq = big_p[front:front + max(nk, 1)], big_s[front:front + max(nb, 1)]
    return b


def pick(a, fields):
    """The named fields of a record array as bytes (a multi-field view would drag the other fields' bytes along)."""
    return b"".join(np.ascontiguousarray(a[f]).tobytes() for f in fields)


IV_F, RV_F = ("b", "e", "cb", "ce", "is_rel"), ("b", "e", "ccb", "cce")


def records(clf, b):
    """Per read: (labels, N, M, interval ends + reliable flags, reliable (b, e, ccb, cce))."""
    from classpro_amd.api import STAGE_CLASS_ALL
    lab = clf.classify(b).copy()
    clf.run(b, STAGE_CLASS_ALL)
    ivs = clf.intervals(b)
    so = b.seq_off_h
    out = []
    for r, (iv, rv) in enumerate(ivs):
        out.append((lab[so[r]:so[r + 1]].tobytes(), pick(iv, IV_F), pick(rv, RV_F)))
    return out


def neighbour(rng, first, last):
    L = int(rng.integers(60, 400))
    s = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)])
    p = rng.integers(1, 80, L - K + 1).astype(np.uint16)
    p[0], p[-1] = first, last
    return s, p


@pytest.mark.parametrize("read_len,gen", [(20000, "tail"), (2000, "tail"), (20000, "adv")])
def test_result_is_a_function_of_the_read(torch_dev, read_len, gen):
    from classpro_amd.api import Classifier
    from oracle.oracle import Oracle
    from adversarial import tail_run_reads, adversarial_reads
    rng = np.random.default_rng(5)
    seqs, profs = tail_run_reads(21, n=160) if gen == "tail" else adversarial_reads(9, n=160)
    O = Oracle(K, read_len, 20, 40)
    keep, want = [], []
    for s, p in zip(seqs, profs):
        try:
            lab, iv, M = O.classify_read(s, p, want_intvl=True)
        except OverflowError:
            continue
        rel = iv[iv["is_rel"] != 0]
        keep.append((s, p))
        want.append((lab, pick(iv, IV_F), pick(rel, RV_F)))
    assert len(keep) > 100
    if gen == "tail":
        assert sum(1 for w in want if len(w[2])) > 60            # reliable intervals present: ccb/cce are compared
    clf = Classifier(K=K, read_len=read_len, hcov=20, dcov=40)
    S, P = [k[0] for k in keep], [k[1] for k in keep]
    # (a) every read alone, its buffers followed (and, second pass, preceded) by poison
    for front in (0, 64):
        for r in range(0, len(keep), 3 if front else 1):
            got = records(clf, poisoned_batch(torch_dev, [S[r]], [P[r]], front=front))
            assert got[0] == want[r], "read %d alone (front poison %d)" % (r, front)
    # (b) all of them in one batch, in order and reversed: every read is somebody's neighbour, one is last
    for order in (list(range(len(keep))), list(range(len(keep) - 1, -1, -1))):
        got = records(clf, poisoned_batch(torch_dev, [S[i] for i in order], [P[i] for i in order]))
        for pos, i in enumerate(order):
            assert got[pos] == want[i], "read %d at place %d of the batch" % (i, pos)
    # (c) between neighbours whose last / first counts are 0, 1, 32767
    for first in (0, 1, 32767):
        ss, pp, idx = [], [], []
        for i in range(len(keep)):
            nb = neighbour(rng, first, first)
            ss += [nb[0], S[i]]; pp += [nb[1], P[i]]; idx.append(len(ss) - 1)
        nb = neighbour(rng, first, first)
        ss.append(nb[0]); pp.append(nb[1])
        got = records(clf, poisoned_batch(torch_dev, ss, pp, front=16))
        for i, at in enumerate(idx):
            assert got[at] == want[i], "read %d between neighbours with count %d" % (i, first)
    clf.close()


def _small_reads(seed, n, lo=80, hi=260):
    """Many short reads with walls: cheap for the oracle, enough reads to cross prefix-sum tile boundaries."""
    rng = np.random.default_rng(seed)
    al = np.frombuffer(b"ACGT", np.uint8)
    seqs, profs = [], []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        plen = L - K + 1
        c = np.full(plen, 40, np.int64)
        for _k in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, plen)); e = min(plen, a + int(rng.integers(1, 90)))
            c[a:e] = int(rng.choice([1, 2, 19, 21, 60, 85, 300]))
        c += (rng.random(plen) < 0.05) * rng.integers(-3, 4, plen)
        seqs.append(bytes(al[rng.integers(0, 4, L)]))
        profs.append(np.clip(c, 1, 32767).astype(np.uint16))
    return seqs, profs


def _oracle_labels(O, seqs, profs):
    from classpro_amd.synth import pack_batch
    seq, so, prof, po = pack_batch(seqs, profs)
    return O.classify_batch(seq, so, prof, po, nthreads=8).tobytes()


def test_workspace_grows_between_calls(torch_dev):
    """500 reads, then 30 000 (29 more prefix-sum tiles than the state array had room for), then -s, then 60 000: each
    call's labels equal the oracle's.  Before the fix the second call ran with the first call's totals."""
    from classpro_amd.api import Classifier, Batch
    from oracle.oracle import Oracle
    O = Oracle(K, 20000, 20, 40)
    clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
    for step, (seed, n) in enumerate([(1, 500), (2, 30000), (3, 700), (4, 60000), (5, 40)]):
        seqs, profs = _small_reads(seed, n)
        b = Batch.from_reads(seqs, profs)
        got = clf.classify(b).tobytes()
        assert got == _oracle_labels(O, seqs, profs), "call %d (%d reads)" % (step, n)
        if step in (0, 2):                                  # a seeds call in between: its prefix sums share the state array
            seeds, reps = clf.find_seeds(b)
            so = b.seq_off_h
            for j in range(0, n, 37):
                sas, rep = O.find_seeds(seqs[j], got[so[j]:so[j + 1]], profs[j])
                assert np.array_equal(seeds[so[j] + K - 1:so[j + 1]], sas) and np.array_equal(reps[j].reshape(-1, 2), rep.reshape(-1, 2))
    clf.close()


def test_batch_limit_is_refused(torch_dev):
    """More k-mer positions than CP_MAX_BATCH_KMERS in one call: CP_EINVAL from the host check, nothing launched."""
    import ctypes as C
    from classpro_amd.api import Classifier, Batch
    from classpro_amd._lib import ClassProError, check
    clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
    seqs, profs = _small_reads(7, 4)
    b = Batch.from_reads(seqs, profs)
    with pytest.raises(ClassProError) as ei:
        check(clf.L.cp_classify_batch(clf.p, clf.ws, b.seq.data_ptr(), b.seq_off.data_ptr(), b.prof.data_ptr(),
                                      b.prof_off.data_ptr(), b.nreads, C.c_int64(b.total_bases), C.c_int64((1 << 35) + 1),
                                      b.labels.data_ptr(), clf._stream()))
    assert "split" in str(ei.value)
    from oracle.oracle import Oracle
    assert clf.classify(b).tobytes() == _oracle_labels(Oracle(K, 20000, 20, 40), seqs, profs)     # and the workspace is fine
    clf.close()


@pytest.mark.parametrize("Kx", [21, 25, 63])
def test_other_k(torch_dev, Kx):
    """The whole path at k-mer lengths other than 40: labels, interval ends, reliable flags and corrected counts equal the
    oracle's on generated, adversarial and tail-run reads made for that K -- alone in poisoned buffers and all in one batch."""
    from classpro_amd import synth
    from classpro_amd.api import Classifier
    from oracle.oracle import Oracle
    from adversarial import tail_run_reads, adversarial_reads
    ds = synth.make_dataset(genome_len=80000, cov=30, read_len=6000, K=Kx, seed=70 + Kx)
    a_s, a_p = adversarial_reads(80 + Kx, n=120, K=Kx)
    t_s, t_p = tail_run_reads(90 + Kx, n=60, K=Kx)
    O = Oracle(Kx, 20000, 15, 30)
    S, P, want = [], [], []
    for s_, p_ in zip(list(ds["seqs"]) + a_s + t_s, list(ds["profiles"]) + a_p + t_p):
        try:
            lab, iv, M = O.classify_read(s_, p_, want_intvl=True)
        except OverflowError:
            continue
        rel = iv[iv["is_rel"] != 0]
        S.append(s_); P.append(p_); want.append((lab, pick(iv, IV_F), pick(rel, RV_F)))
    assert len(S) > 200
    clf = Classifier(K=Kx, read_len=20000, hcov=15, dcov=30)
    got = records(clf, poisoned_batch(torch_dev, S, P, front=16))
    for i in range(len(S)):
        assert got[i] == want[i], "K = %d, read %d of the batch" % (Kx, i)
    for r in range(0, len(S), 7):
        assert records(clf, poisoned_batch(torch_dev, [S[r]], [P[r]]))[0] == want[r], "K = %d, read %d alone" % (Kx, r)
    clf.close()


def _many_intervals(rng, nint, seglen, noisy):
    """A read of `nint` stretches alternating between the diploid and the haploid level (each a reliable interval when it
    is K or longer); `noisy`: a third of the stretches shorter than K, some at error / repeat levels."""
    lens = np.full(nint, seglen)
    lv = np.where(np.arange(nint) % 2 == 0, 40, 20) + rng.integers(-2, 3, nint)
    if noisy:
        short = rng.random(nint) < 0.33
        lens = np.where(short, rng.integers(5, K - 2, nint), lens)
        lv = np.where(rng.random(nint) < 0.1, rng.choice([1, 2, 90, 300], nint), lv)
    c = np.repeat(lv, lens)
    s = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(c) + K - 1)])
    return s, c.astype(np.uint16)


def test_large_interval_counts(torch_dev):
    """Every size class of the classify kernels against the oracle: M (reliable intervals) of 150 and 600 (the one-read-per-wave
    class above 112), 1100 and 1400 (beyond 1024: the sequential kernel), N (all intervals) up to ~2000 with short and
    odd-level stretches in between, and a read beyond 65535 k-mers -- alone and together in one batch."""
    from classpro_amd.api import Classifier
    from oracle.oracle import Oracle
    rng = np.random.default_rng(77)
    O = Oracle(K, 20000, 20, 40)
    S, P, want, sizes = [], [], [], []
    for nint, seglen, noisy in ((150, 60, False), (600, 60, False), (1100, 45, False), (1400, 45, False), (1500, 60, False),
                                (300, 60, True), (900, 50, True), (2000, 44, True), (120, 60, True), (40, 70, False)):
        s_, p_ = _many_intervals(rng, nint, seglen, noisy)
        try:
            lab, iv, M = O.classify_read(s_, p_, want_intvl=True)
        except OverflowError:
            continue
        rel = iv[iv["is_rel"] != 0]
        S.append(s_); P.append(p_); want.append((lab, pick(iv, IV_F), pick(rel, RV_F))); sizes.append((len(iv), M, len(p_)))
    Ms, Ns = [m for _, m, _ in sizes], [n for n, _, _ in sizes]
    assert max(Ms) > 1024 and any(112 < m <= 1024 for m in Ms) and max(Ns) > 1024 and any(256 < n <= 1024 for n in Ns)
    assert any(pl > 65535 for _, _, pl in sizes)
    clf = Classifier(K=K, read_len=20000, hcov=20, dcov=40)
    got = records(clf, poisoned_batch(torch_dev, S, P, front=16))
    for i in range(len(S)):
        assert got[i] == want[i], "read %d of the batch (N, M, plen = %s)" % (i, sizes[i])
    for i in range(len(S)):
        assert records(clf, poisoned_batch(torch_dev, [S[i]], [P[i]]))[0] == want[i], "read %d alone (N, M, plen = %s)" % (i, sizes[i])
    clf.close()
