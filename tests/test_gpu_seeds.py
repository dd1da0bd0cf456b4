"""The -s seed path on a real MI355X (`-m gpu`): cp_find_seeds_batch (k_seed_caps + k_find_seeds, one wave per read) through the C ABI
against the oracle, which is pinned against the reference's own seed.c (tests/test_oracle_seeds.py), and against the
reference-generated golden vectors directly.  Bit-exact: seed labels and .rep intervals are integer/byte outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K = 40


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def test_seeds_golden_on_device(torch_dev):
    """The reference's own outputs (tests/golden/seeds.npz): labels are handed to the device as they are."""
    torch = torch_dev
    from conftest import load_golden
    from classpro_amd.api import Classifier, Batch
    g = load_golden("seeds.npz")
    byK = {}
    for i in range(int(g["n"])):
        byK.setdefault(int(g["K%d" % i]), []).append(i)
    for Kx, idx in byK.items():
        clf = Classifier(K=Kx, read_len=20000, hcov=20, dcov=40)
        idx = [i for i in idx if len(g["prof%d" % i]) >= 1]
        b = Batch.from_reads([g["seq%d" % i].tobytes() for i in idx], [g["prof%d" % i] for i in idx])
        lab = np.concatenate([g["lab%d" % i] for i in idx])
        b.labels = torch.from_numpy(lab.copy()).to(b.device)
        seeds, reps = clf.find_seeds(b)
        so = b.seq_off_h
        for j, i in enumerate(idx):
            got = seeds[so[j]:so[j + 1]]
            assert got[:Kx - 1].tobytes() == b"N" * (Kx - 1)
            assert np.array_equal(got[Kx - 1:], g["sasgn%d" % i]), (Kx, i)
            assert np.array_equal(reps[j].reshape(-1, 2), g["rep%d" % i].reshape(-1, 2)), (Kx, i)
        clf.close()


def test_seeds_after_classification_60x(torch_dev):
    """BASELINE configs[4] in small: 60x, r=25000; classify then find seeds on the device, oracle does the same."""
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=400000, cov=60, read_len=25000, seed=17, het=0.002, n_repeats=12, min_len=4000)
    seqs, profs = ds["seqs"], ds["profiles"]
    clf = Classifier(K=K, read_len=25000, hcov=30, dcov=60)
    b = Batch.from_reads(seqs, profs)
    lab = clf.classify(b)
    seeds, reps = clf.find_seeds(b)
    O = Oracle(K, 25000, 30, 60)
    so = b.seq_off_h
    nseed = 0
    for j, (s, p) in enumerate(zip(seqs, profs)):
        want_lab = O.classify_read(s, p)
        assert lab[so[j]:so[j + 1]].tobytes() == want_lab
        sas, rep = O.find_seeds(s, want_lab, p)
        assert np.array_equal(seeds[so[j] + K - 1:so[j + 1]], sas), j
        assert np.array_equal(reps[j].reshape(-1, 2), rep.reshape(-1, 2)), j
        nseed += int((sas != ord("E")).sum())
    assert nseed > 10 * len(seqs)
    clf.close()


def _check_against_oracle(clf, O, Kx, cases):
    """cases: (seq bytes, label bytes incl. the N prefix, profile).  Labels go to the device as they are."""
    import torch
    from classpro_amd.api import Batch
    b = Batch.from_reads([c[0] for c in cases], [c[2] for c in cases])
    b.labels = torch.from_numpy(np.frombuffer(b"".join(c[1] for c in cases), np.uint8).copy()).to(b.device)
    seeds, reps = clf.find_seeds(b)
    so = b.seq_off_h
    for j, (seq, labs, prof) in enumerate(cases):
        want, wrep = O.find_seeds(seq, labs, prof)
        got = seeds[so[j]:so[j + 1]]
        assert got[:Kx - 1].tobytes() == b"N" * min(Kx - 1, len(got))
        assert np.array_equal(got[Kx - 1:], want), (Kx, j, len(prof))
        assert np.array_equal(reps[j].reshape(-1, 2), wrep.reshape(-1, 2)), (Kx, j)


def test_seeds_random_labels_odd_letters(torch_dev):
    """Random label strings, count profiles up to 1500 and sequences with every kind of letter nthash.h knows, K = 40,
    21 and 70 (longer than the kernel's rotated-seed table: the byte-wise fold)."""
    from classpro_amd.api import Classifier
    from oracle.oracle import Oracle
    rng = np.random.default_rng(123)
    letters = np.frombuffer(b"ACGTacgtNnUuRYKM", np.uint8)
    for Kx in (40, 21, 70):
        O = Oracle(Kx, 20000, 20, 40)
        clf = Classifier(K=Kx, read_len=20000, hcov=20, dcov=40)
        cases = []
        for rep_i in range(25):
            plen = int(rng.integers(1, 6000))
            seq = bytes(letters[rng.choice(len(letters), plen + Kx - 1, p=[.22, .22, .22, .22] + [.01] * 12)])
            lab = np.repeat(np.frombuffer(b"EHDR", np.uint8)[rng.integers(0, 4, plen)], rng.integers(1, 300, plen))[:plen]
            prof = np.repeat(rng.integers(1, 1500 if rep_i % 5 == 0 else 70, plen), rng.integers(1, 12, plen))[:plen].astype(np.uint16)
            cases.append((seq, b"N" * (Kx - 1) + lab.tobytes(), prof))
        _check_against_oracle(clf, O, Kx, cases)
        clf.close()


def test_seeds_beyond_the_on_chip_sizes(torch_dev):
    """Every on-chip bound of the kernel is passed: staircase profiles make the monotone deque deeper than its LDS ring;
    counts above 1000 give window counts above the window (the sorted head is ordered by insertion); hundreds of short
    label islands make the masked-interval list longer than its LDS part, in the first selection and in a later one;
    more than 64 repetitive stretches; groups with more than 64 members to take (equal window counts all over the
    read); a read of more than 65535 k-mers; a selection of more than 65535 segments."""
    from classpro_amd.api import Classifier
    from oracle.oracle import Oracle
    rng = np.random.default_rng(9)
    AL = np.frombuffer(b"ACGT", np.uint8)
    O = Oracle(40, 20000, 20, 40)
    clf = Classifier(K=40, read_len=20000, hcov=20, dcov=40)
    cases = []

    def add(prof, lab):
        plen = len(prof)
        seq = bytes(AL[rng.integers(0, 4, plen + 39)])
        cases.append((seq, b"N" * 39 + bytes(lab), np.ascontiguousarray(prof, np.uint16)))
    plen = 5000
    for case in range(6):
        if case < 3:                                                  # deep deque
            step = [1, 2, 3][case]
            prof = (3000 - (np.arange(plen) // step) % 400).astype(np.uint16)
            if case == 2:
                prof = prof[::-1].copy()
            lab = np.repeat(np.frombuffer(b"DDHDR", np.uint8)[rng.integers(0, 5, plen)], rng.integers(50, 900, plen))[:plen]
        else:                                                         # window counts above the window
            prof = np.repeat(rng.integers(900, 1400, plen), rng.integers(1, 6, plen))[:plen].astype(np.uint16)
            lab = np.repeat(np.frombuffer(b"HD", np.uint8)[rng.integers(0, 2, plen)], rng.integers(100, 2000, plen))[:plen]
        add(prof, lab.tobytes())
    for plen, isl in ((30000, 12), (60000, 25), (30000, 5)):          # long masked-interval lists
        lab = np.repeat(np.frombuffer(b"DEHEDRHE", np.uint8)[np.arange(plen) % 8], rng.integers(1, 2 * isl, plen))[:plen]
        prof = np.repeat(rng.integers(10, 70, plen), rng.integers(1, 8, plen))[:plen]
        add(prof, lab.tobytes())
    plen = 40000                                                      # > 64 repetitive stretches: R islands between unique stretches
    lab = np.tile(np.frombuffer(b"D" * 150 + b"R" * 60 + b"H" * 120 + b"R" * 30, np.uint8), plen // 360 + 1)[:plen]
    add(np.repeat(rng.integers(10, 70, plen), rng.integers(1, 8, plen))[:plen], lab.tobytes())
    plen = 50000                                                      # big groups: every segment the same length and the same stand-in
    prof = np.tile(np.repeat(np.array([30, 31, 30, 29], np.uint16), 3), plen // 12 + 1)[:plen]
    add(prof, b"D" * plen)
    prof = np.tile(np.repeat(np.array([30, 40, 30, 20], np.uint16), 40), plen // 160 + 1)[:plen]
    add(prof, (b"D" * 300 + b"R" * 200) * (plen // 500))
    plen = 70000                                                      # more than 65535 k-mers
    lab = np.repeat(np.frombuffer(b"EHDR", np.uint8)[rng.choice(4, plen, p=[.1, .2, .5, .2])], rng.integers(1, 400, plen))[:plen]
    add(np.repeat(rng.integers(1, 80, plen), rng.integers(1, 9, plen))[:plen], lab.tobytes())
    plen = 90000                                                      # more than 65535 SEGMENTS in one selection: the sort's counters
    prof = (20 + (np.arange(plen) % 2) * 7 + (np.arange(plen) // 3000) % 5).astype(np.uint16)   # no longer fit 16 bits (two rounds of keys)
    add(prof, b"D" * plen)
    _check_against_oracle(clf, O, 40, cases)
    clf.close()
