"""The product's seed-path routine (classpro_amd/csrc/cp_seed.h, the code one lane of k_find_seeds runs) compiled
for the host (tests/host_harness.cpp) against the reference-generated golden vectors and against the oracle on fresh
inputs, including sequences with non-ACGT letters."""
import ctypes as C

import numpy as np

from conftest import load_golden


FAST = {"fast": 0, "plain": 0}


def _run(H, seq, lab, prof, K):
    """The flat form first; a read it hands back goes through the plain form, as in the kernel."""
    s = np.frombuffer(seq, np.uint8)
    cls = np.ascontiguousarray(np.frombuffer(lab, np.uint8)[K - 1:])
    prof = np.ascontiguousarray(prof, np.uint16)
    plen = len(prof)
    st = np.zeros(max(plen, 1), np.uint8)
    rep = np.zeros((plen + 2, 2), np.int32)
    runs = 1 + int((prof[1:] != prof[:-1]).sum()) if plen > 1 else 1
    lruns = 1 + int((cls[1:] != cls[:-1]).sum()) if plen > 1 else 1
    n = H.hh_find_seeds_fast(s.ctypes.data_as(C.c_char_p), cls.ctypes.data_as(C.c_char_p), prof.ctypes.data_as(C.c_void_p),
                             C.c_int(plen), C.c_int(K), st.ctypes.data_as(C.c_char_p), rep.ctypes.data_as(C.c_void_p),
                             C.c_int(plen + 2), C.c_int(runs + lruns + 6))
    assert n != -2
    if n >= 0:
        FAST["fast"] += 1
        return st[:plen], rep[:n]
    FAST["plain"] += 1
    return _run_plain(H, seq, lab, prof, K)


def _run_plain(H, seq, lab, prof, K):
    s = np.frombuffer(seq, np.uint8)
    cls = np.ascontiguousarray(np.frombuffer(lab, np.uint8)[K - 1:])
    prof = np.ascontiguousarray(prof, np.uint16)
    plen = len(prof)
    st = np.zeros(max(plen, 1), np.uint8)
    rep = np.zeros((plen + 2, 2), np.int32)
    runs = 1 + int((prof[1:] != prof[:-1]).sum()) if plen > 1 else 1
    lruns = 1 + int((cls[1:] != cls[:-1]).sum()) if plen > 1 else 1
    cap = runs + lruns + 4
    n = H.hh_find_seeds(s.ctypes.data_as(C.c_char_p), cls.ctypes.data_as(C.c_char_p), prof.ctypes.data_as(C.c_void_p),
                        C.c_int(plen), C.c_int(K), st.ctypes.data_as(C.c_char_p), rep.ctypes.data_as(C.c_void_p),
                        C.c_int(plen + 2), C.c_int(cap))
    assert n >= 0, "scratch overflow with cap = count runs + label runs + 4"
    return st[:plen], rep[:n]


def test_seed_routine_on_golden_vectors(harness):
    g = load_golden("seeds.npz")
    for i in range(int(g["n"])):
        K = int(g["K%d" % i])
        st, rep = _run(harness, g["seq%d" % i].tobytes(), g["lab%d" % i].tobytes(), g["prof%d" % i], K)
        assert np.array_equal(st, g["sasgn%d" % i]), i
        assert np.array_equal(rep.reshape(-1, 2), g["rep%d" % i].reshape(-1, 2)), i
        seq = g["seq%d" % i].tobytes()
        hs = g["hash%d" % i]
        for j in range(0, len(hs), max(1, len(hs) // 50)):
            assert harness.hh_kmer_hash(seq, j, K) == int(hs[j])


def test_seed_routine_against_oracle_fresh_inputs(harness):
    from oracle.oracle import Oracle
    rng = np.random.default_rng(123)
    letters = np.frombuffer(b"ACGTacgtNnUuRYKM", np.uint8)
    for K in (40, 21):
        O = Oracle(K, 20000, 20, 40)
        for rep_i in range(25):
            plen = int(rng.integers(1, 6000))
            seq = bytes(letters[rng.choice(len(letters), plen + K - 1, p=[.22, .22, .22, .22] + [.01] * 12)])
            lab = np.repeat(np.frombuffer(b"EHDR", np.uint8)[rng.integers(0, 4, plen)], rng.integers(1, 300, plen))[:plen]
            prof = np.repeat(rng.integers(1, 1500 if rep_i % 5 == 0 else 70, plen), rng.integers(1, 12, plen))[:plen].astype(np.uint16)
            labs = b"N" * (K - 1) + lab.tobytes()
            st, rp = _run(harness, seq, labs, prof, K)
            want, wrep = O.find_seeds(seq, labs, prof)
            assert np.array_equal(st, want), (K, rep_i, plen)
            assert np.array_equal(rp.reshape(-1, 2), wrep.reshape(-1, 2))
            h = O.kmer_hash(seq, K)
            for j in range(0, plen, max(1, plen // 20)):
                assert harness.hh_kmer_hash(seq, j, K) == int(h[j])


def test_flat_form_deep_deque_and_large_window_counts(harness):
    """Staircase profiles make the monotone deque deeper than the on-chip ring (it moves to HBM scratch); counts above
    1000 under H/D labels give window counts above the window (ordered by insertion).  Neither hands the read back."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(9)
    O = Oracle(40, 20000, 20, 40)
    FAST["fast"] = FAST["plain"] = 0
    for case in range(6):
        plen = 5000
        seq = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, plen + 39)])
        if case < 3:
            step = [1, 2, 3][case]
            prof = (3000 - (np.arange(plen) // step) % 400).astype(np.uint16)          # 400 strictly decreasing runs in a row
            if case == 2:
                prof = prof[::-1].copy()
            lab = np.repeat(np.frombuffer(b"DDHDR", np.uint8)[rng.integers(0, 5, plen)], rng.integers(50, 900, plen))[:plen]
        else:
            prof = np.repeat(rng.integers(900, 1400, plen), rng.integers(1, 6, plen))[:plen].astype(np.uint16)
            lab = np.repeat(np.frombuffer(b"HD", np.uint8)[rng.integers(0, 2, plen)], rng.integers(100, 2000, plen))[:plen]
        labs = b"N" * 39 + lab.tobytes()
        st, rp = _run(harness, seq, labs, prof, 40)
        want, wrep = O.find_seeds(seq, labs, prof)
        assert np.array_equal(st, want), case
        assert np.array_equal(rp.reshape(-1, 2), wrep.reshape(-1, 2))
    assert FAST["plain"] == 0, FAST


def test_flat_form_is_the_usual_path_and_plain_form_agrees(harness):
    """On realistic reads (labels from the classifier) the flat form must do nearly all the work; every read also goes
    through the plain form, which must give the same bytes."""
    from oracle.oracle import Oracle
    from classpro_amd import synth
    FAST["fast"] = FAST["plain"] = 0
    ds = synth.make_dataset(genome_len=250000, cov=60, read_len=25000, seed=4, het=0.002, n_repeats=8, min_len=4000)
    O = Oracle(40, 25000, 30, 60)
    for s, p in list(zip(ds["seqs"], ds["profiles"]))[:200]:
        lab = O.classify_read(s, p)
        st, rp = _run(harness, s, lab, p, 40)
        st2, rp2 = _run_plain(harness, s, lab, p, 40)
        want, wrep = O.find_seeds(s, lab, p)
        assert np.array_equal(st, want) and np.array_equal(st2, want)
        assert np.array_equal(rp.reshape(-1, 2), wrep.reshape(-1, 2)) and np.array_equal(rp2.reshape(-1, 2), wrep.reshape(-1, 2))
    assert FAST["plain"] == 0, FAST
