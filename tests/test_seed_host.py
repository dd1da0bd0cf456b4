"""The product's seed-path routine (classpro_amd/csrc/cp_seed.h, the code one lane of k_find_seeds runs) compiled
for the host (tests/host_harness.cpp) against the reference-generated golden vectors and against the oracle on fresh
inputs, including sequences with non-ACGT letters."""
import ctypes as C

import numpy as np

from conftest import load_golden


def _run(H, seq, lab, prof, K):
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
