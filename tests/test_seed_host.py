"""The scalar pieces of the seed path that compile for the host (classpro_amd/csrc/cp_seed.h through
tests/host_harness.cpp): the canonical ntHash of a k-mer against the hashes the reference itself computed
(tests/golden/seeds.npz, all of nthash.h's seedTab rows) and against the oracle on fresh sequences with odd letters.
The seed selection itself is a wave-level device routine (cp_seed_wave.h); tests/test_gpu_seeds.py holds its cases."""
import ctypes as C

import numpy as np

from conftest import load_golden


def test_kmer_hash_on_golden_vectors(harness):
    g = load_golden("seeds.npz")
    for i in range(int(g["n"])):
        K = int(g["K%d" % i])
        seq = g["seq%d" % i].tobytes()
        hs = g["hash%d" % i]
        for j in range(0, len(hs), max(1, len(hs) // 50)):
            assert harness.hh_kmer_hash(seq, j, K) == int(hs[j])


def test_kmer_hash_against_oracle_odd_letters(harness):
    from oracle.oracle import Oracle
    rng = np.random.default_rng(123)
    letters = np.frombuffer(b"ACGTacgtNnUuRYKM", np.uint8)
    for K in (40, 21, 63):
        O = Oracle(K, 20000, 20, 40)
        for _ in range(5):
            plen = int(rng.integers(1, 3000))
            seq = bytes(letters[rng.choice(len(letters), plen + K - 1, p=[.22, .22, .22, .22] + [.01] * 12)])
            h = O.kmer_hash(seq, K)
            for j in range(0, plen, max(1, plen // 40)):
                assert harness.hh_kmer_hash(seq, j, K) == int(h[j])
