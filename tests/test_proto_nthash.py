"""The prefix-XOR restatement of ntHash (scripts/proto/nthash_prefix.py: DESIGN 8.7's next form of the seed kernel's marks)
against the oracle's kmer_hash, which is pinned against the reference's own seed.c / nthash.h (tests/golden/seeds.npz)."""
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "proto"))
import nthash_prefix as nt
from oracle.oracle import Oracle


@pytest.mark.parametrize("K", [21, 40, 63])
def test_prefix_form_is_the_oracles_hash(K):
    O = Oracle(K, 20000, 20, 40)
    rng = random.Random(100 + K)
    letters = b"ACGTacgtNnUuRYKM"
    for trial in range(6):
        n = rng.randrange(K, K + 700)
        p_odd = (0.0, 0.03, 0.4)[trial % 3]
        seq = bytes(rng.choice(letters) if rng.random() < p_odd else rng.choice(b"ACGT") for _ in range(n))
        want = O.kmer_hash(seq, K)
        got = np.array(nt.chunked(seq, K), dtype=np.int64)
        assert np.array_equal(got, want.astype(np.int64)), (K, trial)


def test_split_rotation_is_a_group_action():
    rng = random.Random(3)
    for m in (0, 1, 30, 31, 32, 33, 34, 63, 64, 1022, 1023, 1024):
        v = rng.getrandbits(64)
        w = v
        for _ in range(m):
            w = nt.srol1(w)
        assert nt.srol(v, m) == w and nt.srol(w, -m) == v
    assert nt.srol(0x123456789abcdef0, 33 * 31) == 0x123456789abcdef0
