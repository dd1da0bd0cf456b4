"""The oracle's restatement of the -s seed path (oracle/classpro_oracle_seed.c: find_seeds, anno_repeat, ntHash)
against the reference's OWN seed.c: through the committed golden vectors (tests/golden/seeds.npz, made by
oracle/gen_golden.py from oracle/_ref) and, when oracle/_ref is present, live on fresh inputs."""
import numpy as np
import pytest

from conftest import load_golden


def test_seeds_golden(built):
    from oracle.oracle import Oracle
    g = load_golden("seeds.npz")
    n = int(g["n"])
    assert n >= 60
    kinds = set()
    for i in range(n):
        K = int(g["K%d" % i])
        O = Oracle(K, 20000, 20, 40)
        seq, lab, prof = g["seq%d" % i].tobytes(), g["lab%d" % i].tobytes(), g["prof%d" % i]
        sas, rep = O.find_seeds(seq, lab, prof)
        assert np.array_equal(sas, g["sasgn%d" % i]), i
        assert np.array_equal(rep.reshape(-1, 2), g["rep%d" % i].reshape(-1, 2)), i
        assert np.array_equal(O.kmer_hash(seq, K), g["hash%d" % i]), i
        kinds |= set(np.unique(sas).tolist())
    assert kinds == set(b"EHDR")                          # seeds of every kind occur in the vectors


def test_seeds_live_against_reference(built):
    from oracle.oracle import Oracle, Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built here (reference tree absent); covered by the golden vectors")
    from classpro_amd import synth
    R = Ref(20000, 20, 40)
    ds = synth.make_dataset(genome_len=150000, cov=40, read_len=8000, seed=91, het=0.003, n_repeats=10)
    O = Oracle(40, 20000, 20, 40)
    for s, p in list(zip(ds["seqs"], ds["profiles"]))[:120]:
        lab = O.classify_read(s, p)
        a, ra = O.find_seeds(s, lab, p)
        b, rb, hb = R.find_seeds(s, lab, p)
        assert np.array_equal(a, b) and np.array_equal(ra.reshape(-1, 2), rb.reshape(-1, 2))
        assert np.array_equal(O.kmer_hash(s), hb)
