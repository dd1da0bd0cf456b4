"""The -s seed path on a real MI355X (`-m gpu`): cp_find_seeds_batch (k_seed_caps + k_find_seeds) through the C ABI
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
