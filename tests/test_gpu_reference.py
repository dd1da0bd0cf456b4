"""The HIP path (through the C ABI) against records produced by the REFERENCE's own text -- not via the oracle.  `-m gpu`.

tests/golden/wall.npz: per read the reference's find_wall + find_rel_intvl output (wall.c:570-958, 960-1051, the GSL-free
part of wall.c compiled as it stands; oracle/gen_golden.py).  tests/golden/labels.npz: label strings and interval classes
of whole reads through context.c -> wall.c -> class_rel.c -> class_unrel.c -> paint (ClassPro.c:229-271).  K = 21 / 25 /
40 / 63, -r 2 000 - 25 000, coverages (12,25) - (30,60), one set under a -M error model; generated, adversarial, tail-run, edge and tiny reads; the reads
on which the reference exit(1)s ("# E-intvls >= plen") must raise CP_EOVERFLOW, each on its own.
Bar: bit-exact, the three doubles of a record included.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

DBL_FIELDS = ("pe", "peo_b", "peo_e")


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def reads_of_set(g, si):
    so, po = g["seq_off"], g["prof_off"]
    idx = [i for i in range(len(so) - 1) if int(g["set"][i]) == si]
    return idx, [g["seq"][so[i]:so[i + 1]].tobytes() for i in idx], [np.ascontiguousarray(g["prof"][po[i]:po[i + 1]]) for i in idx]


def classifier_for(g, si):
    """A Classifier for parameter set si whose tables are the ones the reference's find_wall was handed."""
    from classpro_amd.api import Classifier
    K, rl, h, d, m = (int(x) for x in g["psets"][si])
    # the -M set: the reference's find_wall was handed the oracle's fit of the synthetic HIsim model; the product gets the
    # same table (cp_params_create_pe) -- its own fit of that file agrees to rtol 1e-12, not to the bit (the fit is the one
    # part without a reference: GSL; tests/test_error_model.py), and this test is about find_wall, not about the fit
    clf = Classifier(K, rl, h, d, pe=(g["pe"][si] if m else None))
    ex = clf.export()
    assert np.array_equal(ex["cthres"], g["cthres"][si]) and np.array_equal(ex["pe"], g["pe"][si])
    assert ex["cmax"] == g["cmax"][si] and ex["hc_erate"] == g["hc_erate"][si]
    return clf, K


def assert_aborts(clf, seqs, profs):
    from classpro_amd.api import Batch
    from classpro_amd._lib import ClassProError
    for s, p in zip(seqs, profs):
        with pytest.raises(ClassProError) as ei:
            clf.classify(Batch.from_reads([s], [p]))
        assert ei.value.code == -5                          # CP_EOVERFLOW = the reference's exit(1)


@pytest.mark.parametrize("si", range(7))
def test_wall_and_rel_stage_against_reference_records(torch_dev, si):
    from classpro_amd.api import Batch, STAGE_WALL, STAGE_REL
    from oracle.oracle import INTVL_DTYPE
    g = load_golden("wall.npz")
    clf, K = classifier_for(g, si)
    idx, seqs, profs = reads_of_set(g, si)
    st = g["status"][idx]
    assert_aborts(clf, [s for s, t in zip(seqs, st) if t == 1], [p for p, t in zip(profs, st) if t == 1])
    keep = [k for k, t in enumerate(st) if t == 0]
    b = Batch.from_reads([seqs[k] for k in keep], [profs[k] for k in keep])
    io, ro = g["intvl_off"], g["rintvl_off"]
    iv_all, rv_all = g["intvl"].view(INTVL_DTYPE), g["rintvl"].view(INTVL_DTYPE)

    clf.run(b, STAGE_WALL)                                  # find_wall alone: Intvl[N] before find_rel_intvl touches it
    clf.check()
    for k, (iv, _) in zip(keep, clf.intervals(b)):
        want = iv_all[io[idx[k]]:io[idx[k] + 1]]
        assert len(iv) == len(want), idx[k]
        for f in ("b", "e", "cb", "ce"):
            assert np.array_equal(iv[f], want[f]), (idx[k], f)
        for f in DBL_FIELDS:
            assert np.array_equal(iv[f].view(np.uint64), want[f].view(np.uint64)), (idx[k], f)

    clf.run(b, STAGE_REL)                                   # + find_rel_intvl / correct_wall_cnt
    clf.check()
    n_rel = 0
    for k, (iv, rv) in zip(keep, clf.intervals(b)):
        want, wantr = iv_all[io[idx[k]]:io[idx[k] + 1]], rv_all[ro[idx[k]]:ro[idx[k] + 1]]
        assert len(iv) == len(want) and len(rv) == len(wantr), idx[k]
        for f in ("b", "e", "cb", "ce", "is_rel"):
            assert np.array_equal(iv[f], want[f]), (idx[k], f)
        rel = want["is_rel"] != 0                           # corrected counts are defined for the reliable intervals
        for f in ("ccb", "cce"):
            assert np.array_equal(iv[f][rel], want[f][rel]), (idx[k], f)
        for f in ("b", "e", "cb", "ce", "ccb", "cce", "is_rel"):
            assert np.array_equal(rv[f], wantr[f]), (idx[k], f)
        for f in DBL_FIELDS:
            assert np.array_equal(iv[f].view(np.uint64), want[f].view(np.uint64)), (idx[k], f)
            assert np.array_equal(rv[f].view(np.uint64), wantr[f].view(np.uint64)), (idx[k], f)
        n_rel += len(rv)
    assert n_rel > 100
    clf.close()


@pytest.mark.parametrize("si", range(7))
def test_labels_against_reference_text_only(torch_dev, si):
    from classpro_amd.api import Batch, STAGE_CLASS_ALL
    g = load_golden("labels.npz")
    clf, K = classifier_for(g, si)
    idx, seqs, profs = reads_of_set(g, si)
    st = g["status"][idx]
    assert_aborts(clf, [s for s, t in zip(seqs, st) if t == 1], [p for p, t in zip(profs, st) if t == 1])
    keep = [k for k, t in enumerate(st) if t == 0]
    b = Batch.from_reads([seqs[k] for k in keep], [profs[k] for k in keep])
    lo, ao = g["labels_off"], g["asgn_off"]
    got = clf.classify(b).tobytes()
    want = b"".join(g["labels"][lo[idx[k]]:lo[idx[k] + 1]].tobytes() for k in keep)
    assert got == want
    clf.run(b, STAGE_CLASS_ALL)
    clf.check()
    for k, (iv, _) in zip(keep, clf.intervals(b)):
        assert np.array_equal(iv["asgn"], g["asgn"][ao[idx[k]]:ao[idx[k] + 1]]), idx[k]
    # one read at a time and in reversed order: the same bytes (a read's result is a function of the read)
    rev = keep[::-1]
    b2 = Batch.from_reads([seqs[k] for k in rev], [profs[k] for k in rev])
    got2 = clf.classify(b2).tobytes()
    assert got2 == b"".join(g["labels"][lo[idx[k]]:lo[idx[k] + 1]].tobytes() for k in rev)
    clf.close()
