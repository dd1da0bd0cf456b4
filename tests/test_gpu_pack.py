"""2-bit payloads across PCIe (`-m gpu`): a FASTX batch of pure upper-case ACGT reads goes to the
device as host-packed 2-bit bases (cp_pack_bases -> cp_unpack_bases) + FASTK code strings (cp_decode_profiles) and its
labels come back as 2-bit codes (cp_pack_labels -> cp_unpack_labels): the label bytes equal the character path's and
the oracle's; a batch with a lower-case or ambiguous base is refused by the packer and travels as characters."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K = 40


def test_two_bit_round_trip(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch, pack_bases, unpack_labels, encode_profiles
    from classpro_amd._lib import check
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=150000, cov=30, read_len=7000, seed=21)
    seqs, profs = ds["seqs"], ds["profiles"]
    seqs = list(seqs) + [b"ACGT" * 9 + b"ACG", b"A" * 41]            # plen 0 and plen 2: shorter than four labels past the prefix
    profs = list(profs) + [np.zeros(0, np.uint16), np.array([3, 3], np.uint16)]
    seq, so, prof, po = synth.pack_batch(seqs, profs)
    clf = Classifier(K=K, read_len=20000, hcov=15, dcov=30)
    ref = Batch(seq, so, prof, po)
    want = clf.classify(ref).copy()
    assert np.array_equal(want, Oracle(K, 20000, 15, 30).classify_batch(seq, so, prof, po, nthreads=4))

    packed, pko = pack_bases(seqs)
    codes, co = encode_profiles(profs)
    dev = clf.device
    d_pk = torch.from_numpy(packed).to(dev)
    d_pko = torch.from_numpy(pko).to(dev)
    d_seq = clf.unpack_bases(packed, pko, so)
    assert np.array_equal(d_seq[:so[-1]].cpu().numpy(), seq)
    d_prof = clf.decode_profiles(codes, co, po)
    b = Batch.__new__(Batch)
    b.device, b.nreads, b.total_bases, b.total_kmers = dev, len(seqs), int(so[-1]), int(po[-1])
    b.seq_off_h, b.prof_off_h = so, po
    b.seq, b.prof = d_seq, d_prof
    b.seq_off, b.prof_off = torch.from_numpy(so).to(dev), torch.from_numpy(po).to(dev)
    b.labels = torch.zeros(b.total_bases, dtype=torch.uint8, device=dev)
    clf.run(b)
    clf.check()
    d_out = torch.full((int(pko[-1]),), 0xAA, dtype=torch.uint8, device=dev)
    check(clf.L.cp_pack_labels(b.labels.data_ptr(), b.seq_off.data_ptr(), d_pko.data_ptr(), b.nreads, d_out.data_ptr(), clf._stream()))
    torch.cuda.synchronize()
    got = unpack_labels(d_out.cpu().numpy(), pko, np.diff(so), K)
    assert got.tobytes() == want.tobytes()
    assert int(pko[-1]) * 4 < b.total_bases + 4 * b.nreads                      # 0.25 B/base

    # a batch that cannot be packed: lower case / N / IUPAC -> characters, same labels as the oracle
    s2 = [seqs[0][:3000] + b"n" + seqs[0][3001:], seqs[1].lower(), seqs[2][:100] + b"RYK" + seqs[2][103:]]
    assert pack_bases(s2) is None
    seq2, so2, prof2, po2 = synth.pack_batch(s2, profs[:3])
    got2 = clf.classify(Batch(seq2, so2, prof2, po2))
    assert np.array_equal(got2, Oracle(K, 20000, 15, 30).classify_batch(seq2, so2, prof2, po2, nthreads=2))
    clf.close()


def test_label_runs(built):
    """cp_label_runs + cp_expand_label_runs: the labels of a batch as (end, class) runs equal the painted label string and the
    oracle's, read by read -- long reads, adversarial ones (hundreds of intervals, overflow-free), reads shorter than K,
    a read of exactly K bases; the runs of a read are strictly increasing, neighbouring runs differ in class, and the
    whole batch needs far fewer bytes than the 2-bit form."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from adversarial import adversarial_reads, tail_run_reads
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch, expand_label_runs
    from classpro_amd._lib import ClassProError, check
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=150000, cov=30, read_len=7000, seed=22)
    O = Oracle(K, 20000, 15, 30)
    seqs, profs = list(ds["seqs"]), list(ds["profiles"])
    a_s, a_p = adversarial_reads(31, n=120)
    t_s, t_p = tail_run_reads(32, n=40)
    for s_, p_ in zip(a_s + t_s, a_p + t_p):
        try:
            O.classify_read(s_, p_)
        except OverflowError:
            continue
        seqs.append(s_); profs.append(p_)
    seqs += [b"ACGT" * 9 + b"ACG", b"A" * 40, b"ACGTT" * 8 + b"A", b"C"]
    profs += [np.zeros(0, np.uint16), np.array([5], np.uint16), np.array([3, 3], np.uint16), np.zeros(0, np.uint16)]
    seq, so, prof, po = synth.pack_batch(seqs, profs)
    clf = Classifier(K=K, read_len=20000, hcov=15, dcov=30)
    b = Batch(seq, so, prof, po)
    want = clf.classify(b).copy()
    assert np.array_equal(want, O.classify_batch(seq, so, prof, po, nthreads=4))
    runs = clf.label_runs(Batch(seq, so, prof, po))
    nbytes = 0
    for r, (ends, cls) in enumerate(runs):
        rlen = int(so[r + 1] - so[r])
        assert expand_label_runs(ends, cls, rlen, K) == want[so[r]:so[r + 1]].tobytes(), r
        assert np.all(np.diff(ends) > 0) and np.all(cls[1:] != cls[:-1]), r
        assert (len(ends) == 0) == (rlen < K)
        nbytes += 5 * len(ends) + 12
    assert nbytes * 8 < int(so[-1])                                             # < 0.125 B/base even on this adversarial mix
    # the expander refuses runs that do not cover the read
    ends, cls = runs[0]
    with pytest.raises(ClassProError):
        expand_label_runs(ends[:-1], cls[:-1], int(so[1] - so[0]), K)
    clf.close()
