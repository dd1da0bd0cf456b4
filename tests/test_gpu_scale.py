"""BASELINE configs[2] on a real MI355X (`-m gpu`): the synthetic 200 Mbp diploid, 40x HiFi, r=20000 set
(400 000 distinct reads, 8 Gbases, 16 GB of count profiles) generated in HBM by the device synthesiser,
classified by the HIP path in machine-filling sub-batches through the C ABI, and checked by

  * size-independent properties on ALL 8 Gbases: label alphabet, the exact N-prefix of every read, determinism,
    invariance under a different sub-batch split;
  * read-order invariance and a full comparison with the oracle on a sample of >= 200 Mbases drawn from every
    sub-batch (bit-exact: 0 mismatching positions);
  * the synthesiser itself: range regeneration, shard-wise generation == whole-set slices, histogram peaks.
"""
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


def _prefix_ok(torch, lab, seq_off, nreads):
    """every read: 'N' exactly at its first K-1 positions"""
    starts = seq_off[:-1]
    cnt = torch.bincount(lab.long(), minlength=256)
    if int(cnt[ord("N")]) != nreads * (K - 1):
        return False
    if sum(int(cnt[ord(c)]) for c in "NEHDR") != lab.numel():
        return False
    for k in (0, 1, K - 2):
        if not bool((lab[starts + k] == ord("N")).all()):
            return False
    return bool((lab[starts + (K - 1)] != ord("N")).all())


def test_synth_small_sanity(torch_dev):
    torch = torch_dev
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import hist_covs
    ds = DeviceSynth(genome_len=2_000_000, cov=40, read_len=10000, seed=3)
    assert ds.n_reads == 8000
    hc, dc = hist_covs(ds.hist[4], 1, 32767, 0, 0, 0)
    assert 17 <= hc <= 21 and 36 <= dc <= 41
    rd = ds.reads(0, ds.n_reads, truth=True)
    ds.check()
    seq = rd["seq"][:rd["total_bases"]]
    assert set(torch.unique(seq).tolist()) <= set(b"ACGT")
    prof = rd["prof"][:rd["total_kmers"]]
    assert int(prof.min()) >= 1                                   # uint16 payload <= 32767 reads as positive int16
    # a range generated on its own == the same reads inside the whole set; two ranges concatenate
    a = ds.reads(1000, 500)
    so, po = rd["seq_off_h"], rd["prof_off_h"]
    assert torch.equal(a["seq"][:a["total_bases"]], rd["seq"][so[1000]:so[1500]])
    assert torch.equal(a["prof"][:a["total_kmers"]], rd["prof"][po[1000]:po[1500]])
    # same seed, new object: identical
    ds2 = DeviceSynth(genome_len=2_000_000, cov=40, read_len=10000, seed=3)
    b = ds2.reads(1000, 500)
    assert torch.equal(a["seq"], b["seq"]) and torch.equal(a["prof"], b["prof"])
    # reverse-strand reads exist and truth is 0 exactly where the count is the error count of 1 on clean data
    tr = rd["truth"][:rd["total_kmers"]]
    assert bool(((tr == 0) == (prof == 1)).float().mean() > 0.999)


def test_config2_full_size(torch_dev):
    torch = torch_dev
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import Classifier, Batch, hist_covs
    from classpro_amd.shard import plan_shards
    from oracle.oracle import Oracle
    ds = DeviceSynth(genome_len=200_000_000, cov=40, read_len=20000, seed=1)
    assert ds.n_reads == 400_000 and 7.9e9 < ds.total_bases < 8.1e9
    hc, dc = hist_covs(ds.hist[4], 1, 32767, 0, 0, 0)
    assert (hc, dc) == (19, 38)
    clf = Classifier(K=K, read_len=20000, hcov=hc, dcov=dc)
    plan = ds.plan_batches(800_000_000)
    assert len(plan) == 10
    batches = []
    for a, n in plan:
        rd = ds.reads(a, n)
        batches.append((rd, Batch.from_device(rd)))
    ds.check()
    for _, b in batches:
        clf.run(b)
        clf.check()
    # ---- properties on all 8 Gbases ----
    for rd, b in batches:
        assert _prefix_ok(torch, b.labels[:b.total_bases], b.seq_off, b.nreads)
    # determinism: a second pass gives the same bytes
    rd3, b3 = batches[3]
    first = b3.labels.clone()
    clf.run(b3)
    clf.check()
    assert torch.equal(first, b3.labels)
    # a different split: the last 3000 reads of sub-batch 3 + the first 2000 of sub-batch 4 as one batch
    a3, n3 = plan[3]
    mid = ds.reads(a3 + n3 - 3000, 5000)
    bm = Batch.from_device(mid)
    clf.run(bm)
    clf.check()
    s3, s4 = batches[3][0]["seq_off_h"], batches[4][0]["seq_off_h"]
    want_mid = torch.cat([batches[3][1].labels[s3[n3 - 3000]:s3[n3]], batches[4][1].labels[:s4[2000]]])
    assert torch.equal(bm.labels[:bm.total_bases], want_mid)
    # shards of a 2-rank strong-scaling run regenerate exactly their slice of the whole set
    bounds = plan_shards(ds.seq_off_all, 2)
    assert abs(ds.seq_off_all[bounds[1]] - ds.total_bases / 2) < 60000
    sh = ds.reads(bounds[1], 1000)
    j = next(i for i, (a, n) in enumerate(plan) if a <= bounds[1] < a + n)
    o = bounds[1] - plan[j][0]
    sj = batches[j][0]["seq_off_h"]
    if o + 1000 <= plan[j][1]:
        assert torch.equal(sh["seq"][:sh["total_bases"]], batches[j][0]["seq"][sj[o]:sj[o + 1000]])

    # ---- oracle on a sample from every sub-batch: 10 000 reads of the first, 1 500 of each other (~470 Mbases) ----
    O = Oracle(K, 20000, hc, dc)
    total = bad = 0
    host0 = None
    for i, (rd, b) in enumerate(batches):
        ns = 10000 if i == 0 else 1500
        so, po = rd["seq_off_h"][:ns + 1], rd["prof_off_h"][:ns + 1]
        seq = rd["seq"][:so[-1]].cpu().numpy()
        prof = rd["prof"][:po[-1]].cpu().numpy().view(np.uint16)
        want = O.classify_batch(seq, so, prof, po, nthreads=16)
        got = b.labels[:so[-1]].cpu().numpy()
        bad += int((got != want).sum())
        total += int(so[-1])
        if i == 0:
            host0 = (seq, so, prof, po, got)
    assert total >= 200_000_000
    assert bad == 0, "%d mismatching positions of %d" % (bad, total)

    # ---- read-order invariance: 2000 reads of the sample in reversed order ----
    seq, so, prof, po, got = host0
    order = np.arange(2000)[::-1]
    rs = [seq[so[i]:so[i + 1]].tobytes() for i in order]
    rp = [prof[po[i]:po[i + 1]] for i in order]
    lr = clf.classify(Batch.from_reads(rs, rp))
    o2 = np.concatenate([[0], np.cumsum([len(x) for x in rs])])
    for j, i in enumerate(order):
        assert np.array_equal(lr[o2[j]:o2[j + 1]], got[so[i]:so[i + 1]])
    clf.close()


def test_interval_invariants_at_bench_scale(torch_dev):
    """What the later phases of the reference assume about find_wall's output (and assert under `#define DEBUG`,
    ClassPro.h:17): on every read of a 200-Mbase batch the intervals tile [0,plen) in order, carry the counts at their
    ends, the reliable ones satisfy find_rel_intvl's filters (wall.c:1016-1051), probabilities are logs of numbers <= 1,
    N <= 2*candidates+3 (the scratch bound), and the label string is the paint of the classes (ClassPro.c:265-271)."""
    torch = torch_dev
    import ctypes as C
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import Classifier, Batch, hist_covs, INTVL_DTYPE
    from classpro_amd._lib import check
    ds = DeviceSynth(genome_len=5_000_000, cov=40, read_len=20000, seed=2)
    hc, dc = hist_covs(ds.hist[4], 1, 32767, 0, 0, 0)
    clf = Classifier(K=K, read_len=20000, hcov=hc, dcov=dc)
    rd = ds.reads(0, ds.n_reads)
    b = Batch.from_device(rd)
    lab = clf.classify(b)
    R = clf.export()["cov"][1]
    nc, ni, nr, off = clf.counts(b)
    tot = int(off[-1])
    iv = np.zeros(tot, INTVL_DTYPE)
    rv = np.zeros(tot, INTVL_DTYPE)
    check(clf.L.cp_get_intervals(clf.ws, iv.ctypes.data, rv.ctypes.data, tot))
    n = b.nreads
    assert np.all(ni >= 1) and np.all(ni <= 2 * nc + 3)
    pos = np.arange(tot)
    rid = np.searchsorted(off, pos, side="right") - 1
    k = pos - off[rid]                                    # interval number inside its read
    live = k < ni[rid]
    I, rid, k = iv[live], rid[live], k[live]
    plen = np.diff(b.prof_off_h)[rid]
    first, last = k == 0, k == ni[rid] - 1
    assert np.all(I["b"] < I["e"])
    assert np.all(I["b"][first] == 0) and np.all(I["e"][last] == plen[last])
    assert np.all(I["b"][~first] == I["e"][np.nonzero(~first)[0] - 1])          # consecutive intervals touch
    prof = rd["prof"][:rd["total_kmers"]].cpu().numpy().view(np.uint16)
    po, so = b.prof_off_h[rid], b.seq_off_h[rid]
    assert np.array_equal(I["cb"], prof[po + I["b"]]) and np.array_equal(I["ce"], prof[po + I["e"] - 1])
    assert np.all((I["asgn"] >= 0) & (I["asgn"] <= 3))
    stoc = np.frombuffer(b"ERHD", np.uint8)
    for where in (I["b"], I["e"] - 1, (I["b"] + I["e"]) // 2):
        assert np.array_equal(lab[so + K - 1 + where], stoc[I["asgn"]])
    for f in ("pe", "peo_b", "peo_e"):
        v = I[f]
        assert np.all((v <= 1e-12) | np.isneginf(v))
    rel = I["is_rel"] == 1
    assert rel.sum() == nr.sum() and rel.any()
    assert np.all((I["e"] - I["b"])[rel] >= K)
    assert np.all(np.maximum(I["cb"], I["ce"])[rel] < R)
    assert np.all(I["pe"][rel] < np.log(1e-5))
    assert np.all(np.maximum(I["ccb"], I["cce"])[rel] != 32767)
    # labels change only at interval boundaries
    flat_change = np.nonzero(lab[1:] != lab[:-1])[0] + 1
    bounds = set((so + K - 1 + I["b"]).tolist()) | set(b.seq_off_h.tolist()) | set((b.seq_off_h[:-1] + K - 1).tolist())
    assert set(flat_change.tolist()) <= bounds
    clf.close()
