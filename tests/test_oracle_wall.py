"""find_wall / find_rel_intvl and whole-read labels against the REFERENCE's own text.

tests/golden/wall.npz and labels.npz were produced (oracle/gen_golden.py) by the GSL-free part of the reference's
wall.c -- lines 245-1051, compiled as they stand into oracle/_ref (oracle/Makefile `ref`) -- together with its
context.c, class_rel.c, class_unrel.c: interval records incl. the bits of their doubles, label strings, and the reads on
which the reference's `#define DEBUG` abort fires ("# E-intvls >= plen").  Checked here on the CPU:
  * the oracle (oracle/classpro_oracle.c) against both files;
  * the product's scalar device functions compiled for the host (tests/host_harness.cpp) against labels.npz;
  * live, when oracle/_ref is present: fresh random reads, a negative control, the reference's thread loop.
The -m gpu twin is tests/test_gpu_reference.py.
"""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

from conftest import load_golden
from oracle.oracle import Oracle, Ref, INTVL_DTYPE, ref_wall_available
from adversarial import adversarial_reads, tail_run_reads
from classpro_amd import synth

INT_FIELDS = ("b", "e", "cb", "ce", "ccb", "cce", "is_rel")
DBL_FIELDS = ("pe", "peo_b", "peo_e")
MODEL_GROWTH = (0.0004, 0.0005, 0.0006)      # oracle/gen_golden.py: the -M parameter set's synthetic HIsim model


def golden_model(d):
    p = os.path.join(d, "golden_hifi.model")
    if not os.path.exists(p):
        synth.write_himodel(p, growth=MODEL_GROWTH)
    return p


def same_records(a, b, fields=INT_FIELDS):
    """Integer fields equal and the doubles bit for bit."""
    if len(a) != len(b):
        return False
    ok = all(np.array_equal(a[f], b[f]) for f in fields)
    return ok and all(np.array_equal(a[f].view(np.uint64), b[f].view(np.uint64)) for f in DBL_FIELDS)


def golden_reads(g):
    so, po = g["seq_off"], g["prof_off"]
    for i in range(len(so) - 1):
        yield i, int(g["set"][i]), int(g["status"][i]), g["seq"][so[i]:so[i + 1]].tobytes(), np.ascontiguousarray(g["prof"][po[i]:po[i + 1]])


def oracles_for(g):
    out = []
    td = tempfile.mkdtemp()
    for k, (K, rl, h, d, m) in enumerate(g["psets"]):
        O = Oracle(int(K), int(rl), int(h), int(d), model=(golden_model(td) if m else None))
        # the tables the reference's find_wall was given are the oracle's (and, test_gpu_reference.py, the product's)
        assert np.array_equal(O.cthres(), g["cthres"][k]) and np.array_equal(O.pe(), g["pe"][k])
        assert np.array_equal(O.lmax(), g["lmax"][k]) and O.scalars()[2] == g["cmax"][k] and O.scalars()[3] == g["hc_erate"][k]
        out.append(O)
    return out


def oracle_wall_rel(O, s, p):
    l, r = O.seq_context(s)
    try:
        iv = O.find_wall(p, l, r)
    except RuntimeError:
        return None
    return O.find_rel_intvl(iv, p, l, r)


def test_oracle_against_reference_wall_records():
    g = load_golden("wall.npz")
    Os = oracles_for(g)
    io, ro = g["intvl_off"], g["rintvl_off"]
    iv_all, rv_all = g["intvl"].view(INTVL_DTYPE), g["rintvl"].view(INTVL_DTYPE)
    n_abort = n_iv = 0
    for i, si, status, s, p in golden_reads(g):
        got = oracle_wall_rel(Os[si], s, p)
        if status == 1:                                     # the reference exit(1)s on this read; the oracle must reject it
            assert got is None, i
            n_abort += 1
            continue
        assert got is not None, i
        iv, rv = got
        assert same_records(iv, iv_all[io[i]:io[i + 1]]), i
        assert same_records(rv, rv_all[ro[i]:ro[i + 1]]), i
        n_iv += len(iv)
    assert n_abort >= 10 and n_iv > 30000


def test_oracle_against_reference_labels():
    g = load_golden("labels.npz")
    Os = oracles_for(g)
    lo, ao = g["labels_off"], g["asgn_off"]
    for i, si, status, s, p in golden_reads(g):
        if status == 1:
            with pytest.raises(OverflowError):
                Os[si].classify_read(s, p)
            continue
        lab, iv, _ = Os[si].classify_read(s, p, want_intvl=True)
        assert lab == g["labels"][lo[i]:lo[i + 1]].tobytes(), i
        assert np.array_equal(iv["asgn"], g["asgn"][ao[i]:ao[i + 1]]), i


def test_product_host_functions_against_reference_labels(harness):
    """The product's own scalar code (cp_wall.h, cp_class.h, cp_ctx.h, cp_math.h compiled for the host by
    tests/host_harness.cpp) against the reference's labels and interval classes -- not via the oracle."""
    from test_host_logic import run_harness_read
    g = load_golden("labels.npz")
    lo, ao = g["labels_off"], g["asgn_off"]
    P = {}
    td = tempfile.mkdtemp()
    for i, si, status, s, p in golden_reads(g):
        if si not in P:
            K, rl, h, d, m = (int(x) for x in g["psets"][si])
            harness.hh_params_new_model.restype = C.c_void_p
            P[si] = (harness.hh_params_new_model(K, rl, h, d, golden_model(td).encode()) if m else harness.hh_params_new(K, rl, h, d))
            assert P[si]
        N, lab, iv, *_ = run_harness_read(harness, P[si], s, p)
        if status == 1:
            assert N < 0, i                                 # the product reports the reference's abort
            continue
        assert N >= 0 and lab == g["labels"][lo[i]:lo[i + 1]].tobytes(), i
        assert np.array_equal(iv["asgn"], g["asgn"][ao[i]:ao[i + 1]]), i
    for p_ in P.values():
        harness.hh_params_free(C.c_void_p(p_))


needs_ref = pytest.mark.skipif(not ref_wall_available(), reason="oracle/_ref with the wall.c slice not built (no reference tree)")


@needs_ref
@pytest.mark.parametrize("K,rl,h,d,seed", [(40, 20000, 20, 40, 1001), (40, 2000, 19, 38, 1002), (25, 20000, 30, 60, 1003),
                                           (63, 25000, 15, 30, 1004), (21, 20000, 45, 90, 1005)])
def test_oracle_vs_reference_wall_live(K, rl, h, d, seed):
    """Randomised live comparison with the reference's find_wall / find_rel_intvl and its whole-read labels."""
    O = Oracle(K, rl, h, d)
    R = Ref(rl, h, d).wall_setup_from(O)
    ds = synth.make_dataset(genome_len=50000, cov=d, read_len=6000, K=K, seed=seed, het=0.003, n_repeats=5)
    reads = list(zip(ds["seqs"], ds["profiles"]))[:25]
    reads += list(zip(*adversarial_reads(seed, n=60, K=K))) + list(zip(*tail_run_reads(seed + 7, n=40, K=K)))
    n_abort = 0
    for s, p in reads:
        got = oracle_wall_rel(O, s, p)
        if got is None:
            assert R.find_wall_exit_status(s, p, K) == 1
            n_abort += 1
            continue
        assert R.find_wall_exit_status(s, p, K) == 0 if len(p) < 12 else True
        iv, rv = R.find_wall_rel(s, p, K)
        assert same_records(got[0], iv) and same_records(got[1], rv)
        assert O.classify_read(s, p) == R.classify_read(s, p, K)
    assert n_abort < len(reads) // 4


@needs_ref
def test_negative_control_a_wrong_threshold_is_seen():
    """The comparison bites: the reference side with one corrupted threshold row disagrees on many reads."""
    O = Oracle(40, 20000, 20, 40)
    ct = O.cthres()
    ct[:, :, :, 1, :] = np.maximum(ct[:, :, :, 1, :].astype(np.int32) - 2, 0).astype(np.uint8)   # FINAL thresholds two lower
    cov, _, cmax, hc = O.scalars()
    R = Ref(20000, 20, 40).wall_setup(ct, O.pe(), O.lmax(), cmax, hc)
    S, P = adversarial_reads(3, n=80)
    bad = 0
    for s, p in zip(S, P):
        got = oracle_wall_rel(O, s, p)
        if got is None or R.find_wall_exit_status(s, p, 40) != 0:
            continue
        iv, rv = R.find_wall_rel(s, p, 40)
        bad += not (same_records(got[0], iv) and same_records(got[1], rv))
    assert bad >= 10


@needs_ref
def test_reference_thread_loop_equals_per_read_calls():
    """ref_classify_batch (the reference's functions in its own thread loop, scratch reused across reads with the
    three per-read resets of hazards 1, 2, 8) == the same functions on fresh buffers per read == the oracle's batch;
    any thread count gives the same bytes.  This loop is bench.py's cpu_baseline (kind "reference")."""
    O = Oracle(40, 20000, 20, 40)
    R = Ref(20000, 20, 40).wall_setup_from(O)
    ds = synth.make_dataset(genome_len=60000, cov=40, read_len=7000, seed=77, het=0.002)
    seqs, profs = list(ds["seqs"][:40]), list(ds["profiles"][:40])
    S, P = tail_run_reads(5, n=40)
    A, B = adversarial_reads(6, n=60)
    for s, p in list(zip(S, P)) + list(zip(A, B)):
        if oracle_wall_rel(O, s, p) is not None:              # (an aborting read would end this process)
            seqs.append(s); profs.append(p)
    order = np.random.default_rng(1).permutation(len(seqs))   # long and short reads interleaved: stale scratch would show
    seqs, profs = [seqs[i] for i in order], [profs[i] for i in order]
    seq, so, prof, po = synth.pack_batch(seqs, profs)
    rmax = int(np.diff(so).max()) + 1
    want = b"".join(R.classify_read(s, p, 40) for s, p in zip(seqs, profs))
    for nt in (1, 3):
        lab, (t_alloc, t_run) = R.classify_batch(seq, so, prof, po, 40, nthreads=nt, rlen_max=rmax, defined=True)
        assert lab.tobytes() == want and t_run > 0
    assert O.classify_batch(seq, so, prof, po, nthreads=2).tobytes() == want
