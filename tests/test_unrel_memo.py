"""The re-evaluation rule of classify_unrel's second sweep (k_classify_unrel_grp) is exact.  CPU.

class_unrel.c:239-274 sweeps the non-fixed intervals twice.  update_state(idx) (class_unrel.c:192-236) reads the constant
fields of interval idx, "asgn == H" / "asgn == D" of its two neighbours (:126,:145) and the nearest reliable-H / reliable-D
interval on either side (find_nn_u, :11-25) -- not the interval's own class.  The device kernel therefore skips, in the
second sweep, every interval none of whose inputs changed since the first sweep evaluated it.  tests/host_harness.cpp
(hh_unrel_memo) runs the reference's two plain sweeps (the product's scalar cp_update_state, which tests/test_oracle_wall.py
holds against the reference's labels) and the two sweeps with the rule on the same records: the classes must be the same
for every read, whatever the read.
"""
import ctypes as C

import numpy as np
import pytest

from adversarial import adversarial_reads, tail_run_reads
from classpro_amd import synth
from oracle.oracle import INTVL_DTYPE
from test_host_logic import run_harness_read


def memo_on_reads(H, P, reads):
    tot = np.zeros(4, np.int64)
    n = 0
    for s, p in reads:
        N, lab, iv, *_ = run_harness_read(H, P, s, p)
        if N <= 0:
            continue
        pre = np.zeros(N, INTVL_DTYPE)
        assert H.hh_last_pre_unrel(pre.ctypes.data_as(C.c_void_p), N) == N
        st = np.zeros(4, np.int64)
        bad = H.hh_unrel_memo(C.c_void_p(P), pre.ctypes.data_as(C.c_void_p), N, st.ctypes.data_as(C.c_void_p))
        assert bad == 0
        assert np.array_equal(pre["asgn"], iv["asgn"])         # and both are what the whole path gave
        tot += st
        n += 1
    return n, tot


@pytest.mark.parametrize("K,rl,h,d,seed", [(40, 20000, 20, 40, 11), (40, 20000, 30, 60, 12), (21, 20000, 20, 40, 13),
                                           (63, 25000, 15, 30, 14), (25, 2000, 19, 38, 15)])
def test_second_sweep_rule_is_exact(harness, K, rl, h, d, seed):
    P = harness.hh_params_new(K, rl, h, d)
    ds = synth.make_dataset(genome_len=60000, cov=d, read_len=8000, K=K, seed=seed, het=0.004, n_repeats=6)
    reads = list(zip(ds["seqs"], ds["profiles"]))[:40]
    reads += list(zip(*adversarial_reads(seed, n=120, K=K))) + list(zip(*tail_run_reads(seed + 3, n=60, K=K)))
    n, tot = memo_on_reads(harness, P, reads)
    assert n > 100 and tot[0] > 1000
    assert tot[1] > 0                                          # the rule does skip something
    harness.hh_params_free(C.c_void_p(P))


def test_rule_statistics_on_bench_like_reads(harness, capsys):
    """How much of the second sweep the rule skips on reads like the bench's (40x, r = 20 000): reported, and at least a third."""
    P = harness.hh_params_new(40, 20000, 19, 38)
    ds = synth.make_dataset(genome_len=300000, cov=40, read_len=20000, K=40, seed=5)
    n, tot = memo_on_reads(harness, P, list(zip(ds["seqs"], ds["profiles"]))[:60])
    with capsys.disabled():
        print(f"\n  second sweep: {tot[0]} evaluations on {n} reads, {tot[1]} skipped by the rule ({100.*tot[1]/tot[0]:.1f} %); "
              f"class changes: {tot[2]} in sweep 1, {tot[3]} in sweep 2")
    assert 3*tot[1] >= tot[0]
    harness.hh_params_free(C.c_void_p(P))
