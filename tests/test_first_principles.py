"""Checks of the restatement of calc_init_thres (wall.c:167-243: parity-unpinned in this image -- it calls load_emodel ->
load_himodel -> GSL, wall.c:9-165, and GSL is absent; find_wall / find_rel_intvl, wall.c:245-1051, ARE pinned against the
compiled reference: tests/test_oracle_wall.py) that do not rest on anyone's reading of the code: mathematics.

calc_init_thres (wall.c:167-244) defines, for an error rate pe and an outer count cout,
    cthres[..][cout][s][SELF]   = the smallest cin with P(X > cin) < PE_THRES[s][e],  X ~ Binomial(cout, pe)
    cthres[..][cout][s][OTHERS] = cout - that cin            (cout / 0 when no cin qualifies)
The tables of the oracle and of the product (cp_host_fill_params through the host harness) are compared with EXACT
integer arithmetic on the binary values of pe and of the thresholds; a table entry may differ from the exact answer
only where the exact tail lies within 1e-9 (relative) of the threshold, i.e. where floating-point summation order
could legitimately decide either way: exactly the entries with pe = 0.1 and cout = 3 or 5 (0.1^3 and 0.1^5 are the
thresholds 1e-3 and 1e-5)."""
import ctypes as C
from fractions import Fraction
from math import comb

import numpy as np
import pytest

THRES = [[1e-3, 5e-2], [1e-5, 1e-5]]                      # const.c:62-63  PE_THRES[INIT|FINAL][SELF|OTHERS]
LMAX = [20, 10, 6]                                        # wall.c:122-143


def exact_first_cin(cout, pe, thr):
    """smallest cin in [0,cout] with P(X > cin) < thr, exactly; also the relative distance of the nearest tail to thr"""
    p = Fraction(pe)
    q = 1 - p
    den = p.denominator ** cout                           # common denominator of all terms: pe is a binary fraction
    pn, qn = p.numerator, p.denominator - p.numerator
    terms = [comb(cout, x) * pn ** x * qn ** (cout - x) for x in range(cout + 1)]
    t = Fraction(thr)
    tail = sum(terms)                                     # P(X >= 0) * den
    first, margin = None, None
    for cin in range(cout + 1):
        tail -= terms[cin]                                # P(X > cin) * den
        lhs, rhs = tail * t.denominator, t.numerator * den
        rel = abs(Fraction(lhs - rhs, rhs)) if rhs else None
        margin = rel if margin is None or rel < margin else margin
        if first is None and lhs < rhs:
            first = cin
    return first, margin


@pytest.mark.parametrize("hd", [(20, 40), (30, 60)])
def test_cthres_tables_equal_exact_binomial_tails(harness, built, hd):
    from oracle.oracle import Oracle
    h, d = hd
    O = Oracle(40, 20000, h, d)
    cov, _, cmax, _ = O.scalars()
    tab = O.cthres()
    harness.hh_params_new.restype = C.c_void_p
    P = harness.hh_params_new(40, 20000, h, d)
    ct = np.ctypeslib.as_array(C.cast(harness.hh_params_cthres(C.c_void_p(P)), C.POINTER(C.c_uint8)), shape=(3, 21, 256, 2, 2)).copy()
    assert np.array_equal(ct, tab)                        # product tables == oracle tables
    near = []
    for t in range(3):
        for l in range(1, LMAX[t] + 1):
            pe = 0.002 * l * l + 0.002                    # wall.c:142, evaluated in doubles as the reference does
            assert pe == O.pe()[t][l]
            for cout in list(range(1, 12)) + list(range(12, cmax, 7)) + [cmax - 1]:
                for s in range(2):
                    for e in range(2):
                        first, margin = exact_first_cin(cout, pe, THRES[s][e])
                        want = (first if e == 0 else cout - first) if first is not None else (cout if e == 0 else 0)
                        got = int(tab[t, l, cout, s, e])
                        if got != want:
                            assert margin is not None and margin < Fraction(1, 10 ** 9), (t, l, cout, s, e, got, want)
                            near.append((t, l, cout, s, e, got, want))
    # The only entries not decided by the mathematics: l = 7 gives pe = 0.1, and 0.1^3 = 1e-3, 0.1^5 = 1e-5 ARE the
    # thresholds (equal as decimals, 1.5e-16 apart as binary doubles), so `psum < PE_THRES` there is decided by the
    # rounding of exp(log-binomial) in libm.  The oracle and the product use the reference's formula with the same libm.
    assert all(l == 7 and t in (0, 1) and cout in (3, 5) for t, l, cout, s, e, got, want in near), near
    assert len(near) <= 6


def test_error_rate_and_globals_formulas(built):
    """pe[t][l] = 0.002 l^2 + 0.002 (wall.c:142), HC_ERATE = pe[HP][1] (wall.c:180), R = D + floor(5 sqrt(D)) (util.c:9-11,
    ClassPro.c:545), DR_RATIO = 1 + 2 / sqrt(D) (ClassPro.c:548), logfact = running sum of logs (prob.c:14-19)."""
    import math
    from oracle.oracle import Oracle
    for h, d in ((20, 40), (19, 38), (30, 60), (12, 25)):
        O = Oracle(40, 20000, h, d)
        cov, dr, cmax, hc = O.scalars()
        assert cov == [1, d + int(math.sqrt(d) * 5), h, d] and cmax == cov[1]
        assert dr == 1. + 2. * (1. / math.sqrt(d)) and hc == 0.002 * 1 * 1 + 0.002
        lf = O.logfact()
        acc = 0.
        for n in range(1, 2000):
            acc += math.log(n)
            assert lf[n] == acc
