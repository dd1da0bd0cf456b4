"""The oracle (oracle/classpro_oracle.c) against golden vectors produced by the REFERENCE's own code
(oracle/gen_golden.py over oracle/_ref) and, when oracle/_ref is present, against that build live."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.oracle import Oracle, Ref, INTVL_DTYPE, ref_available


@pytest.fixture(scope="module")
def O():
    return Oracle(40, 20000, 20, 40)


def test_primitives_golden(O):
    g = load_golden("prims.npz")
    assert all(O.bessi(int(n), float(x)) == v for n, x, v in zip(g["bess_n"], g["bess_x"], g["bess"]))
    assert all(O.logp_poisson(int(k), int(l)) == v for k, l, v in zip(g["pois_k"], g["pois_l"], g["pois"]))
    assert all(O.logp_skellam(int(k), float(l)) == v for k, l, v in zip(g["sk_k"], g["sk_l"], g["skel"]))
    assert all(O.logp_binom(int(k), int(n), float(p)) == v for k, n, p, v in zip(g["bk"], g["bn"], g["bp"], g["binom"]))
    assert all(O.binom_test_g(int(k), int(n), float(p)) == v for k, n, p, v in zip(g["bk"], g["bn"], g["bp"], g["btest"]))
    assert all(O.logp_trans(int(a), int(b), int(c), int(d), int(e)) == v
               for a, b, c, d, e, v in zip(g["tb"], g["te"], g["tcb"], g["tce"], g["tcov"], g["trans"]))
    assert np.array_equal(O.logfact()[:4096], g["logfact"])


def test_global_covs_golden():
    g = load_golden("prims.npz")["covs"]
    for (h, d), row in zip(((20, 40), (19, 38), (30, 60), (12, 25), (50, 99)), g):
        cov, dr, cmax, hc = Oracle(40, 20000, h, d).scalars()
        assert cov == [int(x) for x in row[:4]] and dr == row[4] and cmax == cov[1] and hc == 0.004


def test_repeat_threshold_limit():
    with pytest.raises(ValueError):
        Oracle(40, 20000, 100, 200)          # R = 200 + 5*sqrt(200) > 255: the reference exits (wall.c:174)


def test_context_golden(O):
    g = load_golden("context.npz")
    off = g["off"]
    for i in range(len(off) - 1):
        s = g["seq"][off[i]:off[i + 1]].tobytes()
        l, r = O.seq_context(s)
        assert np.array_equal(l, g["lctx"][off[i]:off[i + 1]]), s
        assert np.array_equal(r, g["rctx"][off[i]:off[i + 1]]), s


def test_classify_golden():
    g = load_golden("classify.npz")
    for k in range(int(g["n"])):
        h, d, plen = (int(x) for x in g["meta%d" % k])
        O = Oracle(40, 20000, h, d)
        iv = g["intvl%d" % k].view(INTVL_DTYPE).copy()
        riv = g["rintvl%d" % k].view(INTVL_DTYPE).copy()
        ro, io, fw, bw = O.classify_rel(riv, iv, plen)
        assert np.array_equal(ro["asgn"], g["rel_rasgn%d" % k])
        assert np.array_equal(io["asgn"], g["rel_iasgn%d" % k])
        assert np.array_equal(fw, g["fw%d" % k]) and np.array_equal(bw, g["bw%d" % k])
        assert np.array_equal(O.classify_unrel(io)["asgn"], g["all_iasgn%d" % k])


def test_fastk_golden(O):
    g = load_golden("fastk.npz")
    rc, h, d = O.hist_covs(g["hist"], int(g["low"]), int(g["high"]), int(g["ilow"]), int(g["ihigh"]), 0)
    assert rc == 0 and (h, d) == (int(g["covs"][0]), int(g["covs"][1]))
    rc, h, d = O.hist_covs(g["hist"], int(g["low"]), int(g["high"]), int(g["ilow"]), int(g["ihigh"]), int(g["cov_opt"]))
    assert (h, d) == (int(g["covs"][2]), int(g["covs"][3]))
    po, co = g["prof_off"], g["code_off"]
    for i in range(len(po) - 1):
        n, out = O.decode_profile(g["codes"][co[i]:co[i + 1]].tobytes())
        assert n == po[i + 1] - po[i] and np.array_equal(out, g["prof"][po[i]:po[i + 1]])


def test_hist_no_peak(O):
    h = np.zeros(32767, np.int64)
    h[0] = 1000
    h[4] = 50
    rc, _, _ = O.hist_covs(h, 1, 32767, 0, 0, 0)
    assert rc == 1                            # "Could not find any peak count >= 10" (hist.c:66)


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no reference tree)")
def test_oracle_vs_reference_live(small_ds):
    """Randomised live comparison against the reference's own compiled code."""
    rng = np.random.default_rng(7)
    O, R = Oracle(40, 20000, 20, 40), Ref(20000, 20, 40)
    for _ in range(3000):
        n, x = int(rng.integers(0, 80)), float(rng.uniform(0, 150))
        assert O.bessi(n, x) == R.bessi(n, x)
        nn = int(rng.integers(1, 400)); k = int(rng.integers(0, nn + 1)); pe = float(rng.choice([0.004, 0.1, 0.05, 0.99]))
        assert O.binom_test_g(k, nn, pe) == R.binom_test_g(k, nn, pe)
    for s, p in list(zip(small_ds["seqs"], small_ds["profiles"]))[:120]:
        lo, ro_ = O.seq_context(s)
        lr, rr = R.seq_context(s)
        assert np.array_equal(lo, lr) and np.array_equal(ro_, rr)
        iv = O.find_wall(p, lo, ro_)
        iv2, riv = O.find_rel_intvl(iv, p, lo, ro_)
        a = O.classify_unrel(O.classify_rel(riv, iv2, len(p))[1])
        b = R.classify(riv, iv2, len(p), stage=2)[1]
        assert np.array_equal(a["asgn"], b["asgn"])
