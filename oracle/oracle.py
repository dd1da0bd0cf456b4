"""ctypes bindings for the CPU oracle and the partial reference build.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (classpro_amd/) never does.

  Oracle  -> oracle/libclasspro_oracle.so   (restatement, oracle/classpro_oracle.c)
  Ref     -> oracle/_ref/libclasspro_ref.so (the reference's own GSL-free sources, oracle/ref_driver.c)
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

INTVL_DTYPE = np.dtype({
    "names":   ["b", "e", "cb", "ce", "ccb", "cce", "is_rel", "asgn", "pe", "peo_b", "peo_e"],
    "formats": ["<i4", "<i4", "<u2", "<u2", "<u2", "<u2", "u1", "i1", "<f8", "<f8", "<f8"],
    "offsets": [0, 4, 8, 10, 12, 14, 16, 17, 24, 32, 40],
    "itemsize": 48,
})


# CP_SANITIZE=1 (scripts/sanitize.sh): the restatement is built with AddressSanitizer + UBSan into its own file and
# that file is the one loaded (the process must then run with libasan preloaded).
_SAN = os.environ.get("CP_SANITIZE") == "1"
_SO = "libclasspro_oracle_san.so" if _SAN else "libclasspro_oracle.so"


def _stale(out, deps, force):
    """Content hash of the sources kept beside the output (file times say nothing on a copied snapshot)."""
    import hashlib
    h = hashlib.sha1(os.path.basename(out).encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    key, side = h.hexdigest(), out + ".srchash"
    if not force and os.path.exists(out) and os.path.exists(side) and open(side).read().strip() == key:
        return None
    return side, key


def build(force=False):
    """Compile the oracle (and _ref when /root/reference is present).  Building is not using."""
    so = os.path.join(_HERE, _SO)
    srcs = [os.path.join(_HERE, f) for f in ("classpro_oracle.c", "classpro_oracle_seed.c", "classpro_oracle.h", "Makefile")]
    st = _stale(so, srcs, force)
    if st:
        subprocess.check_call(["make", "-B", "-C", _HERE, _SO], stdout=subprocess.DEVNULL)
        open(st[0], "w").write(st[1] + "\n")
    ref = os.path.join(_HERE, "_ref", "libclasspro_ref.so")
    drv = os.path.join(_HERE, "ref_driver.c")
    if os.path.exists("/root/reference/src/ClassPro.h"):
        st = _stale(ref, [drv, os.path.join(_HERE, "Makefile")], force)
        if st:
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
            open(st[0], "w").write(st[1] + "\n")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, K=40, read_len=20000, hcov=20, dcov=40, model=None):
        build()
        L = C.CDLL(os.path.join(_HERE, _SO))
        self.L = L
        L.cpo_params_new.restype = C.c_void_p
        L.cpo_params_new.argtypes = [C.c_int] * 4
        L.cpo_params_free.argtypes = [C.c_void_p]
        for f in ("cpo_params_cthres", "cpo_params_logfact", "cpo_params_pe", "cpo_params_lmax"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.cpo_bessi.restype = C.c_double
        L.cpo_bessi.argtypes = [C.c_int, C.c_double]
        L.cpo_logp_poisson.restype = C.c_double
        L.cpo_logp_poisson.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cpo_logp_skellam.restype = C.c_double
        L.cpo_logp_skellam.argtypes = [C.c_int, C.c_double]
        L.cpo_logp_binom.restype = C.c_double
        L.cpo_logp_binom.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.cpo_binom_test_g.restype = C.c_double
        L.cpo_binom_test_g.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.cpo_logp_trans.restype = C.c_double
        L.cpo_logp_trans.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.cpo_classify_read.restype = C.c_int
        L.cpo_find_wall.restype = C.c_int
        L.cpo_find_rel_intvl.restype = C.c_int
        L.cpo_decode_profile.restype = C.c_int
        L.cpo_hist_covs.restype = C.c_int
        self.K, self.read_len = K, read_len
        if model is None:
            self.p = L.cpo_params_new(K, read_len, hcov, dcov)
        else:
            pe = np.zeros(63, np.float64)
            L.cpo_load_himodel.argtypes = [C.c_char_p, C.c_void_p]
            if L.cpo_load_himodel(model.encode(), pe.ctypes.data) != 0:
                raise IOError("cannot load error model %s" % model)
            L.cpo_params_new_model.restype = C.c_void_p
            L.cpo_params_new_model.argtypes = [C.c_int] * 4 + [C.c_void_p]
            self.p = L.cpo_params_new_model(K, read_len, hcov, dcov, pe.ctypes.data)
            self.model_pe = pe.reshape(3, 21)
        if not self.p:
            raise ValueError("REPEAT coverage > 255 (reference exits, wall.c:174)")

    def __del__(self):
        try:
            self.L.cpo_params_free(self.p)
        except Exception:
            pass

    # ---- tables / scalars ----
    def cthres(self):
        ptr = self.L.cpo_params_cthres(self.p)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(3, 21, 256, 2, 2)).copy()

    def logfact(self):
        ptr = self.L.cpo_params_logfact(self.p)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(32768,)).copy()

    def pe(self):
        ptr = self.L.cpo_params_pe(self.p)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(3, 21)).copy()

    def lmax(self):
        ptr = self.L.cpo_params_lmax(self.p)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int)), shape=(3,)).copy()

    def scalars(self):
        cov = (C.c_int * 4)()
        dr, cmax, hc = C.c_double(), C.c_int(), C.c_double()
        self.L.cpo_params_scalars(C.c_void_p(self.p), cov, C.byref(dr), C.byref(cmax), C.byref(hc))
        return list(cov), dr.value, cmax.value, hc.value

    # ---- primitives ----
    def bessi(self, n, x): return self.L.cpo_bessi(n, x)
    def logp_poisson(self, k, lam): return self.L.cpo_logp_poisson(self.p, k, lam)
    def logp_skellam(self, k, lam): return self.L.cpo_logp_skellam(k, lam)
    def logp_binom(self, k, n, pr): return self.L.cpo_logp_binom(self.p, k, n, pr)
    def binom_test_g(self, k, n, pe, exact=0): return self.L.cpo_binom_test_g(self.p, k, n, pe, exact)
    def logp_trans(self, b, e, cb, ce, cov): return self.L.cpo_logp_trans(self.p, b, e, cb, ce, cov)

    def hist_covs(self, hist, low, high, ilow, ihigh, coverage=0):
        hist = np.ascontiguousarray(hist, dtype=np.int64)
        h, d = C.c_int(), C.c_int()
        rc = self.L.cpo_hist_covs(_p(hist, C.c_int64), C.c_int(low), C.c_int(high), C.c_int64(ilow),
                                  C.c_int64(ihigh), C.c_int(coverage), C.byref(h), C.byref(d))
        return rc, h.value, d.value

    def decode_profile(self, code, cap=60000):
        code = np.ascontiguousarray(np.frombuffer(bytes(code), dtype=np.uint8))
        out = np.zeros(cap, dtype=np.uint16)
        n = self.L.cpo_decode_profile(_p(code, C.c_uint8), C.c_int64(len(code)), _p(out, C.c_uint16), C.c_int(cap))
        return n, out[:min(n, cap)]

    # ---- stages ----
    def seq_context(self, seq):
        s = np.frombuffer(seq if isinstance(seq, bytes) else seq.encode(), dtype=np.uint8)
        rlen = len(s)
        l = np.zeros((rlen, 3), np.uint8)
        r = np.zeros((rlen, 3), np.uint8)
        self.L.cpo_seq_context(_p(s, C.c_char), C.c_int(rlen), _p(l, C.c_uint8), _p(r, C.c_uint8))
        return l, r

    def find_wall(self, profile, lctx, rctx):
        profile = np.ascontiguousarray(profile, np.uint16)
        plen = len(profile)
        out = np.zeros(plen + 2, INTVL_DTYPE)
        n = self.L.cpo_find_wall(C.c_void_p(self.p), _p(profile, C.c_uint16), C.c_int(plen),
                                 _p(lctx, C.c_uint8), _p(rctx, C.c_uint8), out.ctypes.data_as(C.c_void_p),
                                 C.c_int(plen + 2))
        if n < 0:
            raise RuntimeError("E-interval overflow")
        return out[:n].copy()

    def find_rel_intvl(self, intvl, profile, lctx, rctx):
        profile = np.ascontiguousarray(profile, np.uint16)
        intvl = intvl.copy()
        r = np.zeros(len(intvl) + 1, INTVL_DTYPE)
        m = self.L.cpo_find_rel_intvl(C.c_void_p(self.p), intvl.ctypes.data_as(C.c_void_p), C.c_int(len(intvl)),
                                      r.ctypes.data_as(C.c_void_p), _p(profile, C.c_uint16), C.c_int(len(profile)),
                                      _p(lctx, C.c_uint8), _p(rctx, C.c_uint8))
        return intvl, r[:m].copy()

    def classify_rel(self, rintvl, intvl, plen):
        rintvl, intvl = rintvl.copy(), intvl.copy()
        M = len(rintvl)
        fw = np.full(max(M, 1), -1, np.int8)
        bw = np.full(max(M, 1), -1, np.int8)
        self.L.cpo_classify_rel(C.c_void_p(self.p), rintvl.ctypes.data_as(C.c_void_p), C.c_int(M),
                                intvl.ctypes.data_as(C.c_void_p), C.c_int(len(intvl)), C.c_int(plen),
                                _p(fw, C.c_int8), _p(bw, C.c_int8))
        return rintvl, intvl, fw[:M], bw[:M]

    def classify_unrel(self, intvl):
        intvl = intvl.copy()
        self.L.cpo_classify_unrel(C.c_void_p(self.p), intvl.ctypes.data_as(C.c_void_p), C.c_int(len(intvl)))
        return intvl

    def classify_read(self, seq, profile, want_intvl=False):
        s = np.frombuffer(seq if isinstance(seq, bytes) else seq.encode(), dtype=np.uint8)
        profile = np.ascontiguousarray(profile, np.uint16)
        rlen = len(s)
        labels = np.zeros(rlen, np.uint8)
        cap = max(rlen, 1) + 2
        iv = np.zeros(cap, INTVL_DTYPE)
        M = C.c_int()
        n = self.L.cpo_classify_read(C.c_void_p(self.p), _p(s, C.c_char), C.c_int(rlen), _p(profile, C.c_uint16),
                                     _p(labels, C.c_char), iv.ctypes.data_as(C.c_void_p), C.c_int(cap), C.byref(M))
        if n < 0:
            raise OverflowError("# E-intvls >= plen: the reference exits on this read (wall.c:783-788)")
        lab = labels.tobytes()
        return (lab, iv[:n].copy(), M.value) if want_intvl else lab

    def find_seeds(self, seq, labels, profile):
        """seed.c:966-1032 for one read: labels = the read's label string ('N'*(K-1) + E/H/D/R per k-mer).
        Returns (sasgn uint8[plen] of 'E'/'H'/'D'/'R', rep int32[n,2] in read coordinates)."""
        s = np.frombuffer(seq if isinstance(seq, bytes) else seq.encode(), dtype=np.uint8)
        lab = np.frombuffer(labels if isinstance(labels, bytes) else bytes(labels), dtype=np.uint8)
        profile = np.ascontiguousarray(profile, np.uint16)
        plen = len(profile)
        cls = np.ascontiguousarray(lab[self.K - 1:])
        sas = np.zeros(max(plen, 1), np.int32)
        rep = np.zeros((plen + 2, 2), np.int32)
        self.L.cpo_find_seeds.restype = C.c_int
        n = self.L.cpo_find_seeds(_p(s, C.c_char), _p(cls, C.c_char), _p(profile, C.c_uint16), C.c_int(plen), C.c_int(self.K),
                                  _p(sas, C.c_int), _p(rep, C.c_int), C.c_int(plen + 2))
        return sas[:plen].astype(np.uint8), rep[:n].copy()

    def kmer_hash(self, seq, K=None):
        K = K or self.K
        s = np.frombuffer(seq if isinstance(seq, bytes) else seq.encode(), dtype=np.uint8)
        plen = len(s) - K + 1
        h = np.zeros(max(plen, 1), np.int32)
        self.L.cpo_kmer_hash(_p(s, C.c_char), C.c_int(plen), C.c_int(K), _p(h, C.c_int))
        return h[:plen]

    def classify_batch(self, seq, seq_off, prof, prof_off, nthreads=1):
        seq = np.ascontiguousarray(seq, np.uint8)
        prof = np.ascontiguousarray(prof, np.uint16)
        seq_off = np.ascontiguousarray(seq_off, np.int64)
        prof_off = np.ascontiguousarray(prof_off, np.int64)
        labels = np.zeros(len(seq), np.uint8)
        self.L.cpo_classify_batch(C.c_void_p(self.p), _p(seq, C.c_char), _p(seq_off, C.c_int64),
                                  _p(prof, C.c_uint16), _p(prof_off, C.c_int64), C.c_int(len(seq_off) - 1),
                                  _p(labels, C.c_char), C.c_int(nthreads))
        return labels


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libclasspro_ref.so"))


def ref_wall_available():
    """True when oracle/_ref holds the GSL-free part of the reference's wall.c (find_wall, find_rel_intvl)."""
    if not ref_available():
        return False
    build()
    L = C.CDLL(os.path.join(_HERE, "_ref", "libclasspro_ref.so"))
    return hasattr(L, "ref_have_wall") and L.ref_have_wall() == 1


class Ref:
    """The reference's own code (GSL-free files) via oracle/_ref/libclasspro_ref.so."""

    def __init__(self, read_len=20000, hcov=20, dcov=40):
        build()
        L = C.CDLL(os.path.join(_HERE, "_ref", "libclasspro_ref.so"))
        self.L = L
        for f, at in (("ref_bessi", [C.c_int, C.c_double]), ("ref_logp_poisson", [C.c_int, C.c_int]),
                      ("ref_logp_skellam", [C.c_int, C.c_double]), ("ref_logp_binom", [C.c_int, C.c_int, C.c_double]),
                      ("ref_binom_test_g", [C.c_int, C.c_int, C.c_double, C.c_int]),
                      ("ref_logp_trans", [C.c_int] * 5), ("ref_p_errorin", [C.c_int, C.c_double, C.c_int, C.c_int])):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = at
        L.ref_logfact.restype = C.c_void_p
        L.ref_open_profiles.restype = C.c_void_p
        L.ref_open_profiles.argtypes = [C.c_char_p]
        L.ref_free_profiles.argtypes = [C.c_void_p]
        L.ref_profiles_nreads.argtypes = [C.c_void_p]
        L.ref_profiles_kmer.argtypes = [C.c_void_p]
        L.ref_fetch_profile.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_uint16)]
        self._setup = (read_len, hcov, dcov)
        self._wall = None
        L.ref_setup(read_len, hcov, dcov)
        Ref._current = self

    _current = None

    def _activate(self):
        """The reference keeps its parameters in process-wide globals: make this instance's the live ones."""
        if Ref._current is not self:
            self.L.ref_setup(*self._setup)
            if self._wall is not None:
                self._wall_apply()
            Ref._current = self

    # ---- wall.c:245-1051 (GSL-free part): find_wall, find_rel_intvl, the whole per-read path ----
    def wall_setup(self, cthres, pe, lmax, cmax, hc_erate):
        """Fill the Error_Model find_wall takes as a parameter (what calc_init_thres, wall.c:167-243, would leave):
        cthres uint8[3][21][256][2][2], pe float64[3][21], lmax int[3], CMAX, HC_ERATE."""
        self._wall = (np.ascontiguousarray(cthres, np.uint8).copy(), np.ascontiguousarray(pe, np.float64).copy(),
                      np.ascontiguousarray(lmax, np.int32).copy(), int(cmax), float(hc_erate))
        self._wall_apply()
        return self

    def wall_setup_from(self, O):
        """Tables of an Oracle built for the same (read_len, H, D) -- the ones tests/test_first_principles.py checks
        entry by entry with exact integer arithmetic."""
        cov, _, cmax, hc = O.scalars()
        assert (cov[2], cov[3]) == self._setup[1:] and O.read_len == self._setup[0]
        return self.wall_setup(O.cthres(), O.pe(), O.lmax(), cmax, hc)

    def _wall_apply(self):
        ct, pe, lm, cmax, hc = self._wall
        self.L.ref_wall_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        self.L.ref_wall_setup(ct.ctypes.data, pe.ctypes.data, lm.ctypes.data, cmax, hc)

    def find_wall_rel(self, seq, profile, K=40):
        """find_wall + find_rel_intvl on fresh, zeroed buffers: (intvl[N] after find_rel_intvl, rintvl[M])."""
        self._activate()
        s = (seq if isinstance(seq, bytes) else seq.encode()) + b"\0"
        rlen = len(s) - 1
        profile = np.ascontiguousarray(profile, np.uint16)
        assert len(profile) == rlen - K + 1
        cap = len(profile) + 2
        iv = np.zeros(cap, INTVL_DTYPE)
        rv = np.zeros(cap, INTVL_DTYPE)
        M = C.c_int()
        self.L.ref_find_wall_rel.restype = C.c_int
        n = self.L.ref_find_wall_rel(C.c_char_p(s), C.c_int(rlen), _p(profile, C.c_uint16), C.c_int(K),
                                     iv.ctypes.data_as(C.c_void_p), C.c_int(cap), C.byref(M),
                                     rv.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise RuntimeError("ref_find_wall_rel: %d" % n)
        return iv[:n].copy(), rv[:M.value].copy()

    def find_wall_exit_status(self, seq, profile, K=40):
        """The same call in a child process: 0 = returned, 1 = the reference's own exit(1) ('# E-intvls >= plen')."""
        self._activate()
        s = (seq if isinstance(seq, bytes) else seq.encode()) + b"\0"
        profile = np.ascontiguousarray(profile, np.uint16)
        self.L.ref_find_wall_exit_status.restype = C.c_int
        return self.L.ref_find_wall_exit_status(C.c_char_p(s), C.c_int(len(s) - 1), _p(profile, C.c_uint16), C.c_int(K))

    def classify_read(self, seq, profile, K=40, want_intvl=False):
        """The whole per-read path in reference text only (context.c -> wall.c slice -> class_rel.c -> class_unrel.c ->
        paint, ClassPro.c:229-271) on fresh buffers: the label string."""
        self._activate()
        s = (seq if isinstance(seq, bytes) else seq.encode()) + b"\0"
        rlen = len(s) - 1
        profile = np.ascontiguousarray(profile, np.uint16)
        lab = np.zeros(max(rlen, 1), np.uint8)
        cap = max(rlen, 1) + 2
        iv = np.zeros(cap, INTVL_DTYPE)
        self.L.ref_classify_read.restype = C.c_int
        n = self.L.ref_classify_read(C.c_char_p(s), C.c_int(rlen), _p(profile, C.c_uint16), C.c_int(K), _p(lab, C.c_char),
                                     iv.ctypes.data_as(C.c_void_p), C.c_int(cap))
        if n < 0:
            raise RuntimeError("ref_classify_read: %d" % n)
        out = lab[:rlen].tobytes()
        return (out, iv[:n].copy()) if want_intvl else out

    def classify_batch(self, seq, seq_off, prof, prof_off, K=40, nthreads=1, rlen_max=0, defined=True):
        """The reference's thread loop without its I/O (ref_classify_batch): labels, (alloc seconds, run seconds)."""
        self._activate()
        seq = np.ascontiguousarray(seq, np.uint8)
        prof = np.ascontiguousarray(prof, np.uint16)
        seq_off = np.ascontiguousarray(seq_off, np.int64)
        prof_off = np.ascontiguousarray(prof_off, np.int64)
        labels = np.zeros(len(seq), np.uint8)
        sec = np.zeros(2, np.float64)
        self.L.ref_classify_batch.restype = C.c_int
        rc = self.L.ref_classify_batch(_p(seq, C.c_char), _p(seq_off, C.c_longlong), _p(prof, C.c_uint16),
                                       _p(prof_off, C.c_longlong), C.c_int(len(seq_off) - 1), C.c_int(K),
                                       _p(labels, C.c_char), C.c_int(nthreads), C.c_int(rlen_max),
                                       C.c_int(1 if defined else 0), _p(sec, C.c_double))
        if rc != 0:
            raise RuntimeError("ref_classify_batch: %d" % rc)
        return labels, (float(sec[0]), float(sec[1]))

    def globals(self):
        self._activate()
        cov = (C.c_int * 4)()
        dr = C.c_double()
        self.L.ref_globals(cov, C.byref(dr))
        return list(cov), dr.value

    def logfact(self):
        ptr = self.L.ref_logfact()
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(32768,)).copy()

    def bessi(self, n, x): return self.L.ref_bessi(n, x)
    def logp_poisson(self, k, lam): return self.L.ref_logp_poisson(k, lam)
    def logp_skellam(self, k, lam): return self.L.ref_logp_skellam(k, lam)
    def logp_binom(self, k, n, pr): return self.L.ref_logp_binom(k, n, pr)
    def binom_test_g(self, k, n, pe, exact=0): return self.L.ref_binom_test_g(k, n, pe, exact)
    def logp_trans(self, b, e, cb, ce, cov):
        self._activate()
        return self.L.ref_logp_trans(b, e, cb, ce, cov)

    def p_errorin(self, e, erate, cout, cin): return self.L.ref_p_errorin(e, erate, cout, cin)

    def hist_covs(self, fk_root, coverage=0):
        h, d = C.c_int(), C.c_int()
        self.L.ref_hist_covs(C.c_char_p(fk_root.encode()), C.c_int(coverage), C.byref(h), C.byref(d))
        return h.value, d.value

    def fetch_profiles(self, fk_root, cap=60000):
        P = self.L.ref_open_profiles(fk_root.encode())
        if not P:
            raise IOError("cannot open %s.prof" % fk_root)
        n = self.L.ref_profiles_nreads(C.c_void_p(P))
        out = []
        buf = np.zeros(cap, np.uint16)
        for i in range(n):
            plen = self.L.ref_fetch_profile(C.c_void_p(P), i, cap, _p(buf, C.c_uint16))
            out.append(buf[:plen].copy())
        k = self.L.ref_profiles_kmer(C.c_void_p(P))
        self.L.ref_free_profiles(C.c_void_p(P))
        return k, out

    def seq_context(self, seq):
        s = (seq if isinstance(seq, bytes) else seq.encode()) + b"\0"
        rlen = len(s) - 1
        l = np.zeros((rlen, 3), np.uint8)
        r = np.zeros((rlen, 3), np.uint8)
        self.L.ref_seq_context(C.c_char_p(s), C.c_int(rlen), _p(l, C.c_uint8), _p(r, C.c_uint8))
        return l, r

    def classify(self, rintvl, intvl, plen, stage=2):
        self._activate()
        rintvl, intvl = rintvl.copy(), intvl.copy()
        self.L.ref_classify(rintvl.ctypes.data_as(C.c_void_p), C.c_int(len(rintvl)),
                            intvl.ctypes.data_as(C.c_void_p), C.c_int(len(intvl)), C.c_int(plen), C.c_int(stage))
        return rintvl, intvl

    def find_seeds(self, seq, labels, profile, K=40):
        """The reference's own find_seeds (seed.c:966) through ref_find_seeds: (sasgn, rep pairs, hashes)."""
        s = (seq if isinstance(seq, bytes) else seq.encode()) + b"\0"
        lab = (labels if isinstance(labels, bytes) else bytes(labels))
        cls = lab[K - 1:] + b"\0"
        profile = np.ascontiguousarray(profile, np.uint16)
        plen = len(profile)
        sas = np.zeros(max(plen, 1) + 1, np.int32)
        hsh = np.zeros(max(plen, 1) + 1, np.int32)
        rep = np.zeros((plen + 2, 2), np.int32)
        self.L.ref_find_seeds.restype = C.c_int
        n = self.L.ref_find_seeds(C.c_char_p(s), C.c_char_p(cls), _p(profile, C.c_uint16), C.c_int(plen), C.c_int(K),
                                  _p(sas, C.c_int), _p(hsh, C.c_int), _p(rep, C.c_int), C.c_int(plen + 2))
        return sas[:plen].astype(np.uint8), rep[:n].copy(), hsh[:plen].copy()

    def classify_rel_dir(self, rintvl, plen, forward):
        self._activate()
        rintvl = rintvl.copy()
        M = len(rintvl)
        out = np.zeros(max(M, 1), np.int8)
        hdrr = C.c_double()
        self.L.ref_classify_rel_dir(rintvl.ctypes.data_as(C.c_void_p), C.c_int(M), C.c_int(plen),
                                    C.c_int(1 if forward else 0), _p(out, C.c_int8), C.byref(hdrr))
        return out[:M], hdrr.value
