"""Host-side Python mirror of the reference's per-read interface, batched, over the C ABI.

Names follow the reference (ClassPro.c:229-271): calc_seq_context, find_wall, find_rel_intvl,
classify_rel, classify_unrel; `Classifier.classify` is the whole read loop body for a batch.
PyTorch is used only to own device memory and streams; every computation is a HIP kernel in
libclasspro_amd.so.  Nothing here imports oracle/.
"""
import ctypes as C
import numpy as np
import torch

from ._lib import lib, check

STAGE_SCAN, STAGE_WALL, STAGE_REL, STAGE_CLASS_REL, STAGE_CLASS_ALL, STAGE_LABELS = 1, 2, 3, 4, 5, 6

INTVL_DTYPE = np.dtype({
    "names":   ["b", "e", "cb", "ce", "ccb", "cce", "is_rel", "asgn", "pe", "peo_b", "peo_e"],
    "formats": ["<i4", "<i4", "<u2", "<u2", "<u2", "<u2", "u1", "i1", "<f8", "<f8", "<f8"],
    "offsets": [0, 4, 8, 10, 12, 14, 16, 17, 24, 32, 40],
    "itemsize": 48,
})


def hist_covs(hist, low, high, ilowcnt=0, ihighcnt=0, coverage=0):
    """process_global_hist (hist.c:28): (H,D) coverage from a FASTK histogram or from -c."""
    h = np.ascontiguousarray(hist, dtype=np.int64)
    hc, dc = C.c_int(), C.c_int()
    check(lib().cp_hist_covs(h.ctypes.data, low, high, ilowcnt, ihighcnt, coverage, C.byref(hc), C.byref(dc)))
    return hc.value, dc.value


def decode_profile(code, cap=60000):
    """Fetch_Profile's decoder (libfastk.c:1467) for one read's code string."""
    buf = np.frombuffer(bytes(code), dtype=np.uint8)
    out = np.zeros(cap, np.uint16)
    n = check(lib().cp_decode_profile(buf.ctypes.data if len(buf) else None, len(buf), out.ctypes.data, cap))
    return n, out[:min(n, cap)]


def load_error_model(path):
    """-M: pe[t][l] fitted from a HIsim error-model file (wall.c:55-115); double[3][21]."""
    pe = np.zeros((3, 21), np.float64)
    check(lib().cp_load_error_model(path.encode(), pe.ctypes.data))
    return pe


def encode_profiles(profiles):
    """FASTK code strings for a list of count arrays: (uint8 codes, int64 code_off[n+1])."""
    L = lib()
    chunks, off = [], [0]
    for p in profiles:
        p = np.ascontiguousarray(p, np.uint16)
        buf = np.empty(2 * len(p) + 2, np.uint8)
        n = check(L.cp_encode_profile(p.ctypes.data, len(p), buf.ctypes.data, len(buf)))
        chunks.append(buf[:n].copy())
        off.append(off[-1] + n)
    codes = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return codes, np.array(off, np.int64)


def pack_bases(seqs):
    """2-bit packing of a list of reads (bytes) for cp_unpack_bases: (uint8 packed, int64 pack_off[n+1]) or None when a
    read holds a letter other than upper-case A, C, G, T (the batch then travels as characters)."""
    L = lib()
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([(len(x) + 3) // 4 for x in seqs], out=off[1:])
    out = np.zeros(max(int(off[-1]), 1), np.uint8)
    for i, x in enumerate(seqs):
        b = np.frombuffer(bytes(x), np.uint8)
        if len(b) and check(L.cp_pack_bases(b.ctypes.data, len(b), out[off[i]:].ctypes.data)) == 0:
            return None
    return out, off


def unpack_labels(packed, pack_off, rlens, K):
    """cp_unpack_labels for every read of a batch: the concatenated label bytes."""
    L = lib()
    packed = np.ascontiguousarray(packed, np.uint8)
    so = np.zeros(len(rlens) + 1, np.int64)
    np.cumsum(rlens, out=so[1:])
    out = np.zeros(max(int(so[-1]), 1), np.uint8)
    for i, n in enumerate(rlens):
        if n:
            check(L.cp_unpack_labels(packed[pack_off[i]:].ctypes.data, int(n), K, out[so[i]:].ctypes.data))
    return out[:so[-1]]


def math_eval(fn, x, x2=None, device="cuda:0"):
    """cp_math_eval: the device's own exp (fn 0) / log (1) / sqrt (2) / bessi (3, n = x2) / logp_skellam (4, k = x2) on an
    array of doubles; returns a host float64 array."""
    dev = torch.device(device)
    xd = torch.from_numpy(np.ascontiguousarray(x, np.float64)).to(dev)
    x2d = torch.from_numpy(np.ascontiguousarray(x2, np.float64)).to(dev) if x2 is not None else None
    yd = torch.empty_like(xd)
    check(lib().cp_math_eval(fn, xd.data_ptr(), x2d.data_ptr() if x2d is not None else None, yd.data_ptr(), xd.numel(),
                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize(dev)
    return yd.cpu().numpy()


def expand_label_runs(ends, cls, rlen, K):
    """cp_expand_label_runs: one read's label string from its runs."""
    ends = np.ascontiguousarray(ends, np.int32)
    cls = np.ascontiguousarray(cls, np.uint8)
    out = np.zeros(max(rlen, 1), np.uint8)
    check(lib().cp_expand_label_runs(ends.ctypes.data if len(ends) else None, cls.ctypes.data if len(cls) else None, len(ends), rlen, K,
                                     out.ctypes.data))
    return out[:rlen].tobytes()


class Batch:
    """A batch of reads resident in HBM in the flat layout of include/classpro_amd.h."""

    def __init__(self, seq, seq_off, prof, prof_off, device="cuda:0"):
        self.device = torch.device(device)
        self.nreads = len(seq_off) - 1
        self.total_bases = int(seq_off[-1])
        self.total_kmers = int(prof_off[-1])
        self.seq_off_h = np.ascontiguousarray(seq_off, np.int64)
        self.prof_off_h = np.ascontiguousarray(prof_off, np.int64)
        dev = self.device
        self.seq = torch.from_numpy(np.ascontiguousarray(seq, np.uint8)).to(dev)
        self.prof = torch.from_numpy(np.ascontiguousarray(prof, np.uint16).view(np.int16)).to(dev)
        self.seq_off = torch.from_numpy(self.seq_off_h).to(dev)
        self.prof_off = torch.from_numpy(self.prof_off_h).to(dev)
        self.labels = torch.zeros(max(self.total_bases, 1), dtype=torch.uint8, device=dev)

    @classmethod
    def from_device(cls, rd):
        """A batch over tensors that are already in HBM (a `DeviceSynth.reads()` result); nothing is copied."""
        b = cls.__new__(cls)
        b.device = rd["seq"].device
        b.nreads, b.total_bases, b.total_kmers = rd["nreads"], rd["total_bases"], rd["total_kmers"]
        b.seq_off_h, b.prof_off_h = rd["seq_off_h"], rd["prof_off_h"]
        b.seq, b.prof, b.seq_off, b.prof_off = rd["seq"], rd["prof"], rd["seq_off"], rd["prof_off"]
        b.labels = torch.zeros(max(b.total_bases, 1), dtype=torch.uint8, device=b.device)
        return b

    @classmethod
    def from_reads(cls, seqs, profiles, device="cuda:0"):
        from .synth import pack_batch
        return cls(*pack_batch(seqs, profiles), device=device)


class Classifier:
    """Global setup (ClassPro.c:536-554) + batched hot path."""

    def __init__(self, K=40, read_len=20000, hcov=20, dcov=40, device="cuda:0", model=None, pe=None):
        self.L = lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("classpro_amd runs on a HIP device only")
        torch.cuda.set_device(self.device)
        self.K, self.read_len = K, read_len
        p = C.c_void_p()
        if pe is not None:                                  # the error model as a table (cp_params_create_pe)
            pe = np.ascontiguousarray(pe, np.float64).reshape(3, 21)
            check(self.L.cp_params_create_pe(K, read_len, hcov, dcov, pe.ctypes.data, C.byref(p)))
        else:
            check(self.L.cp_params_create_model(K, read_len, hcov, dcov, model.encode() if model else None, C.byref(p)))
        self.p = p
        w = C.c_void_p()
        check(self.L.cp_workspace_create(C.byref(w)))
        self.ws = w

    def close(self):
        if getattr(self, "ws", None):
            self.L.cp_workspace_destroy(self.ws)
            self.ws = None
        if getattr(self, "p", None):
            self.L.cp_params_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables ----
    def export(self):
        cov = (C.c_int * 4)()
        dr, cmax, hc = C.c_double(), C.c_int(), C.c_double()
        cth = np.zeros((3, 21, 256, 2, 2), np.uint8)
        pe = np.zeros((3, 21), np.float64)
        lf = np.zeros(32768, np.float64)
        check(self.L.cp_params_export(self.p, cov, C.byref(dr), C.byref(cmax), C.byref(hc),
                                      cth.ctypes.data, pe.ctypes.data, lf.ctypes.data))
        return dict(cov=list(cov), dr_ratio=dr.value, cmax=cmax.value, hc_erate=hc.value,
                    cthres=cth, pe=pe, logfact=lf)

    def tables(self):
        """Device bytes of the look-up tables in use (0 = computed on the spot): dict(skel, uerr, petab)."""
        v = [C.c_size_t() for _ in range(3)]
        check(self.L.cp_params_tables(self.p, *[C.byref(x) for x in v]))
        return dict(skel=v[0].value, uerr=v[1].value, petab=v[2].value)

    # ---- pipeline ----
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def run(self, b, last_stage=STAGE_LABELS):
        check(self.L.cp_run_stages(self.p, self.ws, b.seq.data_ptr(), b.seq_off.data_ptr(),
                                   b.prof.data_ptr(), b.prof_off.data_ptr(), b.nreads, b.total_bases,
                                   b.total_kmers, b.labels.data_ptr(), last_stage, self._stream()))

    def classify(self, b, check_overflow=True):
        """ClassPro.c:229-271 for every read of the batch; returns the label bytes (host)."""
        check(self.L.cp_classify_batch(self.p, self.ws, b.seq.data_ptr(), b.seq_off.data_ptr(),
                                       b.prof.data_ptr(), b.prof_off.data_ptr(), b.nreads, b.total_bases,
                                       b.total_kmers, b.labels.data_ptr(), self._stream()))
        if check_overflow:
            check(self.L.cp_workspace_check(self.ws))
        return b.labels[:b.total_bases].cpu().numpy()

    def check(self):
        check(self.L.cp_workspace_check(self.ws))

    def label_runs(self, b, rerun=True):
        """The batch's labels as runs (cp_label_runs): classification without painting the label string, then per read
        (ends int32[n], cls uint8[n]); `expand_label_runs` rebuilds the strings on the host.  rerun=False: the runs of the
        batch last classified on this workspace (after classify() / run())."""
        if rerun:
            self.run(b, STAGE_CLASS_ALL)
        cap = int(self.L.cp_label_runs_capacity(self.ws))
        dev = self.device
        d_end = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        d_cls = torch.empty(max(cap, 1), dtype=torch.uint8, device=dev)
        d_nr = torch.empty(max(b.nreads, 1), dtype=torch.int32, device=dev)
        d_off = torch.empty(b.nreads + 1, dtype=torch.int64, device=dev)
        check(self.L.cp_label_runs(self.p, self.ws, d_end.data_ptr(), d_cls.data_ptr(), d_nr.data_ptr(), d_off.data_ptr(), self._stream()))
        check(self.L.cp_workspace_check(self.ws))
        ends, cls, nr, off = d_end.cpu().numpy(), d_cls.cpu().numpy(), d_nr.cpu().numpy(), d_off.cpu().numpy()
        return [(ends[off[r]:off[r] + nr[r]].copy(), cls[off[r]:off[r] + nr[r]].copy()) for r in range(b.nreads)]

    def find_seeds(self, b):
        """-s (seed.c:966-1032) on a batch already labelled by classify()/run(): returns (seed labels as host bytes
        in the layout of the labels, list per read of int32[n,2] repeat-mask intervals in read coordinates)."""
        seeds = torch.zeros(max(b.total_bases, 1), dtype=torch.uint8, device=self.device)
        check(self.L.cp_find_seeds_batch(self.p, self.ws, b.seq.data_ptr(), b.seq_off.data_ptr(), b.prof.data_ptr(),
                                         b.prof_off.data_ptr(), b.labels.data_ptr(), b.nreads, b.total_bases, b.total_kmers,
                                         seeds.data_ptr(), self._stream()))
        check(self.L.cp_workspace_check(self.ws))
        cap = int(self.L.cp_rep_masks_capacity(self.ws))
        cnt = np.zeros(max(b.nreads, 1), np.int32)
        off = np.zeros(b.nreads + 1, np.int64)
        pairs = np.zeros((max(cap, 1), 2), np.int32)
        check(self.L.cp_get_rep_masks(self.ws, cnt.ctypes.data, off.ctypes.data, pairs.ctypes.data, max(cap, 1)))
        b.seeds = seeds
        return seeds[:b.total_bases].cpu().numpy(), [pairs[off[r]:off[r] + cnt[r]].copy() for r in range(b.nreads)]

    def decode_profiles(self, codes, code_off, prof_off):
        """Fetch_Profile on the device: returns a device tensor of counts (int16 view of uint16)."""
        dev = self.device
        c = torch.from_numpy(np.ascontiguousarray(codes, np.uint8)).to(dev)
        co = torch.from_numpy(np.ascontiguousarray(code_off, np.int64)).to(dev)
        po = torch.from_numpy(np.ascontiguousarray(prof_off, np.int64)).to(dev)
        out = torch.zeros(max(int(prof_off[-1]), 8), dtype=torch.int16, device=dev)
        check(self.L.cp_decode_profiles(self.ws, c.data_ptr(), co.data_ptr(), po.data_ptr(), len(code_off) - 1,
                                        out.data_ptr(), self._stream()))
        check(self.L.cp_workspace_check(self.ws))
        return out

    def unpack_bases(self, packed, pack_off, seq_off):
        """Dazzler 2-bit bases -> upper-case characters on the device (uint8 tensor)."""
        dev = self.device
        pk = torch.from_numpy(np.ascontiguousarray(packed, np.uint8)).to(dev)
        po = torch.from_numpy(np.ascontiguousarray(pack_off, np.int64)).to(dev)
        so = torch.from_numpy(np.ascontiguousarray(seq_off, np.int64)).to(dev)
        out = torch.zeros(max(int(seq_off[-1]), 8), dtype=torch.uint8, device=dev)
        check(self.L.cp_unpack_bases(pk.data_ptr(), po.data_ptr(), so.data_ptr(), len(seq_off) - 1, out.data_ptr(), self._stream()))
        torch.cuda.synchronize(dev)
        return out

    def workspace_bytes(self):
        return int(self.L.cp_workspace_bytes(self.ws))

    # ---- stage read-back (parity tests) ----
    def counts(self, b):
        n = b.nreads
        nc, ni, nr = (np.zeros(n, np.int32) for _ in range(3))
        off = np.zeros(n + 1, np.int64)
        check(self.L.cp_get_counts(self.ws, nc.ctypes.data, ni.ctypes.data, nr.ctypes.data, off.ctypes.data))
        return nc, ni, nr, off

    def intervals(self, b):
        """Per-read lists of (intvl[N], rintvl[M]) after a run of at least STAGE_WALL / STAGE_REL."""
        nc, ni, nr, off = self.counts(b)
        tot = int(off[-1])
        iv = np.zeros(max(tot, 1), INTVL_DTYPE)
        rv = np.zeros(max(tot, 1), INTVL_DTYPE)
        check(self.L.cp_get_intervals(self.ws, iv.ctypes.data, rv.ctypes.data, max(tot, 1)))
        out = []
        for r in range(b.nreads):
            o = int(off[r])
            out.append((iv[o:o + ni[r]].copy(), rv[o:o + nr[r]].copy()))
        return out

    def rel_asgn(self, b):
        nc, ni, nr, off = self.counts(b)
        tot = int(off[-1])
        fw = np.zeros(max(tot, 1), np.int8)
        bw = np.zeros(max(tot, 1), np.int8)
        check(self.L.cp_get_rel_asgn(self.ws, fw.ctypes.data, bw.ctypes.data, max(tot, 1)))
        return [(fw[int(off[r]):int(off[r]) + nr[r]].copy(), bw[int(off[r]):int(off[r]) + nr[r]].copy())
                for r in range(b.nreads)]

    def bitmap(self, b):
        nw = b.total_kmers // 64 + 1
        w = np.zeros(nw, np.uint64)
        check(self.L.cp_get_bitmap(self.ws, w.ctypes.data, nw))
        return w

    def seq_context(self, b):
        """calc_seq_context (context.c:8), dense, per read: list of (lctx[rlen,3], rctx[rlen,3])."""
        l = torch.zeros((max(b.total_bases, 1), 3), dtype=torch.uint8, device=self.device)
        r = torch.zeros_like(l)
        check(self.L.cp_seq_context(b.seq.data_ptr(), b.seq_off.data_ptr(), b.nreads, b.total_bases,
                                    l.data_ptr(), r.data_ptr(), self._stream()))
        lh, rh = l.cpu().numpy(), r.cpu().numpy()
        so = b.seq_off_h
        return [(lh[so[i]:so[i + 1]], rh[so[i]:so[i + 1]]) for i in range(b.nreads)]
