"""Device-side synthetic read sets: TEST / BENCH INFRASTRUCTURE, not the product.

`DeviceSynth` drives the kernels of csrc/synth/synth_gen.hip (libcp_synth.so): a seeded diploid genome with
segmental repeats, low-complexity runs and homozygous blocks, HiFi-like reads with substitution / indel errors,
FASTK-like k-mer count profiles and the ground-truth relative profile -- all generated in HBM, a few seconds
for the 8-Gbase (200 Mbp x 40x) set of BASELINE configs[2].  Every read is a pure function of (seed, read id),
so a rank of a multi-GPU run regenerates exactly the read range it owns and the reads are all distinct (nothing is
tiled).  torch owns the device memory and does the prefix sums between the kernels.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcp_synth.so")
SLOT_SHIFT = 14
SLOT = 1 << SLOT_SHIFT


class SgParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("G", C.c_int64), ("n_reads", C.c_int64), ("K", C.c_int32),
                ("copy_off", C.c_int32), ("copy_len", C.c_int32), ("lc_span", C.c_int32), ("homo_every", C.c_int32),
                ("t_het", C.c_uint32), ("t_sub", C.c_uint32), ("t_del", C.c_uint32), ("t_ins", C.c_uint32),
                ("hp_mult", C.c_int32), ("read_mean", C.c_int32), ("read_sd", C.c_int32), ("min_len", C.c_int32),
                ("max_len", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("classpro_amd: %s not found; build it with `python -m classpro_amd.build`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.sg_last_error.restype = C.c_char_p
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        pp = C.POINTER(SgParams)
        L.sg_genome.argtypes = [pp, vp, vp, vp]
        L.sg_reads_pass1.argtypes = [pp, vp, vp, vp, vp, vp, vp]
        L.sg_famtot.argtypes = [pp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
        L.sg_totals.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sg_reads_pass2.argtypes = [pp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("synth_gen: " + lib().sg_last_error().decode(errors="replace"))


def _cumsum_into(src, out, chunk=1 << 28):
    """out[i] = sum(src[0..i]) as int32, chunk by chunk with a carried total: the arrays of a 3-Gbp genome hold more
    than 2^31 elements and a chunk's temporary stays at 1 GB.  `out` may be `src` itself (int32, in place)."""
    carry = 0
    n = src.numel()
    for a in range(0, n, chunk):
        b = min(a + chunk, n)
        t = torch.cumsum(src[a:b], 0, dtype=torch.int32)
        if carry:
            t += carry
        carry = int(t[-1])
        out[a:b] = t
        del t
    return out


def _thr(rate):
    return int(min(max(rate, 0.0), 0.999999) * 4294967296.0)


class DeviceSynth:
    """A synthetic diploid read set living on one HIP device.

    After construction: n_reads, rlen (int32 device tensor, all reads), total_bases, hist (the FASTK `.hist`
    tuple low, high, ilowcnt, ihighcnt, int64[high-low+1]).  `reads(first, count)` generates that read range."""

    def __init__(self, genome_len=200_000_000, cov=40, read_len=20000, K=40, het=0.001, err_sub=0.0006,
                 err_indel=0.0006, hp_mult=5, min_len=3000, max_len=60000, copy_len=3000, n_families=None,
                 homo_every=64, seed=1, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceSynth needs a HIP device")
        torch.cuda.set_device(self.device)
        self.L = lib()
        G = int(genome_len)
        self.G, self.K = G, K
        n_reads = max(1, int(cov * G / read_len))
        p = SgParams()
        p.seed, p.G, p.n_reads, p.K = seed, G, n_reads, K
        p.copy_off, p.copy_len, p.lc_span, p.homo_every = 12000, copy_len, 9000, homo_every
        p.t_het, p.t_sub = _thr(het), _thr(err_sub)
        p.t_del = p.t_ins = _thr(err_indel / 2)
        p.hp_mult = hp_mult
        p.read_mean, p.read_sd, p.min_len, p.max_len = read_len, int(0.2 * read_len), min(min_len, G), max_len
        self.p = p
        self.n_reads = n_reads
        dev = self.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        # segmental-repeat families: copies placed in distinct slots (host tables, seeded)
        nslots = G >> SLOT_SHIFT
        rng = np.random.default_rng(seed)
        nf = max(3, G // 80000) if n_families is None else n_families
        ncopy = rng.integers(3, 6, size=nf)
        if ncopy.sum() > nslots:                         # small genomes: as many whole families as fit
            nf = int(np.searchsorted(np.cumsum(ncopy), nslots, side="right"))
            ncopy = ncopy[:nf]
        slots = rng.permutation(nslots)[:int(ncopy.sum())].astype(np.int32)
        fam_off = np.zeros(nf + 1, np.int32)
        fam_off[1:] = np.cumsum(ncopy)
        slot_fam = np.full(max(nslots, 1), -1, np.int32)
        for f in range(nf):
            s = np.sort(slots[fam_off[f]:fam_off[f + 1]])
            slots[fam_off[f]:fam_off[f + 1]] = s
            slot_fam[s] = f * 8 + np.arange(len(s))
        self.n_fam = nf
        d_slot_fam = torch.from_numpy(slot_fam).to(dev)
        d_fam_off = torch.from_numpy(fam_off).to(dev)
        d_fam_slots = torch.from_numpy(slots if len(slots) else np.zeros(1, np.int32)).to(dev)

        self.gen = torch.empty(G, dtype=torch.uint8, device=dev)
        _chk(self.L.sg_genome(C.byref(p), d_slot_fam.data_ptr(), self.gen.data_ptr(), st))
        snpcum = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        for a in range(0, G, 1 << 28):                     # SNP flags -> prefix counts, 256 M positions at a time
            b = min(a + (1 << 28), G)
            t = torch.cumsum((self.gen[a:b] >> 4) & 1, 0, dtype=torch.int32)
            t += snpcum[a]
            snpcum[a + 1:b + 1] = t
            del t

        diffA = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        diffB = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        self.rlen = torch.zeros(n_reads, dtype=torch.int32, device=dev)
        nerr = torch.zeros(1, dtype=torch.int64, device=dev)
        _chk(self.L.sg_reads_pass1(C.byref(p), self.gen.data_ptr(), diffA.data_ptr(), diffB.data_ptr(),
                                   self.rlen.data_ptr(), nerr.data_ptr(), st))
        cntA = _cumsum_into(diffA, diffA)                   # in place: clean coverage per window and haplotype
        cntB = _cumsum_into(diffB, diffB)
        del diffA, diffB

        nwin = copy_len - K + 1
        famtot = torch.zeros(max(nf * nwin, 1), dtype=torch.int32, device=dev)
        fammult = torch.zeros(max(nf * nwin, 1), dtype=torch.uint8, device=dev)
        _chk(self.L.sg_famtot(C.byref(p), snpcum.data_ptr(), cntA.data_ptr(), cntB.data_ptr(), d_fam_off.data_ptr(),
                              d_fam_slots.data_ptr(), nf, famtot.data_ptr(), fammult.data_ptr(), st))
        self.totA = torch.empty(G, dtype=torch.int16, device=dev)          # uint16 payloads
        self.totB = torch.empty(G, dtype=torch.int16, device=dev)
        self.relA = torch.empty(G, dtype=torch.uint8, device=dev)
        self.relB = torch.empty(G, dtype=torch.uint8, device=dev)
        hist = torch.zeros(32768, dtype=torch.int64, device=dev)
        _chk(self.L.sg_totals(C.byref(p), snpcum.data_ptr(), cntA.data_ptr(), cntB.data_ptr(), d_slot_fam.data_ptr(),
                              famtot.data_ptr(), fammult.data_ptr(), self.totA.data_ptr(), self.totB.data_ptr(),
                              self.relA.data_ptr(), self.relB.data_ptr(), hist.data_ptr(), st))
        torch.cuda.synchronize(dev)
        del snpcum, cntA, cntB, famtot, fammult
        h = hist.cpu().numpy()
        self.n_err_kmers = int(nerr.item())
        h[1] += self.n_err_kmers                         # every k-mer with a read error is a distinct k-mer seen once
        self.hist = (1, 32767, 0, 0, h[1:32768].copy())
        self.rlen_h = self.rlen.cpu().numpy().astype(np.int64)
        self.seq_off_all = np.zeros(n_reads + 1, np.int64)
        np.cumsum(self.rlen_h, out=self.seq_off_all[1:])
        self.total_bases = int(self.seq_off_all[-1])
        self._err = torch.zeros(1, dtype=torch.int32, device=dev)

    # ---- read ranges -------------------------------------------------------------------------------------
    def plan_batches(self, target_bases, first=0, last=None):
        """Contiguous read ranges [(first, count)] of about `target_bases` bases each covering [first, last)."""
        last = self.n_reads if last is None else last
        so = self.seq_off_all
        out, a = [], first
        while a < last:
            b = int(np.searchsorted(so, so[a] + target_bases, side="left"))
            b = min(max(b, a + 1), last)
            if last - b < (b - a) // 4:                  # do not leave a small ragged batch at the end
                b = last
            out.append((a, b - a))
            a = b
        return out

    def reads(self, first, count, truth=False):
        """Reads [first, first+count) in the flat layout of include/classpro_amd.h, as device tensors:
        dict(seq uint8, seq_off int64, prof int16 (uint16 payload), prof_off int64, truth uint8 or None,
        seq_off_h / prof_off_h numpy, nreads, total_bases, total_kmers)."""
        dev = self.device
        Km1 = self.K - 1
        so = self.seq_off_all[first:first + count + 1] - self.seq_off_all[first]
        po = so - np.arange(count + 1, dtype=np.int64) * Km1
        nb, nk = int(so[-1]), int(po[-1])
        d_so = torch.from_numpy(np.ascontiguousarray(so)).to(dev)
        d_po = torch.from_numpy(np.ascontiguousarray(po)).to(dev)
        seq = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        prof = torch.empty(max(nk, 8), dtype=torch.int16, device=dev)
        tr = torch.empty(max(nk, 1), dtype=torch.uint8, device=dev) if truth else None
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _chk(self.L.sg_reads_pass2(C.byref(self.p), self.gen.data_ptr(), self.totA.data_ptr(), self.totB.data_ptr(),
                                   self.relA.data_ptr(), self.relB.data_ptr(), first, count, d_so.data_ptr(),
                                   seq.data_ptr(), prof.data_ptr(), tr.data_ptr() if truth else None,
                                   self._err.data_ptr(), st))
        return dict(seq=seq, seq_off=d_so, prof=prof, prof_off=d_po, truth=tr, seq_off_h=so, prof_off_h=po,
                    nreads=count, total_bases=nb, total_kmers=nk, first=first)

    def check(self):
        """Raises if a generated read's length disagreed with the length pass (cannot happen: same walk)."""
        if int(self._err.item()) != 0:
            raise RuntimeError("synth_gen: read length mismatch between the passes")

    def to_host(self, rd, names=True):
        """Host copy of a `reads()` result in the form classpro_amd.synth.make_dataset returns (lists per read)."""
        seq = rd["seq"].cpu().numpy()
        prof = rd["prof"].cpu().numpy().view(np.uint16)
        so, po = rd["seq_off_h"], rd["prof_off_h"]
        n = rd["nreads"]
        out = dict(seqs=[seq[so[i]:so[i + 1]].tobytes() for i in range(n)],
                   profiles=[prof[po[i]:po[i + 1]] for i in range(n)], K=self.K, hist=self.hist)
        if rd["truth"] is not None:
            tr = rd["truth"].cpu().numpy()
            out["rel_profiles"] = [tr[po[i]:po[i + 1]].astype(np.uint16) for i in range(n)]
        if names:
            out["names"] = ["read%d" % (rd["first"] + i + 1) for i in range(n)]
        return out
