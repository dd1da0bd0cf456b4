"""Dazzler database files for tests and tooling: a writer for .db / .dam (+ hidden .idx / .bps / .hdr) in
the layout DAZZ_DB's Open_DB / Load_Read read (DB.h:287-297, 392-437; DB.c:319-363, 690-900), and a
reader for the data tracks ClassPro writes (.class.anno / .class.data).

The writer is pinned by the reference's own DB.c: tests run the reference-built prof2class (oracle/_ref)
on databases written here and compare its records with the FASTA the database was made from."""
import os
import struct

import numpy as np

_CODE = np.full(256, 0, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[ord(chr(_c).lower())] = _i


def pack_2bit(vals):
    """4 values per byte, first in the top bits (Compress_Read, DB.c:319-338)."""
    v = np.asarray(vals, np.uint8)
    n = len(v)
    pad = np.zeros((-n) % 4, np.uint8)
    q = np.concatenate([v, pad]).reshape(-1, 4)
    return ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)


def unpack_2bit(buf, n):
    b = np.frombuffer(bytes(buf), np.uint8)
    out = np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], 1).reshape(-1)
    return out[:n]


def _db_header(nreads, maxlen, totlen, freq):
    # DAZZ_DB on x86-64: 112 bytes (DB.h:392-422); pointers are meaningless on disk
    h = bytearray(112)
    struct.pack_into("<iiii4f", h, 0, nreads, nreads, -1, 1, *freq)      # ureads treads cutoff allarr freq
    struct.pack_into("<i", h, 32, maxlen)
    struct.pack_into("<q", h, 40, totlen)
    struct.pack_into("<iiiii", h, 48, nreads, 0, 0, 0, 0)                # nreads trimmed part ufirst tfirst
    return bytes(h)


def write_db(dirpath, root, seqs, files, dam=False, hdr_lines=None):
    """files = [(n_reads_in_file, fname, prolog)], wells/pulses are synthesised: read i of a file gets
    well i and pulses [0, rlen).  For a .dam pass hdr_lines (one '>...' line per read)."""
    os.makedirs(dirpath, exist_ok=True)
    n = len(seqs)
    ext = ".dam" if dam else ".db"
    with open(os.path.join(dirpath, root + ext), "w") as f:
        f.write("files = %9d\n" % len(files))
        last = 0
        for cnt, fname, prolog in files:
            last += cnt
            f.write("  %9d %s %s\n" % (last, fname, prolog))
        assert last == n
        f.write("blocks = %9d\n" % 1)
        f.write("size = %11d cutoff = %9d all = %1d\n" % (200, 0, 1))
        f.write(" %9d %9d\n" % (0, 0))
        f.write(" %9d %9d\n" % (n, n))
    recs, boff, coff = [], 0, 0
    hdr = b""
    with open(os.path.join(dirpath, "." + root + ".bps"), "wb") as f:
        fi, within = 0, 0
        for i, s in enumerate(seqs):
            while within >= files[fi][0]:
                fi, within = fi + 1, 0
            packed = pack_2bit(_CODE[np.frombuffer(s, np.uint8)])
            f.write(packed.tobytes())
            if dam:
                line = hdr_lines[i].encode() + b"\n"
                recs.append((i, len(s), 0, boff, len(hdr), 0))
                hdr += line
            else:
                recs.append((within + 1, len(s), 7 * (i % 5), boff, 0, 0x0800 | 850))
            boff += len(packed)
            within += 1
    if dam:
        open(os.path.join(dirpath, "." + root + ".hdr"), "wb").write(hdr)
    counts = np.bincount(np.concatenate([_CODE[np.frombuffer(s, np.uint8)] for s in seqs]) if n else np.zeros(0, np.uint8),
                         minlength=4).astype(np.float64)
    freq = (counts / max(counts.sum(), 1)).tolist()
    with open(os.path.join(dirpath, "." + root + ".idx"), "wb") as f:
        f.write(_db_header(n, max((len(s) for s in seqs), default=0), sum(len(s) for s in seqs), freq))
        for origin, rlen, fpulse, bo, co, flags in recs:
            f.write(struct.pack("<iiiiqqii", origin, rlen, fpulse, 0, bo, co, flags, 0))
    return recs


def db_headers(files, recs, dam=False, hdr_lines=None):
    """The header ClassPro / prof2class print for every read (ClassPro.c:166-180)."""
    if dam:
        return ["@" + h[1:] for h in hdr_lines]
    out, fi, within = [], 0, 0
    for origin, rlen, fpulse, _bo, _co, _fl in recs:
        while within >= files[fi][0]:
            fi, within = fi + 1, 0
        out.append("@%s/%d/%d_%d" % (files[fi][2], origin, fpulse, fpulse + rlen))
        within += 1
    return out


def read_class_track(dirpath, root, kind="class"):
    """(nreads, size, int64 offsets[nreads+1], data bytes) of .<root>.<kind>.anno/.data."""
    a = open(os.path.join(dirpath, ".%s.%s.anno" % (root, kind)), "rb").read()
    nreads, size = struct.unpack_from("<ii", a, 0)
    offs = np.frombuffer(a, "<i8", offset=8)
    data = open(os.path.join(dirpath, ".%s.%s.data" % (root, kind)), "rb").read()
    return nreads, size, offs, data
