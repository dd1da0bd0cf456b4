"""FASTK on-disk formats that ClassPro consumes (`<root>.hist`, `<root>.prof`, `.<root>.pidx.N`,
`.<root>.prof.N`) -- Python writer/reader used to build test and benchmark inputs.

Format as read by the reference (all little-endian):
  .hist    int kmer, int low, int high, int64 ilowcnt, int64 ihighcnt, int64 hist[high-low+1]   (libfastk.c:72-83)
  .prof    int kmer, int nparts                                                              (libfastk.c:1291-1292)
  .pidx.N  int kmer, int64 first, int64 n, int64 endoff[n]  (cumulative byte ends in .prof.N) (libfastk.c:1305-1333)
  .prof.N  per read: first count 1 byte (<128) or 2 bytes (0x80|hi, lo); then
           00rrrrrr run of r equal counts, 01sxxxxx 6-bit signed delta, 1sxxxxxx yyyyyyyy 15-bit delta
                                                                                             (libfastk.c:1467-1534)
The production host path (C++ CLI, classpro_amd/csrc/host) has its own readers; this module is tooling.
"""
import gzip
import os
import struct
import numpy as np


def encode_profile(c):
    """Inverse of Fetch_Profile's decoder for one read (counts <= 32767)."""
    out = bytearray()
    n = len(c)
    if n == 0:
        return bytes(out)
    c = [int(x) for x in c]
    d = c[0]
    if d < 128:
        out.append(d)
    else:
        out.append(0x80 | (d >> 8))
        out.append(d & 0xFF)
    i = 1
    while i < n:
        if c[i] == d:
            run = 1
            while i + run < n and c[i + run] == d and run < 63:
                run += 1
            out.append(run)
            i += run
        else:
            dl = c[i] - d
            if 1 <= dl <= 31:
                out.append(0x40 | dl)
            elif -32 <= dl <= -1:
                out.append(0x60 | (dl + 32))
            else:
                x = dl & 0x7FFF
                out.append(0x80 | (x >> 8))
                out.append(x & 0xFF)
            d = c[i]
            i += 1
    return bytes(out)


def decode_profile(code):
    """Pure-Python decoder (tooling / cross-check of the encoder)."""
    code = bytes(code)
    if not code:
        return np.zeros(0, np.uint16)
    out = []
    p = 0
    x = code[p]; p += 1
    if x & 0x80:
        d = ((x & 0x7F) << 8) | code[p]; p += 1
    else:
        d = x
    out.append(d)
    while p < len(code):
        x = code[p]; p += 1
        if (x & 0xC0) == 0:
            out.extend([d] * x)
        else:
            if x & 0x80:
                v = ((x << 8) & 0xFFFF) if (x & 0x40) else ((x << 8) & 0x7FFF)
                v |= code[p]; p += 1
                d = (d + v) & 0x7FFF
            else:
                if x & 0x20:
                    d = (d + ((x & 0x1F) | 0xFFE0)) & 0xFFFF
                else:
                    d = (d + (x & 0x1F)) & 0xFFFF
            out.append(d)
    return np.array(out, dtype=np.uint16)


def write_fastk(dirpath, root, K, profiles, hist, nparts=1):
    """hist = (low, high, ilowcnt, ihighcnt, int64 array[high-low+1])."""
    os.makedirs(dirpath, exist_ok=True)
    low, high, ilow, ihigh, h = hist
    with open(os.path.join(dirpath, root + ".hist"), "wb") as f:
        f.write(struct.pack("<iii", K, low, high))
        f.write(struct.pack("<qq", ilow, ihigh))
        f.write(np.asarray(h, dtype="<i8").tobytes())
    with open(os.path.join(dirpath, root + ".prof"), "wb") as f:
        f.write(struct.pack("<ii", K, nparts))
    n = len(profiles)
    per = (n + nparts - 1) // nparts if nparts else n
    first = 0
    for part in range(nparts):
        chunk = profiles[part * per:(part + 1) * per]
        codes = [encode_profile(c) for c in chunk]
        ends = np.cumsum([len(c) for c in codes], dtype=np.int64) if codes else np.zeros(0, np.int64)
        with open(os.path.join(dirpath, ".%s.pidx.%d" % (root, part + 1)), "wb") as f:
            f.write(struct.pack("<i", K))
            f.write(struct.pack("<qq", first, len(chunk)))
            f.write(ends.astype("<i8").tobytes())
        with open(os.path.join(dirpath, ".%s.prof.%d" % (root, part + 1)), "wb") as f:
            f.write(b"".join(codes))
        first += len(chunk)


def read_fastk_hist(path):
    with open(path, "rb") as f:
        K, low, high = struct.unpack("<iii", f.read(12))
        ilow, ihigh = struct.unpack("<qq", f.read(16))
        h = np.frombuffer(f.read(8 * (high - low + 1)), dtype="<i8").copy()
    return K, low, high, ilow, ihigh, h


def read_fastk_codes(dirpath, root):
    """Returns (K, [code bytes per read]) without decoding."""
    with open(os.path.join(dirpath, root + ".prof"), "rb") as f:
        K, nparts = struct.unpack("<ii", f.read(8))
    codes = []
    for part in range(nparts):
        with open(os.path.join(dirpath, ".%s.pidx.%d" % (root, part + 1)), "rb") as f:
            _k = struct.unpack("<i", f.read(4))[0]
            _first, n = struct.unpack("<qq", f.read(16))
            ends = np.frombuffer(f.read(8 * n), dtype="<i8")
        with open(os.path.join(dirpath, ".%s.prof.%d" % (root, part + 1)), "rb") as f:
            blob = f.read()
        o = 0
        for e in ends:
            codes.append(blob[o:int(e)])
            o = int(e)
    return K, codes


def write_fasta(path, names, seqs, gz=None, comments=None):
    gz = path.endswith(".gz") if gz is None else gz
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, (n, s) in enumerate(zip(names, seqs)):
            hdr = n if not comments else "%s %s" % (n, comments[i])
            f.write(b">" + hdr.encode() + b"\n" + s + b"\n")
