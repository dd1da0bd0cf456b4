"""Dazzler database inputs and track outputs.

* the test-side writer classpro_amd/dazz.py is pinned by the reference's own DB.c (Open_DB / Load_Read
  through oracle/_ref, and the reference-built prof2class run on the database);
* the product's reader (csrc/host/dazz_db.h, used by ClassPro and prof2class) must give the same records;
* the product's track writer must produce files DAZZ_DB's own Open_Track / Load_All_Track_Data load, with
  the 2-bit codes ClassPro.c:290-304 defines (K-1 zeros, then E=0 R=1 H=2 D=3).
The sha256 of the reference prof2class output is kept in tests/golden/dazz_db.json for machines where
/root/reference (hence oracle/_ref) is absent."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import eval_case
from conftest import ROOT, build_if_changed

K = 40
TOOLS = os.path.join(ROOT, "classpro_amd")
REF = os.path.join(ROOT, "oracle", "_ref")
FILES = lambda n: [(n // 3, "cell_one", "m64011_190830_220126"), (n - n // 3, "cell_two.subreads", "m54238_180901_011437")]


def make_case(d, dam):
    from classpro_amd import synth, fastk, dazz
    ds = synth.make_dataset(genome_len=30000, cov=20, read_len=3000, min_len=800, seed=909)
    seqs, rels = list(ds["seqs"]), list(ds["rel_profiles"])
    seqs.insert(4, b"ACGTTGCAAC"); rels.insert(4, np.zeros(0, np.uint16))          # shorter than K
    n = len(seqs)
    hdr = [">scaf%d ctg=%d len=%d" % (i // 2, i % 2, len(s)) for i, s in enumerate(seqs)] if dam else None
    recs = dazz.write_db(d, "reads", seqs, FILES(n), dam=dam, hdr_lines=hdr)
    fastk.write_fastk(d, "truth", K, rels, ds["hist"], nparts=2)
    fastk.write_fastk(d, "reads", K, [p for p in ds["profiles"][:4]] + [np.zeros(0, np.uint16)] + list(ds["profiles"][4:]),
                      ds["hist"], nparts=1)
    heads = dazz.db_headers(FILES(n), recs, dam=dam, hdr_lines=hdr)
    want = b""
    for h, s, r in zip(heads, seqs, rels):
        lab = b"N" * len(s) if len(s) < K else b"N" * (K - 1) + bytes(b"EHDR"[min(int(c), 3)] for c in r)
        want += h.encode() + b"\n" + s + b"\n+\n" + lab + b"\n"
    return seqs, heads, recs, want


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "dazz_db.json")))


@pytest.mark.parametrize("dam", [False, True], ids=["db", "dam"])
def test_prof2class_on_database(built, tmp_path, golden, dam):
    d = str(tmp_path)
    seqs, heads, recs, want = make_case(d, dam)
    src = os.path.join(d, "reads.dam" if dam else "reads.db")
    subprocess.check_call([os.path.join(TOOLS, "prof2class"), os.path.join(d, "truth"), src])
    ours = open(os.path.join(d, "truth.class"), "rb").read()
    assert ours == want
    assert eval_case.sha(os.path.join(d, "truth.class")) == golden["prof2class_sha256_dam" if dam else "prof2class_sha256_db"]
    if os.path.exists(os.path.join(REF, "prof2class")):
        os.remove(os.path.join(d, "truth.class"))
        subprocess.check_call([os.path.join(REF, "prof2class"), os.path.join(d, "truth"), src])
        assert open(os.path.join(d, "truth.class"), "rb").read() == ours
    # error contract: read-count mismatch
    from classpro_amd import fastk
    fastk.write_fastk(d, "fewer", K, [np.zeros(5, np.uint16)] * 3, (1, 2, 0, 0, np.zeros(2, np.int64)))
    r = subprocess.run([os.path.join(TOOLS, "prof2class"), os.path.join(d, "fewer"), src], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Inconsistent # of reads: .prof (3) != .db (%d)\n" % len(seqs)


def _ref_lib():
    p = os.path.join(REF, "libclasspro_ref.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    L = C.CDLL(p)
    L.ref_db_track.restype = C.c_longlong
    L.ref_db_track.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_longlong]
    return L


@pytest.mark.parametrize("dam", [False, True], ids=["db", "dam"])
def test_writer_and_tracks_against_reference_db_library(built, tmp_path, dam):
    from classpro_amd import dazz
    L = _ref_lib()
    d = str(tmp_path)
    seqs, heads, recs, want = make_case(d, dam)
    src = os.path.join(d, "reads.dam" if dam else "reads.db").encode()
    assert L.ref_db_open(src) == (1 if dam else 0)
    assert L.ref_db_nreads() == len(seqs) and L.ref_db_maxlen() == max(len(s) for s in seqs)
    buf = C.create_string_buffer(L.ref_db_maxlen() + 8)
    o, f, c = C.c_int(), C.c_int(), C.c_longlong()
    for i, s in enumerate(seqs):
        n = L.ref_db_read(i, buf, C.byref(o), C.byref(f), C.byref(c))
        assert buf.raw[:n] == s and (o.value, f.value) == (recs[i][0], recs[i][2])
    # tracks written by the product's ClassTrack from a .class file, loaded back by DAZZ_DB's Open_Track
    open(os.path.join(d, "est.class"), "wb").write(want)
    exe = os.path.join(ROOT, "tests", "_track_harness")
    srcs = [os.path.join(ROOT, "tests", "track_harness.cpp"), os.path.join(TOOLS, "csrc", "host", "dazz_db.h"),
            os.path.join(TOOLS, "csrc", "host", "host_io.h")]
    build_if_changed(exe, ["g++", "-O2", "-std=c++17", srcs[0], "-o", exe, "-lz"], srcs)
    subprocess.check_call([exe, d, "reads", os.path.join(d, "est.class")])
    alen = (C.c_int * len(seqs))()
    data = C.create_string_buffer(sum(len(s) for s in seqs))
    tot = L.ref_db_track(b"class", alen, data, len(data))
    assert tot == sum((len(s) + 3) // 4 for s in seqs)
    code = {ord("N"): 0, ord("E"): 0, ord("R"): 1, ord("H"): 2, ord("D"): 3}
    off = 0
    labels = [rec.split(b"\n")[3] for rec in want.split(b"\n@")]
    for i, s in enumerate(seqs):
        assert alen[i] == (len(s) + 3) // 4
        got = dazz.unpack_2bit(data.raw[off:off + alen[i]], len(s))
        assert np.array_equal(got, np.array([code[ch] for ch in labels[i]], np.uint8))
        off += alen[i]
    nreads, size, offs, raw = dazz.read_class_track(d, "reads")
    assert (nreads, size) == (len(seqs), 8) and offs[0] == 0 and offs[-1] == len(raw) == tot
    nr, sz, roffs, rraw = dazz.read_class_track(d, "reads", "rep")
    assert (nr, sz, len(roffs), len(rraw)) == (len(seqs), 0, 1, 0)          # header-only mask track (io.c:306-311)
    L.ref_db_close()
