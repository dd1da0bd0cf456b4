"""The command line's windowed, chunk-parallel FASTX indexer (csrc/host/fastx_index.h) == the sequential
kseq-semantics reader (kseq.h:177-216 as restated in host_io.h) on awkward inputs: wrapped sequences, records
without comments (kseq prints the previous record's comment), tabs, CR LF, blank lines, lower case, FASTQ with
quality lines that start with '@' or '>', windows that end inside a record, both window protocols (in place for
an mmap'ed file, carried tail for a .gz)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, build_if_changed


@pytest.fixture(scope="module")
def checker():
    src = os.path.join(ROOT, "tests", "fx_index_check.cpp")
    out = os.path.join(ROOT, "tests", "_fx_index_check")
    deps = [src] + [os.path.join(ROOT, "classpro_amd", "csrc", "host", f) for f in ("fastx_index.h", "host_io.h", "thread_pool.h")]
    return build_if_changed(out, ["g++", "-O2", "-std=c++17", "-o", out, src, "-lz", "-lpthread"], deps)


def _reads(rng, n, lo=50, hi=4000):
    al = np.frombuffer(b"ACGTacgtN", np.uint8)
    return [bytes(al[rng.integers(0, len(al), int(rng.integers(lo, hi)))]) for _ in range(n)]


def _fasta(rng, reads, wrap, comments, crlf=False, blanks=False):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i, s in enumerate(reads):
        h = b">r%d" % i
        c = comments[i % len(comments)]
        if c is not None:
            h += c
        out.append(h + nl)
        if wrap:
            for k in range(0, len(s), wrap):
                out.append(s[k:k + wrap] + nl)
        else:
            out.append(s + nl)
        if blanks and i % 7 == 3:
            out.append(nl)
    return b"".join(out)


def _fastq(rng, reads, evil=True):
    out = []
    for i, s in enumerate(reads):
        q = bytearray(rng.integers(33, 74, len(s)).astype(np.uint8).tobytes())
        if evil and len(q) > 2:
            q[0] = ord("@") if i % 3 == 0 else (ord(">") if i % 3 == 1 else ord("+"))
        out.append(b"@q%d %s\n%s\n+\n%s\n" % (i, b"c%d" % i if i % 4 else b"", s, bytes(q)))
    return b"".join(out)


CASES = [
    ("single_line", lambda r: _fasta(r, _reads(r, 300), 0, [b" cm1", None, b"\tcm two words", None, None, b" x"])),
    ("wrapped", lambda r: _fasta(r, _reads(r, 300), 60, [None, None, b" late comment", None])),
    ("crlf_blank", lambda r: _fasta(r, _reads(r, 200), 70, [b" a b", None], crlf=True, blanks=True)),
    ("no_comments", lambda r: _fasta(r, _reads(r, 250), 0, [None])),
    ("no_final_newline", lambda r: _fasta(r, _reads(r, 50), 0, [b" z"])[:-1]),
    ("leading_junk", lambda r: b"\n\n  junk line\n" + _fasta(r, _reads(r, 80), 80, [b" k", None])),
    ("fastq_evil", lambda r: _fastq(r, _reads(r, 300))),
    ("empty_seq", lambda r: b">a c\n\n>b\nACGT\n>c\n>d x\nAC\nGT\n"),
]


@pytest.mark.parametrize("name,gen", CASES)
def test_indexer_matches_sequential_reader(checker, tmp_path, name, gen):
    rng = np.random.default_rng(sum(name.encode()))
    data = gen(rng)
    p = tmp_path / (name + ".fx")
    p.write_bytes(data)
    first = None
    for threads, window, carry, minpar in ((1, 1 << 30, 0, 0), (4, 1 << 30, 0, 0), (4, 9000, 0, 0), (3, 9000, 1, 0),
                                           (4, 50000, 1, 0), (2, 20011, 0, 1 << 22)):
        out = subprocess.run([checker, str(p), str(threads), str(window), str(carry), str(minpar)],
                             capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("OK"), (name, threads, window, carry, out.stdout, out.stderr)
        first = first or out.stdout
        assert out.stdout == first


def test_indexer_bad_quality_and_carried_windows(checker, tmp_path):
    rng = np.random.default_rng(5)
    good = _fastq(rng, _reads(rng, 40), evil=False)
    bad = good + b"@broken\nACGTACGT\n+\nIIII\n"
    p = tmp_path / "bad.fq"
    p.write_bytes(bad)
    out = subprocess.run([checker, str(p), "2", "100000", "0", "0"], capture_output=True, text=True)
    assert "BADQUAL" in out.stdout
    p2 = tmp_path / "ok.fa"
    p2.write_bytes(_fasta(rng, _reads(rng, 100), 60, [b" c", None]))
    out = subprocess.run([checker, str(p2), "4", "7000", "1", "0"], capture_output=True, text=True)
    assert out.stdout.startswith("OK 100")
