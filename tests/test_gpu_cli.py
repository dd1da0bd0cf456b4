"""The drop-in `ClassPro` binary on a real MI355X (`-m gpu`): one input sharded over several device shards and
several host threads gives the same bytes as one shard and one thread (the reference's -T invariance,
SURVEY section 4), for plain FASTA, FASTQ and .gz inputs, FASTK profiles in several parts, windows and batches
much smaller than the input."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K = 40


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.fixture(scope="module")
def cli_set(torch_dev, tmp_path_factory):
    from classpro_amd import synth, fastk, build
    from classpro_amd.api import hist_covs
    from oracle.oracle import Oracle
    d = str(tmp_path_factory.mktemp("cli"))
    ds = synth.make_dataset(genome_len=300000, cov=40, read_len=9000, seed=21)
    seqs, profs, names = list(ds["seqs"]), list(ds["profiles"]), list(ds["names"])
    seqs.insert(7, b"ACGTACGTACGT"); profs.insert(7, np.zeros(0, np.uint16)); names.insert(7, "tiny")
    seqs.insert(400, b"A" * 39); profs.insert(400, np.zeros(0, np.uint16)); names.insert(400, "km1")
    comments = [None] * len(seqs)
    comments[2] = "first comment"
    comments[300] = "later\tcomment"
    with open(os.path.join(d, "reads.fasta"), "wb") as f:
        for n, s, c in zip(names, seqs, comments):
            f.write(b">" + n.encode() + ((b" " + c.encode()) if c else b"") + b"\n" + s + b"\n")
    with open(os.path.join(d, "readsq.fastq"), "wb") as f:
        for i, (n, s, c) in enumerate(zip(names, seqs, comments)):
            q = b"@" + b"I" * (len(s) - 1) if i % 2 else b"I" * len(s)
            f.write(b"@" + n.encode() + ((b" " + c.encode()) if c else b"") + b"\n" + s + b"\n+\n" + q + b"\n")
    fastk.write_fastk(d, "reads", K, profs, ds["hist"], nparts=3)
    fastk.write_fastk(d, "readsq", K, profs, ds["hist"], nparts=1)
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    O = Oracle(K, 20000, hc, dc)
    exp, last = [], "(null)"
    for n, s, p, c in zip(names, seqs, profs, comments):
        if c:
            last = c
        lab = O.classify_read(s, p) if len(s) >= K else b"N" * len(s)
        exp.append(b"@" + n.encode() + b" " + last.encode() + b"\n" + s + b"\n+\n" + lab + b"\n")
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    return d, cli, b"".join(exp)


def _run(cli, d, src, threads, devices=None, window_kb=None, batch_kb=None, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    if devices:
        env["CLASSPRO_DEVICES"] = devices
    if window_kb:
        env["CLASSPRO_WINDOW_KB"] = str(window_kb)
    if batch_kb:
        env["CLASSPRO_BATCH_KBASES"] = str(batch_kb)
    out = os.path.join(d, os.path.basename(src).split(".")[0] + ".class")
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([cli, "-v", "-T%d" % threads, os.path.join(d, src)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    return open(out, "rb").read(), r.stderr


def test_one_vs_many_shards_and_threads(cli_set):
    d, cli, exp = cli_set
    base, err = _run(cli, d, "reads.fasta", 1, devices="0")
    assert base == exp
    assert "1 device shard(s), 1 host threads" in err
    for threads, devices, win, bat in ((4, "0,0", None, None), (7, "0,0,0", 700, 900), (3, "0", 300, 500), (16, "0,0", 5000, 1500)):
        got, err = _run(cli, d, "reads.fasta", threads, devices, win, bat)
        assert got == base, (threads, devices, win, bat)
        assert "%d device shard(s), %d host threads" % (len(devices.split(",")), threads) in err


def test_fastq_and_gz_inputs(cli_set):
    d, cli, exp = cli_set
    got, _ = _run(cli, d, "readsq.fastq", 4, "0,0", 400, 600)
    assert got == exp
    subprocess.check_call(["gzip", "-kf", os.path.join(d, "readsq.fastq")])
    os.rename(os.path.join(d, "readsq.fastq"), os.path.join(d, "readsq.fastq.bak"))
    got, _ = _run(cli, d, "readsq.fastq.gz", 3, "0,0", 350, 500)
    assert got == exp


def test_tiny_inputs(torch_dev, tmp_path):
    """One read; only reads shorter than K (printed with N labels, nothing goes to the device); more threads and shards
    than reads."""
    from classpro_amd import synth, fastk, build
    from classpro_amd.api import hist_covs
    from oracle.oracle import Oracle
    d = str(tmp_path)
    ds = synth.make_dataset(genome_len=60000, cov=30, read_len=5000, seed=5)
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    O = Oracle(K, 20000, hc, dc)
    # one read
    s0, p0 = ds["seqs"][0], ds["profiles"][0]
    open(os.path.join(d, "one.fa"), "wb").write(b">solo\n" + s0 + b"\n")
    fastk.write_fastk(d, "one", K, [p0], ds["hist"], nparts=1)
    got, _ = _run(cli, d, "one.fa", 8, "0,0,0")
    assert got == b"@solo (null)\n" + s0 + b"\n+\n" + O.classify_read(s0, p0) + b"\n"
    # only short reads
    shorts = [b"ACGT", b"A" * 39, b"C"]
    with open(os.path.join(d, "short.fa"), "wb") as f:
        for i, s in enumerate(shorts):
            f.write(b">s%d x\n" % i + s + b"\n")
    fastk.write_fastk(d, "short", K, [np.zeros(0, np.uint16)] * 3, ds["hist"], nparts=2)
    got, _ = _run(cli, d, "short.fa", 4, "0,0")
    assert got == b"".join(b"@s%d x\n" % i + s + b"\n+\n" + b"N" * len(s) + b"\n" for i, s in enumerate(shorts))
    # read count mismatch between FASTA and profiles: message + exit 1 (ClassPro.c:148-154's consistency)
    open(os.path.join(d, "two.fa"), "wb").write(b">a\n" + s0 + b"\n>b\n" + s0 + b"\n")
    fastk.write_fastk(d, "two", K, [p0], ds["hist"], nparts=1)
    r = subprocess.run([cli, os.path.join(d, "two.fa")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Inconsistent # of reads" in r.stderr


def test_output_larger_than_the_upfront_estimate(torch_dev, tmp_path):
    """A plain FASTA's output gets its pages up front from an estimate (2 x the input + 64 MB); records that print an
    inherited comment (kseq keeps the last one) can make the output far larger than that: the rest is allocated window
    by window.  Same bytes with the up-front allocation switched off."""
    from classpro_amd import synth, fastk, build
    d = str(tmp_path)
    ds = synth.make_dataset(genome_len=60000, cov=30, read_len=5000, seed=5)
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    n, cmt = 70000, b"c" * 1000
    rng = np.random.default_rng(3)
    seqs = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 30)]) for _ in range(n)]
    with open(os.path.join(d, "wide.fa"), "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">r%d" % i + (b" " + cmt if i == 0 else b"") + b"\n" + s + b"\n")
    fastk.write_fastk(d, "wide", K, [np.zeros(0, np.uint16)] * n, ds["hist"], nparts=2)
    exp = b"".join(b"@r%d " % i + cmt + b"\n" + s + b"\n+\n" + b"N" * 30 + b"\n" for i, s in enumerate(seqs))
    assert len(exp) > 2 * os.path.getsize(os.path.join(d, "wide.fa")) + (64 << 20)
    got, _ = _run(cli, d, "wide.fa", 4, "0,0", 600, None)
    assert got == exp
    got, _ = _run(cli, d, "wide.fa", 4, "0,0", 600, None, {"CLASSPRO_OUT_ALLOC": "window"})
    assert got == exp


@pytest.mark.parametrize("mapping", [3, 70])
def test_full_output_device_in_any_window(cli_set, mapping):
    """A page of the mapped output that cannot be allocated shows up as SIGBUS at the first store into it.  The handler
    must recognise a fault in ANY output window as its own -- also the 71st mapping of a run (the registry of mapped
    ranges was a 64-entry table until round 5: ADVICE r4) -- say "no space", remove the partial output, exit 1.
    CLASSPRO_DEBUG_ENOSPC_MAPPING backs the chosen mapping with an empty anonymous file, the same fault as a full tmpfs."""
    d, cli, exp = cli_set
    env = dict(os.environ, CLASSPRO_DEVICES="0", CLASSPRO_WINDOW_KB="100", CLASSPRO_OUT_ALLOC="window",
               CLASSPRO_DEBUG_ENOSPC_MAPPING=str(mapping))
    assert os.path.getsize(os.path.join(d, "reads.fasta")) > 75 * 100 * 1024      # more than 70 windows
    out = os.path.join(d, "reads.class")
    r = subprocess.run([cli, "-T4", os.path.join(d, "reads.fasta")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr)
    assert "no space left on the output device" in r.stderr
    assert not os.path.exists(out)
    env.pop("CLASSPRO_DEBUG_ENOSPC_MAPPING")
    r = subprocess.run([cli, "-T4", os.path.join(d, "reads.fasta")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and open(out, "rb").read() == exp
