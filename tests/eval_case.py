"""A small evaluation scenario for the prof2class / class2acc tools: a synthetic
read set, its FASTK read profile, the *relative* profile against the true genome (ground truth), and
an estimate .class file written with labels from the oracle.  Shared by oracle/gen_golden.py (which runs
the reference's own tools on it) and tests/test_eval_tools.py (which runs ours)."""
import hashlib
import os

import numpy as np

K = 40
# (scenario has a read shorter than K, class2acc options): defaults; per-read report with the label
# strings; repeat split and E-rate filter; with the read profile (per-read / per-window coverages).
# With -p the reference stops at a read shorter than K ("inconsist lengths", exit 1): kept as a case.
ACC_CASES = [
    (True, []),
    (True, ["-e4", "-s", "-m0", "-n100"]),
    (True, ["-r5", "-f50"]),
    (True, ["-e0", "-p{dir}/reads"]),
    (False, ["-e1", "-p{dir}/reads"]),
    (False, ["-w2000", "-p{dir}/reads.prof", "-e0", "-m20", "-n60"]),
]


def build_case(d, labels_fn, tiny=True):
    """labels_fn(seqs, profiles, hcov, dcov) -> list of label bytes per read (len == rlen)."""
    from classpro_amd import synth, fastk
    ds = synth.make_dataset(genome_len=40000, cov=30, read_len=4000, min_len=1500, seed=404, het=0.003)
    seqs, profs, rels, names = list(ds["seqs"]), list(ds["profiles"]), list(ds["rel_profiles"]), list(ds["names"])
    if tiny:
        seqs.insert(2, b"ACGTTGCA"); profs.insert(2, np.zeros(0, np.uint16)); rels.insert(2, np.zeros(0, np.uint16))
        names.insert(2, "tiny")
    comments = [None] * len(seqs)
    comments[1] = "a comment"
    with open(os.path.join(d, "reads.fasta"), "wb") as f:
        for n, s, c in zip(names, seqs, comments):
            f.write(b">" + n.encode() + ((b" " + c.encode()) if c else b"") + b"\n" + s + b"\n")
    fastk.write_fastk(d, "reads", K, profs, ds["hist"], nparts=2)
    fastk.write_fastk(d, "truth", K, rels, ds["hist"], nparts=1)
    low, high, il, ih, h = ds["hist"]
    labels = labels_fn(seqs, profs, ds["hist"])
    last = "(null)"
    with open(os.path.join(d, "est.class"), "wb") as f:
        for n, s, c, lab in zip(names, seqs, comments, labels):
            if c:
                last = c
            f.write(b"@" + n.encode() + b" " + last.encode() + b"\n" + s + b"\n+\n" + lab + b"\n")
    return dict(names=names, seqs=seqs, profiles=profs, rel_profiles=rels, comments=comments)


def oracle_labels(seqs, profs, hist):
    from oracle.oracle import Oracle
    low, high, il, ih, h = hist
    _rc, hc, dc = Oracle(K, 20000, 20, 40).hist_covs(h, low, high, il, ih, 0)
    O = Oracle(K, 20000, hc, dc)
    return [O.classify_read(s, p) if len(s) >= K else b"N" * len(s) for s, p in zip(seqs, profs)]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()
