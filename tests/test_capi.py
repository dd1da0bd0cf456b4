"""The C-ABI library builds for gfx950, loads, and exports every symbol include/classpro_amd.h declares.
No compute calls here (no GPU in this container)."""
import numpy as np
import os
import re

from conftest import ROOT


def test_library_exports_header_symbols(built):
    from classpro_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "classpro_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cp_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libclasspro_amd.so does not export %s" % name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert b"gfx950" in L.cp_version()


def test_host_entry_points_without_gpu(built):
    """cp_hist_covs / cp_decode_profile are host functions of the ABI: usable without a device."""
    import numpy as np
    from classpro_amd.api import hist_covs, decode_profile
    from classpro_amd import fastk
    from conftest import load_golden
    g = load_golden("fastk.npz")
    assert hist_covs(g["hist"], int(g["low"]), int(g["high"]), int(g["ilow"]), int(g["ihigh"]), 0) == (int(g["covs"][0]), int(g["covs"][1]))
    c = np.array([5, 5, 5, 9, 200, 200, 1, 32767, 3], np.uint16)
    n, out = decode_profile(fastk.encode_profile(c))
    assert n == len(c) and np.array_equal(out, c)


def test_product_does_not_touch_oracle():
    """The product path must never import, link or call anything under oracle/."""
    for root, _d, files in os.walk(os.path.join(ROOT, "classpro_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "from oracle" not in txt and "import oracle" not in txt and "classpro_oracle" not in txt, f


def test_profile_encoder_matches_fastk_writer_and_round_trips(built):
    """cp_encode_profile (tooling) == classpro_amd.fastk.encode_profile == inverse of cp_decode_profile,
    including 15-bit deltas, negative small deltas and runs longer than one token."""
    from classpro_amd import fastk
    from classpro_amd.api import encode_profiles, decode_profile
    rng = np.random.default_rng(11)
    ps = [np.array([5], np.uint16), np.array([200] * 300, np.uint16),
          np.array([1, 32767, 0, 31, 0, 63, 31, 32, 0, 127, 128], np.uint16)]
    for _ in range(40):
        n = int(rng.integers(2, 400))
        c = rng.integers(0, 60, n)
        c[rng.integers(0, n, 3)] = rng.integers(0, 32768, 3)
        ps.append(np.repeat(c, rng.integers(1, 90, n))[:n].astype(np.uint16))
    codes, off = encode_profiles(ps)
    for i, p in enumerate(ps):
        code = codes[off[i]:off[i + 1]].tobytes()
        assert code == fastk.encode_profile(p)
        n, out = decode_profile(code)
        assert n == len(p) and np.array_equal(out, p)
