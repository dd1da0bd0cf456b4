"""classpro_amd -- MI355X-native per-read k-mer classifier (ClassPro's hot path on gfx950).

The product is the C-ABI library libclasspro_amd.so (include/classpro_amd.h) built from csrc/;
`api` is the Python mirror of the reference's per-read interface, `synth`/`fastk` are input tooling.
"""
__version__ = "0.1.0"
