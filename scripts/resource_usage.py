"""Per-kernel registers / spills / occupancy / LDS of the product library (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/resource_usage.py [extra hipcc flags]     (cross-compiles: no GPU needed)
"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "classpro_amd", "csrc")
with tempfile.TemporaryDirectory() as d:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                        "-Rpass-analysis=kernel-resource-usage", "capi.hip", "-o", os.path.join(d, "lib.so")] + sys.argv[1:],
                       cwd=csrc, capture_output=True, text=True)
t = p.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split(" ")[0]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    print("%-44s VGPR %4s AGPR %3s SGPR %4s spillS %3s spillV %3s scratch %4s occ %2s LDS %6s" % (
        dn[:44], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
