"""Registers / scratch / spill instructions of every device FUNCTION (kernels and non-inlined callees) of the library,
from the compiler's assembly (the resource-usage remarks only cover kernels).   python scripts/func_regs.py [filter] [hipcc flags]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "classpro_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                    "--save-temps=obj", "capi.hip", "-o", os.path.join(d, "lib.so")] + sys.argv[2:], cwd=csrc, capture_output=True)
    asm = open(os.path.join(d, "capi-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    if os.environ.get("KEEP_ASM"):
        open(os.environ["KEEP_ASM"], "w").write(asm)
for m in re.finditer(r"^(_Z\w+):\s+; @.*?\n(.*?)^\.Lfunc_end\d+:\n(.*?)(?=^\t\.(?:text|section|globl|p2align|protected|type|weak|hidden))", asm, re.S | re.M):
    name, body, tail = m.group(1), m.group(2), m.group(3)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if flt not in dn:
        continue
    def g(k):
        x = re.search(r"; " + k + r": (\d+)", tail)
        return x.group(1) if x else "?"
    print("%-40s lines %6d VGPR %4s SGPR %4s scratch %4s | scratch ld/st %4d  v_readlane %4d v_writelane %4d" % (
        dn[:40], body.count("\n"), g("NumVgprs"), g("NumSgprs"), g("ScratchSize"),
        len(re.findall(r"\tscratch_(load|store)", body)), body.count("v_readlane_b32"), body.count("v_writelane_b32")))
