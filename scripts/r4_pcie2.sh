#!/bin/bash
# the PCIe pipeline: default twice, the whole 8-Gbase set once, and a memory-copy trace of a short run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_pcie2; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=j["extras"].get("pcie") or j["extras"]
    print(sys.argv[2], json.dumps(p))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
}
B="python bench.py --steps 2 --warmup 1 --no-cpu --only-pcie"
run base $B
run base_again $B
run full8 $B --pcie-gbases 8
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 1 --warmup 1 --no-cpu --only-pcie --pcie-seconds 0.3 > $O/trace.log 2>&1
echo "trace rc=$?"; ls $O/trace/*/ | head
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4_pcie2/trace/**/*memory_copy_stats.csv",recursive=True)
for x in f:
    print(open(x).read()[:1500])
PY
