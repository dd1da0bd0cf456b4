#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_pack; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pack.py tests/test_gpu_parity.py -x -q -k "pack or unpack or dazz or db or cli or track" > $O/pytest.log 2>&1; echo "pack tests rc=$?"; tail -3 $O/pytest.log
python bench.py --steps 2 --warmup 1 --no-cpu --only-pcie > $O/pcie.json 2> $O/pcie.err; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4_pack/pcie.json").read().strip().splitlines()[-1]); p=j["extras"]["pcie"]; print(p["mbases_per_s"], p["link_gb_per_s_in_plus_out"], p["frac"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 1 --warmup 1 --no-cpu --only-pcie --pcie-seconds 0.3 > $O/trace.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4_pack/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print("%-50s %5s %10.1f us avg  %5.1f%%" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
