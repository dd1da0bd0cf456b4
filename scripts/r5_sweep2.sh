#!/bin/bash
# classify_unrel's second sweep restricted to the intervals whose inputs changed (default) against CLASSPRO_UNREL_SWEEP2=full,
# (and, when it was measured, the auxiliary stream at high priority: 215.7-217.0 against 218.5-219.1 Gbases/s, dropped): parity tests first,
# then bench A/B and a kernel trace of each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_neighbours.py -m gpu -x -q 2>&1 | tail -3 || exit 1
run() {   # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s value %.1f Gb/s  step %.2f ms' % ('$name', d['value']/1e3, d['ms_per_step']))"
}
for i in 1 2; do
  run "default" X=1
  run "SWEEP2=full" CLASSPRO_UNREL_SWEEP2=full
done
for v in default full; do
  rm -rf gpurun_out/kt_$v
  case $v in
    default) unset CLASSPRO_UNREL_SWEEP2;;
    full) export CLASSPRO_UNREL_SWEEP2=full;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$v -- python bench.py --steps 3 --warmup 2 --no-cpu --no-extras > gpurun_out/kt_$v.log 2>&1
  f=$(find gpurun_out/kt_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "unrel" in n or "classify_rel" in n or "find_wall" in n or "wall_tasks" in n:
        print("  %-48s calls %4s avg %9.1f us  %5s %%"%(n.split("(")[0].replace("void ","")[:48],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
done
