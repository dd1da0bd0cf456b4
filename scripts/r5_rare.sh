#!/bin/bash
# the rare-class kernels' durations in the pipeline (kernel trace of the default bench), default library against build_diag/lib_prev.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_interval or compact or second_sweep" 2>&1 | tail -2 || exit 1
for l in default build_diag/lib_prev.so; do
  if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
  rm -rf gpurun_out/kt_rare
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_rare -- python bench.py --steps 3 --warmup 2 --no-cpu --no-extras > gpurun_out/kt_rare.log 2>&1
  f=$(find gpurun_out/kt_rare -name '*kernel_stats.csv' | head -1)
  echo "== $l: $(grep -o '"value": [0-9.]*' gpurun_out/kt_rare.log | head -1)"; python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_","void k_")) and "_table" not in r["Name"] and "k_sg_" not in r["Name"]]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
rare=0
for r in rows:
    n=r["Name"].split("(")[0].replace("void ","")
    sh=100*float(r["TotalDurationNs"])/tot
    israre = n in ("k_classify_rel","k_classify_unrel") or "<112, 1024" in n or "<256, 1024" in n
    rare+= sh if israre else 0
    if israre or sh>3: print("  %-44s calls %4s avg %9.1f us  %5.1f %%%s"%(n[:44],r["Calls"],float(r["AverageNs"])/1e3,sh," (rare)" if israre else ""))
print("  rare classes together: %.1f %% of the summed kernel time"%rare)
PY
done
