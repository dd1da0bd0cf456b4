#!/bin/bash
# Diagnostic: per-kernel average durations of the default bench under rocprofv3 --kernel-trace, one stream (kernels alone
# on the machine: the low-noise figure to optimise a kernel against) and two (the pipeline), for each library given.
#   bash scripts/kt.sh default build_diag/lib_x.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/kt
for l in "$@"; do
  if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
  for ns in ${KT_STREAMS:-1 2}; do
    rm -rf gpurun_out/kt/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt/tmp -- python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --streams $ns > gpurun_out/kt/log.txt 2>&1
    python - "$l" "$ns" <<'PY'
import csv,glob,sys,json
f=glob.glob("gpurun_out/kt/tmp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
keep=[r for r in rows if any(x in r["Name"] for x in ("k_wall_tasks","k_find_wall","k_find_rel","k_classify_rel_grp<0","k_classify_unrel_grp<0","k_paint","k_scan")) ]
try:
    j=[json.loads(l) for l in open("gpurun_out/kt/log.txt") if l.startswith("{")][-1]; v="%.1f Gb/s" % (j["value"]/1e3)
except Exception: v="?"
print("%-28s streams=%s %s | " % (sys.argv[1][-28:], sys.argv[2], v) + "  ".join("%s %.0f" % (r["Name"].split("(")[0].replace("void ","").replace("k_","")[:22], float(r["AverageNs"])/1e3) for r in sorted(keep,key=lambda r:r["Name"])))
PY
  done
done
