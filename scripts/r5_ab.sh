#!/bin/bash
# A/B inside one call: bench value, step and the per-kernel averages of the traced run (two streams), for each library given
#   bash scripts/r5_ab.sh default build_diag/lib_x.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab5
for l in "$@"; do
  if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
  for i in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s value %.1f Gb/s  step %.2f ms' % ('$l', d['value']/1e3, d['ms_per_step']))"
  done
  rm -rf gpurun_out/ab5/tmp
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab5/tmp -- python bench.py --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/ab5/log.txt 2>&1
  python - "$l" <<'PY'
import csv,glob,collections,sys
f=sorted(glob.glob('gpurun_out/ab5/tmp/*/*_kernel_trace.csv'))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace(" ","")
    d[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
keys=["k_wall_tasks","k_find_wall","k_classify_rel_grp<0,112,4,4>","k_classify_unrel_grp<0,256,2>","k_classify_rel","k_classify_rel_grp<112,1024,1,1>","k_classify_unrel","k_classify_unrel_grp<256,1024,1>","k_paint_labels"]
print("   traced:", "  ".join("%s %.0f" % (k.replace("k_classify_","").replace("k_","")[:18], sum(d[k])/max(1,len(d[k]))) for k in keys))
PY
done
