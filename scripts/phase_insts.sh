#!/bin/bash
# Diagnostic: instructions per phase of k_wall_tasks / k_find_wall.  Builds of the library whose kernel ends after phase k
# (-DCP_STOP_AT=k -> build_diag/libstop_k.so, made beforehand on the build host) run one sub-batch each under a counter
# pass; the differences between consecutive variants are the phases' instruction counts.
#   bash scripts/phase_insts.sh 0 7 6 1 2 3 4 full
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/phase_insts
rm -rf $OUT && mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = full ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/build_diag/libstop_$v.so; fi
  rm -rf $OUT/tmp
  rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES} --output-format csv -d $OUT/tmp -- python scripts/stage_once.py ${STAGE:-wall} > $OUT/run_$v.log 2>&1 || true
  python - "$OUT" "$v" <<'PY'
import csv, glob, collections, sys
out, v = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/tmp/**/*counter_collection.csv", recursive=True)
w = open("%s/summary.txt" % out, "a")
if not f:
    w.write("variant %s: no counter output\n" % v); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc:
    if not any(x in k for x in ("k_wall_tasks", "k_find_wall", "k_find_rel", "k_classify")): continue
    w.write("variant %-5s %-36s " % (v, k) + " ".join("%s %.1fM" % (c[3:].replace("INSTS_", ""), x / 1e6) for c, x in sorted(acc[k].items())) + "\n")
PY
done
rm -rf $OUT/tmp
cat $OUT/summary.txt
