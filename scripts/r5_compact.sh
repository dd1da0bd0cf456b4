#!/bin/bash
# whole-path calls with compact reliable-interval records (default) against CLASSPRO_COMPACT_REL=0: parity tests, bench A/B, bytes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_neighbours.py tests/test_gpu_pack.py -m gpu -x -q 2>&1 | tail -3
for v in 2 1 0 2 1 0; do
  CLASSPRO_COMPACT_REL=$v python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('COMPACT_REL=$v: value %.1f Gb/s  step %.2f ms' % (d['value']/1e3, d['ms_per_step']))"
done
for v in 2 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_tmp
    CLASSPRO_COMPACT_REL=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_tmp -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras > gpurun_out/pmc_tmp.log 2>&1
    python - $v $c <<'PY'
import csv,glob,collections,sys
f=glob.glob("gpurun_out/pmc_tmp/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace(" ","")
    acc[k].append(float(r["Counter_Value"]))
tot=0
out=[]
for k,v in acc.items():
    if not k.startswith("k_") or "_table" in k or k.startswith("k_sg_"): continue
    n=4 if k!="k_scan_candidates" else None
    per = sum(v)/len(v)*(4 if k=="k_scan_candidates" else 1) if k=="k_scan_candidates" else sum(v)/4
    tot+=per
    if any(x in k for x in ("find_wall","rel_grp<0","unrel_grp<0")): out.append("%s %.2f GB"%(k.replace("k_classify_","")[:18],per*1024/1e9))
print("COMPACT_REL=%s %s: sum per 4-Gbase sub-batch %.2f GB (scan as reported) | "%(sys.argv[1],sys.argv[2],tot*1024/1e9)+"  ".join(out))
PY
  done
done
