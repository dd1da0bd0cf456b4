#!/bin/bash
# Round evidence (run on the GPU box through gpurun; every rocprofv3 command has the program right after `--`):
#   $OUT/bench.json      the default bench line
#   $OUT/stats           rocprofv3 --kernel-trace --stats of the same command (CPU legs and extras off) -> timeline.txt
#   $OUT/fetch, write    PMC passes FETCH_SIZE / WRITE_SIZE (own runs)
#   $OUT/sq_*.txt        SQ counter passes, per kernel and launch (own runs; rocprofv3 --pmc serialises the dispatches whatever
#                        the stream count -- checked on the round-4 traces: 0 overlapping kernels -- so counters are per kernel ALONE;
#                        what kernels do to each other when they overlap: scripts/antagonist_ab.py)
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $OUT/stats.log 2>&1
python scripts/timeline.py $(ls $OUT/stats/*/*_kernel_trace.csv | tail -1) $OUT/timeline.txt --skip=4
echo "stats done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/write.log 2>&1
echo "hbm passes done" >> $OUT/progress.txt
sq_pass() {
  name=$1; shift
  rm -rf $OUT/sq_tmp
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_tmp -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --streams 1 > $OUT/sq_$name.log 2>&1 || true
  python - "$OUT" "$name" "$@" <<'PY'
import csv, glob, collections, sys
out, name = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/sq_tmp/**/*counter_collection.csv", recursive=True)
w = open("%s/sq_%s.txt" % (out, name), "w")
w.write("# rocprofv3 --kernel-trace --pmc %s -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --streams 1\n" % " ".join(sys.argv[3:]))
if not f:
    w.write("no counter output:\n" + open("%s/sq_%s.log" % (out, name)).read()[-1500:])
    raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if not (k.startswith("k_") or k.startswith("void k_")) or "k_sg_" in k or "_table" in k: continue
    w.write(k + "\n")
    for c, v in acc[k].items():
        w.write("    %-28s %16.0f per launch (%d launches)\n" % (c, v / n[(k, c)], n[(k, c)]))
PY
  echo "sq $name done" >> $OUT/progress.txt
}
sq_pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
sq_pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
sq_pass fp64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
sq_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
# round 5: how full the scalar unit is (SQ_ACTIVE_INST_SCA, SQ_INST_CYCLES_SALU against SQ_BUSY_CU_CYCLES) and where the loads hit
sq_pass scalar SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
sq_pass l2 TCC_HIT_sum TCC_MISS_sum
rm -rf $OUT/sq_tmp
ls $OUT
