#!/bin/bash
# Produces the rocprofv3 evidence for a round (run on the GPU box through gpurun):
#   gpurun_out/prof_<tag>/bench.json  the default bench line (python bench.py --steps 10 --warmup 3)
#   gpurun_out/prof_<tag>/stats       kernel-trace + stats of the same command (CPU legs and extras off)
#   gpurun_out/prof_<tag>/fetch       PMC pass FETCH_SIZE  (own run: TCC slots, no other trace domains)
#   gpurun_out/prof_<tag>/write       PMC pass WRITE_SIZE
# Copy the summaries you want judged into profiles/ with scripts/summarize_profile.py.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/write.log 2>&1
ls -R $OUT | head -40
