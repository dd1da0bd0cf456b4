#!/bin/bash
# kernel trace of the default bench -> per-stream timeline (scripts/timeline.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-trace}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $O/stats.log 2>&1
python scripts/timeline.py $(ls $O/stats/*/*_kernel_trace.csv | tail -1) $O/timeline.txt --skip=4
sed -n 1,40p $O/timeline.txt
