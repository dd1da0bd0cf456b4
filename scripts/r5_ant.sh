#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ant && mkdir -p gpurun_out/ant
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ant -- python scripts/antagonist_ab.py run > gpurun_out/ant/run.log 2>&1
echo "rc=$?"; tail -20 gpurun_out/ant/run.log
python scripts/antagonist_ab.py report $(ls gpurun_out/ant/*/*_kernel_trace.csv | tail -1) gpurun_out/ant/report.txt
