#!/bin/bash
# Diagnostic: PC sampling (rocprofv3 beta, host trap) of one 800-Mbase run of the whole path; the samples per kernel and
# code offset go to gpurun_out/pcsamp/.  Bounded by a short timeout: this is a beta feature.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pcsamp; rm -rf $O; mkdir -p $O
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${1:-500} --kernel-trace --output-format csv -d $O/out -- python scripts/stage_once.py labels > $O/log.txt 2>&1
echo "rc=$?"; tail -5 $O/log.txt; ls -la $O/out/* | head; 
