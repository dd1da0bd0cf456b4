#!/bin/bash
# soak after the round-5 work reductions (second-sweep rule, multi-error pre-filter, unrel sort): adversarial + tail-run reads over
# many seeds against the oracle, with fixed and with varying K / -r; then the bounds-check build on the parity tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_soak; rm -rf $O; mkdir -p $O
timeout -k 10 420 python scripts/fuzz_parity.py 5000 400 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log; grep -c "differs\|did not reject\|wrong error" $O/fuzz.log
timeout -k 10 420 python scripts/fuzz_parity.py 7000 400 params > $O/fuzz_params.log 2>&1; echo "fuzz params rc=$?"; tail -1 $O/fuzz_params.log; grep -c "differs\|did not reject\|wrong error" $O/fuzz_params.log
