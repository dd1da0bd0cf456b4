#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_soak; rm -rf $O; mkdir -p $O
timeout -k 10 700 python scripts/fuzz_parity.py 400 60 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log; grep -c "differs\|did not reject\|wrong error" $O/fuzz.log
timeout -k 10 300 python scripts/fuzz_seeds.py 900 > $O/fuzz_seeds.log 2>&1; echo "fuzz_seeds rc=$?"; tail -1 $O/fuzz_seeds.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
