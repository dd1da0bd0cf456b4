#!/bin/bash
# round 4, first GPU call: the new neighbour / regrowth tests, the fuzz soak, then the whole gpu suite and the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_first; rm -rf $O; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_neighbours.py -x -q > $O/neigh.log 2>&1; echo "neighbours rc=$?"; tail -5 $O/neigh.log
timeout -k 10 400 python scripts/fuzz_parity.py 300 12 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_neighbours.py > $O/pytest.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
