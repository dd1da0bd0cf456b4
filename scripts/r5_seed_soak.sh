#!/bin/bash
# soak of the -s seed kernels at the END of round 5 (cp_seed_wave.h got the checked views of the -DCP_BOUNDS build this round):
# scripts/fuzz_seeds.py with fresh generator seeds, one process per K (40 / 21 / 63), each for at most SOAK_SECONDS; then the
# seed rate of the 60x set once (scripts/seed_bench.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_seed_soak; rm -rf $O; mkdir -p $O
T=${SOAK_SECONDS:-150}
timeout -k 10 $T python scripts/fuzz_seeds.py 100000 5040 40 > $O/fuzz_k40.log 2>&1 &
timeout -k 10 $T python scripts/fuzz_seeds.py 100000 5021 21 > $O/fuzz_k21.log 2>&1 &
timeout -k 10 $T python scripts/fuzz_seeds.py 100000 5063 63 > $O/fuzz_k63.log 2>&1 &
wait
for k in 40 21 63; do echo "fuzz_k$k: $(grep 'ok so far\|done' $O/fuzz_k$k.log | tail -1)  [lines naming a difference: $(grep -c DIFFERS $O/fuzz_k$k.log)]"; done | tee $O/summary.txt
timeout -k 10 120 python scripts/seed_bench.py > $O/seed_bench.log 2>&1; echo "seed_bench rc=$?" | tee -a $O/summary.txt; tail -3 $O/seed_bench.log | tee -a $O/summary.txt
