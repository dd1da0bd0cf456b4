#!/bin/bash
# soak at the END of round 5 (after the on-chip memo / E-interval / off-list SELF wall changes of k_find_wall and the word forms of
# the sequence contexts): adversarial + tail-run reads over fresh seeds against the oracle, four processes side by side
# (two with K = 40 / -r 20000, two with K and -r changing by the seed), each for at most SOAK_SECONDS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_soak_head; rm -rf $O; mkdir -p $O
T=${SOAK_SECONDS:-270}
timeout -k 10 $T python scripts/fuzz_parity.py 9000 400 > $O/fuzz_a.log 2>&1 &
timeout -k 10 $T python scripts/fuzz_parity.py 9400 400 > $O/fuzz_b.log 2>&1 &
timeout -k 10 $T python scripts/fuzz_parity.py 11000 400 params > $O/fuzz_pa.log 2>&1 &
timeout -k 10 $T python scripts/fuzz_parity.py 11400 400 params > $O/fuzz_pb.log 2>&1 &
wait
for f in a b pa pb; do echo "fuzz_$f: $(tail -1 $O/fuzz_$f.log)  [lines naming a difference: $(grep -c 'differs\|did not reject\|wrong error' $O/fuzz_$f.log)]"; done
