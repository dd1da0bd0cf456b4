#!/bin/bash
# Diagnostic: instructions per phase of k_find_seeds from builds whose selections end after phase k
# (-DCP_SEED_STOP_AT=k -> build_diag/libseedstop_k.so); differences of consecutive variants = the phases.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/seed_phase_insts.txt
rm -f $OUT
for v in "$@"; do
  if [ "$v" = full ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/build_diag/libseedstop_$v.so; fi
  echo "variant $v" >> $OUT
  bash scripts/pmc_seed.sh SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES >> $OUT 2>&1
  grep "seeds pass 2" gpurun_out/pmc_seed.log >> $OUT
done
cat $OUT
