#!/bin/bash
# Diagnostic: wait / busy cycle counters of k_find_seeds for truncated builds (see seed_phase_insts.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/seed_phase_waits.txt
rm -f $OUT
for v in "$@"; do
  if [ "$v" = full ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/build_diag/libseedstop_$v.so; fi
  echo "variant $v" >> $OUT
  bash scripts/pmc_seed.sh SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA >> $OUT 2>&1
  bash scripts/pmc_seed.sh SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_BRANCH >> $OUT 2>&1
done
cat $OUT
