#!/bin/bash
# Diagnostic builds of the library on the build host (cross-compiles; no GPU): build_diag/<name>.so from extra -D flags.
#   bash scripts/build_diag.sh libseedstop_1 -DCP_SEED_STOP_AT=1
# Point the Python layer at one with CLASSPRO_AMD_LIB=build_diag/<name>.so (travels to the GPU box with gpurun).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_diag
name=$1; shift
cd $ROOT/classpro_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "$@" capi.hip -o $ROOT/build_diag/$name.so 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|generated" || true
ls -la $ROOT/build_diag/$name.so
