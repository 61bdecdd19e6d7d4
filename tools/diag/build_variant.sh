#!/bin/bash
# Diagnostic build of the library with extra compiler flags for the key-switch translation unit, e.g.
#   tools/diag/build_variant.sh nostore -DMOAI_DIAG_NOSTORE      (strided passes: arithmetic without the stores)
#   tools/diag/build_variant.sh nocompute -DMOAI_DIAG_NOCOMPUTE  (strided passes: loads, exchange and stores without the butterflies)
# Results of such a build are wrong by construction; it exists to time the halves of a kernel (tools/diag/run_variants.sh on the
# GPU box, which loads it through MOAI_HIP_LIB).  Needs the regular build's object files (make -C .../csrc).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/moai-fhe-transformerinference-public_amd/csrc
out=$(mktemp -d)
for f in context ntt elementwise encoder; do cp $src/$f.o $out/$f.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $src/keyswitch.hip -o $out/keyswitch.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/diag/libmoai_hip_$name.so $out/*.o
rm -rf $out
ls -la $root/tools/diag/libmoai_hip_$name.so
