#!/bin/bash
# experiment helper: tests/mkvariant_obj.sh <name> <source stem, e.g. lisreg_index> [extra hipcc flags] — like tests/mkvariant.sh for a
# translation unit other than lisreg_assoc.hip: lib/variants/liblisreg_<name>.so = the regular objects with <stem>.o rebuilt under the flags.
set -e
N=$1; STEM=$2; shift; shift
cd "$(dirname "$0")/../lis-slam_amd/csrc"
make -s
mkdir -p ../lib/variants
FLAGS=""; case $STEM in lisreg_solve|lisreg_nn1|lisreg_features) FLAGS="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $FLAGS "$@" -c $STEM.hip -o ../lib/variants/${STEM}_$N.o
OBJS=$(ls ../lib/*.o | grep -v "$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/liblisreg_$N.so $OBJS ../lib/variants/${STEM}_$N.o -ldl
rm ../lib/variants/${STEM}_$N.o
echo built variants/liblisreg_$N.so
