#!/bin/bash
# experiment helper (not part of the product build): tests/mkvariant.sh <name> [extra hipcc flags for lisreg_assoc.hip]
# builds lis-slam_amd/lib/variants/liblisreg_<name>.so from the CURRENT lisreg_assoc.hip (production arithmetic) + the other objects
# of the regular build; tests/ab.sh then times the variants back to back on one GPU box.
set -e
N=$1; shift
cd "$(dirname "$0")/../lis-slam_amd/csrc"
make -s
mkdir -p ../lib/variants
SRC=${ASSOC_SRC:-lisreg_assoc.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=on -fno-slp-vectorize "$@" -c $SRC -o ../lib/variants/assoc_$N.o
OBJS=$(ls ../lib/*.o | grep -v "lisreg_assoc.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/liblisreg_$N.so $OBJS ../lib/variants/assoc_$N.o -ldl
rm ../lib/variants/assoc_$N.o
echo built variants/liblisreg_$N.so
