#!/bin/bash
# build_kvariant.sh NAME "FLAGS" [UNIT]: like build_variant.sh, but only the kernel unit UNIT (default l3_tile) is compiled
# with the extra flags; the other objects are the shipped ones (tokenizer_amd/lib/obj, `make` first).
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
U=${3:-l3_tile}
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include $2 -c -o ../lib/variants/k_$1.o $U.hip
OBJS=$(ls ../lib/obj/*.o | grep -v "/$U.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libl3hip_$1.so $OBJS ../lib/variants/k_$1.o
rm -f ../lib/variants/k_$1.o
