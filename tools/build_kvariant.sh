#!/bin/bash
# build_kvariant.sh NAME "FLAGS": like build_variant.sh but only l3_kernels.hip is compiled with the extra flags; the other
# objects are the shipped ones (tokenizer_amd/lib/obj, `make` first).  For flags that only l3_kernels.hip reads.
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include $2 -c -o ../lib/variants/k_$1.o l3_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libl3hip_$1.so ../lib/obj/l3_api.o ../lib/variants/k_$1.o ../lib/obj/l3_giant.o ../lib/obj/l3_tables.o
rm -f ../lib/variants/k_$1.o
