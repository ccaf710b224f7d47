#!/bin/bash
# build_variant.sh NAME "FLAGS": builds tokenizer_amd/lib/variants/libl3hip_NAME.so with extra compile flags (A/B timing)
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
mkdir -p ../lib/variants ../lib/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c -o ../lib/variants/k_$1.o l3_kernels.hip
make -s ../lib/obj/l3_api.o ../lib/obj/l3_tables.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libl3hip_$1.so ../lib/obj/l3_api.o ../lib/variants/k_$1.o ../lib/obj/l3_tables.o
rm -f ../lib/variants/k_$1.o
