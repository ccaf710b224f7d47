#!/bin/bash
# build_variant.sh NAME "FLAGS": builds tokenizer_amd/lib/variants/libl3hip_NAME.so with extra compile flags (A/B timing).
# Use it with L3HIP_LIB=tokenizer_amd/lib/variants/libl3hip_NAME.so.  -DL3_DEBUG_FLAGS compiles in the
# phase-skipping knobs tools/phase_timing.py needs (results are then wrong on purpose).  Delete variants after use.
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
mkdir -p ../lib/variants ../lib/obj
HIPCC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include"
$HIPCC $2 -c -o ../lib/variants/k_$1.o l3_kernels.hip
$HIPCC $2 -c -o ../lib/variants/a_$1.o l3_api.hip
make -s ../lib/obj/l3_tables.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libl3hip_$1.so ../lib/variants/a_$1.o ../lib/variants/k_$1.o ../lib/obj/l3_tables.o
rm -f ../lib/variants/k_$1.o ../lib/variants/a_$1.o
