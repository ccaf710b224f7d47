#!/bin/bash
# build_variant.sh NAME "FLAGS": builds tokenizer_amd/lib/variants/libl3hip_NAME.so with extra compile flags for A/B
# timing (tools/ab.sh NAME ... runs bench.py with each).  Knobs the sources honour: -DL3_TILE_BYTES (3840),
# -DL3_TILE_OCC (7), -DL3_LOOK_U (4), -DL3_EMIT_OCC (6), -DL3_EMIT_PER (4), -DL3_MERGE_THREADS (256).
# Delete the variants after use: they travel to the GPU box with every gpurun call.
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
mkdir -p ../lib/variants ../lib/obj
HIPCC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include"
$HIPCC $2 -c -o ../lib/variants/k_$1.o l3_kernels.hip
$HIPCC $2 -c -o ../lib/variants/a_$1.o l3_api.hip
$HIPCC $2 -c -o ../lib/variants/g_$1.o l3_giant.hip
make -s ../lib/obj/l3_tables.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libl3hip_$1.so ../lib/variants/a_$1.o ../lib/variants/k_$1.o ../lib/variants/g_$1.o ../lib/obj/l3_tables.o
rm -f ../lib/variants/k_$1.o ../lib/variants/a_$1.o ../lib/variants/g_$1.o
