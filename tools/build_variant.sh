#!/bin/bash
# build_variant.sh NAME "FLAGS": builds tokenizer_amd/lib/variants/libl3hip_NAME.so with extra compile flags in every HIP unit,
# for A/B timing (tools/ab.sh NAME ... runs bench.py with each).  Knobs the sources honour: -DL3_TILE_BYTES (3840),
# -DL3_TILE_OCC (7), -DL3_LOOK_U (4), -DL3_EMIT_OCC (8), -DL3_MERGE_THREADS (256), -DL3_SLOW_OCC (1), -DL3_TILE_EXIT=i,
# -DL3_PHASE_PROF.  Delete the variants after use: they travel to the GPU box with every gpurun call.
set -e
cd "$(dirname "$0")/../tokenizer_amd/csrc"
V=../lib/variants
mkdir -p $V/obj_$1 $V/out_$1
make -s -j8 OBJ=$V/obj_$1 OUT=$V/out_$1 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2" $V/out_$1/libl3hip.so
mv $V/out_$1/libl3hip.so $V/libl3hip_$1.so
rm -rf $V/obj_$1 $V/out_$1
