#!/bin/bash
# c3_tile_counters.sh: SQ counters of tile_kernel on C3 (tools/other_timing.py c3), one rocprofv3 pass; run through gpurun.
set -u
OUT=gpurun_out/c3tile
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 tools/other_timing.py c3 > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -3 $OUT/sq.log; exit 1; }
python3 tools/pmc_summary.py $OUT/sq | grep tile_kernel
