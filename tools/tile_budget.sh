#!/bin/bash
# tile_budget.sh: SQ instruction counters of the tile kernel for the shipped library and the early-exit variants
# exit0..exit7 (tools/build_kvariant.sh exitN -DL3_TILE_EXIT=N), one rocprofv3 pass each.  Run through gpurun from the
# repository root; output: gpurun_out/budget/<name>.txt (tools/pmc_summary.py format, tile_kernel lines only).
set -u
OUT=gpurun_out/budget
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-c5"
for v in base exit1 exit2 exit3 exit4 exit5 exit6 exit7; do
  if [ "$v" = base ]; then unset L3HIP_LIB; else export L3HIP_LIB=$PWD/tokenizer_amd/lib/variants/libl3hip_$v.so; fi
  rm -rf $OUT/raw_$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/raw_$v -- $BENCH > $OUT/$v.log 2>&1 || { echo "$v pass failed"; tail -5 $OUT/$v.log; exit 1; }
  python3 tools/pmc_summary.py $OUT/raw_$v | grep tile_kernel > $OUT/$v.txt
  rm -rf $OUT/raw_$v
  echo "== $v"; cat $OUT/$v.txt
done
