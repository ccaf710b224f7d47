#!/bin/bash
# profile_round.sh TAG: the rocprofv3 passes of one round on the GPU box (run it through gpurun from the repository root):
# kernel trace + statistics, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes), two passes of SQ counters
# (instruction mix; LDS bank conflicts against LDS-active cycles).  Raw output under gpurun_out/prof_TAG/, the summaries
# (which are what gets copied to profiles/) beside it.  Every pass is a run of its own; the program follows `--` directly.
set -u
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-c5"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || { echo "sq1 pass failed"; tail -5 $OUT/sq1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || { echo "sq2 pass failed"; tail -5 $OUT/sq2.log; exit 1; }
python3 tools/pmc_hbm_summary.py $OUT/fetch $OUT/write $OUT/${TAG}_pmc_hbm_summary.json > $OUT/hbm.txt
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 > $OUT/${TAG}_sq_counters.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
cat $OUT/hbm.txt; head -12 $OUT/${TAG}_kernel_stats.csv
