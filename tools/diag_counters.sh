#!/bin/bash
# diag_counters.sh TAG: two more rocprofv3 passes over the bench command (instruction cache; L1->L2 latency, TLB, L2 hit rate).
set -u
TAG=$1
OUT=gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-c5"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1 || { echo "p1 failed"; tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1 || { echo "p2 failed"; tail -5 $OUT/p2.log; exit 1; }
python3 tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/${TAG}_diag.txt
grep -E "^(tile|merge|emit)_kernel" $OUT/${TAG}_diag.txt
