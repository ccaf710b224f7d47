#!/bin/bash
# profile_other.sh TAG: rocprofv3 kernel trace + statistics of the C3 (16 x 4 MiB multilingual) and C4 (pathological batch)
# configurations and of the single giant documents (tools/other_timing.py), one run each; SQ counters of the slow kernel on C3.
# Run through gpurun from the repository root; summaries: gpurun_out/other_TAG/TAG_{c3,c4,giants}_kernel_stats.csv, TAG_c3_sq.txt
set -u
TAG=$1
OUT=gpurun_out/other_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for c in c3 c4 giants; do
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -- python3 tools/other_timing.py $c > $OUT/$c.log 2>&1 || { echo "$c trace failed"; tail -5 $OUT/$c.log; exit 1; }
  f=$(find $OUT/trace_$c -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${TAG}_${c}_kernel_stats.csv
  grep -v "^\[\|^W2\|rocprof" $OUT/$c.log | cut -c1-400 > $OUT/${TAG}_${c}_timing.txt
  rm -rf $OUT/trace_$c
done
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 tools/other_timing.py c3 > $OUT/sq1.log 2>&1 || { echo "sq1 failed"; tail -5 $OUT/sq1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- python3 tools/other_timing.py c3 > $OUT/sq2.log 2>&1 || { echo "sq2 failed"; tail -5 $OUT/sq2.log; exit 1; }
# texture-addresser / L1 side of the slow kernel's divergent gathers (prefix-table probes): its own pass, may be refused
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 tools/other_timing.py c3 > $OUT/ta.log 2>&1 || { echo "ta pass failed (continuing)"; tail -3 $OUT/ta.log; rm -rf $OUT/ta; }
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 $( [ -d $OUT/ta ] && echo $OUT/ta ) > $OUT/${TAG}_c3_sq.txt
rm -rf $OUT/ta
rm -rf $OUT/sq1 $OUT/sq2
head -8 $OUT/${TAG}_c3_kernel_stats.csv; head -8 $OUT/${TAG}_c4_kernel_stats.csv
