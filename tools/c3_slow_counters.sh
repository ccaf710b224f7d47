#!/bin/bash
# c3_slow_counters.sh: per-dispatch counters of slow_kernel on C3 (tools/other_timing.py c3), two rocprofv3 passes; prints
# every dispatch of the kernel (the first of a run grows the workspace and is not representative).
set -u
OUT=gpurun_out/c3slow
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 tools/other_timing.py c3 > $OUT/ta.log 2>&1 || { echo "ta pass failed"; tail -3 $OUT/ta.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 tools/other_timing.py c3 > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -3 $OUT/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/l2 -- python3 tools/other_timing.py c3 > $OUT/l2.log 2>&1 || { echo "l2 pass failed"; tail -3 $OUT/l2.log; }
python3 - <<'PY'
import csv, glob, collections
for d in ("ta", "sq", "l2"):
    rows = collections.OrderedDict()
    for f in glob.glob("gpurun_out/c3slow/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "slow_kernel" not in r["Kernel_Name"]:
                continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["grid"] = r.get("Grid_Size", "")
    print("==", d)
    for k, v in rows.items():
        print(k, {a: (b if isinstance(b, str) else "%.4g" % b) for a, b in v.items()})
PY
