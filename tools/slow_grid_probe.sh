#!/bin/bash
# slow_grid_probe.sh V...: C3 timing with each variant library libl3hip_V.so, then SQ_WAVES / SQ_WAVE_CYCLES of slow_kernel with the last one
for v in "$@"; do export L3HIP_LIB=$PWD/tokenizer_amd/lib/variants/libl3hip_$v.so; timeout -k 10 120 python -u tools/other_timing.py c3 2>/dev/null | cut -c1-190 || exit 1; done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/sgp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/sgp -- python3 tools/other_timing.py c3 > gpurun_out/sgp.log 2>&1
python3 - <<'PY'
import csv,glob,collections
rows=collections.OrderedDict()
for f in glob.glob("gpurun_out/sgp/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "slow_kernel" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"]); rows[int(r["Dispatch_Id"])]["grid"]=r["Grid_Size"]
for k,v in list(rows.items())[-3:]: print(k,v)
PY
