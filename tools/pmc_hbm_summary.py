"""HBM bytes per launch of every kernel from two rocprofv3 passes (csv output):
     pmc_hbm_summary.py FETCH_DIR WRITE_DIR OUT.json
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  On gfx950 FETCH_SIZE counts wide coalesced reads at half their
bytes (MI355X_MICROARCH.md, HBM section), so bytes = 2 * FETCH_SIZE + WRITE_SIZE; for kernels whose reads are
random 16-byte probes (merge_kernel) the factor is uncalibrated and the figure is an upper bound."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"::(\w+_kernel)\(", r["Kernel_Name"])
            if m:
                a = acc[m.group(1)]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
import bench  # noqa: E402  (source_digest: bench.py quotes a summary only if it was taken from the same kernel sources)

out = {"source_digest": bench.source_digest(), "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs (tools/profile_round.sh)",
       "note": "gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md), so bytes = "
               "2*FETCH_SIZE + WRITE_SIZE; counter unit KB", "kernels": {}}
for k in fetch:
    f, nf = fetch[k]
    w, nw = write.get(k, [0.0, 1])
    out["kernels"][k] = {"FETCH_SIZE_KB_avg": round(f / nf, 1), "dispatches_FETCH_SIZE": nf,
                         "WRITE_SIZE_KB_avg": round(w / max(nw, 1), 1), "dispatches_WRITE_SIZE": nw,
                         "hbm_bytes_per_launch_corrected": int(2 * f / nf * 1024 + w / max(nw, 1) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-22s fetch %.1f KB  write %.1f KB  -> %.1f MB" % (k, v["FETCH_SIZE_KB_avg"], v["WRITE_SIZE_KB_avg"], v["hbm_bytes_per_launch_corrected"] / 1e6))
