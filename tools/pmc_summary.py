"""Per-kernel averages of rocprofv3 --pmc passes (csv output).  usage: pmc_summary.py DIR [DIR ...]
Also prints VGPR / LDS / scratch of every kernel (from the dispatch records)."""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
meta = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"::(\w+_kernel)\(", r["Kernel_Name"]) or re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
for k in sorted(meta):
    print("%-22s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % ((k,) + meta[k]))
for (k, c), (s, n) in sorted(acc.items()):
    print("%-22s %-24s %.4e  (n=%d)" % (k, c, s / n, n))
# LDS bank conflicts as a share of the cycles the LDS was busy (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
for k in sorted(meta):
    c, a = acc.get((k, "SQ_LDS_BANK_CONFLICT")), acc.get((k, "SQ_LDS_IDX_ACTIVE"))
    if c and a and a[0] > 0:
        print("%-22s LDS conflict cycles / LDS active cycles = %.3f" % (k, (c[0] / c[1]) / (a[0] / a[1])))
