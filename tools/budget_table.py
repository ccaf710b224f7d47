#!/usr/bin/env python3
"""budget_table.py DIR [N_TILES]: per-phase counters of the tile kernel from the outputs of tools/tile_budget.sh
(differences between consecutive early-exit variants), per tile."""
import sys
d0 = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 17477
names = ['exit1', 'exit2', 'exit3', 'exit4', 'exit5', 'exit6', 'exit7', 'base']
label = {'exit1': 'load, LDS set-up (+ what the compiler keeps of the class table step)', 'exit2': 'ASCII classes by table, non-ASCII, contractions, specials, mask stores',
         'exit3': 'start masks (one wave)', 'exit4': 'list of own starts', 'exit5': 'probes, stage 1', 'exit6': 'stage 2 (second bucket, rare kinds)',
         'exit7': 'list-space atomics, document items', 'base': 'deferred records / write-out'}
rows = {}
for n in names:
    d = {}
    for l in open('%s/%s.txt' % (d0, n)):
        p = l.split()
        if len(p) >= 3 and p[1].startswith('SQ_'):
            d[p[1]] = float(p[2])
    rows[n] = d
prev = {k: 0 for k in rows['base']}
print("| phase | VALU | SALU | LDS | wave-cycles (10^6, whole launch) |")
print("|---|---|---|---|---|")
for n in names:
    d = rows[n]
    print("| %s | %.0f | %.0f | %.0f | %.1f |" % (label[n], (d['SQ_INSTS_VALU'] - prev['SQ_INSTS_VALU']) / T, (d['SQ_INSTS_SALU'] - prev['SQ_INSTS_SALU']) / T,
                                             (d['SQ_INSTS_LDS'] - prev['SQ_INSTS_LDS']) / T, (d['SQ_WAVE_CYCLES'] - prev['SQ_WAVE_CYCLES']) / 1e6))
    prev = d
d = rows['base']
print("| **whole kernel** | %.0f | %.0f | %.0f | %.1f |" % (d['SQ_INSTS_VALU'] / T, d['SQ_INSTS_SALU'] / T, d['SQ_INSTS_LDS'] / T, d['SQ_WAVE_CYCLES'] / 1e6))
print()
print("SQ_BUSY_CYCLES %.4g, SQ_ACTIVE_INST_VALU %.4g" % (d['SQ_BUSY_CYCLES'], d.get('SQ_ACTIVE_INST_VALU', 0)))
