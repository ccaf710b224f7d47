"""Per-phase wall cycles of the tile kernel on the C2 batch (argument c3: the C3 batch), from a -DL3_PHASE_PROF build:
    tools/build_variant.sh phase -DL3_PHASE_PROF
    L3HIP_LIB=$PWD/tokenizer_amd/lib/variants/libl3hip_phase.so python tools/phase_prof.py
Cycles are per workgroup (thread 0's clock between the barriers that end the phases), averaged over all tiles: with 7
workgroups sharing a CU they show where a tile WAITS as much as where it computes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizer_amd import capi  # noqa: E402
from tools import corpora  # noqa: E402

NAMES = ["0 load + LDS set-up", "1 doc bits + ASCII classes", "2 non-ASCII, apostrophes, specials, mask store", "3 compute_starts (wave 0)",
         "4 list compaction", "5 vocabulary probes", "6 generic loop (rare)", "7 list space atomics + doc items", "8 stream + entry records",
         "9 (of 2) thread 0: UTF-8 walk", "10 (of 2) thread 0: classes of its characters"]


def main():
    ctx = capi.Context(0)
    text, off = corpora.multilingual(16, 4 << 20) if "c3" in sys.argv[1:] else corpora.english(65536, 1024)
    ctx.encode_batch(text, off)
    out = (C.c_ulonglong * 16)()
    ctx.lib.l3_debug_phases(out, 16, 1)
    reps = 5
    for _ in range(reps):
        ctx.encode_batch(text, off)
    ctx.lib.l3_debug_phases(out, 16, 1)
    tiles = ctx.last_counters()["tiles"] * reps
    v = np.array(list(out)[:len(NAMES)], dtype=np.float64) / tiles
    for n, x in zip(NAMES, v):
        print("%-50s %8.0f cycles  %5.1f %%" % (n, x, 100 * x / v.sum()))
    print("%-50s %8.0f cycles per tile" % ("sum", v.sum()))


if __name__ == "__main__":
    main()
