"""Kernel times of the C3 / C4 configurations, device-resident (what bench.py reports as other_configs), one line each.
usage: other_timing.py [c3] [c4] [c4docs] [giants]   (L3HIP_LIB selects a library variant, see tools/build_variant.sh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import bench  # noqa: E402
from tokenizer_amd import capi  # noqa: E402
from tools import corpora  # noqa: E402

what = sys.argv[1:] or ["c3", "c4", "giants"]
ctx = capi.Context(0)
cases = []
if "c3" in what:
    cases.append(("c3_16x4MiB", corpora.multilingual(16, 4 << 20), 5))
if "c4" in what:
    cases.append(("c4_batch", capi.pack_docs(corpora.pathological(1 << 20)), 3))
if "giants" in what:
    for name, d in (("1MiB_a", b"a" * (1 << 20)), ("1MiB_0xFF", b"\xff" * (1 << 20)), ("1MiB_blank", b" " * (1 << 20)),
                    ("1MiB_acgt", (b"acgtgca" * 150000)[:1 << 20])):
        cases.append((name, capi.pack_docs([d]), 3))
C4_NAMES = ["blanks", "tabs", "nl_sp", "digits", "one_letter", "apostrophe_s", "random_bytes", "invalid_sequences", "bot_every_64",
            "all_specials", "near_misses", "unicode_mix", "ff_quarter", "bangs_newlines"]
if "c4docs" in what:   # every document of the C4 batch on its own
    for name, d in zip(C4_NAMES, corpora.pathological(1 << 20)):
        cases.append(("c4:" + name, capi.pack_docs([d]), 2))
for name, (t, o), steps in cases:
    r, _ = bench.time_resident(ctx, torch, capi, t, o, steps=steps, warmup=2)
    print("%s %-12s ms %.3f kernels %.3f tok %d" % (os.path.basename(os.environ.get("L3HIP_LIB", "base")), name, r["ms"], r["kernels_ms"], r["tokens"]),
          {k: round(v, 3) for k, v in r["kernel_ms"].items()}, ctx.last_counters(), flush=True)
