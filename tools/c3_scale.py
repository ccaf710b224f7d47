"""slow_kernel on C3 against the number of 4 MiB documents: does its time follow the work (throughput-bound) or stay (one long item)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import bench
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0)
for n in (16, 8, 4, 2, 1):
    for max_run in (4096, 256):
        t, o = corpora.multilingual(n, 4 << 20, max_run)
        r, _ = bench.time_resident(ctx, torch, capi, t, o, steps=3, warmup=2)
        c = ctx.last_counters()
        print("docs %2d max_run %4d: ms %.3f slow %.3f merge %.3f scan %.3f tile %.3f window_items %d slow_pretokens %d" % (
            n, max_run, r["ms"], r["kernel_ms"]["slow"], r["kernel_ms"]["merge"], r["kernel_ms"]["scan"], r["kernel_ms"]["tile"], c["window_items"], c["slow_pretokens"]), flush=True)
