import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0); ctx.set_profiling(True)
t, o = corpora.multilingual(4, 4 << 20)
for i in range(3):
    t0 = time.perf_counter(); tok, toff = ctx.encode_batch(t, o, True, True); t1 = time.perf_counter()
    print(i, "wall %.1f ms" % ((t1 - t0) * 1e3), ctx.last_timings(), ctx.last_counters(), flush=True)
