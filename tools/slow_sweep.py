"""C3 (multilingual) with different caps on the CJK run length: time of the kernels per batch, slow pre-token count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0); ctx.set_profiling(True)
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (16, 64, 256, 1024, 2048, 4096)
for max_run in runs:
    t, o = corpora.multilingual(n_docs, 4 << 20, max_run)
    for rep in range(3):
        tok, toff = ctx.encode_batch(t, o, True, True)
    tm = ctx.last_timings(); c = ctx.last_counters()
    print("max_run=%5d tok=%9d tile=%.3f merge=%.3f slow=%.3f emit=%.3f total=%.3f slow_n=%d short_n=%d" % (
        max_run, tok.size, tm["tile"], tm["merge"], tm["slow"], tm["emit"], tm["total"], c["slow_pretokens"],
        c["short_pretokens"]), flush=True)
