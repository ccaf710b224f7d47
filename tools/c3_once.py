"""Three encodes of a C3 batch (profiling target for the slow kernel).  usage: c3_once.py [n_docs] [max_run]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0); ctx.set_profiling(True)
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
max_run = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
t, o = corpora.multilingual(n_docs, 4 << 20, max_run)
for rep in range(3):
    tok, toff = ctx.encode_batch(t, o, True, True)
print(ctx.last_timings(), ctx.last_counters())
