"""Times the C3 / C4 configurations document by document (host API) and checks each against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokenizer_amd import capi
from tools import corpora
from tests.oracle import Oracle
ctx = capi.Context(0); ctx.set_profiling(True)
orc = Oracle()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
def run(name, text, off, threads=16):
    t0 = time.perf_counter(); tok, toff = ctx.encode_batch(text, off, True, True); t1 = time.perf_counter()
    tm = ctx.last_timings(); c = ctx.last_counters()
    e_tok, e_off = orc.encode_batch(text, off, True, True, threads=threads); t2 = time.perf_counter()
    ok = np.array_equal(tok, e_tok) and np.array_equal(toff, e_off)
    print("%-28s bytes=%9d tok=%9d ok=%s gpu_wall=%.1fms tile=%.2f slow=%.2f emit=%.2f slow_n=%d retries=%d oracle=%.0fms" % (
        name, text.size, tok.size, ok, (t1 - t0) * 1e3, tm["tile"], tm["slow"], tm["emit"], c["slow_pretokens"], c["retries"], (t2 - t1) * 1e3), flush=True)
names = ["spaces", "tabs", "nl-sp", "digits", "a*", "'s*", "random bytes", "bad utf8", "english+BOT", "all specials", "near misses", "unicode ws/N", "0xFF*", "!*+nl"]
docs = corpora.pathological(size)
for nm, d in zip(names, docs):
    t, o = capi.pack_docs([d]); run("C4 " + nm, t, o, 1)
t, o = capi.pack_docs(docs); run("C4 batch", t, o)
t, o = corpora.multilingual(1, 4 << 20); run("C3 1x4MiB", t, o, 1)
t, o = corpora.multilingual(16, 4 << 20); run("C3 16x4MiB", t, o)
t, o = corpora.english(65536, 1024); run("C2 64MiB", t, o)
