import os, sys
os.environ["SOAK_BIG"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tools import fuzz_soak as fs
from tests.oracle import Oracle
from tokenizer_amd import capi

orc = Oracle()
def batches(upto):
    fs.rng = np.random.default_rng(102)
    return [fs.gen_batch() for _ in range(upto)]
B = batches(25)
docs, bos, eos = B[24]
text, off = capi.pack_docs(docs)
exp_tok, exp_off = orc.encode_batch(text, off, bos, eos, threads=16)
print("round 25: docs", len(docs), "bytes", text.size, "bos/eos", bos, eos, flush=True)

def check(ctx, tag):
    got_tok, got_off = ctx.encode_batch(text, off, bos, eos)
    ok = np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok)
    print(tag, "equal" if ok else "DIFFERENT", ctx.last_counters(), flush=True)
    if not ok:
        print("  offsets equal:", np.array_equal(got_off, exp_off), " sizes", got_tok.size, exp_tok.size)
        if not np.array_equal(got_off, exp_off):
            d = int(np.nonzero(got_off != exp_off)[0][0])
            print("  first offset difference at doc", d, int(got_off[d]), int(exp_off[d]))
        n = min(got_tok.size, exp_tok.size)
        bad = np.nonzero(got_tok[:n] != exp_tok[:n])[0]
        print("  token differences:", bad.size, "first at", int(bad[0]) if bad.size else None, "last at", int(bad[-1]) if bad.size else None)
        if bad.size:
            p = int(bad[0])
            d = int(np.searchsorted(exp_off.astype(np.int64), p, side="right") - 1)
            print("  in doc", d, "doc token range", int(exp_off[d]), int(exp_off[d + 1]), "doc bytes", int(off[d]), int(off[d + 1]))
            print("  got", got_tok[max(p - 4, 0):p + 8].tolist())
            print("  exp", exp_tok[max(p - 4, 0):p + 8].tolist())
            print("  prev doc tail bytes", bytes(text[max(int(off[d]) - 60, 0):int(off[d])].tobytes())[-60:])
            print("  doc head bytes", text[int(off[d]):int(off[d]) + 40].tobytes())
    return ok

# (a) same history as the soak: rounds 1..24 then 25, on a fresh context; three times
for attempt in range(3):
    ctx = capi.Context(0)
    for (dd, b, e) in B[:24]:
        t, o = capi.pack_docs(dd)
        ctx.encode_batch(t, o, b, e)
    check(ctx, "history attempt %d:" % attempt)
    check(ctx, "   repeat:")
    ctx.close()
# (b) only the previous round, then 25
ctx = capi.Context(0)
t, o = capi.pack_docs(B[23][0]); ctx.encode_batch(t, o, B[23][1], B[23][2])
check(ctx, "after round 24 only:")
ctx.close()
# (c) fresh context
ctx = capi.Context(0)
check(ctx, "fresh context:")
check(ctx, "   repeat:")
ctx.close()
