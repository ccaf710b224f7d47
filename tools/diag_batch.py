"""Diagnose a batch on which GPU and oracle differ: usage diag_batch.py TEXT.npy OFF.npy BOS EOS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.oracle import Oracle
from tokenizer_amd import capi
text, off = np.load(sys.argv[1]), np.load(sys.argv[2])
bos, eos = sys.argv[3] == "1", sys.argv[4] == "1"
ctx, o = capi.Context(0), Oracle()
exp_tok, exp_off = o.encode_batch(text, off, bos, eos, threads=8)
for rep in range(4):
    got_tok, got_off = ctx.encode_batch(text, off, bos, eos)
    same = np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok)
    print("rep", rep, "equal" if same else "DIFFERENT", ctx.last_counters(), flush=True)
    if not same:
        go, eo = got_off.astype(np.int64), exp_off.astype(np.int64)
        for i in range(off.size - 1):
            g, e = got_tok[go[i]:go[i + 1]].tolist(), exp_tok[eo[i]:eo[i + 1]].tolist()
            if g == e:
                continue
            k = next(j for j in range(min(len(g), len(e)) + 1) if j >= min(len(g), len(e)) or g[j] != e[j])
            base = int(eo[i])
            print(" doc %d (global offset %d, %d bytes): first differing token index %d of %d/%d" % (i, int(off[i]), int(off[i + 1] - off[i]), k, len(g), len(e)))
            print("  exp:", [(t, o.decode([t])) for t in e[max(0, k - 6):k + 6]])
            print("  got:", [(t, o.decode([t])) for t in g[max(0, k - 6):k + 6]])
            pre = o.decode(e[:k])
            print("  byte position in doc ~", len(pre), "global", int(off[i]) + len(pre), "= tile", (int(off[i]) + len(pre)) // 3840, "+", (int(off[i]) + len(pre)) % 3840)
            break
