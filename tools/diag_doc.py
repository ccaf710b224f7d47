"""Diagnose one document on which GPU and oracle differ: usage diag_doc.py FILE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.oracle import Oracle
from tokenizer_amd import capi
d = open(sys.argv[1], "rb").read()
ctx, o = capi.Context(0), Oracle()
for rep in range(3):
    for bos, eos in ((True, True), (False, False)):
        got = ctx.encode_docs([d], bos, eos)[0]
        exp = o.encode(d, bos, eos)
        print("rep", rep, bos, eos, "equal" if got == exp else "DIFFERENT", len(got), len(exp), flush=True)
st = o.pretokenize(d) + [len(d)]
got = ctx.encode_docs([d], False, False)[0]
pos = 0
for a, b in zip(st[:-1], st[1:]):
    e = o.encode(d[a:b], False, False)
    g = got[pos:pos + len(e)]
    if g != e:
        print("first differing pre-token: bytes [%d, %d) len %d" % (a, b, b - a), d[a:b][:60], "...", d[a:b][-20:])
        print("  exp", e[:12], len(e)); print("  got", g[:12])
        print("  tile of start:", a // 3840, "offset in tile", a % 3840, "context before:", d[max(0, a - 80):a])
        alone = ctx.encode_docs([d[a:b]], False, False)[0]
        print("  the pre-token alone on the GPU:", "equal" if alone == e else "DIFFERENT")
        break
    pos += len(e)
