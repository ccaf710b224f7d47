"""Diagnose tests/test_gpu_c5.py::test_forty_consecutive_batches_on_one_context: the batches it compares with the oracle,
device-resident as the test runs them, document by document.  usage: diag_c5.py [batch ...] (default 0 17 39)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.oracle import Oracle
from tokenizer_amd import capi
from tools import corpora

DOC = 1024
batches = [int(a) for a in sys.argv[1:]] or [0, 17, 39]
ctx, o = capi.Context(0), Oracle()
n_docs, n_bytes = 65536, 65536 * DOC
d_text = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
d_off = torch.arange(n_docs + 1, dtype=torch.int64, device="cuda") * DOC
d_out = torch.empty(n_bytes // 2 + 2 * n_docs, dtype=torch.int32, device="cuda")
d_out_off = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
for b in batches:
    corpora.english_device(d_text, (1000 + b) * n_docs, n_docs, DOC)
    text, hoff = corpora.english(n_docs, DOC, first_doc=(1000 + b) * n_docs)
    torch.cuda.synchronize()
    print("batch", b, "device text == host text:", bool(np.array_equal(d_text.cpu().numpy(), text)), flush=True)
    exp_tok, exp_off = o.encode_batch(text, hoff, True, True, threads=16)
    eo = exp_off.astype(np.int64)
    for rep in range(3):
        rc, n = ctx.encode_batch_device(d_text.data_ptr(), n_bytes, d_off.data_ptr(), n_docs, 3, d_out.data_ptr(), d_out.numel(),
                                        d_out_off.data_ptr(), None)
        got_tok, go = d_out[:n].cpu().numpy(), d_out_off.cpu().numpy().astype(np.int64)
        same = n == exp_tok.size and np.array_equal(go, eo) and np.array_equal(got_tok, exp_tok)
        print(" rep", rep, "rc", rc, "n", n, "exp", exp_tok.size, "equal" if same else "DIFFERENT", ctx.last_counters(), flush=True)
        if same:
            continue
        shown = 0
        ndiff = 0
        for i in range(n_docs):
            g, e = got_tok[go[i]:go[i + 1]].tolist(), exp_tok[eo[i]:eo[i + 1]].tolist()
            if g == e:
                continue
            ndiff += 1
            if shown >= 4:
                continue
            shown += 1
            k = next(j for j in range(min(len(g), len(e)) + 1) if j >= min(len(g), len(e)) or g[j] != e[j])
            pre = o.decode(e[1:k]) if k >= 1 else b""
            gp = i * DOC + len(pre)
            print("  doc %d: first differing token %d of %d/%d; byte in doc ~%d, global %d = tile %d + %d" % (i, k, len(g), len(e), len(pre), gp, gp // 3840, gp % 3840))
            print("   exp:", [(t, o.decode([t])) for t in e[max(0, k - 4):k + 6]])
            print("   got:", [(t, o.decode([t]) if 0 <= t < 128256 else None) for t in g[max(0, k - 4):k + 6]])
        print("  documents that differ:", ndiff, flush=True)
