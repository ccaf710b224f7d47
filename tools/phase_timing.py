"""Timing experiment: tile-kernel time with phases disabled through debug flag bits (results are then wrong).
Needs a library built with the knobs: tools/build_variant.sh dbg "-DL3_DEBUG_FLAGS", then
L3HIP_LIB=tokenizer_amd/lib/variants/libl3hip_dbg.so python tools/phase_timing.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0)
n_docs = 65536
text, off = corpora.english(n_docs, 1024)
d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off.astype(np.int64)).cuda()
cap = text.size
d_out = torch.empty(cap, dtype=torch.int32, device="cuda"); d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
ctx.set_profiling(True)
for name, fl in [("full", 3), ("no-mergeloop", 3 | 0x100), ("words-noprobe", 3 | 0x400), ("no-vocab", 3 | 0x200)]:
    for _ in range(3):
        rc, n = ctx.encode_batch_device(d_text.data_ptr(), text.size, d_off.data_ptr(), n_docs, fl, d_out.data_ptr(), cap, d_oo.data_ptr(), None)
    print(name, rc, n, ctx.last_timings())
