"""Host-fed (PCIe-inclusive) rate of l3_encode_batch on C2, with pageable and with pinned caller buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tokenizer_amd import capi
from tools import corpora

ctx = capi.Context(0)
n_docs = 65536
text, off = corpora.english(n_docs, 1024)
cap = text.size // 2 + 2 * n_docs
for pinned in (False, True):
    if pinned:
        t_text = torch.empty(text.size, dtype=torch.uint8).pin_memory(); t_text.numpy()[:] = text
        t_off = torch.empty(off.size, dtype=torch.int64).pin_memory(); t_off.numpy()[:] = off.astype(np.int64)
        t_out = torch.empty(cap, dtype=torch.int32).pin_memory()
        t_oo = torch.empty(n_docs + 1, dtype=torch.int64).pin_memory()
        a_text, a_off, a_out, a_oo = t_text.numpy(), t_off.numpy().view(np.uint64), t_out.numpy(), t_oo.numpy().view(np.uint64)
    else:
        a_text, a_off = text, off.astype(np.uint64)
        a_out, a_oo = np.empty(cap, dtype=np.int32), np.empty(n_docs + 1, dtype=np.uint64)
    import ctypes as C
    n = C.c_uint64(0)
    def call():
        rc = ctx.lib.l3_encode_batch(ctx.h, a_text.ctypes.data, a_off.ctypes.data, n_docs, 3, a_out.ctypes.data, cap, a_oo.ctypes.data, C.byref(n))
        assert rc == 0, rc
    for _ in range(3): call()
    t0 = time.perf_counter()
    for _ in range(10): call()
    dt = (time.perf_counter() - t0) / 10
    print("pinned=%s  %.2f ms per 64 MiB batch  %.2f G tokens/s  %.1f GB/s input" % (pinned, dt * 1e3, n.value / dt / 1e9, text.size / dt / 1e9), flush=True)
