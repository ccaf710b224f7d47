"""SURVEY 8(d) C5, host-fed leg: >= 8 GiB of the C5 corpus per GPU held in HOST memory and pushed through l3_encode_batch
batch by batch (PCIe-inclusive; never bench.py's `value`).  Inside a call the batch moves in pieces of whole documents, the
upload of piece k + 1 and the download of piece k - 1 overlapping the kernels of piece k (double-buffered both ways).
    python tools/c5_host_fed.py [GiB] [pinned|pageable]
Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizer_amd import capi  # noqa: E402
from tools import corpora  # noqa: E402

N_DOCS, DOC = 65536, 1024


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    pinned = (sys.argv[2] if len(sys.argv) > 2 else "pinned") == "pinned"
    n_bytes = N_DOCS * DOC
    batches = int(gib * (1 << 30)) // n_bytes
    host = torch.empty(batches * n_bytes, dtype=torch.uint8)
    if pinned:
        host = host.pin_memory()
    d = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    g0 = time.perf_counter()
    for b in range(batches):   # document i from its seed, on the GPU, then down to the host once
        corpora.english_device(d, b * N_DOCS, N_DOCS, DOC)
        torch.cuda.synchronize()
        host[b * n_bytes:(b + 1) * n_bytes].copy_(d)
    gen_s = time.perf_counter() - g0
    ctx = capi.Context(0)
    cap = n_bytes // 2 + 2 * N_DOCS
    out = torch.zeros(cap, dtype=torch.int32)
    ooff = torch.zeros(N_DOCS + 1, dtype=torch.int64)
    if pinned:
        out, ooff = out.pin_memory(), ooff.pin_memory()
    off = np.arange(N_DOCS + 1, dtype=np.uint64) * np.uint64(DOC)
    n = C.c_uint64(0)
    base = host.data_ptr()

    def call(b):
        rc = ctx.lib.l3_encode_batch(ctx.h, C.c_void_p(base + b * n_bytes), off.ctypes.data, N_DOCS, 3, C.c_void_p(out.data_ptr()), cap,
                                     C.c_void_p(ooff.data_ptr()), C.byref(n))
        if rc != capi.L3_OK:
            raise RuntimeError("l3_encode_batch: %d" % rc)
        return n.value

    call(0)
    call(1)
    t0 = time.perf_counter()
    total = 0
    for b in range(batches):
        total += call(b)
    dt = time.perf_counter() - t0
    # the last batch once more, device-resident, for comparison
    want_tok, want_off = ctx.encode_batch(host[(batches - 1) * n_bytes:].numpy(), off)
    same = bool(np.array_equal(out.numpy()[:n.value], want_tok) and np.array_equal(ooff.numpy().view(np.uint64), want_off))
    print(json.dumps({"workload": "C5 host-fed: %d batches of %d docs x %d B (%.1f GiB) from %s host memory through l3_encode_batch"
                      % (batches, N_DOCS, DOC, batches * n_bytes / 2 ** 30, "page-locked" if pinned else "pageable"),
                      "seconds": dt, "ms_per_batch": dt / batches * 1e3, "tokens": total, "tokens_per_s": total / dt,
                      "input_gb_per_s": batches * n_bytes / dt / 1e9, "output_gb_per_s": 4 * total / dt / 1e9,
                      "corpus_generation_s": gen_s, "last_batch_equals_one_pass": same}))


if __name__ == "__main__":
    main()
