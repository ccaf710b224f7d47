"""How much does running the kernels of consecutive batches on DIFFERENT streams gain?  Probe with N contexts (each has its
own workspace and stream) fed round-robin from one host thread, against the one-context submit/wait loop of bench.py."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizer_amd import capi  # noqa: E402
from tools import corpora  # noqa: E402

N_DOCS, DOC = 65536, 1024
STEPS = 24


def main():
    n_bytes = N_DOCS * DOC
    d_text = torch.empty(STEPS * n_bytes, dtype=torch.uint8, device="cuda")
    for i in range(STEPS):
        corpora.english_device(d_text[i * n_bytes:(i + 1) * n_bytes], i * N_DOCS, N_DOCS, DOC)
    torch.cuda.synchronize()
    d_off = torch.arange(N_DOCS + 1, dtype=torch.int64, device="cuda") * DOC
    cap = n_bytes // 2 + 2 * N_DOCS
    for n_ctx in (1, 2, 3, 4):
        ctxs = [capi.Context(0) for _ in range(n_ctx)]
        depth = max(2, n_ctx)          # batches in flight overall
        outs = [(torch.empty(cap, dtype=torch.int32, device="cuda"), torch.empty(N_DOCS + 1, dtype=torch.int64, device="cuda"))
                for _ in range(depth + 1)]
        n_out = C.c_uint64(0)
        tk = C.c_int(-1)

        def submit(i):
            c = ctxs[i % n_ctx]
            o, oo = outs[i % len(outs)]
            rc = c.lib.l3_encode_submit(c.h, C.c_void_p(d_text.data_ptr() + i * n_bytes), n_bytes, C.c_void_p(d_off.data_ptr()), N_DOCS, 3,
                                        C.c_void_p(o.data_ptr()), cap, C.c_void_p(oo.data_ptr()), None, C.byref(tk))
            assert rc == 0, rc
            return (c, tk.value)

        def wait(p):
            c, t = p
            rc = c.lib.l3_encode_wait(c.h, t, C.byref(n_out))
            assert rc == 0, rc
            return n_out.value

        def run():
            pend, tot = [], 0
            for i in range(STEPS):
                pend.append(submit(i))
                if len(pend) >= depth:
                    tot += wait(pend.pop(0))
            while pend:
                tot += wait(pend.pop(0))
            return tot

        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("contexts %d: %.4f ms/step, %.3e tok/s, tokens %d" % (n_ctx, dt / STEPS * 1e3, tot / dt, tot))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
