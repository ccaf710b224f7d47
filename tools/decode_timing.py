"""Kernel time of l3_decode_batch on the C2 and C3 token streams: output GB/s next to the encode figure.
Run on the GPU box: python tools/decode_timing.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizer_amd import capi  # noqa: E402
from tools import corpora  # noqa: E402


def main():
    ctx = capi.Context(0)
    res = {}
    for name, (text, off) in (("c2_64k_x_1KiB", corpora.english(65536, 1024)), ("c3_16_x_4MiB", corpora.multilingual(16, 4 << 20)),
                              ("one_4MiB_document", corpora.multilingual(1, 4 << 20))):
        tok, toff = ctx.encode_batch(text, off, False, False)
        ctx.set_profiling(1)
        ms = []
        for _ in range(6):
            out, ooff = ctx.decode_batch(tok, toff)
            ms.append(ctx.last_timings()["decode"])
        ctx.set_profiling(0)
        assert np.array_equal(out, text)
        best = float(np.median(ms[1:]))
        alg = tok.size * 4 + out.size + 8 * toff.size * 2      # ids read + bytes written + both offset arrays
        res[name] = {"ids": int(tok.size), "bytes_out": int(out.size), "kernel_ms": round(best, 4),
                     "out_GBps": round(out.size / best / 1e6, 1), "alg_GBps": round(alg / best / 1e6, 1),
                     "frac_of_8TBps": round(alg / best / 1e6 / 8000, 4)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
