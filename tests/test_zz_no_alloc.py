"""l3_encode_batch_device_noalloc (include/l3hip.h): the entry point whose workspace is the library's file-scope
__device__ array, for callers that cannot allocate.  Checked through the C ABI against l3_encode_batch_device on the same
inputs, bit for bit, on two calls in a row -- the second with a smaller input, so that whatever the first call left in
the shared array must not leak into it -- and against the oracle.  Collected last (file name) so that a failure here
hides no other test.
"""
import numpy as np
import pytest

from tests import corpus
from tools import corpora

pytestmark = pytest.mark.gpu


def _device_encode(ctx, torch, text, off, flags, noalloc):
    from tokenizer_amd import capi
    d_text = torch.from_numpy(np.ascontiguousarray(text)).cuda() if text.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    cap = int(text.size) * 3 + 2 * (off.size - 1) + 16
    d_out = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    d_out_off = torch.full((off.size,), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rc, n = ctx.encode_batch_device(d_text.data_ptr(), int(text.size), d_off.data_ptr(), off.size - 1, flags, d_out.data_ptr(),
                                    cap, d_out_off.data_ptr(), None, noalloc=noalloc)
    assert rc == capi.L3_OK, (rc, ctx._err(rc))
    return d_out[:n].cpu().numpy(), d_out_off.cpu().numpy().astype(np.uint64)


def _inputs():
    import random
    rng = random.Random(77)
    from tokenizer_amd import capi
    # first call: close to the limits -- 3,000 ragged fuzz documents (special tokens, invalid UTF-8, empty documents) in
    # front of ~3 MiB of C2 text; second call: a much smaller batch of other documents
    docs = [corpus.random_doc(rng, max_pieces=30) for _ in range(3000)] + [b"", b"", "中文字符".encode() * 200, b"a" * 3000, b" " * 5000 + b"x"]
    t1, o1 = capi.pack_docs(docs)
    e_text, e_off = corpora.english(3000, 1024)
    big = (np.concatenate([t1, e_text]), np.concatenate([o1, e_off[1:] + o1[-1]]))
    docs2 = [corpus.random_doc(rng, max_pieces=12) for _ in range(200)] + [b"Hello, world!", b""]
    small = capi.pack_docs(docs2)
    return big, small


def test_noalloc_entry_point_matches_the_allocating_one(oracle):
    import torch
    from tokenizer_amd import capi
    ctx = capi.Context(device=0)
    try:
        big, small = _inputs()
        assert big[0].size <= capi.L3_NOALLOC_MAX_BYTES and big[1].size - 1 <= capi.L3_NOALLOC_MAX_DOCS
        for flags in (capi.L3_BOS | capi.L3_EOS, 0):
            for text, off in (big, small):     # two calls in a row on the shared array, the second smaller
                got_tok, got_off = _device_encode(ctx, torch, text, off, flags, noalloc=True)
                ref_tok, ref_off = _device_encode(ctx, torch, text, off, flags, noalloc=False)
                assert np.array_equal(got_off, ref_off) and np.array_equal(got_tok, ref_tok)
                exp_tok, exp_off = oracle.encode_batch(text, off, bool(flags & 1), bool(flags & 2), threads=8)
                assert np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok)
        # an empty batch, and one document that is empty
        for docs in ([], [b""]):
            text, off = capi.pack_docs(docs)
            got_tok, got_off = _device_encode(ctx, torch, text, off, 3, noalloc=True)
            ref_tok, ref_off = _device_encode(ctx, torch, text, off, 3, noalloc=False)
            assert np.array_equal(got_off, ref_off) and np.array_equal(got_tok, ref_tok)
        # beyond the limits: refused with a clear error, nothing launched
        n_big = capi.L3_NOALLOC_MAX_BYTES + 1
        d = torch.zeros(n_big, dtype=torch.uint8, device="cuda")
        o = torch.tensor([0, n_big], dtype=torch.int64, device="cuda")
        out = torch.empty(16, dtype=torch.int32, device="cuda")
        oo = torch.empty(2, dtype=torch.int64, device="cuda")
        rc, _ = ctx.encode_batch_device(d.data_ptr(), n_big, o.data_ptr(), 1, 3, out.data_ptr(), 16, oo.data_ptr(), None, noalloc=True)
        assert rc == capi.L3_E_INVALID_ARG and "static workspace" in str(ctx._err(rc))
        # a second context on the same GPU shares the array: still correct after the first one's calls
        ctx2 = capi.Context(device=0)
        try:
            text, off = small
            got_tok, got_off = _device_encode(ctx2, torch, text, off, 3, noalloc=True)
            exp_tok, exp_off = oracle.encode_batch(text, off, True, True, threads=8)
            assert np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok)
        finally:
            ctx2.close()
    finally:
        ctx.close()
