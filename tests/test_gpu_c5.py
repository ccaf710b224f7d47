"""C5 (BASELINE.json configs[4]): the 64 GiB corpus is 67,108,864 C2 documents, document i generated from its
seed on the GPU that owns it and encoded batch after batch on one context.  Here: the device generator against the host
generator, many consecutive batches on one context (workspace reuse; more than 2^31 bytes in all) and ONE call over more
than 2^31 bytes, against the oracle on sampled batches / documents.
"""
import numpy as np
import pytest

from tools import corpora

pytestmark = pytest.mark.gpu

DOC = 1024


@pytest.fixture(scope="module")
def ctx():
    from tokenizer_amd import capi
    c = capi.Context(device=0)
    yield c
    c.close()


def test_device_generator_writes_the_host_generator_s_bytes():
    import torch
    for first, n in ((0, 4096), (123456789, 1000), (67108864 - 70, 70), (5, 1)):
        d = torch.zeros(n * DOC + 64, dtype=torch.uint8, device="cuda")
        corpora.english_device(d, first, n, DOC)
        torch.cuda.synchronize()
        host, _ = corpora.english(n, DOC, first_doc=first)
        got = d.cpu().numpy()
        assert np.array_equal(got[:n * DOC], host), (first, n)
        assert not got[n * DOC:].any()            # nothing written past the last document
    d = torch.zeros(300 * 100, dtype=torch.uint8, device="cuda")     # another document size (not a multiple of 16)
    corpora.english_device(d, 42, 300, 100)
    torch.cuda.synchronize()
    host, _ = corpora.english(300, 100, first_doc=42)
    assert np.array_equal(d.cpu().numpy(), host)


def _encode_resident(ctx, torch, d_text, n_bytes, d_off, n_docs, d_out, d_out_off):
    from tokenizer_amd import capi
    torch.cuda.synchronize()    # the generator ran on the default stream; the context's own stream does not wait for that one
    rc, n = ctx.encode_batch_device(d_text.data_ptr(), n_bytes, d_off.data_ptr(), n_docs, 3, d_out.data_ptr(), d_out.numel(),
                                    d_out_off.data_ptr(), None)
    assert rc == capi.L3_OK, ctx._err(rc)
    return n


def test_forty_consecutive_batches_on_one_context(ctx, oracle):
    """40 x 64 MiB = 2.5 GiB (> 2^31 bytes) through one context, batch after batch: the workspace is reused, nothing of
    one batch may leak into the next.  Batches 0, 17 and 39 against the oracle; every batch: offsets consistent, BOS/EOS
    in place, and the same answer when its text is encoded again after all the others."""
    import torch
    n_docs, n_bytes = 65536, 65536 * DOC
    d_text = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    d_off = torch.arange(n_docs + 1, dtype=torch.int64, device="cuda") * DOC
    d_out = torch.empty(n_bytes // 2 + 2 * n_docs, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    first_batch = 1000                      # somewhere inside the corpus
    totals, sums = [], []
    total_bytes = 0
    for b in range(40):
        corpora.english_device(d_text, (first_batch + b) * n_docs, n_docs, DOC)
        n = _encode_resident(ctx, torch, d_text, n_bytes, d_off, n_docs, d_out, d_out_off)
        total_bytes += n_bytes
        off = d_out_off.cpu().numpy()
        assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 2).all()
        tok = d_out[:n]
        assert bool((tok[d_out_off[:-1]] == 128000).all()) and bool((tok[d_out_off[1:] - 1] == 128001).all())
        totals.append(n)
        sums.append(int(tok.to(torch.int64).sum().item()))
        if b in (0, 17, 39):
            text, hoff = corpora.english(n_docs, DOC, first_doc=(first_batch + b) * n_docs)
            exp_tok, exp_off = oracle.encode_batch(text, hoff, True, True, threads=16)
            assert n == exp_tok.size and np.array_equal(off.astype(np.uint64), exp_off)
            assert np.array_equal(tok.cpu().numpy(), exp_tok)
    assert total_bytes > 2 ** 31
    for b in (3, 28):                       # again, after the context has seen all the others
        corpora.english_device(d_text, (first_batch + b) * n_docs, n_docs, DOC)
        n = _encode_resident(ctx, torch, d_text, n_bytes, d_off, n_docs, d_out, d_out_off)
        assert n == totals[b] and int(d_out[:n].to(torch.int64).sum().item()) == sums[b]


def test_one_call_over_more_than_2_31_bytes(ctx, oracle):
    """One batch of 2,359,296 documents = 2.25 GiB: every byte offset past 2^31 and every index past 2^31 / 4 has to be
    64-bit all the way.  Against the sum of its 36 parts encoded separately, and against the oracle on documents from the
    front, the middle (around byte 2^31) and the end."""
    import torch
    part_docs, parts = 65536, 36
    n_docs = part_docs * parts
    n_bytes = n_docs * DOC
    assert n_bytes > 2 ** 31
    d_text = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    for p in range(parts):
        corpora.english_device(d_text[p * part_docs * DOC:], p * part_docs, part_docs, DOC)
    d_off = torch.arange(n_docs + 1, dtype=torch.int64, device="cuda") * DOC
    d_out = torch.empty(n_bytes // 2 + 2 * n_docs, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    n = _encode_resident(ctx, torch, d_text, n_bytes, d_off, n_docs, d_out, d_out_off)
    off = d_out_off.cpu().numpy()
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 2).all()
    # parts encoded on their own
    p_out = torch.empty(part_docs * DOC // 2 + 2 * part_docs, dtype=torch.int32, device="cuda")
    p_off = torch.empty(part_docs + 1, dtype=torch.int64, device="cuda")
    sub_off = torch.arange(part_docs + 1, dtype=torch.int64, device="cuda") * DOC
    base = 0
    for p in (0, 1, 15, 16, 31, 32, 35):    # byte 2^31 is the start of part 32
        m = _encode_resident(ctx, torch, d_text[p * part_docs * DOC:], part_docs * DOC, sub_off, part_docs, p_out, p_off)
        base = int(off[p * part_docs])
        assert int(off[(p + 1) * part_docs]) - base == m
        assert bool(torch.equal(d_out[base:base + m], p_out[:m]))
        assert np.array_equal(off[p * part_docs:(p + 1) * part_docs + 1] - base, p_off.cpu().numpy())
    for first in (0, 32 * part_docs - 300, n_docs - 600):
        text, hoff = corpora.english(600, DOC, first_doc=first)
        exp_tok, exp_off = oracle.encode_batch(text, hoff, True, True, threads=8)
        lo, hi = int(off[first]), int(off[first + 600])
        assert hi - lo == exp_tok.size and np.array_equal(d_out[lo:hi].cpu().numpy(), exp_tok)
        assert np.array_equal((off[first:first + 601] - lo).astype(np.uint64), exp_off)
