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


def test_submit_wait_pipeline(ctx, oracle):
    """l3_encode_submit / l3_encode_wait: up to four batches pending on one context, waited for in and out of order,
    every batch with its own output buffers; one of them holds a giant pre-token and documents with invalid UTF-8 (its
    wait runs it again synchronously), one is empty, one reports a too small output buffer."""
    import torch
    from tokenizer_amd import capi
    from tests import corpus
    import random
    rng = random.Random(5)
    batches = []
    for b in range(7):
        if b == 3:
            docs = [b"a" * 200000, b"\xff" * 30000, b"tail"] + [corpus.random_doc(rng, max_pieces=20) for _ in range(50)]
            text, off = capi.pack_docs(docs)
        elif b == 5:
            text, off = capi.pack_docs([])
        else:
            text, off = corpora.english(4096 + 512 * b, DOC, first_doc=77000 * b)
        batches.append((text, off))
    dev = []
    for text, off in batches:
        d_text = torch.from_numpy(np.ascontiguousarray(text)).cuda() if text.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
        d_off = torch.from_numpy(off.astype(np.int64)).cuda()
        cap = int(text.size) * 3 + 2 * (off.size - 1) + 16
        dev.append((d_text, d_off, torch.full((cap,), -3, dtype=torch.int32, device="cuda"),
                    torch.full((off.size,), -3, dtype=torch.int64, device="cuda"), cap))
    torch.cuda.synchronize()

    def submit(i, cap=None):
        d_text, d_off, d_out, d_oo, c = dev[i]
        rc, t = ctx.encode_submit(d_text.data_ptr(), int(batches[i][0].size), d_off.data_ptr(), batches[i][1].size - 1, 3,
                                  d_out.data_ptr(), c if cap is None else cap, d_oo.data_ptr(), None)
        assert rc == capi.L3_OK, ctx._err(rc)
        return t

    def check(i, t):
        rc, n = ctx.encode_wait(t)
        assert rc == capi.L3_OK, ctx._err(rc)
        exp_tok, exp_off = oracle.encode_batch(batches[i][0], batches[i][1], True, True, threads=8)
        assert n == exp_tok.size
        assert np.array_equal(dev[i][2][:n].cpu().numpy(), exp_tok)
        assert np.array_equal(dev[i][3].cpu().numpy().astype(np.uint64), exp_off)

    t0, t1, t2, t3 = submit(0), submit(1), submit(2), submit(3)
    rc, _ = ctx.encode_submit(dev[4][0].data_ptr(), int(batches[4][0].size), dev[4][1].data_ptr(), batches[4][1].size - 1, 3,
                              dev[4][2].data_ptr(), dev[4][4], dev[4][3].data_ptr(), None)
    assert rc == capi.L3_E_INVALID_ARG                       # a fifth pending batch is refused
    check(1, t1)                                             # out of order
    t4 = submit(4)
    check(0, t0)
    check(3, t3)                                             # the giant: run again inside its wait
    t5 = submit(5)
    check(2, t2)
    t6 = submit(6, cap=100)                                  # too small an output buffer
    check(4, t4)
    check(5, t5)
    rc, need = ctx.encode_wait(t6)
    exp_tok, _ = oracle.encode_batch(batches[6][0], batches[6][1], True, True, threads=8)
    assert rc == capi.L3_E_OUTPUT_CAP and need == exp_tok.size
    rc, _ = ctx.encode_wait(t6)
    assert rc == capi.L3_E_INVALID_ARG                       # nothing pending under that ticket any more
    # the synchronous call still works on the same context, and with a batch pending
    t = submit(0)
    rc, n = ctx.encode_batch_device(dev[1][0].data_ptr(), int(batches[1][0].size), dev[1][1].data_ptr(), batches[1][1].size - 1, 3,
                                    dev[1][2].data_ptr(), dev[1][4], dev[1][3].data_ptr(), None)
    assert rc == capi.L3_OK
    check(0, t)


@pytest.mark.gpu
def test_submit_alternating_streams(ctx, oracle):
    """Batches submitted in turn on the context's own stream (stream = NULL) and on caller streams while the ones before
    are still pending: they share the workspace and the status word, so each must start behind the one before it whatever
    the streams are (the context orders them by an event recorded behind every batch).  Many rounds, outputs against the
    oracle."""
    import torch
    from tokenizer_amd import capi
    n_b = 4
    batches = [corpora.english(8192 + 1024 * b, DOC, first_doc=123000 * (b + 1)) for b in range(n_b)]
    exp = [oracle.encode_batch(t, o, True, True, threads=8) for t, o in batches]
    dev = []
    for text, off in batches:
        cap = int(text.size) // 2 + 2 * (off.size - 1) + 16
        dev.append((torch.from_numpy(np.ascontiguousarray(text)).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(),
                    torch.full((cap,), -3, dtype=torch.int32, device="cuda"), torch.full((off.size,), -3, dtype=torch.int64, device="cuda"), cap))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    streams = [None, s1.cuda_stream, None, s2.cuda_stream, s1.cuda_stream, None, s2.cuda_stream, s2.cuda_stream]
    for rnd in range(12):
        for g in range(0, len(streams), n_b):
            tickets = []
            for b in range(n_b):
                d_text, d_off, d_out, d_oo, cap = dev[b]
                d_out.fill_(-3)
                torch.cuda.synchronize()
                rc, t = ctx.encode_submit(d_text.data_ptr(), int(batches[b][0].size), d_off.data_ptr(), batches[b][1].size - 1, 3,
                                          d_out.data_ptr(), cap, d_oo.data_ptr(), streams[(g + b + rnd) % len(streams)])
                assert rc == capi.L3_OK, ctx._err(rc)
                tickets.append(t)
            for b in range(n_b):
                rc, n = ctx.encode_wait(tickets[b])
                assert rc == capi.L3_OK, ctx._err(rc)
                assert n == exp[b][0].size
                assert np.array_equal(dev[b][2][:n].cpu().numpy(), exp[b][0]), (rnd, g, b)
                assert np.array_equal(dev[b][3].cpu().numpy().astype(np.uint64), exp[b][1])
