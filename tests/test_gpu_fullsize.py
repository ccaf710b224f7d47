"""GPU parity at BASELINE.json's full sizes (C2, C3, C4), bit-exact against the oracle, plus size-independent
properties: decode(encode(x)) == x on valid UTF-8, batch == concatenation of single-document encodes,
offsets consistent with counts.  The oracle runs on the box's host cores (a few seconds per case)."""
import numpy as np
import pytest

from tools import corpora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tokenizer_amd import capi
    c = capi.Context(device=0)
    yield c
    c.close()


def _same(ctx, oracle, text, off, threads=16):
    tok, toff = ctx.encode_batch(text, off, True, True)
    e_tok, e_off = oracle.encode_batch(text, off, True, True, threads=threads)
    assert np.array_equal(toff, e_off)
    assert tok.size == e_tok.size and np.array_equal(tok, e_tok)
    return tok, toff


def test_c2_english_64k_docs(ctx, oracle):
    text, off = corpora.english(65536, 1024)
    tok, toff = _same(ctx, oracle, text, off)
    assert int(toff[-1]) == tok.size and tok[0] == 128000 and tok[-1] == 128001
    # round trip of the whole batch without BOS/EOS
    tok2, toff2 = ctx.encode_batch(text, off, False, False)
    out, ooff = ctx.decode_batch(tok2, toff2)
    assert np.array_equal(ooff, off) and np.array_equal(out, text)


def test_c3_multilingual_4mib_docs(ctx, oracle):
    text, off = corpora.multilingual(4, 4 << 20)
    tok, toff = _same(ctx, oracle, text, off)
    # a document encodes the same alone as inside the batch
    d = 2
    one, _ = ctx.encode_batch(text[int(off[d]):int(off[d + 1])], np.array([0, 4 << 20], dtype=np.uint64), True, True)
    assert np.array_equal(one, tok[int(toff[d]):int(toff[d + 1])])
    tok2, toff2 = ctx.encode_batch(text, off, False, False)
    out, ooff = ctx.decode_batch(tok2, toff2)
    assert np.array_equal(ooff, off) and np.array_equal(out, text)


def test_c3_sixteen_docs_full_size(ctx, oracle):
    # C3 at its full size, 16 x 4 MiB: all 41.1 M ids bit for bit against the oracle (5.6 s on the box's 16 host threads),
    # then the size-independent properties
    text, off = corpora.multilingual(16, 4 << 20)
    tok, toff = ctx.encode_batch(text, off, False, False)
    e_tok, e_off = oracle.encode_batch(text, off, False, False, threads=16)
    assert np.array_equal(toff, e_off) and np.array_equal(tok, e_tok)
    out, ooff = ctx.decode_batch(tok, toff)
    assert np.array_equal(ooff, off) and np.array_equal(out, text)
    first4, off4 = ctx.encode_batch(text[:4 * (4 << 20)], off[:5], False, False)
    assert np.array_equal(first4, tok[:first4.size]) and np.array_equal(off4, toff[:5])


def test_c4_pathological_1mib(ctx, oracle):
    from tokenizer_amd import capi
    docs = corpora.pathological(size=1 << 20)
    for i, d in enumerate(docs):
        t, o = capi.pack_docs([d])
        _same(ctx, oracle, t, o, threads=1)
    t, o = capi.pack_docs(docs)
    _same(ctx, oracle, t, o)


def test_long_digit_runs_are_linear(ctx, oracle):
    # every tile inside a digit run needs the run's length before it (1-3 digit grouping, state_machine.go:178-191);
    # tiles pass it on (Workspace::tile_digits), so 16 MiB of digits must not take 256 times as long as 1 MiB
    from tokenizer_amd import capi
    rng = np.random.default_rng(12)
    ascii_run = rng.integers(48, 58, 16 << 20, dtype=np.uint8).tobytes()
    mixed = ("12٣٤٥" * 50000 + "7" * 400001 + "٣" * 300000 + " x " + "9" * 100000).encode()
    docs = [ascii_run, b"ab" + ascii_run[:5000001], mixed, b"1" * 3839, b"1" * 3841, b"22" + "٣".encode() * 7000]
    ctx.set_profiling(True)
    for i, d in enumerate(docs):
        text, off = capi.pack_docs([d])
        _same(ctx, oracle, text, off, threads=1)
        if i == 0:
            assert ctx.last_timings()["tile"] < 20.0, ctx.last_timings()     # quadratic: ~900 ms
    text, off = capi.pack_docs(docs)
    _same(ctx, oracle, text, off)
    ctx.set_profiling(False)


def test_host_batches_in_pieces(ctx, oracle):
    """l3_encode_batch moves batches of >= 16 MiB in pieces of whole documents (overlapped copies): same result as
    the device-pointer entry point in one pass, ragged documents included, and the capacity protocol still holds."""
    import ctypes as C

    import torch

    from tokenizer_amd import capi
    rng = np.random.default_rng(5)
    base, boff = corpora.english(40000, 1024)
    # ragged: cut the same text at random places instead (documents of 1 B .. 64 KiB, some empty)
    cuts = np.unique(np.concatenate([[0, base.size], rng.integers(0, base.size, 3000), np.arange(0, 1 << 22, 97)]))
    off = np.sort(np.concatenate([cuts, cuts[100:200]])).astype(np.uint64)   # duplicates = empty documents
    # page-locked caller buffers select the path in pieces
    p_text = torch.empty(base.size, dtype=torch.uint8).pin_memory()
    p_text.numpy()[:] = base
    p_out = torch.empty(base.size + 2 * (off.size - 1) + 64, dtype=torch.int32).pin_memory()
    p_off = np.empty(off.size, dtype=np.uint64)
    n_p = C.c_uint64(0)
    rc = ctx.lib.l3_encode_batch(ctx.h, p_text.numpy().ctypes.data, off.ctypes.data, off.size - 1, 3, p_out.numpy().ctypes.data,
                                 p_out.numel(), p_off.ctypes.data, C.byref(n_p))
    assert rc == capi.L3_OK
    tok, toff = p_out.numpy()[:n_p.value].copy(), p_off
    tok1, toff1 = ctx.encode_batch(base, off, True, True)   # pageable buffers: one pass
    assert np.array_equal(tok1, tok) and np.array_equal(toff1, toff)
    # the same with buffers from the library's own allocator (what a cgo shim would use)
    l_text = ctx.host_alloc(base.size)
    l_text[:] = base
    l_out = ctx.host_alloc(4 * p_out.numel(), dtype=np.int32)
    rc = ctx.lib.l3_encode_batch(ctx.h, l_text.ctypes.data, off.ctypes.data, off.size - 1, 3, l_out.ctypes.data, l_out.size,
                                 p_off.ctypes.data, C.byref(n_p))
    assert rc == capi.L3_OK and n_p.value == tok.size and np.array_equal(l_out[:tok.size], tok) and np.array_equal(p_off, toff1)
    ctx.host_free(l_text)
    ctx.host_free(l_out)
    d_text = torch.from_numpy(base).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    n_docs = off.size - 1
    d_out = torch.empty(base.size + 2 * n_docs + 64, dtype=torch.int32, device="cuda")
    d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    rc, n = ctx.encode_batch_device(d_text.data_ptr(), base.size, d_off.data_ptr(), n_docs, 3, d_out.data_ptr(), d_out.numel(),
                                    d_oo.data_ptr())
    assert rc == capi.L3_OK and n == tok.size
    assert np.array_equal(d_oo.cpu().numpy().astype(np.uint64), toff)
    assert np.array_equal(d_out[:n].cpu().numpy(), tok)
    e_tok, e_off = oracle.encode_batch(base[:1 << 22], off[off <= (1 << 22)], True, True, threads=16)   # and a slice against the oracle
    k = e_off.size - 1
    assert np.array_equal(toff[:k + 1], e_off) and np.array_equal(tok[:int(e_off[-1])], e_tok)
    # capacity protocol
    with pytest.raises(capi.L3Error) as ei:
        ctx.encode_batch(base, off, True, True, out_cap=1000)
    assert ei.value.code == capi.L3_E_OUTPUT_CAP
    ooff = np.empty(n_docs + 1, dtype=np.uint64)
    need = C.c_uint64(0)
    rc = ctx.lib.l3_encode_batch(ctx.h, p_text.numpy().ctypes.data, off.ctypes.data, n_docs, 3, p_out.numpy().ctypes.data, 16,
                                 ooff.ctypes.data, C.byref(need))   # in pieces
    assert rc == capi.L3_E_OUTPUT_CAP and need.value == tok.size and np.array_equal(ooff, toff)
    tok2, toff2 = ctx.encode_batch(base, off, True, True, out_cap=tok.size)
    assert np.array_equal(tok2, tok) and np.array_equal(toff2, toff)


def test_giant_pieces_grid_wide_merge(ctx, oracle):
    """Pieces of 32,768 symbols and more that no cut divides take the grid-wide merge (l3_giant.hip): one character
    repeated, a DNA-like string, invalid bytes (three symbols each), several such pieces in one pre-token / document /
    batch, next to ordinary text, around the threshold, twice in a row on one context."""
    from tokenizer_amd import capi
    rng = np.random.default_rng(5)
    acgt = bytes(rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 300000))
    e_text, _ = corpora.english(64, 1024)
    eng = e_text.tobytes()
    docs = [
        b"a" * 32767, b"a" * 32768, b"a" * 32769, b"b" * 100000, b" " * 70001 + b"x", acgt,
        b"\xff" * 40000,                                        # 120,000 symbols EF BF BD ...
        eng[:3000] + b"z" * 50000 + eng[3000:9000] + acgt[:60000] + b" " + b"!" * 45000 + eng[9000:12000],
        "中".encode() * 20000 + b"q" * 40000 + "文".encode() * 20000,     # cuts around a giant piece inside ONE pre-token
        b"", eng[:5000], b"ab" * 30000, b"\t" * 33000 + b"\n" * 40000,
    ]
    text, off = capi.pack_docs(docs)
    for _ in range(2):
        _same(ctx, oracle, text, off)
        assert ctx.last_counters()["giant_symbols"] >= 32768 * 8
    for d in (docs[1], docs[7]):
        t, o = capi.pack_docs([d])
        _same(ctx, oracle, t, o, threads=1)
    # a batch without such pieces right after: nothing of the path may linger
    t, o = capi.pack_docs([eng[:7000], b"a" * 2500])
    _same(ctx, oracle, t, o)
    assert ctx.last_counters()["giant_symbols"] == 0
