"""Decode / DecodeBytes (tokenizer.go:339-354) on the GPU: the block-wise kernels (blocks of 4,096 ids whatever the sequence
borders are) against the oracle -- ragged and empty sequences, borders on and next to block and round borders, ids outside
the vocabulary, long vocabulary entries, the capacity protocol, and the C2 / C3 token streams at full size."""
import numpy as np
import pytest

from tokenizer_amd import capi
from tools import corpora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(device=0)
    yield c
    c.close()


def _check(ctx, oracle, ids, off):
    out, ooff = ctx.decode_batch(ids, off)
    want = [oracle.decode(ids[int(off[i]):int(off[i + 1])].tolist()) for i in range(len(off) - 1)]
    assert out.tobytes() == b"".join(want)
    assert ooff.tolist() == [0] + np.cumsum([len(w) for w in want]).tolist()


def test_borders_around_blocks_and_rounds(ctx, oracle):
    rng = np.random.default_rng(2)
    n = 3 * 4096 + 700
    ids = rng.integers(0, 128256, n).astype(np.int32)
    ids[rng.integers(0, n, 300)] = rng.choice(np.array([-1, -2**31, 128256, 2**31 - 1, 200000], dtype=np.int64), 300).astype(np.int32)
    cuts = sorted(set([0, n] + [k * 1024 + d for k in range(1, 13) for d in (-1, 0, 1)] + [4096, 8192, 12288, 12289]))
    off = np.array([c for c in cuts if 0 <= c <= n], dtype=np.uint64)
    _check(ctx, oracle, ids, off)
    # empty sequences: at the front, in the middle (on a block border), at the end; more than 64 in a row
    off2 = np.array([0, 0, 0, 5] + [4096] * 70 + [4097, n] + [n] * 130, dtype=np.uint64)
    _check(ctx, oracle, ids, off2)
    # one id per sequence
    k = 2000
    _check(ctx, oracle, ids[:k], np.arange(k + 1, dtype=np.uint64))


def test_empty_inputs(ctx, oracle):
    none = np.zeros(0, dtype=np.int32)
    _check(ctx, oracle, none, np.array([0], dtype=np.uint64))          # no sequences at all
    _check(ctx, oracle, none, np.array([0, 0, 0], dtype=np.uint64))    # two empty sequences
    _check(ctx, oracle, np.array([-5, 999999], dtype=np.int32), np.array([0, 2], dtype=np.uint64))   # nothing decodes


def test_long_entries_take_the_unstaged_path(ctx, oracle):
    # the longest vocabulary entries: a round of them outgrows the LDS staging buffer
    lens = np.array([len(ctx.token_bytes(i) or b"") for i in range(0, 128000, 1)])
    longest = np.argsort(lens)[-40:]
    assert lens[longest].min() >= 20
    ids = np.tile(longest.astype(np.int32), 200)         # 8,000 ids of 20+ bytes: > 16 KiB per round of 1,024
    _check(ctx, oracle, ids, np.array([0, 3, 4097, ids.size], dtype=np.uint64))


def test_capacity_protocol(ctx, oracle):
    import ctypes as C
    ids = np.arange(1000, 9000, dtype=np.int32)
    off = np.array([0, 10, 8000], dtype=np.uint64)
    want = oracle.decode(ids.tolist())
    out = np.empty(16, dtype=np.uint8)
    ooff = np.empty(3, dtype=np.uint64)
    n = C.c_uint64(0)
    rc = ctx.lib.l3_decode_batch(ctx.h, ids.ctypes.data, off.ctypes.data, 2, out.ctypes.data, 16, ooff.ctypes.data, C.byref(n))
    assert rc == capi.L3_E_OUTPUT_CAP and n.value == len(want)
    assert ooff.tolist() == [0, len(oracle.decode(ids[:10].tolist())), len(want)]      # offsets are filled in all the same


def test_c2_and_c3_token_streams_round_trip(ctx):
    # full-size streams: 65,536 sequences of ~212 ids, and 16 sequences of ~2.6 M ids (one block grid for both)
    for text, off in (corpora.english(65536, 1024), corpora.multilingual(16, 4 << 20)):
        tok, toff = ctx.encode_batch(text, off, False, False)
        out, ooff = ctx.decode_batch(tok, toff)
        assert np.array_equal(ooff, off) and np.array_equal(out, text)
