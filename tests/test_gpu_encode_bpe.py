"""BPE.EncodeBPE (reference llama3/tokenizer.go:80-84,478-488 -> internal/bpe/bpe.go:18-71) on the GPU: L3_ONE_PRETOKEN makes
every document ONE pre-token of raw bytes, so each merge path of the pipeline is compared with the oracle's merge loop
(l3o_bpe_raw) directly, length by length -- without a pre-tokenizer in between to choose the lengths:

  1 ..  12 bytes   the tile kernel's probes and the 16-element lane slots
 13 ..  64         lane slots of 64 elements
 65 .. 512         one wave per entry (mid_pass)
 513 .. 32,767     window items: cuts, lane pieces, round-based merge by a workgroup
 32,768 ..         the grid-wide merge (l3_giant.hip)

The reference holds no vectors for EncodeBPE (tokenizer_test.go tests it through Encode only): parity here is
oracle-vs-GPU, the oracle's merge loop being pinned by the Encode vectors (tests/test_oracle_kat.py)."""
import random

import numpy as np
import pytest

from tokenizer_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _word(rng, n, kind):
    if kind == 0:      # lower-case letters: many merges
        return bytes(rng.choice(b"etaoinshrdlucmfwypvbgkqjxz") for _ in range(n))
    if kind == 1:      # any byte at all: invalid UTF-8, NUL, 0xFF -- bytes are symbols here, nothing is replaced
        return bytes(rng.randrange(256) for _ in range(n))
    if kind == 2:      # what the pre-tokenizer would never keep together: blanks, digits, punctuation, newlines
        return bytes(rng.choice(b"  \n\t0123456789.,;:'\"!?-()ab") for _ in range(n))
    if kind == 3:      # one byte repeated / a short period: the alternating rule of the round-based merge
        unit = bytes(rng.choice(b"ab \xe4\xb8\xad") for _ in range(rng.randrange(1, 4)))
        return (unit * (n // len(unit) + 1))[:n]
    text = "Grüße 你好世界 Привет мир — ".encode()     # kind 4: UTF-8 text cut anywhere
    o = rng.randrange(len(text))
    return ((text[o:] + text) * (n // len(text) + 1))[:n]


def _check(ctx, oracle, words, what):
    got = ctx.encode_bpe(words)
    for w, g in zip(words, got):
        exp = oracle.bpe_raw(w) if w else []
        assert g == exp, (what, len(w), w[:80], g[:20], exp[:20])


def test_every_length_up_to_600(ctx, oracle):
    rng = random.Random(80)
    words = [b""]
    for n in range(1, 601):
        for kind in range(5):
            w = _word(rng, n, kind)
            if b"<|" not in w:
                words.append(w)
    _check(ctx, oracle, words, "lengths 1..600")
    # whole-string hits of every length the vocabulary has (bpe.go:27-33), and near misses of them
    toks = [oracle.decode([i]) for i in rng.sample(range(128000), 3000)]
    _check(ctx, oracle, toks + [t + b"x" for t in toks[:500]] + [t[:-1] for t in toks[:500] if len(t) > 1], "vocabulary entries")
    assert all(len(g) == 1 for g in ctx.encode_bpe(toks))


def test_hand_over_points_and_long_pre_tokens(ctx, oracle):
    rng = random.Random(81)
    words = []
    for n in (640, 1023, 1024, 1025, 1919, 1920, 1921, 3839, 3840, 3841, 4000, 7679, 7680, 7681, 12000, 32767, 32768, 32769, 50000):
        for kind in (0, 1, 3, 4):
            words.append(_word(rng, n, kind))
    _check(ctx, oracle, words, "long pre-tokens")
    _check(ctx, oracle, [b"a" * 200000, _word(rng, 150000, 0)], "giants")


def test_flags_and_empty_documents(ctx, oracle):
    words = [b"", b"hello", b"", b"", b" world", b""]
    text, off = capi.pack_docs(words)
    bos_id, eos_id = oracle.encode(b"", True, True)
    for bos, eos in ((False, False), (True, False), (False, True), (True, True)):
        tok, o = ctx.encode_batch(text, off, bos, eos, one_pretoken=True)
        o = o.astype(np.int64)
        for i, w in enumerate(words):
            exp = ([bos_id] if bos else []) + (oracle.bpe_raw(w) if w else []) + ([eos_id] if eos else [])
            assert tok[o[i]:o[i + 1]].tolist() == exp
    # special-token text is bytes like any other in this mode (the C++ mirror looks the whole string up among the special
    # tokens before it comes here, as the reference's lookup table would: tests/test_host_api.py)
    got = ctx.encode_bpe([b"<|begin_of_text|>"])[0]
    assert oracle.decode(got) == b"<|begin_of_text|>" and len(got) > 1
    # and the pre-tokenizer's splits are not made: one merge problem over blanks, digits and newlines
    for w in (b"Hello world! It's 12345", b"a\n\nb", b"     x"):
        assert ctx.encode_bpe([w])[0] == oracle.bpe_raw(w)
