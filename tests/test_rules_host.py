"""CPU tests of the host/device-shared logic (position-local pre-tokenizer rules, table probes,
merge loop) and of the table builder, against the oracle.  No GPU needed."""
import base64
import json
import os
import random

import numpy as np
import pytest

from tests import corpus
from tests.hostshim import HostShim

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def shim():
    return HostShim()


def test_table_stats(shim):
    s = shim.stats()
    assert s["n_tokens"] == 128256 and s["n_merges"] == 280147
    assert s["class_blocks"] < 1024


def test_classes_match_oracle_everywhere(shim, oracle):
    # class codes: 1 letter, 2 digit; white space is 3 (NL), 4 (SP) or 5 (other)
    for cp in list(range(0, 0x3400)) + list(range(0x4DB0, 0x4E10)) + list(range(0x9FF0, 0xD800)) + \
            list(range(0xE000, 0x14700)) + list(range(0x16000, 0x19000)) + list(range(0x1AFF0, 0x20010)) + \
            list(range(0x2A6D0, 0x2A710)) + list(range(0x2B730, 0x2B830)) + list(range(0x2CEA0, 0x2CEC0)) + \
            list(range(0x2EBE0, 0x2EE70)) + list(range(0x2F800, 0x2FA20)) + list(range(0x30000, 0x32400)) + \
            [0xE0001, 0x10FFFF]:
        o = oracle.rune_class(cp)
        c = shim.class_of(cp)
        exp = 1 if o & 1 else 2 if o & 2 else (3 if cp in (10, 13) else 4 if cp == 32 else 5) if o & 4 else 0
        assert c == exp, hex(cp)


def test_vocab_lookup_and_decode_table(shim, oracle):
    rng = random.Random(1)
    ids = list(range(0, 300)) + [rng.randrange(128000) for _ in range(3000)] + list(range(127990, 128000))
    for tid in ids:
        raw = shim.token_bytes(tid)
        assert raw == oracle.decode([tid])
        assert shim.vocab_lookup(raw) == tid
    assert shim.vocab_lookup(b"definitely not a token \x00") == -1
    assert shim.token_bytes(128000) == b"<|begin_of_text|>" and shim.token_bytes(128255) == b"<|reserved_special_token_247|>"
    assert shim.token_bytes(128256) is None


def test_every_vocab_entry_found(shim):
    for tid in range(128000):
        assert shim.vocab_lookup(shim.token_bytes(tid)) == tid


def test_pair_priorities_follow_rank_order(shim):
    md = base64.b64decode(open(os.path.join(os.path.dirname(G), "..", "tokenizer_amd", "data", "merges_binary.txt"), "rb").read())
    bits = np.unpackbits(np.frombuffer(md, dtype=np.uint8))
    n = bits.size // 17
    ids = bits[:n * 17].reshape(-1, 17).dot(1 << np.arange(16, -1, -1)).astype(np.int64)
    pairs = ids[:n // 2 * 2].reshape(-1, 2)
    assert pairs.shape[0] == 280147
    rng = random.Random(2)
    idx = sorted(set([0, 1, 2, 280146] + [rng.randrange(280147) for _ in range(20000)]))
    prios = [shim.pair_prio(int(pairs[i, 0]), int(pairs[i, 1])) for i in idx]
    assert all(p >= 0 for p in prios)
    assert all(a < b for a, b in zip(prios, prios[1:]))          # order-isomorphic to rank
    assert shim.pair_prio(5, 5) == -1 or (5, 5) in set(map(tuple, pairs.tolist()))


def _check_split(shim, oracle, text):
    assert shim.split_document(text) == oracle.split_document(text), text


def test_split_on_reference_kat_inputs(shim, oracle):
    for c in json.load(open(os.path.join(G, "ref_pretokenize_kats.json")))["cases"]:
        _check_split(shim, oracle, bytes.fromhex(c["input"]))
    enc = json.load(open(os.path.join(G, "ref_encode_kats.json")))
    for c in enc["encode"]:
        _check_split(shim, oracle, bytes.fromhex(c["input"]))
    _check_split(shim, oracle, bytes.fromhex(enc["large"]["input"]))


def test_split_exhaustive_short(shim, oracle):
    # exhaustive over a small alphabet that exercises every rule, lengths 0..5
    alpha = [b"a", b"s", b"'", b" ", b"\t", b"\n", b"1", b"!", "é".encode(), " ".encode(), b"\xff"]
    import itertools
    for n in range(0, 6):
        for combo in itertools.product(alpha, repeat=n):
            _check_split(shim, oracle, b"".join(combo))


def test_split_fuzz(shim, oracle):
    rng = random.Random(1234)
    for _ in range(30000):
        _check_split(shim, oracle, corpus.random_doc(rng))
    for _ in range(3000):
        _check_split(shim, oracle, corpus.random_bytes_doc(rng, rng.randint(0, 40)))
    for s in corpus.SPECIALS + corpus.NEAR_MISSES:
        _check_split(shim, oracle, s)
        _check_split(shim, oracle, b"x " + s + b"\n y")
        _check_split(shim, oracle, s[:-1])


def test_split_long_runs(shim, oracle):
    for t in (b"1" * 1000, b"a" * 1000, b" " * 1000, b"word " * 200, b"'s" * 100, b"\n \n " * 50, "٣".encode() * 100,
              b"12" + "٣".encode() * 5 + b"7" * 8, b"\t" * 70 + b"x"):
        _check_split(shim, oracle, t)


def test_bpe_matches_oracle(shim, oracle):
    rng = random.Random(7)
    words = [corpus.random_word(rng) for _ in range(20000)]
    words += [w.encode() for w in ["grabbed", " grabbed", "tokenization", " antidisestablishmentarianism", "镇", "🦙", "🦙Ꙋ",
                                   " việc", "ektedir", " Привет", "文字文字文字", "�", "a�b"]]
    words += [corpus.random_bytes_doc(rng, rng.randint(1, 20)).replace(b"\xff", b"a") for _ in range(2000)]
    for w in words:
        assert shim.bpe(w) == oracle.bpe_raw(w), w


def test_bpe_unreachable_tokens(shim, oracle):
    # the 588 vocabulary entries the merge rules cannot build are found by the whole-string lookup
    for tid in (100769, 100827, 101503, 101673, 103084):
        raw = shim.token_bytes(tid)
        assert shim.bpe(raw) == [tid] == oracle.bpe_raw(raw)


def test_round_based_merge_is_exact(shim, oracle):
    """The slow kernel's round algorithm (all local minima below theta per round) against the sequential
    greedy merge of the oracle, with no whole-string shortcut: words, byte soup, runs, CJK, Cyrillic."""
    rng = random.Random(17)
    cases = [corpus.random_word(rng, lo=2, hi=60) for _ in range(20000)]
    cases += [bytes(rng.choice(b"ab") for _ in range(rng.randint(2, 80))) for _ in range(4000)]
    cases += [bytes(rng.choice(b"abcde ") for _ in range(rng.randint(2, 200))) for _ in range(3000)]
    cases += [corpus.random_bytes_doc(rng, rng.randint(2, 120)) for _ in range(3000)]
    cases += ["".join(chr(rng.randrange(0x4E00, 0x9FA5)) for _ in range(rng.randint(1, 300))).encode() for _ in range(300)]
    cases += ["".join(chr(rng.randrange(0x410, 0x450)) for _ in range(rng.randint(1, 200))).encode() for _ in range(500)]
    cases += [b"a" * k for k in range(2, 70)] + [b"ab" * 500, b"aab" * 300, b" " * 1000, b"\t" * 777, b"!" * 999, b"0" * 500]
    cases += [("the quick brown fox jumps over the lazy dog " * 20).encode(), "文字".encode() * 400, ("ab" * 40 + "c") .encode() * 30]
    total_rounds = total_merges = 0
    for w in cases:
        got, nr = shim.bpe_rounds(w)
        exp = oracle.bpe_raw(w) if len(w) > 128 or shim.vocab_lookup(w) < 0 else None
        if exp is None:  # the oracle would take the whole-string shortcut: compare with the plain merge loop
            exp = shim.bpe(w) if shim.vocab_lookup(w) < 0 else None
        if exp is not None:
            assert got == exp, w
            total_rounds += nr
            total_merges += len(w) - len(got)
    assert total_rounds < total_merges  # rounds really do batch several merges


def test_cutting_long_pretokens_is_exact(shim, oracle):
    """Pre-tokens cut at the positions no vocabulary entry can span (l3_common.h: prefix tables, reach, cuts), every
    piece merged on its own, against the oracle's PerformBPE of the whole pre-token (heap, linked list, whole-string
    shortcut): ideographs, Cyrillic, byte soup incl. U+FFFD expansions, words glued together, runs, every single token."""
    rng = random.Random(23)
    cases = [corpus.random_word(rng, lo=2, hi=60) for _ in range(20000)]
    cases += ["".join(chr(rng.randrange(0x4E00, 0x9FA5)) for _ in range(rng.randint(1, 400))).encode() for _ in range(1500)]
    cases += ["".join(chr(rng.randrange(0x410, 0x450)) for _ in range(rng.randint(1, 200))).encode() for _ in range(1000)]
    cases += ["".join(chr(rng.randrange(0x1F300, 0x1F700)) for _ in range(rng.randint(1, 60))).encode() for _ in range(500)]
    cases += [corpus.random_bytes_doc(rng, rng.randint(2, 300)).decode("utf-8", "replace").encode() for _ in range(4000)]
    cases += [corpus.random_bytes_doc(rng, rng.randint(2, 300)) for _ in range(3000)]        # raw bytes as symbols
    cases += [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 300))) for _ in range(3000)]
    cases += [bytes(rng.choice(b"!@#$%^&*()-=+[]{};:,.<>/?") for _ in range(rng.randint(2, 300))) for _ in range(2000)]
    words = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "words_4096.txt")).read().split()
    cases += ["".join(rng.choice(words) for _ in range(rng.randint(2, 40))).encode() for _ in range(2000)]
    cases += [b"a" * k for k in range(2, 70)] + [b"ab" * 500, b"aab" * 300, b" " * 1000, b"\t" * 777, b"!" * 999, b"\xef\xbf\xbd" * 300]
    cases += ["中文".encode() * 300, ("ab" * 40 + "c").encode() * 30, "пример".encode() * 90 + "汉".encode() * 50]
    cases += [shim.token_bytes(t) for t in range(256, 128000, 7)]                              # whole-string hits
    pieces = cut_cases = 0
    for w in cases:
        got, npc = shim.bpe_pieces(w)
        assert got == oracle.bpe_raw(w), w
        pieces += npc
        cut_cases += npc > 1
    assert cut_cases > 10000 and pieces > 100000    # the cuts are really exercised


def test_sequence_length_by_word_matches_the_generic_walk(shim):
    """seq_len_word (the tile kernel's register path for threads with no document start nearby) against seq_len
    (Go's utf8.DecodeRune acceptance) over every lead and second byte, third and fourth bytes from the
    edges of the byte classes."""
    assert shim.lib.l3t_seq_len_word_check() == 0


def test_direct_class_table_equals_the_two_stage_table(shim):
    """Tables::cls_direct (planes 0 and 1, one load per character in the tile kernel) holds the bytes of the two-stage
    table that tests/test_unicode_tables.py pins: every code point below 0x20000."""
    assert shim.lib.l3t_class_direct_check(shim.h) == 0
