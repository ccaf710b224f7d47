"""Custom vocabulary data files (the reference's WithDataFiles option, options.go / vocab.go:46-118): the table builder
and the oracle take the two files' contents, nothing in the path assumes the shipped vocabulary.

CPU part: the oracle and the host-side table builder on a truncated vocabulary (tests/custom_vocab.py).
GPU part (marked gpu): the HIP path created from the same files against that oracle, bit-exact.
The reference holds no stored answers for a custom vocabulary: parity here is oracle-vs-GPU only.
"""
import random

import numpy as np
import pytest

from tests import c1_inputs, corpus
from tests.custom_vocab import truncated_vocabulary

N_BASE = 30000


@pytest.fixture(scope="module")
def files():
    v, m, n_merges = truncated_vocabulary(N_BASE)
    assert n_merges > 50000
    return v, m


@pytest.fixture(scope="module")
def small_oracle(files):
    from tests.oracle import Oracle
    return Oracle(*files)


def test_special_tokens_follow_the_base_vocabulary(small_oracle, oracle):
    o = small_oracle
    assert o.lib.l3o_vocab_size(o.h) == N_BASE + 256                       # tokenizer.go:194-198
    assert o.encode(b"", True, True) == [N_BASE, N_BASE + 1]              # GetSpecialTokenID: tokenizer.go:226,259
    assert o.encode(b"<|eot_id|>", False, False) == [N_BASE + 9]
    assert oracle.encode(b"<|eot_id|>", False, False) == [128009]
    text = b"Tokenization of uncommon words: floccinaucinihilipilification, \xe4\xb8\xad\xe6\x96\x87!"
    a, b = o.encode(text, False, False), oracle.encode(text, False, False)
    assert a != b and len(a) > len(b)                                       # fewer merges: more tokens
    assert max(a) < N_BASE
    assert o.decode(a) == text and o.decode([N_BASE] + a + [N_BASE + 1]) == b"<|begin_of_text|>" + text + b"<|end_of_text|>"


def test_repacked_full_vocabulary_is_the_shipped_one(oracle):
    # the packer of tests/custom_vocab.py against the shipped merges file: same tokenizer
    from tests.oracle import Oracle
    v, m, n = truncated_vocabulary(128000)
    assert n == 280147
    o = Oracle(v, m)
    rng = random.Random(3)
    for _ in range(300):
        d = corpus.random_doc(rng)
        assert o.encode(d, True, True) == oracle.encode(d, True, True)


def test_host_table_builder_on_custom_files(files, small_oracle):
    from tests.hostshim import HostShim
    h = HostShim(*files)
    st = h.stats()
    assert st["n_tokens"] == N_BASE + 256 and st["n_merges"] > 50000
    rng = random.Random(11)
    for _ in range(3000):
        w = corpus.random_word(rng)
        assert h.bpe(w) == small_oracle.bpe_raw(w), w
    for w in (b"a" * 300, "中文".encode() * 40, b" " * 90, bytes(range(33, 127)) * 2):
        got, _ = h.bpe_rounds(w)
        assert got == small_oracle.bpe_raw(w), w[:20]


def test_truncated_vocabulary_with_the_full_merges_file(files, small_oracle):
    """What the reference loads, this loads: the shipped merges file beside a truncated vocabulary.  Pairs that name ids
    beyond the token list are skipped (decoder.go:61-63); rules whose product is missing rank after every productive rule
    and change no output (bpe.go:157-161), so the tokenizer equals the one built from the filtered merges."""
    from tests.hostshim import HostShim
    from tests.oracle import Oracle
    v, m_full, _ = truncated_vocabulary(N_BASE, full_merges=True)
    assert v == files[0]
    o_full = Oracle(v, m_full)
    h = HostShim(v, m_full)
    assert h.stats()["n_tokens"] == N_BASE + 256
    rng = random.Random(12)
    words = [corpus.random_word(rng) for _ in range(3000)] + [b"a" * 300, "中文".encode() * 40, b" " * 90, bytes(range(33, 127)) * 2]
    for w in words:
        exp = o_full.bpe_raw(w)
        assert exp == small_oracle.bpe_raw(w), w[:20]      # the reference's behaviour on the full file == the filtered file
        assert h.bpe(w) == exp, w[:20]


@pytest.fixture(scope="module")
def small_ctx(files):
    from tokenizer_amd import capi
    c = capi.Context(0, *files)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_custom_vocabulary_parity(small_ctx, small_oracle):
    from tests.test_gpu_parity import check_batch
    ctx, o = small_ctx, small_oracle
    assert ctx.vocab_size() == N_BASE + 256
    assert ctx.special_token_id(b"<|begin_of_text|>") == N_BASE and ctx.special_token_id(b"<|python_tag|>") == N_BASE + 10
    assert ctx.encode_docs([b""], True, True)[0] == [N_BASE, N_BASE + 1]
    rng = random.Random(21)
    docs = [corpus.random_doc(rng, max_pieces=30) for _ in range(6000)]
    for bos, eos in ((True, True), (False, False), (True, False)):
        check_batch(ctx, o, docs, bos, eos, "custom vocabulary, fuzz")
    check_batch(ctx, o, c1_inputs.generated() + c1_inputs.comparison(), True, True, "custom vocabulary, C1 inputs")
    long_runs = [b"a" * 5000, "中".encode() * 3000, b" " * 9000 + b"y", b"\xff" * 700, b"ab" * 150, b"-" * 100,
                 bytes(rng.randrange(256) for _ in range(20000)), b"word " * 4000,
                 b"".join(b"x<|reserved_special_token_%d|>" % i for i in range(0, 248, 7))]
    check_batch(ctx, o, long_runs, True, True, "custom vocabulary, long runs")
    from tools import corpora
    text, off = corpora.english(2048, 1024)
    got_tok, got_off = ctx.encode_batch(text, off, True, True)
    exp_tok, exp_off = o.encode_batch(text, off, True, True, threads=8)
    assert np.array_equal(got_tok, exp_tok) and np.array_equal(got_off, exp_off)
    # decode with the same tables
    ids = got_tok[:int(got_off[3])]
    out, out_off = ctx.decode_batch(ids, got_off[:4])
    assert out.tobytes() == o.decode(ids.tolist())


# ---- merge lists the packed priority cannot express (rank mode, l3_common.h): rules in a random order, some repeated ----
@pytest.fixture(scope="module")
def shuffled_files():
    v, m, n_merges = truncated_vocabulary(8000, shuffle_seed=11, duplicates=200)
    assert n_merges > 10000
    return v, m


@pytest.fixture(scope="module")
def shuffled_oracle(shuffled_files):
    from tests.oracle import Oracle
    return Oracle(*shuffled_files)


def _shuffled_docs():
    rng = random.Random(21)
    docs = [corpus.random_doc(rng) for _ in range(1500)]
    docs += [b"x" * 700, b"abcabcabc" * 300, ("слово " * 400).encode(), b"The quick brown fox jumps over the lazy dog. " * 200,
             b"a" * 40000, b"", b"\xff\xfe" * 50 + b"tail"]
    return docs


def test_rank_mode_host_merge_equals_the_oracle(shuffled_files, shuffled_oracle, oracle):
    """The host build of the merge loops (the code the kernels share) on a merge list in random order against the oracle's
    heap: the greedy order follows the rank, whatever the ranks are."""
    from tests.hostshim import HostShim
    h = HostShim(*shuffled_files)
    rng = random.Random(5)
    words = set()
    for d in _shuffled_docs()[:400]:
        for w in d.replace(b"\n", b" ").split(b" "):
            if 1 <= len(w) <= 120 and b"<|" not in w:      # (special-token text: matched before BPE, not by it)
                words.add(w)
    words = sorted(words)[:3000] + [b"a" * 300, b"abcabcabc" * 30, bytes(rng.randrange(97, 123) for _ in range(500))]
    differs = 0
    for w in words:
        if h.vocab_lookup(w) >= 0:
            continue
        exp = shuffled_oracle.bpe_raw(w)
        assert h.bpe(w) == exp, w
        assert h.bpe_rounds(w)[0] == exp, w
        assert h.bpe_pieces(w)[0] == exp, w
        differs += exp != oracle.bpe_raw(w)
    assert differs > 100      # (it IS another tokenizer than the shipped one)


@pytest.mark.gpu
def test_rank_mode_gpu_equals_the_oracle(shuffled_files, shuffled_oracle):
    """WithDataFiles / WithDataLoader with a merge list in random order with repeated pairs: the GPU path (rank mode) against
    the oracle built from the same files, all merge paths (lane, wave, window, grid-wide)."""
    from tokenizer_amd import capi
    ctx = capi.Context(device=0, vocab=shuffled_files[0], merges=shuffled_files[1])
    try:
        docs = _shuffled_docs()
        text, off = capi.pack_docs(docs)
        for bos, eos in ((True, True), (False, False)):
            tok, toff = ctx.encode_batch(text, off, bos, eos)
            exp_tok, exp_off = shuffled_oracle.encode_batch(text, off, bos, eos, threads=8)
            assert np.array_equal(toff, exp_off) and np.array_equal(tok, exp_tok)
    finally:
        ctx.close()
