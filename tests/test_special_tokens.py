"""New(WithSpecialTokens(list)) -- options.go:17-37, tokenizer.go:194-204,226,232-239,259,357-368.

The list replaces the 256 default special tokens behind the base vocabulary.  What the reference then does, and what
the oracle and the HIP path both have to reproduce:
  * ids follow the list: base size + position; VocabSize shrinks or grows with it;
  * Encode still splits the text with the regex of the 256 DEFAULT names (tokenizer.go:27,232); a match is special
    only if tokenLookup knows it (:236), otherwise the matched text is pre-tokenized ON ITS OWN as plain text;
  * a name of the list that the default regex does not match never comes out of Encode as one token;
  * BOS / EOS are added only when <|begin_of_text|> / <|end_of_text|> are in the list (:226,259).
The reference holds no golden ids for a custom list (options_test.go:24-63 only checks the option's validation), so
the expectations below that are not plain consequences of those lines are marked as the oracle's reading."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle import Oracle
from tokenizer_amd import capi

CUSTOM = [b"<|begin_of_text|>", b"<|my_tok|>", b"<|start_header_id|>"]   # no <|end_of_text|>, no <|eot_id|>
TEXTS = [
    b"",
    b"Hello <|start_header_id|>user<|end_header_id|>\n\nHi<|eot_id|>",
    b"a <|my_tok|> b<|my_tok|>",
    b" <|eot_id|>\n<|eot_id|> x<|end_of_text|>",
    b"<|begin_of_text|><|begin_of_text|>x<|start_header_id|>",
    b"<|reserved_special_token_5|>1234<|reserved_special_token_250|>",
    "mixed éè <|eom_id|>你好<|python_tag|>'re".encode(),
    b"<|start_header_id|" + b"<" * 40 + b"|start_header_id|>" * 3,
]


@pytest.fixture(scope="module")
def custom_oracle():
    return Oracle(specials=CUSTOM)


def test_option_validation_is_the_references():
    # options_test.go:24-63 through the C ABI (l3_create_ex checks the list before it touches a device)
    lib = capi.load_library()
    vocab, merges = capi.read_data_files()

    def create(names):
        h = C.c_void_p()
        arr = (C.c_char_p * max(len(names), 1))(*names)
        rc = lib.l3_create_ex(vocab, len(vocab), merges, len(merges), arr, len(names), 0, C.byref(h))
        if rc == capi.L3_OK:
            lib.l3_destroy(h)
        return rc

    assert create([b"<|test|>", b"<|another|>"]) in (capi.L3_OK, capi.L3_E_NO_DEVICE)   # "valid special tokens"
    assert create([b"<|test|>", b"invalid"]) == capi.L3_E_INVALID_ARG                    # "invalid special token format"
    assert b"invalid" in lib.l3_last_error(None)
    assert create([b"<|test|>", b"<|test|>"]) == capi.L3_E_INVALID_ARG                   # "duplicate special tokens"
    assert create([b"<||>"]) == capi.L3_E_INVALID_ARG                                    # len < 5 (options.go:21)
    assert create([b"<|a|>"]) in (capi.L3_OK, capi.L3_E_NO_DEVICE)
    assert create([]) in (capi.L3_OK, capi.L3_E_NO_DEVICE)                               # empty, non-nil: no special tokens


def test_oracle_custom_list_consequences(custom_oracle):
    o, d = custom_oracle, Oracle()
    assert o.vocab_size() == 128000 + len(CUSTOM)
    # ids follow the list
    assert o.encode(b"<|start_header_id|>", False, False) == [128002]
    assert o.encode(b"<|begin_of_text|>", False, False) == [128000]
    # BOS is in the list, EOS is not: only BOS is added
    assert o.encode(b"", True, True) == [128000]
    assert o.encode(b"hi", True, True) == [128000] + d.encode(b"hi", False, False)
    # a listed name outside the default regex is plain text
    assert o.encode(b"<|my_tok|>", False, False) == d.encode(b"<|my_tok|>", False, False)
    # a default name that is not in the list: plain text, but split off from its surroundings first
    def plain(b):   # the bare pre-tokenizer (knows no special tokens) + BPE per piece
        return [t for piece in d.pretokenize_strings(b) for t in d.bpe_raw(piece)]
    assert o.encode(b" <|eot_id|>\nx", False, False) == plain(b" ") + plain(b"<|eot_id|>") + plain(b"\nx")
    assert o.encode(b" <|eot_id|>\nx", False, False) != plain(b" <|eot_id|>\nx")   # " <|" and "|>\n" would be single pieces
    # decode follows the list too
    assert o.decode([128001]) == b"<|my_tok|>" and o.decode([128003]) == b""


def test_oracle_empty_list_has_no_special_tokens():
    o = Oracle(specials=[])
    assert o.vocab_size() == 128000
    assert o.encode(b"x", True, True) == Oracle().encode(b"x", False, False)
    assert len(o.encode(b"<|begin_of_text|>", False, False)) > 1


@pytest.mark.gpu
def test_hip_matches_oracle_with_custom_list(custom_oracle):
    ctx = capi.Context(0, specials=CUSTOM)
    try:
        assert ctx.vocab_size() == 128003
        assert ctx.special_token_id(b"<|start_header_id|>") == 128002
        assert ctx.special_token_id(b"<|my_tok|>") == 128001
        with pytest.raises(capi.L3Error):
            ctx.special_token_id(b"<|end_of_text|>")
        for bos in (False, True):
            for eos in (False, True):
                got = ctx.encode_docs(TEXTS, bos=bos, eos=eos)
                for t, g in zip(TEXTS, got):
                    assert g == custom_oracle.encode(t, bos, eos), (t, bos, eos)
        # a batch big enough for several tiles, special-looking text at tile seams
        rng = np.random.default_rng(5)
        words = [b"<|eot_id|>", b"<|start_header_id|>", b"<|my_tok|>", b" the", b" 12345", b"\n\n", b"<|", b"|>", "é".encode(), b" x"]
        docs = [b"".join(words[i] for i in rng.integers(0, len(words), size=int(rng.integers(1, 3000)))) for _ in range(24)]
        got = ctx.encode_docs(docs, bos=True, eos=True)
        for t, g in zip(docs, got):
            assert g == custom_oracle.encode(t, True, True)
        # decode follows the list
        out, off = ctx.decode_batch([128000, 128001, 128002, 128003, 9906], [0, 5])
        assert out.tobytes() == b"<|begin_of_text|><|my_tok|><|start_header_id|>Hello" and off.tolist() == [0, len(out)]
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hip_empty_list():
    ctx = capi.Context(0, specials=[])
    o = Oracle(specials=[])
    try:
        assert ctx.vocab_size() == 128000
        texts = [b"", b"<|begin_of_text|>x<|eot_id|> y", b"plain"]
        assert ctx.encode_docs(texts, bos=True, eos=True) == [o.encode(t, True, True) for t in texts]
    finally:
        ctx.close()
