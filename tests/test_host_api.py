"""include/llama3.hpp -- the C++ mirror of the reference's Go API -- driven through tests/cpp/host_shim.cpp.

CPU part: the scanner's chunking (scanner.go:124-181,281-311,337-478) against the Python restatement in
tests/scanner_model.py, on the readers of the reference's scanner_test.go.
GPU part: the reference's own API tests (tokenizer_test.go, performance_test.go, scanner_test.go) restated
against llama3::Tokenizer, with the oracle as the expected value wherever the reference test compares against
its own Encode."""
import os
import random
import struct

import numpy as np
import pytest

from tests import hostapi, scanner_model

FOX = "The quick brown fox jumps over the lazy dog. "


def _model(data, read_sizes=(), buf_size=4096, max_buffer=1 << 20):
    return scanner_model.chunks(data, list(read_sizes), buf_size, max_buffer)


def _both(data, read_sizes=(), buf_size=4096, max_buffer=1 << 20):
    got = hostapi.chunks(data, list(read_sizes), buf_size, max_buffer)
    assert got == _model(data, read_sizes, buf_size, max_buffer)
    return got


# ---------------------------------------------------------------------------------------- CPU: chunking
def test_chunker_reference_cases():
    # scanner_test.go TestScanner inputs: one chunk, EOF known when it is encoded
    for s in ["Hello world", "Hello 世界 🦙 café"]:
        assert _both(s.encode()) == [(s.encode(), True)]
    assert _both(b"") == [(b"", True)]
    # large_text (4,500 B): first read fills 4,096 B and does not end in white space, but is > bufSize / 2
    big = (FOX * 100).encode()
    got = _both(big)
    # (the rest ends in a blank, so it is cut before EOF is seen and the empty read follows)
    assert [len(c) for c, _ in got] == [4096, 404, 0] and [l for _, l in got] == [False, False, True]
    # input ending in white space: the chunk is cut before EOF is seen, the empty read follows
    assert _both(b"hello ") == [(b"hello ", False), (b"", True)]


def test_chunker_one_byte_reader_keeps_utf8_whole():
    # scanner_test.go:168-192 utf8_boundary (slowReader, 1 byte per read)
    data = "Hello 世界 world".encode()
    got = _both(data, [1] * 64)
    assert b"".join(c for c, _ in got) == data
    for c, _ in got:
        c.decode("utf-8")  # no chunk splits a sequence
    assert [c for c, _ in got] == [b"Hello ", "世界 ".encode(), b"world"]


def test_chunker_buffer_limits():
    # scanner_test.go:194-319: bufSize 32 / maxBuffer 64 and 128
    cases = [
        (("This is a very long word: " + "a" * 100 + " and more text").encode(), (), 32, 64),
        (("a" * 62 + "世界").encode(), (), 32, 64),
        (("a" * 63 + "世界").encode(), (64, 10), 64, 64),
        ((FOX * 20).encode(), (), 32, 128),
        (("test " * 1000).encode(), (), 128, 512),
    ]
    for data, sizes, bs, mb in cases:
        got = _both(data, sizes, bs, mb)
        assert b"".join(c for c, _ in got) == data
        assert all(len(c) <= mb + 4 for c, _ in got)
        for c, _ in got:
            c.decode("utf-8")


def test_chunker_reader_error():
    # scanner_test.go:154-166 errorReader; an error after some data loses nothing that was already chunked
    assert hostapi.chunks(b"", [-1]) is None
    assert hostapi.chunks(b"abc def", [4, -1]) is None


def test_chunker_fuzz_against_model():
    rng = random.Random(1234)
    alphabet = ["a", "b", " ", "\n", "é", "世", "🦙", "\t", "x" * 9, "\r\n", "\x80", "\xff"]
    for it in range(400):
        n = rng.randrange(0, 300)
        parts = []
        for _ in range(n):
            p = rng.choice(alphabet)
            parts.append(p.encode("utf-8") if ord(p[0]) < 0x80 or len(p.encode("utf-8")) > 1 and p not in ("\x80", "\xff")
                         else bytes([ord(p)]))
        data = b"".join(parts)
        bs = rng.choice([1, 2, 3, 4, 7, 8, 16, 32, 64, 4096])
        mb = rng.choice([1, 4, 5, 16, 33, 64, 128, 1 << 20])
        sizes = [rng.randrange(1, 2 * bs + 2) for _ in range(rng.randrange(0, 40))]
        got = _both(data, sizes, bs, mb)
        joined = b"".join(c for c, _ in got)
        # the reference drops nothing except (its quirk) an incomplete tail split off exactly at maxBuffer at EOF
        assert data.startswith(joined) and len(data) - len(joined) <= 3
        assert got[-1][1] or len(got[-1][0]) >= mb


# ---------------------------------------------------------------------------------------- GPU: the API
@pytest.fixture(scope="module")
def tok():
    t = hostapi.Tokenizer(0)
    yield t
    t.close()


NO = (False, False)


@pytest.mark.gpu
def test_new_missing_data_files():
    with pytest.raises(RuntimeError, match="error kind %d" % hostapi.ERR_DATA_NOT_FOUND):
        hostapi.Tokenizer(0, "/nonexistent")


@pytest.mark.gpu
def test_encode_defaults_and_options(tok, oracle):
    # tokenizer_test.go TestTokenizerEncode: nil options = BOS + EOS; explicit options honoured
    for text in [b"", b"Hello world!", "Unicode: 你好世界 🦙".encode(), b"<|begin_of_text|>Hello<|end_of_text|>",
                 b"I've 1234567 things\r\n\r\n  to do   ", b"\xff\xfe broken \xe4\xb8"]:
        assert tok.encode(text) == oracle.encode(text, True, True)
        for bos in (False, True):
            for eos in (False, True):
                assert tok.encode(text, (bos, eos)) == oracle.encode(text, bos, eos)
    assert tok.encode(b"") == [128000, 128001] and tok.encode(b"", NO) == []
    assert tok.encode(b"Hello world!", NO) == [9906, 1917, 0]


@pytest.mark.gpu
def test_decode_cases(tok):
    # tokenizer_test.go:151-197
    assert tok.decode([9906, 1917, 0]) == b"Hello world!"
    assert tok.decode([128000, 40, 128001]) == b"<|begin_of_text|>I<|end_of_text|>"
    assert tok.decode([]) == b""
    assert tok.decode([-1, 999999999]) == b""
    assert tok.decode([128000, 128006, 128004, 128008, 128010]) == \
        b"<|begin_of_text|><|start_header_id|><|finetune_right_pad_id|><|eom_id|><|python_tag|>"
    # tokenizer_test.go:448-471
    text = "Hello, world! 🦙".encode()
    assert tok.decode(tok.encode(text, NO)) == text


@pytest.mark.gpu
def test_special_token_ids_and_errors(tok):
    # tokenizer_test.go:199-267
    assert tok.special(b"<|begin_of_text|>") == (128000, hostapi.ERR_NONE)
    assert tok.special(b"<|end_of_text|>") == (128001, hostapi.ERR_NONE)
    assert tok.special(b"not_a_special_token")[1] == hostapi.ERR_INVALID_TOKEN
    assert tok.special(b"<|unknown_token|>")[1] == hostapi.ERR_TOKEN_NOT_FOUND
    assert tok.special(b"")[1] == hostapi.ERR_INVALID_TOKEN
    assert tok.vocab_size() == 128256


@pytest.mark.gpu
def test_large_text_known_length(tok, oracle):
    # tokenizer_test.go:313-346: 373 tokens incl. BOS/EOS for the llama paragraph (golden fixture holds the text)
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_encode_kats.json")))["large"]
    text = bytes.fromhex(g["input"])
    ids = tok.encode(text)
    assert len(ids) == int(g["expected_len_bos_eos"]) == 373 and ids == oracle.encode(text, True, True)
    assert tok.decode(ids) == b"<|begin_of_text|>" + text + b"<|end_of_text|>"


@pytest.mark.gpu
def test_append_tokens(tok, oracle):
    # tokenizer_test.go:368-406 / performance_test.go:93-166
    assert tok.append_tokens([], b"Hello", NO) == tok.encode(b"Hello", NO)
    assert tok.append_tokens(tok.encode(b"Hello", NO), b" world", NO) == oracle.encode(b"Hello world", False, False)
    assert tok.append_tokens([1, 2, 3], b"Hello", (True, True)) == [1, 2, 3] + oracle.encode(b"Hello", True, True)
    assert tok.append_tokens([7], b"", (True, False)) == [7, 128000]


def _byte_level(raw: bytes) -> bytes:
    """internal/encoding/encoding.go:26-79 (the GPT-2 byte alphabet), built the way the reference builds it"""
    bs = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    m = dict(zip(bs, cs))
    return "".join(chr(m[b]) for b in raw).encode("utf-8")


@pytest.mark.gpu
def test_pretokenize(tok, oracle):
    # Tokenizer.PreTokenize (tokenizer.go:490-506): no special-token split, []rune conversion, byte-level alphabet
    import json
    import os
    texts = [b"Hello world!  It's 12345 o'clock\n\n  x", "名前は 🦙<|eot_id|>ok".encode(), b"", b"bad \xff\xe4\xb8 tail\xf0\x9f",
             bytes(range(256))]
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_pretokenize_kats.json")))
    texts += [bytes.fromhex(c["input"]) for c in g["cases"]]
    for text in texts:
        exp = oracle.pretokenize_strings(text)
        assert tok.pretokenize(text, raw=True) == exp
        assert tok.pretokenize(text) == [_byte_level(p) for p in exp]
    for c in g["cases"]:  # the reference's own expectations (pretokenizer tests), straight from the fixture
        assert tok.pretokenize(bytes.fromhex(c["input"]), raw=True) == [bytes.fromhex(x) for x in c["expected"]]
    assert tok.pretokenize("Hello 世界".encode()) == ["Hello".encode(), "Ġä¸ĸçķĮ".encode()]


@pytest.mark.gpu
def test_encode_bpe(tok, oracle):
    # BPE.EncodeBPE (tokenizer.go:80-84,478-488): one pre-token in the byte-level alphabet -> PerformBPE (bpe.go:18-71)
    import random
    rng = random.Random(9)
    words = [b"hello", b" world", b"Hello", b" the", b"12345", b"\n\n", b"\xff\xfe", bytes(range(256)), b"a" * 70, b" supercalifragilistic" * 30,
             "名前は".encode(), b"", b"x"]
    words += [bytes(rng.randrange(256) for _ in range(rng.randrange(1, 90))) for _ in range(60)]
    for piece in oracle.pretokenize_strings(b"The quick brown fox can't jump over 1234567 lazy dogs!!\n\n  It's done."):
        words.append(piece)
    for w in words:
        if b"<|" in w:
            continue
        assert tok.encode_bpe(_byte_level(w)) == (oracle.bpe_raw(w) if w else []), w
    # the reference's lookup table holds the special tokens: the whole string is found there first (bpe.go:27-33)
    assert tok.encode_bpe(b"<|eot_id|>") == [128009] == oracle.bpe_raw(b"<|eot_id|>")
    assert tok.encode_bpe(b"<|begin_of_text|>") == [128000]
    # runes outside the alphabet are skipped (bpe.go:84-92): a raw blank, an ideograph
    assert tok.encode_bpe("hel你lo wor ld".encode()) == oracle.bpe_raw(b"helloworld")
    # Encode == the special split, then EncodeBPE over PreTokenize's pieces (tokenizer.go:236-252)
    text = b"Tokenizers split text; BPE merges it. \xe4\xbd\xa0\xe5\xa5\xbd 42!"
    ids = []
    for piece in tok.pretokenize(text):
        ids += tok.encode_bpe(piece)
    assert ids == oracle.encode(text, False, False)


@pytest.mark.gpu
def test_optimistic_count(tok, oracle):
    # tokenizer_test.go:408-445 (lower bounds) and tokenizer.go:370-417 restated with the oracle: anything shaped like a
    # special token counts one, the text between is plain text, BOS and EOS always count
    import re
    pat = re.compile(rb"<\|[a-zA-Z0-9_]+\|>")

    def expect(text):
        pieces = pat.split(text)
        return 2 + len(pat.findall(text)) + sum(len(oracle.encode(p, False, False)) for p in pieces if p)

    for text, lo in [(b"Hello world", 3), (b"<|begin_of_text|>Hello<|end_of_text|>", 4), (b"<|custom_token|>Hello<|another_custom|>", 5)]:
        got = tok.optimistic_count(text)
        assert got >= lo and got == expect(text)
    for text in [b"", b"<|x|>", b"<|x|><|y_2|>", b"a<||>b", b"<|not closed", b"<|a b|>", b"<|<|ok|>|>", b"x<|a|", b"tail <|",
                 "名前<|eot_id|>は 🦙 12345<|made_up|>\n\n".encode(), b"<|a|>" * 50 + b" end", b"\xff<|q|>\xfe"]:
        assert tok.optimistic_count(text) == expect(text), text


def _expected_scan(oracle, data, read_sizes=(), buf_size=4096, max_buffer=1 << 20, opts=NO):
    """What the reference scanner yields: Encode per chunk, BOS on the first, EOS once EOF is known."""
    out = []
    sent_bos = False
    for chunk, last in scanner_model.chunks(data, list(read_sizes), buf_size, max_buffer):
        if chunk:
            out += oracle.encode(chunk, opts[0] and not sent_bos, False)
            sent_bos = sent_bos or opts[0]
        elif opts[0] and not sent_bos:
            out.append(128000)
            sent_bos = True
        if last and opts[1]:
            out.append(128001)
    return out


@pytest.mark.gpu
def test_scanner_reference_cases(tok, oracle):
    # scanner_test.go:11-116: equal to direct encoding except large_text (chunk boundaries)
    for text, opts in [("Hello world", NO), ("Hello world", (True, True)), ("", (True, True)),
                       ("Hello 世界 🦙 café", NO)]:
        got = tok.scan(text.encode(), opts=opts)
        assert got == oracle.encode(text.encode(), *opts) == _expected_scan(oracle, text.encode(), opts=opts)
    big = (FOX * 100).encode()
    got = tok.scan(big)
    assert got == _expected_scan(oracle, big) and 500 <= len(got) <= 2000
    assert tok.scan(b"", opts=NO) == []
    assert tok.scan(b"hello ", opts=(True, True)) == [128000] + oracle.encode(b"hello ", False, False) + [128001]


@pytest.mark.gpu
def test_scanner_options_and_edge_cases(tok, oracle):
    # scanner_test.go:118-320
    data = ("test " * 1000).encode()
    got = tok.scan(data, buf_size=128, max_buffer=512)
    assert got == _expected_scan(oracle, data, (), 128, 512) and len(got) > 0
    assert tok.scan(b"", [-1]) is None  # reader_error
    data = "Hello 世界 world".encode()
    got = tok.scan(data, [1] * 64)
    assert tok.decode(got) == data and got == _expected_scan(oracle, data, [1] * 64)
    for data, sizes, bs, mb in [
        (("This is a very long word: " + "a" * 100 + " and more text").encode(), (), 32, 64),
        (("a" * 62 + "世界").encode(), (), 32, 64),
        (("a" * 63 + "世界").encode(), (64, 10), 64, 64),
        ((FOX * 20).encode(), (), 32, 128),
    ]:
        got = tok.scan(data, sizes, bs, mb)
        assert tok.decode(got) == data
        assert got == _expected_scan(oracle, data, sizes, bs, mb)


@pytest.mark.gpu
def test_process_wire_format(tok, oracle):
    # scanner_test.go:322-369 + llama3/scanner.go:77-102: uint32 little-endian per token, count returned
    for text in [b"Hello world", b"", ("test " * 1000).encode()]:
        count, raw = tok.process(text)
        assert len(raw) == 4 * count
        ids = list(struct.unpack("<%dI" % count, raw))
        assert ids == _expected_scan(oracle, text)
        assert (count > 0) == (len(text) > 0)
    count, raw = tok.process(b"Hi", (True, True))
    assert list(struct.unpack("<%dI" % count, raw)) == oracle.encode(b"Hi", True, True)


@pytest.mark.gpu
def test_many_streams_one_batch(tok, oracle):
    # additive API: chunks of many streams encoded by one l3_encode_batch call == one Scanner per stream
    rng = random.Random(7)
    words = ["alpha", "beta", " ", "\n", "世界", "🦙", "1234567", "don't", "   ", "<|eot_id|>", "x" * 70]
    streams = [b""] + ["".join(rng.choice(words) for _ in range(rng.randrange(1, 400))).encode() for _ in range(40)]
    for opts in (NO, (True, True)):
        got = tok.streams(streams, buf_size=256, opts=opts)
        for s, g in zip(streams, got):
            assert g == tok.scan(s, buf_size=256, opts=opts) == _expected_scan(oracle, s, (), 256, 1 << 20, opts)


@pytest.mark.gpu
def test_token_stream_channel(tok, oracle):
    # llama3/scanner.go:107-132 (scanner_test.go TestTokenStream): the tokens of NewScanner(r) -- no BOS/EOS -- through a
    # channel of 100; more than 100 tokens exercises the blocking sender, `take` drops the stream while it is blocked
    for text in [b"Hello world", b"", ("test " * 1000).encode(), (FOX * 50).encode()]:
        assert tok.token_stream(text) == _expected_scan(oracle, text)
    big = ("word " * 5000).encode()
    want = _expected_scan(oracle, big)
    assert len(want) > 1000
    assert tok.token_stream(big, take=7) == want[:7]
    assert tok.token_stream(big, take=0) == []
    assert tok.token_stream(b"", [-1]) is None                    # the reader's error arrives on the error channel
    assert tok.token_stream(big, [64, 64, -1]) is None


@pytest.mark.gpu
def test_new_with_special_tokens(oracle):
    # options.go:17-37 + tokenizer.go:151-155: a bad list fails New with a ConfigError; a good one replaces the defaults
    from tests.oracle import Oracle
    for bad in ([b"<|test|>", b"invalid"], [b"<|test|>", b"<|test|>"], [b"<||>"]):
        with pytest.raises(RuntimeError, match="error kind %d" % hostapi.ERR_CONFIG):
            hostapi.Tokenizer(0, specials=bad)
    names = [b"<|begin_of_text|>", b"<|my_tok|>", b"<|start_header_id|>"]
    t, o = hostapi.Tokenizer(0, specials=names), Oracle(specials=names)
    try:
        assert t.vocab_size() == 128003
        assert t.special(b"<|my_tok|>") == (128001, hostapi.ERR_NONE)
        assert t.special(b"<|end_of_text|>")[1] == hostapi.ERR_TOKEN_NOT_FOUND
        assert t.special(b"<|>")[1] == hostapi.ERR_TOKEN_NOT_FOUND          # tokens/special.go:43-45: prefix and suffix may overlap
        text = b"Hello <|start_header_id|>user<|eot_id|> <|my_tok|>"
        assert t.encode(text) == o.encode(text, True, True) and t.encode(text)[0] == 128000 and t.encode(text)[-1] != 128001
        # the scanner adds BOS / EOS through GetSpecialTokenID too (scanner.go:188,195,232): no <|end_of_text|>, no EOS
        assert t.scan(b"hi there", opts=(True, True)) == o.encode(b"hi there", True, False)
        assert t.scan(b"", opts=(True, True)) == [128000]
        assert t.streams([b"hi there", b""], opts=(True, True)) == [o.encode(b"hi there", True, False), [128000]]
        assert t.decode([128001, 128002]) == b"<|my_tok|><|start_header_id|>"
        # OptimisticCount counts BOS / EOS only when the tokenizer has them (tokenizer.go:378-380,412-414)
        assert t.optimistic_count(b"hi <|whatever|> there") == 1 + len(o.encode(b"hi ", False, False)) + 1 + len(o.encode(b" there", False, False))
    finally:
        t.close()


@pytest.mark.gpu
def test_with_devices_shards_streams(tok, oracle):
    # WithDevices: the same API over several contexts (here three on device 0); EncodeStreams is one sharded batch
    t = hostapi.Tokenizer(devices=[0, 0, 0])
    try:
        rng = random.Random(11)
        words = ["alpha", "beta", " ", "\n", "世界", "1234567", "don't", "<|eot_id|>"]
        streams = ["".join(rng.choice(words) for _ in range(rng.randrange(1, 600))).encode() for _ in range(30)]
        got = t.streams(streams, buf_size=256, opts=(True, True))
        for i, s in enumerate(streams):
            want = _expected_scan(oracle, s, (), 256, 1 << 20, (True, True))
            if got[i] != want:
                k = next((j for j in range(min(len(got[i]), len(want))) if got[i][j] != want[j]), min(len(got[i]), len(want)))
                raise AssertionError("stream %d: %d ids, want %d; first difference at %d: %r != %r; single context right: %r" % (
                    i, len(got[i]), len(want), k, got[i][max(0, k - 4):k + 12], want[max(0, k - 4):k + 12],
                    tok.streams([s], buf_size=256, opts=(True, True))[0] == want))
        assert t.encode(b"Hello world!") == oracle.encode(b"Hello world!", True, True)
        assert t.decode([9906, 1917, 0]) == b"Hello world!" and t.vocab_size() == 128256
    finally:
        t.close()


def test_data_loader_packs_the_reference_file_formats(tmp_path):
    # WithDataLoader (options.go:52-60, vocab.go:14-22): tokens + merge map -> the two file formats the C ABI takes.  A loader
    # over the shipped files must give those files back (the merge priorities only count through their order), and over a
    # truncated vocabulary with the FULL merge list it must drop the rules that name unknown tokens (decoder.go:61-63).
    import base64
    from tests import custom_vocab
    vocab = open(os.path.join(hostapi.DATA, "vocab_base64.txt"), "rb").read().strip()
    merges = open(os.path.join(hostapi.DATA, "merges_binary.txt"), "rb").read().strip()
    def same_vocab(a, b):   # the same token lines (a final newline or not is the same file to decoder.go:16-36)
        return base64.b64decode(a).rstrip(b"\n") == base64.b64decode(b).rstrip(b"\n")

    def same_merges(a, b, n_pairs=None):   # the same 17-bit pairs; what follows the last whole pair is padding
        a, b = base64.b64decode(a), base64.b64decode(b)
        n = min(len(a), len(b)) * 8 // 34 if n_pairs is None else n_pairs
        bits = lambda x: np.unpackbits(np.frombuffer(x, dtype=np.uint8))[:n * 34]   # noqa: E731
        return abs(len(a) - len(b)) <= 4 and np.array_equal(bits(a), bits(b))

    for step in (1, 7):
        v, m = hostapi.pack_loader(rank_step=step)
        assert same_vocab(v, vocab) and same_merges(m, merges)
    tv, tm, n_kept = custom_vocab.truncated_vocabulary(30000)
    tv_full, tm_full, _ = custom_vocab.truncated_vocabulary(30000, full_merges=True)
    d = tmp_path / "trunc"
    d.mkdir()
    (d / "vocab_base64.txt").write_bytes(tv_full)
    (d / "merges_binary.txt").write_bytes(tm_full)
    v, m = hostapi.pack_loader(str(d))
    assert same_vocab(v, tv)
    # the loader keeps every rule whose two operands exist (the product may be missing: such a rule produces nothing)
    assert len(base64.b64decode(m)) * 8 // 34 >= n_kept
    (d / "merges_binary.txt").write_bytes(tm)         # only the rules of the truncated vocabulary: given back as they are
    v, m = hostapi.pack_loader(str(d))
    assert same_vocab(v, tv) and same_merges(m, tm, n_kept)


@pytest.mark.gpu
def test_new_with_data_loader(tok, oracle, tmp_path):
    from tests import custom_vocab
    from tests.oracle import Oracle
    with pytest.raises(RuntimeError, match="error kind %d" % hostapi.ERR_CONFIG):
        hostapi.Tokenizer(0, loader=(1, False, True))                 # WithDataLoader(nil): ConfigError
    with pytest.raises(RuntimeError, match="error kind %d" % hostapi.ERR_DATA_NOT_FOUND):
        hostapi.Tokenizer(0, loader=(1, True, False))                 # the loader's LoadMerges fails: New fails with it
    texts = [b"Hello world!", "Unicode: 你好世界 🦙 don't".encode(), b"<|begin_of_text|>x<|eot_id|>", b"1234567 \n\n  z", b""]
    t = hostapi.Tokenizer(0, loader=(5, False, False))                # the shipped data through the loader interface
    try:
        assert t.vocab_size() == 128256
        for x in texts:
            assert t.encode(x) == oracle.encode(x, True, True)
    finally:
        t.close()
    tv, tm, _ = custom_vocab.truncated_vocabulary(30000)
    d = tmp_path / "trunc"
    d.mkdir()
    (d / "vocab_base64.txt").write_bytes(tv)
    (d / "merges_binary.txt").write_bytes(tm)
    t, o = hostapi.Tokenizer(0, data_dir=str(d), loader=(1, False, False)), Oracle(tv, tm)
    try:
        assert t.vocab_size() == 30256 and t.special(b"<|begin_of_text|>") == (30000, hostapi.ERR_NONE)
        for x in texts:
            assert t.encode(x) == o.encode(x, True, True)
    finally:
        t.close()
