"""The command line of SURVEY 8 f-3 (`tokenizer llama3 encode|decode|info`, llama3/cmd/llama3/*.go) over the C++
mirror: tokenizer_amd/lib/llama3-hip (tools/llama3_cli.cpp).  Output formats as in encode.go:135-201 / decode.go."""
import json
import os
import random
import subprocess

import pytest

from tests import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tokenizer_amd", "lib", "llama3-hip")
ENV = dict(os.environ, L3_DATA_DIR=os.path.join(ROOT, "tokenizer_amd", "data"))


def run(*args, stdin=None):
    p = subprocess.run([CLI] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV, timeout=110)
    return p.returncode, p.stdout, p.stderr


def test_cli_is_built_and_explains_itself():
    assert os.access(CLI, os.X_OK), "run `make -C tokenizer_amd/csrc` (or __graft_entry__.build())"
    rc, out, err = run()
    assert rc == 2 and b"encode|decode|info" in err
    rc, out, err = run("encode", "-o", "xml", "x")
    assert rc == 1 and b"unknown output format: xml" in err                     # encode.go:203-204


@pytest.mark.gpu
def test_cli_encode_formats():
    rc, out, _ = run("encode", "Hello, world!")
    assert rc == 0 and out == b"128000 9906 11 1917 0 128001\n"               # space separated (default)
    rc, out, _ = run("encode", "Hello,", "world!")                              # args joined by one blank
    assert out == b"128000 9906 11 1917 0 128001\n"
    rc, out, _ = run("encode", "--bos=false", "--eos=false", "-o", "json", "--count", "Hello, world!")
    assert json.loads(out) == {"count": 4, "tokens": [9906, 11, 1917, 0]} and out.startswith(b'{"count":4,"tokens":[')
    rc, out, _ = run("encode", "--count-only", "Hello, world!")
    assert out == b"6\n"
    rc, out, _ = run("encode", "--count-only", "--output", "json", "Hello, world!")
    assert out == b'{"count":6}\n'
    rc, out, _ = run("encode", "--count", "--no-bos", "Hello")
    assert out == b"count: 2\ntokens: 9906 128001\n"
    rc, out, _ = run("encode", "-o", "newline", "--count", "--eos=false", "Hello")
    assert out == b"count: 2\n128000\n9906\n"


@pytest.mark.gpu
def test_cli_stdin_metrics_and_round_trip(oracle):
    rng = random.Random(9)
    text = b"".join(corpus.random_doc(rng, max_pieces=40, special_rate=0.0) + b"\n" for _ in range(600))
    rc, out, err = run("encode", "-o", "newline", "--metrics", stdin=text)
    assert rc == 0, err
    lines = out.decode().split("\n")
    k = lines.index("metrics:")
    ids = [int(x) for x in lines[:k]]
    assert ids[0] == 128000 and ids[-1] == 128001
    m = dict(l.strip().split(": ") for l in lines[k + 1:] if l.strip())
    assert set(m) == {"latency", "tps", "input_bytes", "encode_latency", "encode_tps"}
    assert int(m["input_bytes"]) == len(text) and int(m["tps"]) > 0 and m["latency"][-1] == "s"
    # stdin goes through the scanner's chunking (scanner.go:124-181; a chunk may end inside a character, which then
    # encodes as invalid bytes -- the reference's behaviour, modelled by tests/scanner_model.py)
    from tests.test_host_api import _expected_scan
    assert ids == _expected_scan(oracle, text, opts=(True, True))
    rc, back, err = run("decode", "--skip-special", stdin=" ".join(map(str, ids)).encode())
    assert rc == 0 and back == oracle.decode(ids[1:-1]), err
    rc, out, _ = run("encode", "-o", "json", "--metrics", "--count", stdin=b"Hello, world!")
    j = json.loads(out)
    assert j["tokens"] == [128000, 9906, 11, 1917, 0, 128001] and j["count"] == 6 and j["metrics"]["input_bytes"] == 13
    # short inputs are one chunk: same ids as Encode
    doc = "The quick brown fox — 12345 don't! 中文 🦙".encode()
    rc, out, _ = run("encode", stdin=doc)
    assert [int(x) for x in out.split()] == oracle.encode(doc, True, True)


@pytest.mark.gpu
def test_cli_decode_and_info():
    rc, out, _ = run("decode", "128000", "9906", "11", "1917", "0", "128001")
    assert out == b"<|begin_of_text|>Hello, world!<|end_of_text|>"
    rc, out, _ = run("decode", "--skip-special", "128000", "9906", "11", "1917", "0", "128001")
    assert out == b"Hello, world!"
    rc, out, err = run("decode", "12x")
    assert rc == 1 and b'invalid token ID "12x"' in err                         # decode.go:64-67
    rc, out, err = run("decode", stdin=b"  \n")
    assert rc == 1 and b"no token IDs provided" in err
    rc, out, _ = run("info")
    assert rc == 0 and b"128256" in out and b"<|eot_id|>" in out and b"128009" in out
