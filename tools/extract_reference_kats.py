#!/usr/bin/env python3
"""Extract the reference's hard-coded known-answer vectors into tests/golden/*.json.

Reads (as text, never executes) the reference's Go test tables:
  llama3/tokenizer_test.go                         encode / decode KATs, 373-token paragraph
  llama3/internal/pretokenizer/state_machine_test.go   pre-token split KATs, class KATs
Only {input, expected} DATA is extracted (Go literals are parsed into bytes); no reference source
text is kept.  Strings are stored hex-encoded so that invalid UTF-8 survives JSON.

Run in the authoring container (needs /root/reference); the JSON files are committed so that the
tests run on the GPU box where the reference does not exist.
"""
import json
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def parse_go_string(src, i):
    """Parse a Go string literal starting at src[i] (a quote or backquote). Returns (bytes, next_i)."""
    if src[i] == "`":
        j = src.index("`", i + 1)
        return src[i + 1:j].replace("\r", "").encode("utf-8"), j + 1
    assert src[i] == '"', src[i:i + 20]
    out = bytearray()
    i += 1
    simple = {"n": 10, "t": 9, "r": 13, "\\": 92, '"': 34, "'": 39, "a": 7, "b": 8, "f": 12, "v": 11}
    while src[i] != '"':
        c = src[i]
        if c != "\\":
            out += c.encode("utf-8")
            i += 1
            continue
        e = src[i + 1]
        if e in simple:
            out.append(simple[e]); i += 2
        elif e == "x":
            out.append(int(src[i + 2:i + 4], 16)); i += 4
        elif e == "u":
            out += chr(int(src[i + 2:i + 6], 16)).encode("utf-8"); i += 6
        elif e == "U":
            out += chr(int(src[i + 2:i + 10], 16)).encode("utf-8"); i += 10
        elif e in "01234567":
            out.append(int(src[i + 1:i + 4], 8)); i += 4
        else:
            raise ValueError("escape \\" + e)
    return bytes(out), i + 1


def parse_go_rune(src, i):
    """Parse a Go rune literal 'x' at src[i]; returns (codepoint, next_i)."""
    assert src[i] == "'"
    j = i + 1
    if src[j] == "\\":
        e = src[j + 1]
        simple = {"n": 10, "t": 9, "r": 13, "\\": 92, "'": 39, '"': 34, "a": 7, "b": 8, "f": 12, "v": 11}
        if e in simple:
            return simple[e], j + 3
        if e == "x":
            return int(src[j + 2:j + 4], 16), j + 5
        if e == "u":
            return int(src[j + 2:j + 6], 16), j + 7
        if e == "U":
            return int(src[j + 2:j + 10], 16), j + 11
        raise ValueError(e)
    return ord(src[j]), j + 2


def skip_ws(src, i):
    while src[i] in " \t\r\n":
        i += 1
    return i


def parse_string_expr(src, i):
    """string literal [+ string literal ...] | strings.Repeat("x", n)."""
    parts = bytearray()
    while True:
        i = skip_ws(src, i)
        if src.startswith("strings.Repeat(", i):
            i += len("strings.Repeat(")
            s, i = parse_go_string(src, skip_ws(src, i))
            i = skip_ws(src, i); assert src[i] == ","
            m = re.match(r"\s*(\d+)\s*\)", src[i + 1:])
            parts += s * int(m.group(1)); i += 1 + m.end()
        else:
            s, i = parse_go_string(src, i)
            parts += s
        j = skip_ws(src, i)
        if src[j] == "+":
            i = j + 1
            continue
        return bytes(parts), i


def parse_string_slice(src, i):
    """[]string{ "a", "b", ... }"""
    i = skip_ws(src, i)
    assert src.startswith("[]string{", i), src[i:i + 30]
    i += len("[]string{")
    out = []
    while True:
        i = skip_ws(src, i)
        if src[i] == "}":
            return out, i + 1
        if src[i] == ",":
            i += 1; continue
        if src.startswith("//", i):
            i = src.index("\n", i); continue
        s, i = parse_string_expr(src, i)
        out.append(s)


def parse_int_slice(src, i):
    i = skip_ws(src, i)
    assert src.startswith("[]int{", i), src[i:i + 30]
    j = src.index("}", i)
    body = src[i + len("[]int{"):j]
    return [int(x) for x in body.replace("\n", " ").split(",") if x.strip()], j + 1


def hexs(b):
    return b.hex()


def extract_pretokenize():
    src = open(os.path.join(REF, "llama3/internal/pretokenizer/state_machine_test.go"), encoding="utf-8").read()
    cases = []
    for m in re.finditer(r"\binput:\s*", src):
        i = m.end()
        try:
            inp, j = parse_string_expr(src, i)
        except (AssertionError, ValueError):
            continue
        m2 = re.match(r"\s*,(?:\s*//[^\n]*)?\s*expected:\s*", src[j:])
        if not m2:
            continue
        k = j + m2.end()
        if not src.startswith("[]string{", k):
            continue
        exp, _ = parse_string_slice(src, k)
        line = src.count("\n", 0, m.start()) + 1
        cases.append({"line": line, "input": hexs(inp), "expected": [hexs(e) for e in exp]})
    # class KATs: rows {"name", 'x', true|false} inside t.Run("isLetter"|"isNumber"|"isWhitespace", ...)
    classes = []
    marks = [(m.start(), m.group(1)) for m in re.finditer(r't\.Run\("(isLetter|isNumber|isWhitespace)"', src)]
    fn_end = src.index("func TestStateMachineEdgeCases")
    for n, (pos, fn) in enumerate(marks):
        stop = marks[n + 1][0] if n + 1 < len(marks) else fn_end
        key = {"isLetter": "letter", "isNumber": "number", "isWhitespace": "space"}[fn]
        for m in re.finditer(r'\{\s*"[^"]*"\s*,\s*(\'(?:\\\\.[^\']*|[^\'\\\\])\')\s*,\s*(true|false)\s*\}', src[pos:stop]):
            cp, _ = parse_go_rune(m.group(1), 0)
            classes.append({"line": src.count("\n", 0, pos + m.start()) + 1, "class": key, "cp": cp,
                            "expected": m.group(2) == "true"})
    return cases, classes


def extract_encode():
    src = open(os.path.join(REF, "llama3/tokenizer_test.go"), encoding="utf-8").read()
    enc, dec = [], []
    for m in re.finditer(r"\binput:\s*", src):
        i = m.end()
        line = src.count("\n", 0, m.start()) + 1
        if src.startswith("[]int{", i):
            ids, j = parse_int_slice(src, i)
            m2 = re.match(r"\s*,\s*expected:\s*", src[j:])
            if not m2:
                continue
            s, _ = parse_string_expr(src, j + m2.end())
            dec.append({"line": line, "input": ids, "expected": hexs(s)})
            continue
        try:
            inp, j = parse_string_expr(src, i)
        except (AssertionError, ValueError):
            continue
        m2 = re.match(r"\s*,\s*expected:\s*", src[j:])
        if not m2 or not src.startswith("[]int{", j + m2.end()):
            continue
        ids, k = parse_int_slice(src, j + m2.end())
        m3 = re.match(r"\s*,\s*opts:\s*([^\n]*)", src[k:])
        opts = m3.group(1) if m3 else "nil"
        if opts.startswith("nil"):
            bos = eos = True
        else:
            bos = "BOS: true" in opts
            eos = "EOS: true" in opts
        enc.append({"line": line, "input": hexs(inp), "bos": bos, "eos": eos, "expected": ids})
    # 373-token paragraph (TestLargeText)
    m = re.search(r"largeText := ", src)
    text, j = parse_go_string(src, m.end())
    n = int(re.search(r"expectedLength := (\d+)", src[j:]).group(1))
    large = {"line": src.count("\n", 0, m.start()) + 1, "input": hexs(text), "expected_len_bos_eos": n}
    return enc, dec, large


def extract_comparison_inputs():
    """the literal input strings of comparison_test.go (its expected values come from a JS tokenizer the reference
    does not ship, so these are inputs only: the GPU path is compared with the oracle on them)"""
    src = open(os.path.join(REF, "llama3", "comparison_test.go"), encoding="utf-8").read()
    i = src.index("testInputs := []string{")
    i = src.index("[]string{", i)
    vals, _ = parse_string_slice(src, i)
    return [hexs(v) for v in vals]


def main():
    os.makedirs(OUT, exist_ok=True)
    cmp_inputs = extract_comparison_inputs()
    json.dump({"source": "llama3/comparison_test.go:84-164 (inputs only)", "inputs": cmp_inputs},
              open(os.path.join(OUT, "ref_comparison_inputs.json"), "w"), indent=0)
    print("comparison inputs:", len(cmp_inputs))
    pre, classes = extract_pretokenize()
    enc, dec, large = extract_encode()
    # documentation KATs (README.md:113-114, llama3/README.md:44)
    enc.append({"line": "README.md:113", "input": hexs(b"Hello, world!"), "bos": True, "eos": True,
                "expected": [128000, 9906, 11, 1917, 0, 128001]})
    enc.append({"line": "llama3/README.md:44", "input": hexs(b"Hello world!"), "bos": True, "eos": True,
                "expected": [128000, 9906, 1917, 0, 128001]})
    json.dump({"source": "llama3/internal/pretokenizer/state_machine_test.go", "cases": pre},
              open(os.path.join(OUT, "ref_pretokenize_kats.json"), "w"), indent=0)
    json.dump({"source": "llama3/internal/pretokenizer/state_machine_test.go", "cases": classes},
              open(os.path.join(OUT, "ref_class_kats.json"), "w"), indent=0)
    json.dump({"source": "llama3/tokenizer_test.go", "encode": enc, "decode": dec, "large": large},
              open(os.path.join(OUT, "ref_encode_kats.json"), "w"), indent=0)
    print("pretokenize KATs:", len(pre), "class KATs:", len(classes), "encode:", len(enc), "decode:", len(dec))


if __name__ == "__main__":
    sys.exit(main())
