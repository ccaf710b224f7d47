#!/usr/bin/env python3
"""Generate the Unicode 15.0.0 letter / decimal-digit / white-space range tables.

The reference classifies runes with Go's ``unicode.IsLetter`` / ``unicode.IsDigit`` /
``unicode.IsSpace`` (/root/reference/llama3/internal/pretokenizer/state_machine.go:310-320),
which for Go 1.21+ (go.mod pins 1.24.5) are the Unicode 15.0.0 tables.  No Unicode 15.0.0
database exists offline, so the tables are derived:

  * node's ICU (``process.versions.unicode`` == 14.0) gives the exact 14.0 sets for
    General_Category L*, Nd and the White_Space property;
  * the code points first assigned in 15.0 (list below, from DerivedAge 15.0) are classified
    with the ``regex`` module (Unicode 17 data; a character's category is stable once assigned:
    the script checks that no 14.0-assigned code point changed L/Nd membership 14 -> 17);
  * everything first assigned in 15.1 / 16.0 / 17.0 stays "other".

Residual risk (recorded in DESIGN.md): membership of code points assigned in 14.0/15.0 is
not pinned by any reference test ("parity unpinned" for those code points).

Outputs identical content to
  tokenizer_amd/csrc/unicode15_ranges.h   (product: table builder input)
  oracle/unicode15_ranges.h               (oracle: binary-searched directly)
and tests/golden/unicode_additions_14_15.json: the letters and decimal digits first assigned in 14.0 (node's 14.0 sets
against python's unicodedata, which is 13.0 in this image) and in 15.0 (the list below), as ranges -- what
tests/test_unicode_tables.py expects the oracle's classes to differ from unicodedata's by, exactly.

  gen_unicode_tables.py          writes the three files
  gen_unicode_tables.py --check  regenerates and compares with the committed files; exit status 1 on any difference
"""
import json
import os
import subprocess
import sys
import tempfile

import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NODE_JS = r"""
function ranges(re){ const out=[]; let s=-1;
  for(let cp=0; cp<=0x10FFFF; cp++){
    let m=(cp>=0xD800&&cp<=0xDFFF)?false:re.test(String.fromCodePoint(cp));
    if(m&&s<0)s=cp; if(!m&&s>=0){out.push([s,cp-1]);s=-1;} }
  if(s>=0)out.push([s,0x10FFFF]); return out; }
console.log(JSON.stringify({unicode:process.versions.unicode,
  L:ranges(/^\p{L}$/u), Nd:ranges(/^\p{Nd}$/u), Cn:ranges(/^\p{Cn}$/u),
  WS:ranges(/^\p{White_Space}$/u)}));
"""

# Code points first assigned in Unicode 15.0 that can be letters or decimal digits
# (DerivedAge.txt "; 15.0" lines; symbol/emoji/mark-only ranges are irrelevant here but the
# block ranges are listed whole and filtered by category below).
AGE_15_0 = [
    (0x0CF3, 0x0CF3), (0x0ECE, 0x0ECE), (0x10EFD, 0x10EFF), (0x1123F, 0x11241),
    (0x11B00, 0x11B09), (0x11F00, 0x11F10), (0x11F12, 0x11F3A), (0x11F3E, 0x11F59),
    (0x1342F, 0x1342F), (0x13439, 0x13455), (0x1B132, 0x1B132), (0x1B155, 0x1B155),
    (0x1D2C0, 0x1D2D3), (0x1DF25, 0x1DF2A), (0x1E030, 0x1E06D), (0x1E08F, 0x1E08F),
    (0x1E4D0, 0x1E4F9), (0x2B739, 0x2B739), (0x31350, 0x323AF),
]


def to_set(rs):
    s = set()
    for a, b in rs:
        s.update(range(a, b + 1))
    return s


def to_ranges(s):
    out = []
    for c in sorted(s):
        if out and out[-1][1] == c - 1:
            out[-1][1] = c
        else:
            out.append([c, c])
    return out


def main():
    check = "--check" in sys.argv[1:]
    with tempfile.NamedTemporaryFile("w", suffix=".js", delete=False) as f:
        f.write(NODE_JS)
        js = f.name
    d = json.loads(subprocess.check_output(["node", js]))
    os.unlink(js)
    assert d["unicode"] == "14.0", d["unicode"]
    L14, Nd14, Cn14 = to_set(d["L"]), to_set(d["Nd"]), to_set(d["Cn"])
    ws = to_set(d["WS"])

    re_l, re_nd = regex.compile(r"\p{L}"), regex.compile(r"\p{Nd}")
    assigned14 = set(range(0x110000)) - Cn14 - set(range(0xD800, 0xE000))
    # stability check on everything assigned in 14.0
    for cp in assigned14:
        c = chr(cp)
        assert bool(re_l.match(c)) == (cp in L14), hex(cp)
        assert bool(re_nd.match(c)) == (cp in Nd14), hex(cp)

    L15, Nd15 = set(L14), set(Nd14)
    for a, b in AGE_15_0:
        for cp in range(a, b + 1):
            assert cp in Cn14, "claimed 15.0 addition already assigned in 14.0: %x" % cp
            c = chr(cp)
            if re_l.match(c):
                L15.add(cp)
            if re_nd.match(c):
                Nd15.add(cp)

    # Go's unicode.IsSpace set (White_Space property; version-stable since 6.3)
    expect_ws = {9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F,
                 0x3000} | set(range(0x2000, 0x200B))
    assert ws == expect_ws

    def emit(name, rs):
        lines = ["static const unsigned int %s[][2] = {" % name]
        row = []
        for a, b in rs:
            row.append("{0x%X,0x%X}" % (a, b))
            if len(row) == 6:
                lines.append("  " + ",".join(row) + ",")
                row = []
        if row:
            lines.append("  " + ",".join(row) + ",")
        lines.append("};")
        lines.append("static const unsigned int %s_COUNT = %d;" % (name, len(rs)))
        return "\n".join(lines)

    body = "\n".join([
        "/* GENERATED by tools/gen_unicode_tables.py -- do not edit.",
        " * Unicode 15.0.0 sets as used by Go 1.21-1.24 unicode.IsLetter / IsDigit / IsSpace",
        " * (reference: llama3/internal/pretokenizer/state_machine.go:310-320).",
        " * Derivation: ICU 14.0 (node) + code points aged 15.0 classified by the regex module.",
        " * L: %d code points, Nd: %d, White_Space: %d. */" % (len(L15), len(Nd15), len(ws)),
        "#ifndef L3_UNICODE15_RANGES_H",
        "#define L3_UNICODE15_RANGES_H",
        emit("L3_UNI_LETTER", to_ranges(L15)),
        emit("L3_UNI_DIGIT", to_ranges(Nd15)),
        emit("L3_UNI_SPACE", to_ranges(ws)),
        "#endif",
        "",
    ])
    import unicodedata
    assert unicodedata.unidata_version == "13.0.0", unicodedata.unidata_version
    cn13 = {cp for cp in range(0x110000) if unicodedata.category(chr(cp)) == "Cn"}
    new14 = (cn13 - Cn14)                      # first assigned in 14.0
    add = {
        "comment": "letters (L*) and decimal digits (Nd) first assigned in Unicode 14.0 and 15.0, as [first, last] ranges; "
                   "written by tools/gen_unicode_tables.py",
        "unicodedata": unicodedata.unidata_version, "node_icu": d["unicode"],
        "assigned_14_0": len(new14),
        "L_14_0": to_ranges(new14 & L14), "Nd_14_0": to_ranges(new14 & Nd14),
        "L_15_0": to_ranges(L15 - L14), "Nd_15_0": to_ranges(Nd15 - Nd14),
    }
    outputs = [(rel, body) for rel in ("tokenizer_amd/csrc/unicode15_ranges.h", "oracle/unicode15_ranges.h")]
    outputs.append(("tests/golden/unicode_additions_14_15.json", json.dumps(add, indent=1) + "\n"))
    if check:
        bad = 0
        for rel, text in outputs:
            p = os.path.join(ROOT, rel)
            same = os.path.exists(p) and open(p).read() == text
            print("%-45s %s" % (rel, "identical" if same else "DIFFERS from what the generator writes"))
            bad += not same
        return 1 if bad else 0
    for rel, text in outputs:
        p = os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)
        print("wrote", rel, "L=%d Nd=%d" % (len(L15), len(Nd15)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
