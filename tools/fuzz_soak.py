"""Differential soak: random batches that lean on long pre-tokens (runs of letters, ideographs, digits, blanks,
punctuation, invalid bytes, of log-uniform length 1..6,000, glued together with and without separators), GPU through
the C ABI against the oracle.  usage: fuzz_soak.py SECONDS [SEED]   -- prints one line per round, stops at a difference."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests.oracle import Oracle
from tokenizer_amd import capi

rng = None
BIG = int(os.environ.get("SOAK_BIG", "1"))   # > 1: that many times more documents per batch, runs up to 34,000 bytes
KINDS = [int(x) for x in os.environ.get("SOAK_KINDS", "0,1,2,3,4,5,6,7,8,9").split(",")]   # which kinds of runs
SPECIALS = [b"<|begin_of_text|>", b"<|eot_id|>", b"<|reserved_special_token_17|>", b"<|eot_id|", b"<|x|>"]
SEPS = [b"", b"", b" ", b"\n", b"\r\n", b"  ", b"\t", b", ", b".", b"'s", b"'", b"\xe3\x80\x82"]


def run_of(kind, n):
    if kind == 0:
        return bytes(rng.integers(97, 123, n, dtype=np.uint8))                                  # one long word
    if kind == 1:
        return "".join(chr(c) for c in rng.integers(0x4E00, 0x9FA6, max(n // 3, 1))).encode()    # ideographs
    if kind == 2:
        return bytes(rng.integers(48, 58, n, dtype=np.uint8))                                   # ASCII digits
    if kind == 3:
        return bytes([int(rng.choice([32, 32, 32, 9, 10, 13]))]) * n                            # one blank repeated
    if kind == 4:
        return bytes(rng.choice(np.frombuffer(b"!?-=#*/._~", dtype=np.uint8), n))               # punctuation
    if kind == 5:
        return bytes(rng.integers(128, 256, n, dtype=np.uint8))                                 # mostly invalid
    if kind == 6:
        return bytes([int(rng.integers(33, 127))]) * n                                          # one character repeated
    if kind == 7:
        return "".join(chr(c) for c in rng.integers(0x0660, 0x066A, max(n // 2, 1))).encode()    # Arabic-Indic digits
    if kind == 8:
        return (bytes(rng.integers(97, 123, 2, dtype=np.uint8)) * n)[:n]                        # two letters repeated
    w = rng.choice(np.frombuffer(b"abcdefghij \xc3\xa9\xff", dtype=np.uint8), n)
    return bytes(w)                                                                              # letters, blanks, stray bytes


def gen_batch():
    docs = []
    for _ in range(int(rng.integers(1, 60 * BIG))):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            n = int(2 ** rng.uniform(0, 12.55 + (2.5 if BIG > 1 else 0.0)))
            parts.append(run_of(int(KINDS[int(rng.integers(0, len(KINDS)))]), n))
            if rng.random() < 0.05:
                parts.append(SPECIALS[int(rng.integers(0, len(SPECIALS)))])
            parts.append(SEPS[int(rng.integers(0, len(SEPS)))])
        docs.append(b"".join(parts))
    if rng.random() < 0.2:
        docs.append(b"")
    bos, eos = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    return docs, bos, eos


def gen_corpus_batch():
    """SOAK_MODE=corpora: the generators of the benchmark corpora at random sizes (English documents of one random size,
    multilingual documents with a random cap on the run length), cut again at random places."""
    from tools import corpora
    n = int(rng.integers(1, 3000))
    size = int(2 ** rng.uniform(0, 12))
    text, off = corpora.english(n, size, first_doc=int(rng.integers(0, 1 << 40)))
    parts = [text]
    if rng.random() < 0.5:
        m_size = int(2 ** rng.uniform(6, 19))
        t2, _ = corpora.multilingual(int(rng.integers(1, 4)), m_size, max_run=int(2 ** rng.uniform(0, 12)), first_doc=int(rng.integers(0, 1 << 30)))
        parts.append(t2)
    text = np.concatenate(parts)
    cuts = np.unique(np.concatenate([[0, text.size], rng.integers(0, text.size + 1, int(rng.integers(0, 2000)))]))
    docs = [text[int(a):int(b)].tobytes() for a, b in zip(cuts[:-1], cuts[1:])]
    bos, eos = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    return docs, bos, eos


def replay(seed, round_no):
    """The batch of round `round_no` of seed `seed` (no GPU needed)."""
    global rng
    rng = np.random.default_rng(seed)
    for _ in range(round_no):
        batch = gen_batch()
    return batch


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    if os.environ.get("SOAK_VOCAB"):   # a truncated vocabulary (custom data files) on both sides
        from tests.custom_vocab import truncated_vocabulary
        v, m, _ = truncated_vocabulary(int(os.environ["SOAK_VOCAB"]))
        ctx, orc = capi.Context(0, v, m), Oracle(v, m)
    else:
        ctx, orc = capi.Context(0), Oracle()
    t_end = time.time() + budget
    rounds = 0
    total = 0
    while time.time() < t_end:
        docs, bos, eos = gen_corpus_batch() if os.environ.get("SOAK_MODE") == "corpora" else gen_batch()
        text, off = capi.pack_docs(docs)
        got_tok, got_off = ctx.encode_batch(text, off, bos, eos)
        exp_tok, exp_off = orc.encode_batch(text, off, bos, eos, threads=16)
        rounds += 1
        total += text.size
        if not (np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok)):
            go, eo = got_off.astype(np.int64), exp_off.astype(np.int64)
            for i, d in enumerate(docs):
                if got_tok[go[i]:go[i + 1]].tolist() != exp_tok[eo[i]:eo[i + 1]].tolist():
                    print("DIFFERENCE seed %d round %d document %d (%d bytes): replay(%d, %d) regenerates the batch" % (
                        seed, rounds, i, len(d), seed, rounds), flush=True)
                    g = got_tok[go[i]:go[i + 1]].tolist()[(1 if bos else 0):]
                    st = orc.pretokenize(d) + [len(d)]
                    pos = 0
                    for a, b in zip(st[:-1], st[1:]):   # tokens of the special-token split may shift this; good enough
                        ee = orc.encode(d[a:b], False, False)
                        gg = g[pos:pos + len(ee)]
                        if gg != ee:
                            ga = int(off[i]) + a
                            print(" pre-token bytes [%d, %d) len %d, global start %d = tile %d + %d, batch bos=%s" % (
                                a, b, b - a, ga, ga // 3840, ga % 3840, bos))
                            print("  text:", d[a:b][:60], "...", d[a:b][-16:], " before:", d[max(0, a - 40):a])
                            print("  exp", ee[:12], len(ee), " got", gg[:12], flush=True)
                            break
                        pos += len(ee)
                    break
            print("counters of the differing call:", ctx.last_counters(), flush=True)
            for rep in range(3):   # does it persist?
                t2, o2 = ctx.encode_batch(text, off, bos, eos)
                print("  again:", "equal" if np.array_equal(o2, exp_off) and np.array_equal(t2, exp_tok) else "DIFFERENT", ctx.last_counters(), flush=True)
            sys.exit(1)
        if rounds % 20 == 0:
            print("round %d: %.1f MB compared, all equal" % (rounds, total / 1e6), flush=True)
    print("done: %d rounds, %.1f MB, no difference" % (rounds, total / 1e6))
