"""Custom vocabulary data files for the tests of the WithDataFiles path (options.go, vocab.go:46-118): a truncated
Llama-3 vocabulary in the reference's two file formats, derived from the shipped files.

  vocabulary file: base64 of the '\\n'-separated token strings in id order (decoder.go:16-36)
  merges file:     base64 of the pairs (idA, idB) in rank order, 17 bits per id, most significant bit first
                   (decoder.go:41-126)

Keeping the first n tokens and every merge whose two operands and product are among them gives a smaller, different
tokenizer: other token sequences, 256 special tokens at ids n .. n+255 (tokenizer.go:194-198), so BOS / EOS are n
and n+1 (GetSpecialTokenID is a lookup: tokenizer.go:226,259,357-368).
"""
import base64
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tokenizer_amd", "data")


def truncated_vocabulary(n_tokens, full_merges=False, shuffle_seed=None, duplicates=0):
    """Returns (vocab_file_bytes, merges_file_bytes, n_merges_kept).  full_merges: the shipped merges file unchanged (the
    reference loads that too: pairs naming unknown ids are skipped, decoder.go:61-63, and a rule whose product is not in the
    vocabulary produces nothing, bpe.go:157-161).  shuffle_seed: the kept rules in a random order -- for the reference a rule's
    rank is its index in the file and nothing else (decoder.go:41-71), so this is valid data for a different tokenizer, one
    whose pairs are NOT grouped by ascending product id.  duplicates: that many rules are repeated further down the list (the
    later rank counts: the map entry is overwritten)."""
    vraw = base64.b64decode(open(os.path.join(DATA, "vocab_base64.txt"), "rb").read())
    mraw = base64.b64decode(open(os.path.join(DATA, "merges_binary.txt"), "rb").read())
    toks = [t for t in vraw.split(b"\n") if t]
    assert 256 <= n_tokens <= len(toks)
    bits = np.unpackbits(np.frombuffer(mraw, dtype=np.uint8))                      # most significant bit first
    n_ids = bits.size // 17
    w = (1 << np.arange(16, -1, -1)).astype(np.int64)
    ids = bits[:n_ids * 17].reshape(n_ids, 17).astype(np.int64) @ w
    if full_merges:
        return (base64.b64encode(b"\n".join(toks[:n_tokens]) + b"\n"), open(os.path.join(DATA, "merges_binary.txt"), "rb").read(),
                n_ids // 2)
    index = {t: i for i, t in enumerate(toks)}
    kept = []
    for k in range(0, n_ids - 1, 2):
        a, b = int(ids[k]), int(ids[k + 1])
        if a < n_tokens and b < n_tokens and index[toks[a] + toks[b]] < n_tokens:
            kept += [a, b]
    if shuffle_seed is not None:
        rng = np.random.default_rng(shuffle_seed)
        pairs = np.asarray(kept, dtype=np.int64).reshape(-1, 2)
        pairs = pairs[rng.permutation(pairs.shape[0])]
        if duplicates:
            again = pairs[rng.integers(0, pairs.shape[0] // 2, duplicates)]
            pos = np.sort(rng.integers(pairs.shape[0] // 2, pairs.shape[0], duplicates))
            pairs = np.insert(pairs, pos, again, axis=0)
        kept = pairs.reshape(-1).tolist()
    kb = ((np.asarray(kept, dtype=np.int64)[:, None] >> np.arange(16, -1, -1)) & 1).astype(np.uint8).reshape(-1)
    packed = np.packbits(kb).tobytes()                                             # zero bits pad the last byte
    return base64.b64encode(b"\n".join(toks[:n_tokens]) + b"\n"), base64.b64encode(packed), len(kept) // 2
