"""Seeded synthetic corpora for bench.py and the parity tests.

The heavy generators are in tools/l3_corpus.c (plain C, built to tokenizer_amd/lib/libl3corpus.so so
that the binary travels to the GPU box with the tree).  Measurement / test support only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "l3_corpus.c")
LIB = os.path.join(ROOT, "tokenizer_amd", "lib", "libl3corpus.so")
DATA = os.path.join(ROOT, "tokenizer_amd", "data")

DEV_SRC = os.path.join(ROOT, "tools", "l3_corpus_dev.hip")
DEV_LIB = os.path.join(ROOT, "tokenizer_amd", "lib", "libl3corpusdev.so")

_lib = None
_dev_lib = None
_words = None
_dev_words = {}


def build():
    """Builds the host generators and (where hipcc exists: in the build container, not on the GPU box, which gets the
    built file with the tree) the device generator."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not (os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", LIB, SRC])
    if not (os.path.exists(DEV_LIB) and os.path.getmtime(DEV_LIB) >= os.path.getmtime(DEV_SRC)):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-o", DEV_LIB, DEV_SRC])
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.l3c_gen_english.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_void_p]
        _lib.l3c_gen_multilingual.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    return _lib


def word_list():
    """The fixed 4,096-word list tools/words_4096.txt (frequency-ranked lower-case English words taken
    once from the licence texts and the Python language reference that ship in this image), with
    Zipf(s = 1.0) weights by rank."""
    global _words
    if _words is None:
        words = [w.encode() for w in open(os.path.join(ROOT, "tools", "words_4096.txt")).read().split()]
        assert len(words) == 4096
        blob = np.frombuffer(b"".join(words), dtype=np.uint8).copy()
        off = np.zeros(len(words) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(w) for w in words])
        wgt = 1.0 / np.arange(1, len(words) + 1, dtype=np.float64)  # Zipf s = 1.0
        cdf = np.cumsum(wgt) / wgt.sum()
        thr = np.minimum(np.floor(cdf * 18446744073709551615.0), 18446744073709551615.0).astype(np.uint64)
        thr[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
        _words = (blob, off, thr, len(words))
    return _words


def english(n_docs, doc_bytes=1024, first_doc=0):
    """C2: n_docs documents of exactly doc_bytes bytes.  Returns (uint8 text, uint64 offsets)."""
    blob, off, thr, n = word_list()
    out = np.empty(n_docs * doc_bytes, dtype=np.uint8)
    _load().l3c_gen_english(first_doc, n_docs, doc_bytes, blob.ctypes.data, off.ctypes.data, thr.ctypes.data, n,
                            out.ctypes.data)
    return out, (np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(doc_bytes))


def english_device(d_out, first_doc, n_docs, doc_bytes=1024, stream=None):
    """C2 / C5 documents first_doc .. first_doc + n_docs generated ON THE GPU into the torch uint8 CUDA tensor d_out
    (n_docs * doc_bytes bytes): the same bytes english() returns for the same documents, without a host buffer."""
    global _dev_lib
    import torch
    if _dev_lib is None:
        if not os.path.exists(DEV_LIB):
            build()
        _dev_lib = C.CDLL(DEV_LIB)
        _dev_lib.l3c_gen_english_device.restype = C.c_int
        _dev_lib.l3c_gen_english_device.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_uint32, C.c_void_p, C.c_void_p]
    dev = d_out.device.index
    if dev not in _dev_words:
        blob, off, thr, n = word_list()
        _dev_words[dev] = (torch.from_numpy(blob).to(d_out.device), torch.from_numpy(off.astype(np.int32)).to(d_out.device),
                           torch.from_numpy(thr.view(np.int64)).to(d_out.device), n)
    w, wo, cdf, n = _dev_words[dev]
    assert d_out.dtype == torch.uint8 and d_out.is_contiguous() and d_out.numel() >= n_docs * doc_bytes
    rc = _dev_lib.l3c_gen_english_device(first_doc, n_docs, doc_bytes, w.data_ptr(), wo.data_ptr(), cdf.data_ptr(), n,
                                         d_out.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError("l3c_gen_english_device failed: HIP error %d" % rc)


def multilingual(n_docs, doc_bytes=4 << 20, max_run=4096, first_doc=0):
    """C3: CJK / Cyrillic / emoji documents of exactly doc_bytes bytes."""
    out = np.empty(n_docs * doc_bytes, dtype=np.uint8)
    lib = _load()
    for d in range(n_docs):
        lib.l3c_gen_multilingual(first_doc + d, doc_bytes, max_run, out[d * doc_bytes:].ctypes.data)
    return out, (np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(doc_bytes))


SPECIAL_NAMES = (["begin_of_text", "end_of_text", "start_header_id", "end_header_id", "eot_id", "eom_id", "python_tag",
                  "finetune_right_pad_id"] + ["reserved_special_token_%d" % i for i in range(248)])


def pathological(size=1 << 20, seed=4):
    """C4: list of documents (bytes), each about `size` bytes."""
    rng = np.random.default_rng(seed)
    eng, _ = english(max(size // 1024, 1), 1024, first_doc=1 << 40)
    eng = eng.tobytes()
    with_bot = b"".join(eng[i:i + 64] + b"<|begin_of_text|>" for i in range(0, min(len(eng), size), 64))
    all_specials = b" ".join(("<|%s|>" % n).encode() for n in SPECIAL_NAMES)
    near = b"x<|reserved_special_token_248|>y<|reserved_special_token_07|>z<|eot_id|w<|<|eot_id|>v<|eot_id |>" * 50
    uni = ("a b　c d x² ½ Ⅷ ① ٣٤٥٦ \U0001d7d7\U0001d7d8 ").encode() * (size // 64 + 1)
    docs = [
        b" " * size, b"\t" * size, b"\n \n " * (size // 4), bytes(rng.integers(48, 58, size, dtype=np.uint8)),
        b"a" * size, b"'s" * (size // 2), rng.integers(0, 256, size, dtype=np.uint8).tobytes(),
        b"\xc0\x80\xed\xa0\x80\xf4\x90\x80\x80\xe4\xb8\xf0\x9f\x80" * (size // 14),
        with_bot, all_specials, near, uni[:size], b"\xff" * (size // 4), b"!" * (size // 2) + b"\n" * (size // 8),
    ]
    return docs
