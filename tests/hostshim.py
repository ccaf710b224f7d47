"""ctypes binding of tokenizer_amd/lib/libl3hosttest.so: the HOST build of the headers the HIP kernels
share with the host (l3_rules.h, l3_bpe.h) plus the table builder.  Unit-test shim only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tokenizer_amd", "csrc")
LIB = os.path.join(ROOT, "tokenizer_amd", "lib", "libl3hosttest.so")
DATA = os.path.join(ROOT, "tokenizer_amd", "data")


class HostShim:
    def __init__(self, vocab=None, merges=None):
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cpp"))]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            subprocess.check_call(["make", "-C", CSRC, "-s", "hosttest"])
        L = self.lib = C.CDLL(LIB)
        L.l3t_create.restype = C.c_void_p
        L.l3t_create.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.l3t_destroy.argtypes = [C.c_void_p]
        L.l3t_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.l3t_class_of.restype = C.c_int
        L.l3t_class_of.argtypes = [C.c_void_p, C.c_uint32]
        L.l3t_split_document.restype = C.c_int64
        L.l3t_split_document.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        L.l3t_seq_len_word_check.restype = C.c_int64
        L.l3t_class_direct_check.restype = C.c_int64
        L.l3t_class_direct_check.argtypes = [C.c_void_p]
        L.l3t_pair_prio.restype = C.c_int64
        L.l3t_pair_prio.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.l3t_vocab_lookup.restype = C.c_int32
        L.l3t_vocab_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.l3t_bpe.restype = C.c_int64
        L.l3t_bpe.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        L.l3t_bpe_pieces.restype = C.c_int64
        L.l3t_bpe_pieces.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.l3t_bpe_rounds.restype = C.c_int64
        L.l3t_bpe_rounds.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.l3t_token_bytes.restype = C.c_int64
        L.l3t_token_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        if vocab is None or merges is None:
            vocab = open(os.path.join(DATA, "vocab_base64.txt"), "rb").read()
            merges = open(os.path.join(DATA, "merges_binary.txt"), "rb").read()
        self.h = L.l3t_create(vocab, len(vocab), merges, len(merges))
        assert self.h

    def stats(self):
        a = np.zeros(6, dtype=np.int64)
        self.lib.l3t_stats(self.h, a.ctypes.data)
        return dict(zip(["n_tokens", "n_merges", "pair_log2", "vocab_log2", "class_blocks", "blob_bytes"], a.tolist()))

    def class_of(self, cp):
        return self.lib.l3t_class_of(self.h, cp)

    def split_document(self, text: bytes):
        cap = len(text) + 1
        st = np.empty(cap, dtype=np.uint32)
        kd = np.empty(cap, dtype=np.uint8)
        n = self.lib.l3t_split_document(self.h, text, len(text), st.ctypes.data, kd.ctypes.data, cap)
        return st[:n].tolist(), kd[:n].tolist()

    def pair_prio(self, a, b):
        return self.lib.l3t_pair_prio(self.h, a, b)

    def vocab_lookup(self, raw: bytes):
        return self.lib.l3t_vocab_lookup(self.h, raw, len(raw))

    def bpe(self, raw: bytes):
        cap = len(raw) + 1
        out = np.empty(cap, dtype=np.int32)
        n = self.lib.l3t_bpe(self.h, raw, len(raw), out.ctypes.data, cap)
        return out[:n].tolist()

    def bpe_rounds(self, raw: bytes):
        """Round-based merge (the slow kernel's algorithm, emulated sequentially).  Returns (ids, rounds)."""
        cap = len(raw) + 1
        out = np.empty(cap, dtype=np.int32)
        nr = C.c_int64(0)
        n = self.lib.l3t_bpe_rounds(self.h, raw, len(raw), out.ctypes.data, cap, C.byref(nr))
        return out[:n].tolist(), nr.value

    def bpe_pieces(self, raw: bytes):
        """PerformBPE piece by piece, cut where no vocabulary entry can span (the slow kernel's way with long
        pre-tokens).  Returns (ids, pieces)."""
        cap = len(raw) + 1
        out = np.empty(cap, dtype=np.int32)
        npc = C.c_int64(0)
        n = self.lib.l3t_bpe_pieces(self.h, raw, len(raw), out.ctypes.data, cap, C.byref(npc))
        return out[:n].tolist(), npc.value

    def token_bytes(self, tid):
        buf = np.empty(160, dtype=np.uint8)
        n = self.lib.l3t_token_bytes(self.h, tid, buf.ctypes.data, 160)
        return None if n < 0 else buf[:n].tobytes()
