"""ctypes binding of the C ABI in include/l3hip.h (tokenizer_amd/lib/libl3hip.so).

This is plumbing for tests, bench.py and the Python mirror of the reference API: every call goes
straight to the HIP library.  There is no CPU fallback; a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("L3HIP_LIB") or os.path.join(ROOT, "tokenizer_amd", "lib", "libl3hip.so")
DATA_DIR = os.path.join(ROOT, "tokenizer_amd", "data")

L3_OK = 0
L3_E_INVALID_ARG = -1
L3_E_DATA = -2
L3_E_HIP = -3
L3_E_OUTPUT_CAP = -4
L3_E_NO_DEVICE = -5
L3_E_NOT_FOUND = -6
L3_E_SCRATCH = -7
L3_NOALLOC_MAX_BYTES = 4 << 20
L3_NOALLOC_MAX_DOCS = 65536
L3_BOS = 1
L3_EOS = 2
L3_NO_SPECIAL = 4
L3_ONE_PRETOKEN = 8

# every symbol include/l3hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("l3_create", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    ("l3_create_ex", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.POINTER(_P)]),
    ("l3_destroy", None, [_P]),
    ("l3_multi_create", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_int), C.c_int,
                                  C.POINTER(_P)]),
    ("l3_multi_destroy", None, [_P]),
    ("l3_multi_size", C.c_int, [_P]),
    ("l3_multi_context", _P, [_P, C.c_int]),
    ("l3_multi_last_error", C.c_char_p, [_P]),
    ("l3_multi_encode_batch", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    ("l3_partition_by_bytes", C.c_int, [_P, C.c_uint64, C.c_int, _P]),
    ("l3_last_error", C.c_char_p, [_P]),
    ("l3_encode_batch", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    ("l3_encode_batch_device", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P,
                                        C.POINTER(C.c_uint64), _P]),
    ("l3_encode_submit", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, _P, C.POINTER(C.c_int)]),
    ("l3_encode_wait", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    ("l3_encode_batch_device_noalloc", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P,
                                                C.POINTER(C.c_uint64), _P]),
    ("l3_pretokenize_batch", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, _P]),
    ("l3_host_alloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("l3_host_free", C.c_int, [_P, _P]),
    ("l3_decode_batch", C.c_int, [_P, _P, _P, C.c_uint64, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    ("l3_special_token_id", C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]),
    ("l3_vocab_size", C.c_int, [_P]),
    ("l3_token_bytes", C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    ("l3_set_profiling", C.c_int, [_P, C.c_int]),
    ("l3_last_timings", C.c_int, [_P, _P, C.c_int]),
    ("l3_timing_totals", C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_uint64)]),
    ("l3_last_counters", C.c_int, [_P, _P, C.c_int]),
]


class L3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("l3hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """Load libl3hip.so and bind every declared symbol (no GPU call is made)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C tokenizer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        try:
            # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the
            # GPU, so let torch's load first and libl3hip.so binds to that one (same soname).
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def read_data_files(data_dir=DATA_DIR):
    with open(os.path.join(data_dir, "vocab_base64.txt"), "rb") as f:
        vocab = f.read()
    with open(os.path.join(data_dir, "merges_binary.txt"), "rb") as f:
        merges = f.read()
    return vocab, merges


def pack_docs(docs):
    """list of bytes -> (uint8 text, uint64 offsets)"""
    off = np.zeros(len(docs) + 1, dtype=np.uint64)
    if docs:
        off[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    text = np.frombuffer(b"".join(docs), dtype=np.uint8)
    return text, off


class Context:
    """One l3_ctx: the tables of one tokenizer on one GPU."""

    def __init__(self, device=0, vocab=None, merges=None, specials=None):
        """specials: list of bytes -- the special-token list of WithSpecialTokens (l3_create_ex); None: the defaults"""
        self.lib = load_library()
        if vocab is None or merges is None:
            vocab, merges = read_data_files()
        h = _P()
        if specials is None:
            rc = self.lib.l3_create(vocab, len(vocab), merges, len(merges), device, C.byref(h))
        else:
            arr = (C.c_char_p * max(len(specials), 1))(*specials)
            rc = self.lib.l3_create_ex(vocab, len(vocab), merges, len(merges), arr, len(specials), device, C.byref(h))
        if rc != L3_OK:
            raise L3Error(rc, (self.lib.l3_last_error(None) or b"").decode("utf-8", "replace"))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.l3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        return L3Error(rc, (self.lib.l3_last_error(self.h) or b"").decode("utf-8", "replace"))

    # ---- host-buffer API -------------------------------------------------------------------------
    def encode_batch(self, text, doc_off, bos=True, eos=True, out_cap=None, one_pretoken=False):
        """text: uint8 array / bytes; doc_off: uint64 (n+1).  Returns (int32 tokens, uint64 offsets).
        one_pretoken: L3_ONE_PRETOKEN -- every document is one pre-token of raw bytes (BPE.EncodeBPE)."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text,
                                    dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        n_docs = doc_off.size - 1
        flags = (L3_BOS if bos else 0) | (L3_EOS if eos else 0) | (L3_ONE_PRETOKEN if one_pretoken else 0)
        cap = int(out_cap) if out_cap is not None else int(text.size) + 2 * n_docs + 64  # enough unless bytes are invalid UTF-8
        while True:
            out = np.empty(max(cap, 1), dtype=np.int32)
            off = np.empty(n_docs + 1, dtype=np.uint64)
            n = C.c_uint64(0)
            rc = self.lib.l3_encode_batch(self.h, text.ctypes.data, doc_off.ctypes.data, n_docs, flags,
                                          out.ctypes.data, cap, off.ctypes.data, C.byref(n))
            if rc == L3_E_OUTPUT_CAP and out_cap is None:
                cap = int(n.value)
                continue
            if rc != L3_OK:
                raise self._err(rc)
            return out[:n.value], off

    def encode_bpe(self, pretokens):
        """BPE.EncodeBPE for a list of pre-tokens given as RAW bytes: one list of ids each."""
        text, off = pack_docs(pretokens)
        tok, o = self.encode_batch(text, off, False, False, one_pretoken=True)
        o = o.astype(np.int64)
        return [tok[o[i]:o[i + 1]].tolist() for i in range(len(pretokens))]

    def encode_docs(self, docs, bos=True, eos=True):
        text, off = pack_docs(docs)
        tok, o = self.encode_batch(text, off, bos, eos)
        o = o.astype(np.int64)
        return [tok[o[i]:o[i + 1]].tolist() for i in range(len(docs))]

    def pretokenize_batch(self, text, doc_off, specials=True):
        """kinds per byte: 1 pre-token start, 2 special-token start.  specials=False: the bare pre-tokenizer."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        kinds = np.zeros(max(int(text.size), 1), dtype=np.uint8)
        rc = self.lib.l3_pretokenize_batch(self.h, text.ctypes.data, doc_off.ctypes.data, doc_off.size - 1,
                                           0 if specials else L3_NO_SPECIAL, kinds.ctypes.data)
        if rc != L3_OK:
            raise self._err(rc)
        return kinds[:text.size]

    def decode_batch(self, ids, id_off):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        id_off = np.ascontiguousarray(id_off, dtype=np.uint64)
        n_seqs = id_off.size - 1
        cap = int(ids.size) * 8 + 64
        while True:
            out = np.empty(cap, dtype=np.uint8)
            off = np.empty(n_seqs + 1, dtype=np.uint64)
            n = C.c_uint64(0)
            rc = self.lib.l3_decode_batch(self.h, ids.ctypes.data, id_off.ctypes.data, n_seqs, out.ctypes.data, cap,
                                          off.ctypes.data, C.byref(n))
            if rc == L3_E_OUTPUT_CAP:
                cap = int(n.value)
                continue
            if rc != L3_OK:
                raise self._err(rc)
            return out[:n.value], off

    # ---- device-pointer API (torch tensors are only a way to hold HBM here) --------------------------
    def encode_batch_device(self, d_text_ptr, n_bytes, d_off_ptr, n_docs, flags, d_out_ptr, out_cap, d_out_off_ptr,
                            stream=None, noalloc=False):
        """Raw device pointers (ints).  Returns (rc, n_tok).  noalloc: the entry point whose workspace is the
        library's static device array (l3_encode_batch_device_noalloc)."""
        n = C.c_uint64(0)
        fn = self.lib.l3_encode_batch_device_noalloc if noalloc else self.lib.l3_encode_batch_device
        rc = fn(self.h, d_text_ptr, n_bytes, d_off_ptr, n_docs, flags, d_out_ptr, out_cap, d_out_off_ptr, C.byref(n), stream)
        return rc, int(n.value)

    def encode_submit(self, d_text_ptr, n_bytes, d_off_ptr, n_docs, flags, d_out_ptr, out_cap, d_out_off_ptr, stream=None):
        """Stream-ordered first half (l3_encode_submit).  Returns (rc, ticket)."""
        t = C.c_int(-1)
        rc = self.lib.l3_encode_submit(self.h, d_text_ptr, n_bytes, d_off_ptr, n_docs, flags, d_out_ptr, out_cap, d_out_off_ptr,
                                       stream, C.byref(t))
        return rc, t.value

    def encode_wait(self, ticket):
        """Second half (l3_encode_wait).  Returns (rc, n_tok)."""
        n = C.c_uint64(0)
        rc = self.lib.l3_encode_wait(self.h, ticket, C.byref(n))
        return rc, int(n.value)

    # ---- small queries ------------------------------------------------------------------------------
    def host_alloc(self, nbytes, dtype=np.uint8):
        """Page-locked host buffer from the library as a numpy array (free it with host_free(array))."""
        p = _P()
        rc = self.lib.l3_host_alloc(self.h, int(nbytes), C.byref(p))
        if rc != L3_OK:
            raise self._err(rc)
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype)
        self._host_allocs = getattr(self, "_host_allocs", {})
        self._host_allocs[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        rc = self.lib.l3_host_free(self.h, C.c_void_p(self._host_allocs.pop(a.ctypes.data)))
        if rc != L3_OK:
            raise self._err(rc)

    def special_token_id(self, name: bytes):
        v = C.c_int32(0)
        rc = self.lib.l3_special_token_id(self.h, name, len(name), C.byref(v))
        if rc != L3_OK:
            raise self._err(rc)
        return v.value

    def vocab_size(self):
        return self.lib.l3_vocab_size(self.h)

    def token_bytes(self, tid):
        buf = np.empty(160, dtype=np.uint8)
        n = self.lib.l3_token_bytes(self.h, tid, buf.ctypes.data, 160)
        return None if n < 0 else buf[:n].tobytes()

    def set_profiling(self, on=True):
        """True / 1: events between all kernels; 2: only around the tile kernel and first start / last end; False: none"""
        self.lib.l3_set_profiling(self.h, int(on))

    def last_timings(self):
        a = np.zeros(9, dtype=np.float32)
        self.lib.l3_last_timings(self.h, a.ctypes.data, 9)
        return dict(zip(["docs", "tile", "merge", "slow", "scan", "host_passes", "emit", "total", "decode"], a.tolist()))

    def timing_totals(self):
        """(sums of the eight timings over the profiled calls since set_profiling, number of those calls)"""
        a = np.zeros(8, dtype=np.float64)
        n = C.c_uint64(0)
        self.lib.l3_timing_totals(self.h, a.ctypes.data, 8, C.byref(n))
        return dict(zip(["docs", "tile", "merge", "slow", "scan", "host_passes", "emit", "total"], a.tolist())), int(n.value)

    def last_counters(self):
        a = np.zeros(8, dtype=np.uint64)
        self.lib.l3_last_counters(self.h, a.ctypes.data, 8)
        return dict(zip(["tiles", "slow_pretokens", "deferred_tokens", "scratch_bytes", "retries", "short_pretokens", "window_items",
                         "giant_symbols"], a.tolist()))


def partition_by_bytes(doc_off, world):
    """l3_partition_by_bytes: world + 1 document indices (host arithmetic only, no GPU needed)."""
    lib = load_library()
    doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
    cuts = np.zeros(world + 1, dtype=np.uint64)
    rc = lib.l3_partition_by_bytes(doc_off.ctypes.data, doc_off.size - 1, world, cuts.ctypes.data)
    if rc != L3_OK:
        raise L3Error(rc, "l3_partition_by_bytes")
    return cuts.astype(np.int64).tolist()


class MultiContext:
    """l3_multi: one context per entry of `devices`, batches sharded by document over them."""

    def __init__(self, devices, vocab=None, merges=None, specials=None):
        self.lib = load_library()
        if vocab is None or merges is None:
            vocab, merges = read_data_files()
        h = _P()
        dev = (C.c_int * len(devices))(*devices)
        arr = None if specials is None else (C.c_char_p * max(len(specials), 1))(*specials)
        rc = self.lib.l3_multi_create(vocab, len(vocab), merges, len(merges), arr, 0 if specials is None else len(specials),
                                      dev, len(devices), C.byref(h))
        if rc != L3_OK:
            raise L3Error(rc, (self.lib.l3_multi_last_error(None) or b"").decode("utf-8", "replace"))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.l3_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self.lib.l3_multi_size(self.h)

    def last_counters(self, i):
        out = (C.c_uint64 * 8)()
        self.lib.l3_last_counters(self.lib.l3_multi_context(self.h, i), out, 8)
        return list(out)

    def encode_batch(self, text, doc_off, bos=True, eos=True, out_cap=None):
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text,
                                    dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        n_docs = doc_off.size - 1
        flags = (L3_BOS if bos else 0) | (L3_EOS if eos else 0)
        cap = int(out_cap) if out_cap is not None else int(text.size) + 2 * n_docs + 64
        while True:
            out = np.empty(max(cap, 1), dtype=np.int32)
            off = np.empty(n_docs + 1, dtype=np.uint64)
            n = C.c_uint64(0)
            rc = self.lib.l3_multi_encode_batch(self.h, text.ctypes.data, doc_off.ctypes.data, n_docs, flags, out.ctypes.data, cap,
                                                off.ctypes.data, C.byref(n))
            if rc == L3_E_OUTPUT_CAP and out_cap is None:
                cap = int(n.value)
                continue
            if rc != L3_OK:
                raise L3Error(rc, (self.lib.l3_multi_last_error(self.h) or b"").decode("utf-8", "replace"))
            return out[:n.value], off
