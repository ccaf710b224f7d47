"""ctypes binding of tokenizer_amd/lib/libl3hostapi.so: C entry points (tests/cpp/host_shim.cpp) over
include/llama3.hpp, the C++ mirror of the reference's Go API.  Test support only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tokenizer_amd", "lib", "libl3hostapi.so")
DATA = os.path.join(ROOT, "tokenizer_amd", "data")

ERR_NONE, ERR_DATA_NOT_FOUND, ERR_INVALID_TOKEN, ERR_TOKEN_NOT_FOUND, ERR_DEVICE, ERR_READER, ERR_CONFIG = range(7)

_P = C.c_void_p
_LL = C.c_longlong
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError("%s is missing: build it with `make -C tokenizer_amd/csrc`" % LIB)
        from tokenizer_amd import capi
        capi.load_library()  # torch's HIP runtime first, then libl3hip.so (see capi.load_library)
        L = C.CDLL(LIB)
        L.l3h_chunks.restype = _LL
        L.l3h_chunks.argtypes = [_P, _LL, _P, C.c_int, C.c_int, C.c_int, _P, _P, _LL]
        L.l3h_new.restype = _P
        L.l3h_new.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        L.l3h_new_special.restype = _P
        L.l3h_new_special.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
        L.l3h_new_devices.restype = _P
        L.l3h_new_devices.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        L.l3h_new_loader.restype = _P
        L.l3h_new_loader.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.l3h_pack_loader.restype = C.c_int
        L.l3h_pack_loader.argtypes = [C.c_char_p, C.c_int, _P, _LL, C.POINTER(_LL), _P, _LL, C.POINTER(_LL)]
        L.l3h_token_stream.restype = _LL
        L.l3h_token_stream.argtypes = [_P, _P, _LL, _P, C.c_int, _LL, _P, _LL]
        L.l3h_free.argtypes = [_P]
        L.l3h_encode.restype = _LL
        L.l3h_encode.argtypes = [_P, C.c_char_p, _LL, C.c_int, C.c_int, C.c_int, _P, _LL]
        L.l3h_append.restype = _LL
        L.l3h_append.argtypes = [_P, _P, _LL, C.c_char_p, _LL, C.c_int, C.c_int, _P, _LL]
        L.l3h_decode.restype = _LL
        L.l3h_decode.argtypes = [_P, _P, _LL, _P, _LL]
        L.l3h_encode_bpe.restype = _LL
        L.l3h_encode_bpe.argtypes = [_P, C.c_char_p, _LL, _P, _LL]
        L.l3h_special.restype = C.c_int
        L.l3h_special.argtypes = [_P, C.c_char_p, C.POINTER(C.c_int)]
        L.l3h_vocab_size.restype = C.c_int
        L.l3h_vocab_size.argtypes = [_P]
        L.l3h_optimistic_count.restype = C.c_int
        L.l3h_optimistic_count.argtypes = [_P, C.c_char_p, _LL]
        L.l3h_pretokenize.restype = _LL
        L.l3h_pretokenize.argtypes = [_P, C.c_char_p, _LL, C.c_int, _P, _LL, _P, _LL]
        L.l3h_scan.restype = _LL
        L.l3h_scan.argtypes = [_P, _P, _LL, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _LL]
        L.l3h_process.restype = _LL
        L.l3h_process.argtypes = [_P, _P, _LL, C.c_int, C.c_int, _P, _LL]
        L.l3h_streams.restype = _LL
        L.l3h_streams.argtypes = [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _LL, _P]
        _lib = L
    return _lib


def _sizes(read_sizes):
    a = np.ascontiguousarray(read_sizes if read_sizes is not None else [], dtype=np.int32)
    return a, (a.ctypes.data if a.size else None), int(a.size)


def pack_loader(data_dir=DATA, rank_step=1):
    """The two data files llama3::PackLoaderData makes from a VocabularyDataLoader over the files in data_dir (no GPU)."""
    L = load()
    cap = 8 << 20
    v, m = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)
    vl, ml = _LL(0), _LL(0)
    rc = L.l3h_pack_loader(data_dir.encode(), rank_step, v.ctypes.data, cap, C.byref(vl), m.ctypes.data, cap, C.byref(ml))
    if rc != 0:
        raise RuntimeError("PackLoaderData failed, error kind %d" % rc)
    return v[:vl.value].tobytes(), m[:ml.value].tobytes()


def chunks(data: bytes, read_sizes=None, buf_size=0, max_buffer=0):
    """[(chunk, scanner knew it was at EOF)] as the scanner would hand them to Encode (same shape as
    tests/scanner_model.chunks); None on a reader error."""
    L = load()
    d = np.frombuffer(data, dtype=np.uint8)
    a, ap, an = _sizes(read_sizes)
    cap = len(data) + 8
    ends = np.zeros(cap, dtype=np.int64)
    lasts = np.zeros(cap, dtype=np.uint8)
    n = L.l3h_chunks(d.ctypes.data if d.size else None, len(data), ap, an, buf_size, max_buffer, ends.ctypes.data,
                     lasts.ctypes.data, cap)
    if n < 0:
        return None
    assert n <= cap
    e = [0] + ends[:n].tolist()
    return [(data[e[i]:e[i + 1]], bool(lasts[i])) for i in range(n)]


class Tokenizer:
    """llama3::Tokenizer (needs a GPU)."""

    def __init__(self, device=0, data_dir=DATA, specials=None, devices=None, loader=None):
        """specials: list of bytes -> New(WithSpecialTokens(list)); devices: list of ints -> New(WithDevices(list));
        loader: (rank_step, fail_merges, null_loader) -> New(WithDataLoader(a loader over the files in data_dir))"""
        self.L = load()
        k = C.c_int(0)
        if loader is not None:
            self.h = self.L.l3h_new_loader(device, data_dir.encode(), loader[0], int(loader[1]), int(loader[2]), C.byref(k))
        elif devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            self.h = self.L.l3h_new_devices(arr, len(devices), data_dir.encode() if data_dir else None, C.byref(k))
        elif specials is None:
            self.h = self.L.l3h_new(device, data_dir.encode() if data_dir else None, C.byref(k))
        else:
            arr = (C.c_char_p * max(len(specials), 1))(*specials)
            self.h = self.L.l3h_new_special(device, data_dir.encode() if data_dir else None, arr, len(specials), C.byref(k))
        self.err_kind = k.value
        if not self.h:
            raise RuntimeError("llama3::New failed, error kind %d" % k.value)

    def close(self):
        if self.h:
            self.L.l3h_free(self.h)
            self.h = None

    def encode(self, text: bytes, opts=None):
        """opts: None (the reference's nil: BOS and EOS) or (bos, eos)"""
        cap = len(text) * 3 + 16
        out = np.empty(cap, dtype=np.int32)
        bos, eos = opts if opts is not None else (0, 0)
        n = self.L.l3h_encode(self.h, text, len(text), 0 if opts is None else 1, int(bos), int(eos), out.ctypes.data, cap)
        assert n <= cap
        return out[:n].tolist()

    def append_tokens(self, dst, text: bytes, opts=(True, True)):
        d = np.ascontiguousarray(dst, dtype=np.int32)
        cap = len(dst) + len(text) * 3 + 16
        out = np.empty(cap, dtype=np.int32)
        n = self.L.l3h_append(self.h, d.ctypes.data if d.size else None, d.size, text, len(text), int(opts[0]), int(opts[1]),
                              out.ctypes.data, cap)
        return out[:n].tolist()

    def encode_bpe(self, pretoken: bytes):
        """BPE.EncodeBPE: pretoken = UTF-8 of a string in the byte-level alphabet"""
        cap = len(pretoken) + 16
        out = np.empty(cap, dtype=np.int32)
        n = self.L.l3h_encode_bpe(self.h, pretoken, len(pretoken), out.ctypes.data, cap)
        return out[:n].tolist()

    def decode(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        cap = a.size * 160 + 16
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.l3h_decode(self.h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap)
        return out[:n].tobytes()

    def special(self, name: bytes):
        """(id, error kind)"""
        k = C.c_int(0)
        v = self.L.l3h_special(self.h, name, C.byref(k))
        return v, k.value

    def vocab_size(self):
        return self.L.l3h_vocab_size(self.h)

    def optimistic_count(self, text: bytes):
        return self.L.l3h_optimistic_count(self.h, text, len(text))

    def pretokenize(self, text: bytes, raw=False):
        cap = len(text) + 1
        ends = np.zeros(cap, dtype=np.int64)
        buf = np.zeros(len(text) * 6 + 8, dtype=np.uint8)  # U+FFFD per bad byte, two bytes per mapped byte
        n = self.L.l3h_pretokenize(self.h, text, len(text), int(raw), buf.ctypes.data, buf.size, ends.ctypes.data, cap)
        e = [0] + ends[:n].tolist()
        b = buf.tobytes()
        return [b[e[i]:e[i + 1]] for i in range(n)]

    def scan(self, data: bytes, read_sizes=None, buf_size=0, max_buffer=0, opts=(False, False)):
        """Every token Scanner yields; None on a reader error."""
        d = np.frombuffer(data, dtype=np.uint8)
        a, ap, an = _sizes(read_sizes)
        cap = len(data) * 3 + 16
        out = np.empty(cap, dtype=np.int32)
        n = self.L.l3h_scan(self.h, d.ctypes.data if d.size else None, len(data), ap, an, buf_size, max_buffer,
                            int(opts[0]), int(opts[1]), out.ctypes.data, cap)
        return None if n < 0 else out[:n].tolist()

    def token_stream(self, data: bytes, read_sizes=None, take=-1):
        """Tokens received from TokenStream's channel (all, or the first `take` before the stream is dropped);
        None on a reader error."""
        d = np.frombuffer(data, dtype=np.uint8)
        a, ap, an = _sizes(read_sizes)
        cap = len(data) * 3 + 16
        out = np.empty(cap, dtype=np.int32)
        n = self.L.l3h_token_stream(self.h, d.ctypes.data if d.size else None, len(data), ap, an, take, out.ctypes.data, cap)
        return None if n < 0 else out[:n].tolist()

    def process(self, data: bytes, opts=(False, False)):
        """(count, bytes written)"""
        d = np.frombuffer(data, dtype=np.uint8)
        cap = len(data) * 12 + 64
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.l3h_process(self.h, d.ctypes.data if d.size else None, len(data), int(opts[0]), int(opts[1]),
                               out.ctypes.data, cap)
        return n, out[:max(n, 0) * 4].tobytes()

    def streams(self, datas, buf_size=0, opts=(False, False)):
        blob = np.frombuffer(b"".join(datas), dtype=np.uint8)
        off = np.zeros(len(datas) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in datas])
        cap = int(blob.size) * 3 + 4 * len(datas) + 16
        out = np.empty(cap, dtype=np.int32)
        ooff = np.zeros(len(datas) + 1, dtype=np.int64)
        n = self.L.l3h_streams(self.h, blob.ctypes.data if blob.size else None, off.ctypes.data, len(datas), buf_size,
                               int(opts[0]), int(opts[1]), out.ctypes.data, cap, ooff.ctypes.data)
        if n < 0:
            return None
        return [out[ooff[i]:ooff[i + 1]].tolist() for i in range(len(datas))]
