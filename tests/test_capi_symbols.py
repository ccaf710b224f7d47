"""CPU checks of the drop-in boundary: the library loads and exports every symbol include/l3hip.h declares
(no compute call is made without a GPU)."""
import os
import re

from tokenizer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "l3hip.h")).read()
    declared = set(re.findall(r"\b(l3_[a-z_0-9]+)\s*\(", hdr))
    bound = {name for name, _, _ in capi.SYMBOLS}
    assert declared == bound, declared ^ bound


def test_library_exports_every_symbol():
    lib = capi.load_library()
    for name, _, _ in capi.SYMBOLS:
        assert getattr(lib, name) is not None


def test_create_rejects_bad_data_without_gpu():
    import ctypes as C
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.l3_create(b"!!!not base64!!!", 16, b"AAAA", 4, 0, C.byref(h))
    assert rc == capi.L3_E_DATA and not h.value
    assert b"vocabulary" in lib.l3_last_error(None)
    vocab, merges = capi.read_data_files()
    import base64

    import numpy as np

    def pack(pairs):   # 17 bits per id, most significant bit first (decoder.go:75-126)
        ids = np.asarray(pairs, dtype=np.int64).reshape(-1)
        bits = ((ids[:, None] >> np.arange(16, -1, -1)) & 1).astype(np.uint8).reshape(-1)
        return base64.b64encode(np.packbits(bits).tobytes())

    # rules whose products are not in ascending order ('in' = 258 before the rule of token 256): valid for the reference (a
    # rule's rank is its index, decoder.go:41-71) and served in rank mode here, so the call gets as far as looking for a GPU
    unordered = pack([(72, 77), (220, 220)])
    rc = lib.l3_create(vocab, len(vocab), unordered, len(unordered), 0, C.byref(h))
    assert rc in (capi.L3_E_NO_DEVICE, capi.L3_OK)
    if rc == capi.L3_OK:
        lib.l3_destroy(h)
    # pairs naming ids beyond the token list are skipped as the reference does (decoder.go:61-63): valid data, so the
    # call gets as far as looking for a GPU
    skipped = base64.b64encode(b"\xff" * 34)
    rc = lib.l3_create(vocab, len(vocab), skipped, len(skipped), 0, C.byref(h))
    assert rc in (capi.L3_E_NO_DEVICE, capi.L3_OK)
    if rc == capi.L3_OK:
        lib.l3_destroy(h)
