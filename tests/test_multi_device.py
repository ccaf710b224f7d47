"""Several GPUs behind one handle (include/l3hip.h: l3_multi_*): SURVEY 8(e) inside the library, the C twin of
tokenizer_amd/shard.py.  On the one-GPU test box the handle holds several contexts on device 0: sharding, the meeting
point of the shard threads and the placement of every shard's ids are exercised exactly as with N devices."""
import numpy as np
import pytest

from tokenizer_amd import capi, shard
from tools import corpora


def test_partition_is_shard_pys():
    rng = np.random.default_rng(3)
    for n_docs in (0, 1, 2, 7, 1000):
        for world in (1, 2, 3, 8):
            lens = rng.integers(0, 5000, size=n_docs)
            if n_docs > 3:
                lens[rng.integers(0, n_docs, size=n_docs // 3)] = 0   # runs of empty documents
            off = np.zeros(n_docs + 1, dtype=np.uint64)
            off[1:] = np.cumsum(lens)
            got = capi.partition_by_bytes(off, world)
            assert got == shard.partition_by_bytes(off, world)
            assert got[0] == 0 and got[-1] == n_docs and all(a <= b for a, b in zip(got, got[1:]))
    # one huge document among small ones: every shard stays a contiguous range, nothing is lost
    off = np.array([0, 10, 20, 10_000_020, 10_000_030], dtype=np.uint64)
    assert capi.partition_by_bytes(off, 4) == shard.partition_by_bytes(off, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 5])
def test_sharded_batch_equals_one_context(world):
    one = capi.Context(0)
    multi = capi.MultiContext([0] * world)
    try:
        assert multi.size() == world
        text, off = corpora.english(3000, 1024, first_doc=170000)
        keep = np.random.default_rng(world).random(off.size) < 0.4   # documents of 1 to ~10 KiB
        keep[0] = keep[-1] = True
        off = off[keep]
        for bos, eos in ((True, True), (False, False)):
            want_tok, want_off = one.encode_batch(text, off, bos, eos)
            got_tok, got_off = multi.encode_batch(text, off, bos, eos)
            assert np.array_equal(got_off, want_off) and np.array_equal(got_tok, want_tok)
        if world > 1:
            tiles = [multi.last_counters(i)[0] for i in range(world)]
            assert all(t > 0 for t in tiles)          # every context took part
        # capacity protocol: too small -> the size needed, offsets still filled in
        with pytest.raises(capi.L3Error) as e:
            multi.encode_batch(text, off, True, True, out_cap=100)
        assert e.value.code == capi.L3_E_OUTPUT_CAP
        # ragged: empty documents, fewer documents than contexts, one dominant document, an empty batch
        for docs in ([b""], [b"", b"x", b""], [b"a" * 100000, b"b", b"c"], [b"hello world"] * 3, []):
            t, o = capi.pack_docs(docs)
            w = one.encode_batch(t, o)
            g = multi.encode_batch(t, o)
            assert np.array_equal(g[1], w[1]) and np.array_equal(g[0], w[0])
    finally:
        one.close()
        multi.close()


@pytest.mark.gpu
def test_sharded_with_custom_special_tokens():
    from tests.oracle import Oracle
    names = [b"<|begin_of_text|>", b"<|my_tok|>"]
    multi = capi.MultiContext([0, 0, 0], specials=names)
    o = Oracle(specials=names)
    try:
        docs = [b"one<|eot_id|>", b"<|begin_of_text|>two", b" three <|my_tok|>", b"", b"four"]
        t, off = capi.pack_docs(docs)
        tok, ooff = multi.encode_batch(t, off)
        ooff = ooff.astype(np.int64)
        assert [tok[ooff[i]:ooff[i + 1]].tolist() for i in range(len(docs))] == [o.encode(d, True, True) for d in docs]
    finally:
        multi.close()


# ---- failure paths (what a one-GPU box CAN show of the several-device code: there is no second card to run on) ----
@pytest.mark.gpu
def test_devices_that_do_not_exist_are_refused_without_leaks():
    import torch
    n = torch.cuda.device_count()
    one = capi.Context(0)                     # (loads the code object, so that the figures below compare like with like)
    text, off = capi.pack_docs([b"still alive", b"", b"after the refusals"])
    want = one.encode_batch(text, off)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(6):
        with pytest.raises(capi.L3Error) as e:
            capi.Context(device=n)            # l3_create(device = one past the last)
        assert e.value.code == capi.L3_E_NO_DEVICE and "no usable HIP device" in str(e.value)
        with pytest.raises(capi.L3Error) as e:
            capi.MultiContext([0, n])         # the context on device 0 exists by then and must be given back
        assert e.value.code == capi.L3_E_NO_DEVICE
        with pytest.raises(capi.L3Error) as e:
            capi.MultiContext([0, 0, -1])
        assert e.value.code == capi.L3_E_NO_DEVICE
    with pytest.raises(capi.L3Error) as e:
        capi.MultiContext([])
    assert e.value.code == capi.L3_E_INVALID_ARG
    torch.cuda.synchronize()
    leaked = free0 - torch.cuda.mem_get_info(0)[0]
    assert leaked < (4 << 20), leaked         # one context's tables alone are > 10 MB; 12 were made and given back
    got = one.encode_batch(text, off)         # the refusals left the library and device 0 usable
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    one.close()


def test_bench_refuses_more_gpus_than_there_are():
    """`bench.py --gpus N` with fewer than N cards: a clear message and a non-zero exit before any rank starts (here: 0 cards;
    on the one-GPU box the gpu-marked twin below)."""
    import os
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n + 1 if n else 2), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_there_are_on_the_gpu_box():
    test_bench_refuses_more_gpus_than_there_are()
