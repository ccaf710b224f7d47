"""world_size-2 (and 3) gloo tests of the multi-GPU path's host logic: byte-balanced document sharding and the
all-gather of per-shard counts that forms global offsets.  The per-shard encode is stood in for by the oracle
(no GPU here); on the GPU box the same ranks call l3_encode_batch_device (bench.py)."""
import os
import random
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, seed, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests import corpus
    from tests.oracle import Oracle
    from tokenizer_amd import capi, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = random.Random(seed)
    docs = [corpus.random_doc(rng, max_pieces=30) for _ in range(600)] + [b""] * 5
    text, off = capi.pack_docs(docs)
    cuts = shard.partition_by_bytes(off, world)
    s_text, s_off = shard.shard(text, off, cuts, rank)
    tok, toff = Oracle().encode_batch(s_text, s_off, True, True, threads=1)
    tok_base, doc_base, totals = shard.gather_counts(dist, tok.size, s_off.size - 1)
    assert doc_base == cuts[rank]
    q.put((rank, tok_base, doc_base, totals, tok.tolist(), (toff.astype(np.int64) + tok_base).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_matches_whole_batch(world, oracle):
    from tests import corpus
    from tokenizer_amd import capi
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, 42, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = random.Random(42)
    docs = [corpus.random_doc(rng, max_pieces=30) for _ in range(600)] + [b""] * 5
    text, off = capi.pack_docs(docs)
    exp_tok, exp_off = oracle.encode_batch(text, off, True, True, threads=2)
    tokens, offsets = [], []
    for rank, tok_base, doc_base, totals, tok, goff in res:
        assert tok_base == len(tokens) and totals == (exp_tok.size, len(docs))
        tokens += tok
        offsets += goff[:-1]
    offsets.append(len(tokens))
    assert tokens == exp_tok.tolist() and offsets == exp_off.astype(np.int64).tolist()


def test_partition_is_balanced_and_total():
    from tokenizer_amd import shard
    off = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(0, 5000, 10000))]).astype(np.uint64)
    for world in (1, 2, 4, 8):
        cuts = shard.partition_by_bytes(off, world)
        assert cuts[0] == 0 and cuts[-1] == 10000 and all(a <= b for a, b in zip(cuts, cuts[1:]))
        sizes = [int(off[b] - off[a]) for a, b in zip(cuts, cuts[1:])]
        assert max(sizes) - min(sizes) <= 2 * 5000
    assert shard.partition_by_bytes(np.array([0], dtype=np.uint64), 4) == [0, 0, 0, 0, 0]


def test_corpus_batches_are_disjoint_and_cover():
    from tokenizer_amd import shard
    for world in (1, 2, 3, 4, 8):
        got = [shard.corpus_batches("c5", r, world, 0, 1024) for r in range(world)]
        flat = [b for g in got for b in g]
        assert flat == list(range(1024))                                 # contiguous, in rank order, nothing twice
        assert max(map(len, got)) - min(map(len, got)) <= 1
        got = [shard.corpus_batches("c2", r, world, 20, 1024) for r in range(world)]
        assert sorted(b for g in got for b in g) == list(range(20 * world)) and all(len(g) == 20 for g in got)
