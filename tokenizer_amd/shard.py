"""Document sharding across the GPUs of one node.

Documents are independent units: rank r owns a contiguous range of documents balanced by byte count, runs
the encode on its own GPU with its own context, and nothing crosses ranks on the data path.  The only
exchange is an all-gather of per-shard {tokens, documents} (16 B per rank; RCCL on GPUs, gloo in CPU tests)
from which every rank forms the global token offset of its shard."""
import numpy as np


def partition_by_bytes(doc_off, world):
    """Split documents [0, n) into `world` contiguous ranges with ~equal bytes.  Returns world+1 doc indices."""
    doc_off = np.asarray(doc_off, dtype=np.uint64)
    n = doc_off.size - 1
    total = int(doc_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        i = int(np.searchsorted(doc_off, np.uint64(target), side="left"))
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return cuts


def shard(text, doc_off, cuts, rank):
    """The (text, offsets) of rank's shard, offsets rebased to 0."""
    d0, d1 = cuts[rank], cuts[rank + 1]
    b0, b1 = int(doc_off[d0]), int(doc_off[d1])
    return text[b0:b1], (np.asarray(doc_off[d0:d1 + 1], dtype=np.uint64) - np.uint64(b0))


def gather_counts(dist, n_tok, n_docs, device=None):
    """All-gather {tokens, documents} of every rank; returns (token_base, doc_base, totals) for this rank."""
    import torch
    mine = torch.tensor([int(n_tok), int(n_docs)], dtype=torch.int64, device=device)
    world = dist.get_world_size()
    allc = torch.zeros(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)
    allc = allc.cpu().numpy().reshape(world, 2)
    r = dist.get_rank()
    return int(allc[:r, 0].sum()), int(allc[:r, 1].sum()), (int(allc[:, 0].sum()), int(allc[:, 1].sum()))


def corpus_batches(config, rank, world, steps, total_batches):
    """Batches (of 65,536 documents) of the C5 corpus that `rank` encodes, in order (bench.py).
    "c5": the whole corpus of `total_batches`, each rank one contiguous range of whole batches (strong scaling: the ranges
    are disjoint and cover the corpus).  "c2": `steps` batches per rank, step s of rank r = batch s * world + r (weak
    scaling: every step the ranks take `world` consecutive batches)."""
    if config == "c5":
        return list(range(total_batches * rank // world, total_batches * (rank + 1) // world))
    return [s * world + rank for s in range(steps)]
