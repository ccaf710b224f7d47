#!/usr/bin/env python3
"""bench.py -- Llama-3 batch encode throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config c2|c5]

One process per GPU.  Started plainly with --gpus N > 1 it launches N fresh ranks itself (torch.distributed.run on
127.0.0.1) BEFORE touching a GPU and forwards rank 0's line; started under torch.distributed.run it is one of the ranks.

Workload: the C5 corpus of SURVEY.md 8(d) -- 67,108,864 synthetic English documents of 1,024 B, document i generated from
seed + i ON the GPU that owns it (tools/l3_corpus_dev.hip; the same bytes as the host generator) -- cut into batches of
65,536 documents (64 MiB = the C2 configuration of BASELINE.json).  A step = one batch through l3_encode_submit / l3_encode_wait
(the stream-ordered halves of l3_encode_batch_device; the wait of batch k follows the submit of batch k + 1): text resident in HBM, token ids and per-document offsets left in HBM; every step encodes a DIFFERENT batch.

  --config c2 (default)  K steps per rank: rank r's step s is batch s * N + r of the corpus.  Weak scaling (per-GPU work
                         is fixed as N grows).  This is the headline line: `value` = tokens of all ranks / max time.
  --config c5            the WHOLE 64 GiB corpus, strong scaling: rank r encodes the contiguous 1,024 / N batches it owns
                         (--steps is ignored, K = 1,024 / N; --c5-batches shrinks the corpus for rehearsals).

Ranks share nothing on the data path (documents are independent); the only collective is the all-gather of per-rank
{tokens, documents} over RCCL after the timed steps, from which global token offsets follow.

Rank 0 prints ONE JSON line: whole-job tokens/s, the HBM-roofline figures (kernel times from HIP events recorded on the
launch stream between the kernels of the timed steps), C3 / C4 as `other_configs` (N = 1 runs only) and a CPU baseline
(the C restatement of the reference's algorithm on the host cores, threads stated) on a bounded sample of the workload.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
BATCH_DOCS = 65536
DOC_BYTES = 1024
C5_DOCS = 67108864     # 64 GiB
# kernels of one stream-ordered batch (launch_encode + the finish launch); rocprofv3 summaries hold others too (generators)
PIPELINE_KERNELS = ("docs_kernel", "tile_kernel", "merge_kernel", "slow_kernel", "block_totals_kernel", "emit_kernel",
                    "finish_kernel", "emit_generic_kernel", "status_out_kernel", "copy_runs_kernel")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c2", "c5"), default="c2")
    ap.add_argument("--docs", type=int, default=BATCH_DOCS, help="documents per batch (default: the C2 batch)")
    ap.add_argument("--c5-batches", type=int, default=C5_DOCS // BATCH_DOCS, help="batches of the whole C5 corpus (default 1,024 = 64 GiB)")
    ap.add_argument("--cpu-sample-docs", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the strong-scaling leg over the whole C5 corpus that follows the C2 steps")
    ap.add_argument("--c5-slice-batches", type=int, default=128, help="batches of the C5 leg generated and held in HBM at a time (8 GiB)")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="all ranks on GPU 0 with gloo collectives: exercises the N > 1 code path on a one-GPU box (not a measurement)")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh ranks as CHILD processes (this process has not touched the GPU and
    never will), forward their output, exit with their code."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    # Only rank 0's JSON line goes to this process's stdout (the contract: ONE line); whatever else the ranks and the
    # launcher print on theirs (gloo / RCCL banners) is passed on to stderr.
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        out = sys.stdout if line.startswith("{") and line.rstrip().endswith("}") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def source_digest():
    """Digest of the kernel sources: a PMC summary under profiles/ is only quoted if it was taken from these sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tokenizer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


C4_NAMES = ["blanks", "tabs", "newline_blank", "digits", "one_letter", "apostrophe_s", "random_bytes", "invalid_sequences",
            "begin_of_text_every_64", "all_specials", "near_misses", "unicode_mix", "ff_quarter", "bangs_newlines"]


def b_alg(n_bytes, n_tok, n_docs):
    """algorithmic bytes of one launch: input read once + int32 ids written once + uint64 offsets written once"""
    return n_bytes + 4 * n_tok + 8 * (n_docs + 1)


def time_resident(ctx, torch, capi, text, off, steps, warmup, flags=3):
    """Device-resident timing of one batch given as host arrays (other_configs): returns a dict."""
    import ctypes as C
    import numpy as np
    n_bytes, n_docs = int(text.size), int(off.size - 1)
    d_text = torch.from_numpy(np.ascontiguousarray(text)).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    cap = n_bytes * 3 + 2 * n_docs + 16
    d_out = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    n_out = C.c_uint64(0)
    args = (ctx.h, C.c_void_p(d_text.data_ptr()), C.c_uint64(n_bytes), C.c_void_p(d_off.data_ptr()), C.c_uint64(n_docs),
            C.c_uint32(flags), C.c_void_p(d_out.data_ptr()), C.c_uint64(cap), C.c_void_p(d_out_off.data_ptr()), C.byref(n_out), None)
    for _ in range(warmup):
        rc = ctx.lib.l3_encode_batch_device(*args)
        assert rc == capi.L3_OK, rc
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        rc = ctx.lib.l3_encode_batch_device(*args)
        assert rc == capi.L3_OK, rc
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern, n_prof = ctx.timing_totals()
    ctx.set_profiling(False)
    for k in kern:
        kern[k] /= max(n_prof, 1)
    n_tok = int(n_out.value)
    ba = b_alg(n_bytes, n_tok, n_docs)
    return {"ms": dt * 1e3, "kernels_ms": kern["total"], "tokens": n_tok, "input_bytes": n_bytes, "docs": n_docs, "B_alg": ba,
            "frac": ba / (kern["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS if kern["total"] > 0 else None,
            "tokens_per_s": n_tok / dt, "input_gb_per_s": n_bytes / dt / 1e9, "kernel_ms": kern,
            "steps": steps, "workspace_retries_last_call": ctx.last_counters()["retries"]}, (d_out, d_out_off)


def main():
    args = parse_args()
    if args.gpus > 1 and not args.rehearse_one_gpu:
        import torch   # (counting the devices does not initialise the GPU: the launcher below stays free to start children)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible: nothing was run (one rank per GPU; "
                             "--rehearse-one-gpu runs the N-rank code path on one card, not as a measurement)" % (args.gpus, have))
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    device = 0 if args.rehearse_one_gpu else local_rank
    torch.cuda.set_device(device)
    coll_dev = "cpu" if args.rehearse_one_gpu else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from tokenizer_amd import capi
    from tools import corpora
    import ctypes as C

    ctx = capi.Context(device=device)
    n_docs = args.docs
    n_bytes = n_docs * DOC_BYTES
    from tokenizer_amd import shard
    my_batches = shard.corpus_batches(args.config, rank, world, args.steps, args.c5_batches)
    warm_batches = my_batches[:min(args.warmup, len(my_batches))]
    scaling = "strong" if args.config == "c5" else "weak"
    steps = len(my_batches)

    # ---- the rank's slice of the corpus, generated on its GPU and resident before the timed region starts
    d_text = torch.empty(steps * n_bytes, dtype=torch.uint8, device="cuda")
    g0 = time.perf_counter()
    for i, b in enumerate(my_batches):
        corpora.english_device(d_text[i * n_bytes:(i + 1) * n_bytes], b * n_docs, n_docs, DOC_BYTES)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - g0
    d_off = torch.arange(n_docs + 1, dtype=torch.int64, device="cuda") * DOC_BYTES   # every batch: 65,536 x 1,024 B
    out_cap = n_bytes // 2 + 2 * n_docs
    d_out = torch.empty(out_cap, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    flags = capi.L3_BOS | capi.L3_EOS

    # The calls of one step, with their arguments marshalled once (the host side of a step is in the timed region too).
    # A step = l3_encode_submit of its batch + l3_encode_wait of the batch before: the host looks at batch k while the
    # GPU is busy with batch k + 1 (two output buffers in turn), as the loop of a corpus job does.
    n_out = C.c_uint64(0)
    ticket = C.c_int(-1)
    base_ptr = d_text.data_ptr()
    d_out2 = torch.empty(out_cap, dtype=torch.int32, device="cuda")
    d_out_off2 = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
    outs = ((d_out, d_out_off), (d_out2, d_out_off2))

    def submit_args(i):
        o, oo = outs[i & 1]
        return (ctx.h, C.c_void_p(base_ptr + i * n_bytes), C.c_uint64(n_bytes), C.c_void_p(d_off.data_ptr()), C.c_uint64(n_docs),
                C.c_uint32(flags), C.c_void_p(o.data_ptr()), C.c_uint64(out_cap), C.c_void_p(oo.data_ptr()), None, C.byref(ticket))

    all_args = [submit_args(i) for i in range(steps)]
    submit_fn, wait_fn = ctx.lib.l3_encode_submit, ctx.lib.l3_encode_wait

    def submit(i):
        rc = submit_fn(*all_args[i])
        if rc != capi.L3_OK:
            raise RuntimeError("l3_encode_submit failed: %d" % rc)
        return ticket.value

    def wait(t):
        rc = wait_fn(ctx.h, t, C.byref(n_out))
        if rc != capi.L3_OK:
            raise RuntimeError("l3_encode_wait failed: %d" % rc)
        return n_out.value

    def run(first, count):
        """batches first .. first + count, one pending while the next is queued; returns their token total"""
        total, prev = 0, None
        for i in range(first, first + count):
            t = submit(i)
            if prev is not None:
                total += wait(prev)
            prev = t
        if prev is not None:
            total += wait(prev)
        return total

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: HIP events on the launch stream only where the roofline figures need them (before and behind the tile
    # kernel, first kernel start, last kernel end).  An event between two kernels keeps the second from starting early;
    # the full per-kernel breakdown therefore comes from a second, UNTIMED pass over the same batches below.
    ctx.set_profiling(2)
    run(0, len(warm_batches))
    sync()
    ctx.set_profiling(2)      # resets the library's sums: they cover exactly the timed steps
    t0 = time.perf_counter()
    n_tok_total = run(0, steps)
    sync()
    dt = time.perf_counter() - t0
    kern_timed, n_prof = ctx.timing_totals()
    if n_prof != steps:
        raise RuntimeError("profiled %d calls, expected %d (a call was retried with a larger workspace?)" % (n_prof, steps))
    ctx.set_profiling(1)      # untimed: events between all kernels
    run(0, steps)
    sync()
    kern, n_prof2 = ctx.timing_totals()
    ctx.set_profiling(False)
    for k in kern:
        kern[k] /= max(n_prof2, 1)
        kern_timed[k] /= max(steps, 1)
    counters = ctx.last_counters()
    last_tok = int(n_out.value)

    my_docs = steps * n_docs
    my_bytes = steps * n_bytes
    if dist is not None:
        # The path has no exchange step: ranks share nothing while they encode.  What a job needs to address the shards
        # as one array is each shard's {tokens, documents}: 16 bytes per rank over RCCL, after the timed steps.
        global_token_base, global_doc_base, (tot_tok, tot_docs) = shard.gather_counts(dist, n_tok_total, my_docs, device=coll_dev)
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot_bytes = tot_docs * DOC_BYTES
        assert global_token_base <= tot_tok and global_doc_base <= tot_docs
    else:
        tot_tok, tot_bytes, tot_docs = n_tok_total, my_bytes, my_docs

    # ---- C5, strong scaling (BASELINE configs[4]; the reference's only concurrent workload is benchmark_test.go:151-174): the
    #      whole corpus of --c5-batches batches cut into `world` contiguous ranges; every rank generates its range ON its GPU in
    #      slices of --c5-slice-batches batches (untimed) and encodes each slice back to back (timed: the sum over the slices)
    c5 = None
    if args.config == "c2" and not args.no_c5:
        c5_batches = shard.corpus_batches("c5", rank, world, 0, args.c5_batches)
        del d_text
        torch.cuda.empty_cache()
        slice_n = max(1, min(args.c5_slice_batches, len(c5_batches)))
        d_slice = torch.empty(slice_n * n_bytes, dtype=torch.uint8, device="cuda")
        c5_tok, c5_s, c5_gen_s = 0, 0.0, 0.0
        ctx.set_profiling(False)
        for s0 in range(0, len(c5_batches), slice_n):
            part = c5_batches[s0:s0 + slice_n]
            g1 = time.perf_counter()
            for i, b in enumerate(part):
                corpora.english_device(d_slice[i * n_bytes:(i + 1) * n_bytes], b * n_docs, n_docs, DOC_BYTES)
            torch.cuda.synchronize()
            c5_gen_s += time.perf_counter() - g1
            sargs = [(ctx.h, C.c_void_p(d_slice.data_ptr() + i * n_bytes), C.c_uint64(n_bytes), C.c_void_p(d_off.data_ptr()), C.c_uint64(n_docs),
                      C.c_uint32(flags), C.c_void_p(outs[i & 1][0].data_ptr()), C.c_uint64(out_cap), C.c_void_p(outs[i & 1][1].data_ptr()),
                      None, C.byref(ticket)) for i in range(len(part))]
            e0 = time.perf_counter()
            prev = None
            for a in sargs:
                rc = submit_fn(*a)
                if rc != capi.L3_OK:
                    raise RuntimeError("l3_encode_submit failed: %d" % rc)
                t = ticket.value
                if prev is not None:
                    c5_tok += wait(prev)
                prev = t
            if prev is not None:
                c5_tok += wait(prev)
            torch.cuda.synchronize()
            c5_s += time.perf_counter() - e0
        del d_slice
        c5_docs = len(c5_batches) * n_docs
        if dist is not None:
            _tb, _db, (c5_tok_all, c5_docs_all) = shard.gather_counts(dist, c5_tok, c5_docs, device=coll_dev)
            t = torch.tensor([c5_s, c5_gen_s], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c5_s, c5_gen_s = float(t[0].item()), float(t[1].item())
        else:
            c5_tok_all, c5_docs_all = c5_tok, c5_docs
        c5_bytes = c5_docs_all * DOC_BYTES
        c5 = {"corpus_bytes": c5_bytes, "docs": c5_docs_all, "batches_per_gpu": len(c5_batches), "ms_total": c5_s * 1e3,
              "ms_per_batch": c5_s * 1e3 / max(len(c5_batches), 1), "tokens": c5_tok_all, "tokens_per_s": c5_tok_all / c5_s if c5_s > 0 else None,
              "input_gb_per_s": c5_bytes / c5_s / 1e9 if c5_s > 0 else None,
              "frac": (b_alg(c5_bytes, c5_tok_all, c5_docs_all) / c5_s / 1e9 / (HBM_PEAK_GBS * world)) if c5_s > 0 else None,
              "frac_note": "algorithmic bytes of the whole corpus / slowest rank's encode wall time / (n_gpus x 8 TB/s)",
              "generation_s": c5_gen_s, "scaling": "strong", "n_gpus": world,
              "n_ranks_seen_by_rccl": (world if (dist is not None and not args.rehearse_one_gpu) else (0 if dist is None else "gloo rehearsal")),
              "note": "every rank generates its contiguous range of the corpus on its GPU, %d batches (%.0f GiB) at a time, untimed, and "
                      "encodes each slice back to back through l3_encode_submit / l3_encode_wait; ms_total is the slowest rank's sum "
                      "over its slices" % (slice_n, slice_n * n_bytes / 2 ** 30)}
        d_text = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")      # (the legs below want rank 0's first batch again)
        corpora.english_device(d_text, my_batches[0] * n_docs, n_docs, DOC_BYTES)
        base_ptr = d_text.data_ptr()
        all_args = [submit_args(0)]
        torch.cuda.synchronize()

    if rank == 0:
        ms_per_step = dt / steps * 1e3
        value = tot_tok / dt
        tok_per_batch = n_tok_total / steps
        ba = b_alg(n_bytes, tok_per_batch, n_docs)
        dom = max(("tile", "merge", "slow", "emit", "scan", "docs"), key=lambda k: kern[k])   # (from the untimed breakdown)
        t_kernels = kern_timed["total"]                                  # first kernel start -> last kernel end, timed steps
        t_dom = kern_timed[dom] if dom in ("tile", "docs") else kern[dom]  # the timed region brackets the tile kernel only
        pipeline = ba / (t_kernels * 1e-3) / 1e9 if t_kernels > 0 else 0.0
        dom_gbs = ba / (t_dom * 1e-3) / 1e9 if t_dom > 0 else 0.0
        if args.config == "c5":
            workload = ("C5: %d synthetic English docs x %d B (%.1f GiB) generated on the owning GPU, sharded by document over %d "
                        "GPU(s), %d batches of %d docs per GPU encoded back to back, BOS+EOS, inputs and outputs resident in HBM"
                        % (tot_docs, DOC_BYTES, tot_bytes / 2 ** 30, world, steps, n_docs))
        else:
            workload = ("C2: %d synthetic English docs x %d B per batch (64 MiB), one batch per step and GPU, every step a different "
                        "batch of the C5 corpus generated on the GPU, BOS+EOS, inputs and outputs resident in HBM" % (n_docs, DOC_BYTES))
        result = {
            "metric": "encode tokens/sec (whole node), Llama-3 BPE", "value": value, "unit": "tokens/s",
            "n_gpus": world, "steps": steps, "warmup": len(warm_batches), "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": workload, "config": args.config, "docs_per_batch": n_docs, "doc_bytes": DOC_BYTES,
                       "batches_per_gpu": steps, "corpus_bytes": tot_bytes,
                       "parallelism": "dp%d (documents sharded, no data-path collective)" % world,
                       **({"rehearsal": "all ranks share GPU 0, gloo collectives: NOT a measurement"} if args.rehearse_one_gpu else {})},
            "input_gb_per_s": tot_bytes / dt / 1e9,
            "tokens_total": tot_tok, "tokens_per_batch": tok_per_batch,
            "bytes_per_token": my_bytes / max(n_tok_total - 2 * my_docs, 1),
            "corpus_generation_s": gen_s,
            "kernel_ms": kern, "kernel_ms_note": "per-kernel breakdown from an untimed pass over the same batches with events between all kernels",
            "kernel_ms_timed": {"docs": kern_timed["docs"], "tile": kern_timed["tile"], "total": kern_timed["total"]},
            "deferred_pretokens_last_batch": {"short": counters["short_pretokens"], "slow": counters["slow_pretokens"]},
            **({"c5_strong": c5} if c5 is not None else {}),
            # frac: algorithmic bytes / (first kernel start -> last kernel end) / peak, as SURVEY 8(d) defines it
            "roofline": {"bound": "hbm", "achieved": pipeline, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pipeline / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": ba, "kernels_ms_per_launch": t_kernels,
                         "wall_frac": ba / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "dominant": {"kernel": dom + "_kernel", "avg_ms": t_dom, "achieved": dom_gbs, "frac": dom_gbs / HBM_PEAK_GBS,
                                      "measured_in": "timed steps" if dom in ("tile", "docs") else "untimed breakdown pass"}},
        }
        # HBM traffic per launch (all kernels of the pipeline) from the committed rocprofv3 PMC passes of this command --
        # quoted only when that summary was taken from exactly these kernel sources
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_summary.json")))[-1]
            summ = json.load(open(pm))
            if summ.get("source_digest") == source_digest():
                # (only what launch_encode and the hand-over launch: the corpus generator is not part of a step)
                result["roofline"]["traffic"] = sum(k["hbm_bytes_per_launch_corrected"] for name, k in summ["kernels"].items()
                                                    if name in PIPELINE_KERNELS)
                result["roofline"]["traffic_source"] = os.path.relpath(pm, ROOT)
                result["roofline"]["dominant"]["traffic"] = summ["kernels"].get(dom + "_kernel", {}).get("hbm_bytes_per_launch_corrected")
            else:
                result["roofline"]["traffic_source"] = "none: %s was taken from other kernel sources" % os.path.relpath(pm, ROOT)
        except Exception:
            pass

        if world == 1 and not args.no_other_configs:
            other = {}
            try:
                # C3: 16 multilingual documents of 4 MiB (CJK runs of up to 4,096 characters, Cyrillic, emoji)
                text3, off3 = corpora.multilingual(16, 4 << 20)
                other["c3_16x4MiB"], _ = time_resident(ctx, torch, capi, text3, off3, steps=5, warmup=2)
                # C4: the pathological documents as ONE batch (1 MiB each: one blank / tab / letter repeated, digits,
                # newline patterns, random and invalid bytes, special tokens and near misses)
                docs4 = corpora.pathological(1 << 20)
                text4, off4 = capi.pack_docs(docs4)
                other["c4_batch"], _ = time_resident(ctx, torch, capi, text4, off4, steps=3, warmup=1)
                per_doc = []
                for name, d in zip(C4_NAMES, docs4):     # every document of the batch on its own
                    t, o = capi.pack_docs([d])
                    r1, _ = time_resident(ctx, torch, capi, t, o, steps=2, warmup=1)
                    cn = ctx.last_counters()
                    per_doc.append({"doc": name, "bytes": len(d), "tokens": r1["tokens"], "kernels_ms": r1["kernels_ms"],
                                    "slow_ms": r1["kernel_ms"]["slow"], "emit_ms": r1["kernel_ms"]["emit"], "host_passes_ms": r1["kernel_ms"].get("host_passes", 0.0),
                                    "tile_ms": r1["kernel_ms"]["tile"], "merge_ms": r1["kernel_ms"]["merge"],
                                    "window_items": cn["window_items"], "giant_symbols": cn["giant_symbols"]})
                other["c4_batch"]["per_doc"] = per_doc
                for name, d in (("c4_1MiB_one_letter", b"a" * (1 << 20)), ("c4_1MiB_invalid_bytes", b"\xff" * (1 << 20))):
                    t, o = capi.pack_docs([d])
                    other[name], _ = time_resident(ctx, torch, capi, t, o, steps=3, warmup=1)
            except Exception as e:   # the headline must not die with a side measurement
                other["error"] = repr(e)
            result["other_configs"] = other
            side = {}
            try:
                # f-2: decode of this rank's last batch of ids (host-pointer call; the figure is the kernels' time by events)
                last_i = steps - 1 if c5 is None else 0
                wait(submit(last_i))
                torch.cuda.synchronize()
                o, oo = outs[last_i & 1]
                h_off = oo.cpu().numpy().astype(np.uint64)
                h_ids = o[:int(h_off[-1])].cpu().numpy()
                ctx.set_profiling(1)
                dms = []
                for _ in range(4):
                    dec, _doff = ctx.decode_batch(h_ids, h_off)
                    dms.append(ctx.last_timings()["decode"])
                ctx.set_profiling(False)
                dms = sorted(dms[1:])[1]
                side["decode_c2"] = {"ids": int(h_ids.size), "bytes_out": int(dec.size), "kernels_ms": dms,
                                     "out_gb_per_s": dec.size / dms / 1e6,
                                     "frac": (4 * h_ids.size + dec.size + 16 * h_off.size) / dms / 1e6 / HBM_PEAK_GBS,
                                     "note": "ids of the last batch incl. BOS/EOS -> bytes; kernels only, by HIP events"}
                # host-fed (PCIe-inclusive, never `value`): l3_encode_batch on one batch held in pageable host memory
                h_text = d_text[:n_bytes].cpu().numpy()
                h_doff = (np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(DOC_BYTES))
                h_out = np.zeros(out_cap, dtype=np.int32)        # (touched once, so that the timed calls fault no pages in)
                h_ooff = np.zeros(n_docs + 1, dtype=np.uint64)
                h_n = C.c_uint64(0)

                def host_call():
                    rc = ctx.lib.l3_encode_batch(ctx.h, h_text.ctypes.data, h_doff.ctypes.data, n_docs, flags, h_out.ctypes.data,
                                                 out_cap, h_ooff.ctypes.data, C.byref(h_n))
                    if rc != capi.L3_OK:
                        raise RuntimeError("l3_encode_batch failed: %d" % rc)
                for _ in range(3):
                    host_call()
                h0 = time.perf_counter()
                for _ in range(6):
                    host_call()
                hdt = (time.perf_counter() - h0) / 6
                side["host_fed_pageable_c2"] = {"ms": hdt * 1e3, "input_gb_per_s": n_bytes / hdt / 1e9,
                                                "note": "upload + kernels + download through l3_encode_batch, caller buffers pageable"}
            except Exception as e:
                side["error"] = repr(e)
            result["side_figures"] = side

        if not args.no_cpu_baseline:
            from tests.oracle import Oracle
            threads = min(16, os.cpu_count() or 1)
            ns = min(args.cpu_sample_docs, n_docs)
            oracle = Oracle()
            # the first batch of this rank, from the HOST generator (the same bytes the GPU generated: checked)
            s_text, s_off = corpora.english(ns, DOC_BYTES, first_doc=my_batches[0] * n_docs)
            same_bytes = bool(np.array_equal(d_text[:ns * DOC_BYTES].cpu().numpy(), s_text))
            passes = 12  # about 15 core-seconds x 16 threads
            c0 = time.perf_counter()
            for _ in range(passes):  # one tokenizer, its pre-token caches persist as the reference's default cache does
                exp_tok, exp_off = oracle.encode_batch(s_text, s_off, True, True, threads=threads)
            cdt = time.perf_counter() - c0
            wait(submit(0))
            torch.cuda.synchronize()
            got_off = d_out_off[:ns + 1].cpu().numpy().astype(np.uint64)
            got_tok = d_out[:int(got_off[-1])].cpu().numpy()
            parity = bool(same_bytes and np.array_equal(got_off, exp_off) and np.array_equal(got_tok, exp_tok))
            result["cpu_baseline"] = {"value": passes * exp_tok.size / cdt, "unit": "tokens/s", "cores": threads, "kind": "port",
                                      "sample": "%d passes over the first %d documents (%d MiB) of rank 0's first batch; C restatement "
                                                "of the reference algorithm, document-per-thread work queue, per-thread "
                                                "pre-token caches kept across passes" % (passes, ns, ns * DOC_BYTES >> 20),
                                      "seconds": cdt, "core_seconds": cdt * threads, "input_gb_per_s": passes * ns * DOC_BYTES / cdt / 1e9}
            result["parity_vs_oracle_on_sample"] = parity
            result["device_corpus_equals_host_corpus_on_sample"] = same_bytes
            if not parity:
                result["error"] = "GPU output differs from the oracle on the CPU-baseline sample"
        del last_tok
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
