// l3_api.hip -- the C ABI of libl3hip.so (include/l3hip.h): context, device tables, workspace
// management and the host-side control of the encode / decode pipelines.  HIP only: there is no
// CPU fallback, every entry point fails with L3_E_NO_DEVICE / L3_E_HIP when the GPU path is unusable.
#include <hip/hip_runtime.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/l3hip.h"
#include "l3_kernels.h"
#include "l3_tables.h"

using namespace l3;

namespace {

thread_local std::string g_create_err;

// ---- the workspace of l3_encode_batch_device_noalloc (include/l3hip.h) ---------------------------------------------
// A file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the process
// ends.  In host code the NAME of the array is only a handle for hipGetSymbolAddress; kernels, memsets and copies get the
// per-device address that call returned (g_scratch[device].ptr), never the name.
constexpr size_t kNoAllocScratchBytes = (size_t)L3_NOALLOC_SCRATCH_BYTES;
static __device__ __attribute__((used)) __align__(256) unsigned char scratch[kNoAllocScratchBytes];
constexpr int kMaxDevices = 64;
struct DeviceScratch {
    unsigned char* ptr = nullptr;   // device address of `scratch` on this GPU
    std::mutex mu;                  // calls on one GPU share the array: one at a time
};
DeviceScratch g_scratch[kMaxDevices];
std::mutex g_scratch_mu;

// With `device` current: the address of this GPU's copy of the array (asked once per device).
hipError_t scratch_address(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratch[device].ptr) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch[device].ptr = static_cast<unsigned char*>(p);
    }
    *out = g_scratch[device].ptr;
    return hipSuccess;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grow-only; contents are NOT preserved
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class U> U* as() const { return reinterpret_cast<U*>(p); }
};

// Capacities of the lists a call appends to (entries, not bytes; scratch_cap in bytes).
struct Caps { uint64_t slow_cap, miss_shard_cap, mpool_shard_cap, pool_cap, scratch_cap, win_cap; };

}  // namespace

struct l3_ctx {
    int device = 0;
    HostTables H;
    Tables T{};
    DevBuf t_cls1, t_clsb, t_clsd, t_bytetok, t_bytepair, t_pairs, t_vslots, t_vblob, t_special, t_tokoff, t_tokbytes, t_prefix2, t_bloom, t_rankm;
    // The status word is zero between two calls: the kernel that hands it to the host (status_out_kernel) clears it, in
    // stream order behind the batch.  Calls on DIFFERENT streams are ordered by an event the context owns, recorded behind
    // every batch: the caller's stream handle is never kept, it may be destroyed between two calls.
    // slow_kernel's grid (and the wave pass's share of merge_kernel's) for the next batch: all 512 workgroups once a batch had
    // long pre-tokens, L3_IDLE_SLOW_GRID while batches have none.  Measured on C2 (no long pre-tokens): 32 / 128 / 512 idle
    // workgroups cost 8 / 9 / 13 us per batch.  The first batch WITH long pre-tokens after a run without runs on the idle
    // grid: C3's 35,000 window items took 25 ms on 32 workgroups (2.0 ms on 512), so the idle grid is 128, not 32.
#ifndef L3_IDLE_SLOW_GRID
#define L3_IDLE_SLOW_GRID 128
#endif
    uint32_t slow_grid = L3_IDLE_SLOW_GRID;
    bool status_dirty = false;        // the word must be zeroed in front of the next batch: it is new, or a call ended (with an
                                      // error) between two host-driven passes
    bool order_pending = false;       // ev_order has been recorded behind the last batch
    bool order_on_own = false;        // ... on the context's own stream (same-stream order makes the wait unnecessary)
    hipEvent_t ev_order = nullptr;
    DevBuf w_arena, w_status;         // the per-call workspace (one block, carve_workspace) and the status word
    DevBuf w_giant;                   // flat arrays of the grid-wide merge of giant pieces (only batches that have such pieces)
    DevBuf s_text, s_off, s_out, s_outoff, s_kinds, s_ids, s_idoff, s_total, s_decblk;
    hipEvent_t ev_dec[2] = {};          // around the decode kernels when profiling is on
    float last_decode_ms = 0.f;
    Caps caps{0, 0, 0, 0, 0, 0};         // list capacities inside the workspace, grow-only
    unsigned char* na_scratch = nullptr;  // this GPU's copy of the file-scope array `scratch` (hipGetSymbolAddress at l3_create)
    Status* h_status = nullptr;          // pinned
    unsigned long long* h_total = nullptr;  // pinned
    hipStream_t stream = nullptr;
    // host-fed batches in pieces: upload stream, download stream, two output buffers in turn
    hipStream_t stream_in = nullptr, stream_out = nullptr;
    DevBuf p_out[2], p_outoff[2];
    hipEvent_t ev_in[8] = {}, ev_out[2] = {}, ev_piece[8] = {};
    bool have_pipe = false;
    // page-locked staging for callers whose buffers are pageable (Go slices, numpy arrays): grow-only
    void* hs_text = nullptr; size_t hs_text_cap = 0;
    void* hs_out = nullptr; size_t hs_out_cap = 0;
    hipEvent_t ev[kEvCount] = {};
    bool have_events = false, profiling = false, profiling_light = false;
    float last_ms[kEvCount] = {};
    double sum_ms[kEvCount] = {};     // over the profiled calls since profiling was switched on
    uint64_t sum_calls = 0;
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // stream-ordered calls (l3_encode_submit / l3_encode_wait): what each pending batch left for its wait
    struct Pending {
        bool in_use = false, made = false, profiled = false, light = false;
        EncodeArgs A{};
        hipStream_t stream = nullptr;
        Status* h_status = nullptr;   // pinned
        hipEvent_t done = nullptr;
        hipEvent_t ev[kEvCount] = {};
    } pend[L3_MAX_PENDING];
    int n_pending = 0;
    std::unordered_map<std::string, int> special_by_name;
    std::mutex mu;
    std::string err;
};

namespace {

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return L3_E_HIP;                                                                   \
        }                                                                                      \
    } while (0)

template <class V>
int upload(l3_ctx* c, DevBuf& b, const V* src, size_t count) {
    size_t bytes = count * sizeof(V);
    HIPCHK(c, b.ensure(bytes ? bytes : 16));
    if (bytes) HIPCHK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return L3_OK;
}

int upload_tables(l3_ctx* c) {
    HostTables& H = c->H;
    int rc;
    if ((rc = upload(c, c->t_cls1, H.cls_stage1.data(), H.cls_stage1.size()))) return rc;
    if ((rc = upload(c, c->t_clsb, H.cls_blocks.data(), H.cls_blocks.size()))) return rc;
    if ((rc = upload(c, c->t_clsd, H.cls_direct.data(), H.cls_direct.size()))) return rc;
    if ((rc = upload(c, c->t_bytetok, H.byte_tok, 256))) return rc;
    if ((rc = upload(c, c->t_bytepair, H.bytepair.data(), H.bytepair.size()))) return rc;
    if ((rc = upload(c, c->t_pairs, H.pair_buckets.data(), H.pair_buckets.size()))) return rc;
    if ((rc = upload(c, c->t_vslots, H.vocab_slots.data(), H.vocab_slots.size()))) return rc;
    if ((rc = upload(c, c->t_vblob, H.vocab_blob.data(), H.vocab_blob.size()))) return rc;
    if ((rc = upload(c, c->t_special, H.special_id, 256))) return rc;
    if ((rc = upload(c, c->t_tokoff, H.tok_off.data(), H.tok_off.size()))) return rc;
    if ((rc = upload(c, c->t_tokbytes, H.tok_bytes.data(), H.tok_bytes.size()))) return rc;
    if ((rc = upload(c, c->t_prefix2, H.prefix2.data(), H.prefix2.size()))) return rc;
    if ((rc = upload(c, c->t_bloom, H.prefix_bloom.data(), H.prefix_bloom.size()))) return rc;
    if (!H.rank_merged.empty() && (rc = upload(c, c->t_rankm, H.rank_merged.data(), H.rank_merged.size()))) return rc;
    Tables& T = c->T;
    T.cls_stage1 = c->t_cls1.as<uint16_t>();
    T.cls_blocks = c->t_clsb.as<uint8_t>();
    T.cls_direct = c->t_clsd.as<uint8_t>();
    T.byte_tok = c->t_bytetok.as<uint8_t>();
    T.bytepair = c->t_bytepair.as<uint32_t>();
    T.pair_buckets = c->t_pairs.as<PairBucket>();
    T.vocab_slots = c->t_vslots.as<VocabSlot>();
    T.vocab_blob = c->t_vblob.as<uint32_t>();
    T.special_id = c->t_special.as<int32_t>();
    T.prefix2 = c->t_prefix2.as<uint32_t>();
    T.prefix_bloom = c->t_bloom.as<uint32_t>();
    T.rank_merged = H.rank_merged.empty() ? nullptr : c->t_rankm.as<uint32_t>();
    T.rank_count = (uint32_t)H.rank_merged.size();
    T.pair_log2 = H.pair_log2;
    T.vocab_log2 = H.vocab_log2;
    T.bloom_log2 = H.bloom_log2;
    // the copies above went through the null stream; the context's stream is non-blocking and does not wait for it
    HIPCHK(c, hipDeviceSynchronize());
    return L3_OK;
}

// Lays the per-call workspace of a batch of n_bytes / n_docs out in one block of memory starting at `base` (256-byte
// aligned pieces, in a fixed order) and returns the bytes it takes.  base == nullptr: only the size.  The status word is
// not part of the block (the context keeps it across calls); `status` is stored as given.
size_t carve_workspace(uint8_t* base, uint64_t n_bytes, uint64_t n_docs, const Caps& K, Status* status, Workspace& W) {
    const uint64_t n_tiles = num_tiles(n_bytes);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    W.stream = reinterpret_cast<int32_t*>(take(n_tiles * kTileCap * sizeof(int32_t)));
    W.tile_items = reinterpret_cast<uint32_t*>(take(n_tiles * 4));
    W.tile_extra = reinterpret_cast<uint32_t*>(take(n_tiles * 8));  // tile_extra + tile_digits
    W.tile_digits = base ? W.tile_extra + n_tiles : nullptr;
    W.tile_first = reinterpret_cast<uint32_t*>(take(n_tiles * 4));
    W.tile_base = reinterpret_cast<uint64_t*>(take((n_tiles / 1024 + 2) * 8));
    W.tile_rec = reinterpret_cast<TileRec*>(take(n_tiles * sizeof(TileRec)));
    W.tile_flag = reinterpret_cast<uint32_t*>(take(n_tiles * 4));
    W.tile_doc0 = reinterpret_cast<uint32_t*>(take((n_tiles + 1) * 4));
    W.doc_item = reinterpret_cast<uint32_t*>(take((n_docs + 1) * 4));
    W.slow = reinterpret_cast<SlowEntry*>(take(K.slow_cap * sizeof(SlowEntry)));
    W.miss_in = reinterpret_cast<MissIn*>(take(K.miss_shard_cap * kMissShards * sizeof(MissIn)));
    W.miss_res = reinterpret_cast<uint64_t*>(take(K.miss_shard_cap * kMissShards * sizeof(uint64_t)));
    W.pool = reinterpret_cast<int32_t*>(take((K.mpool_shard_cap * kMissShards + K.pool_cap) * sizeof(int32_t)));
    W.scratch = take(K.scratch_cap);
    W.allow_giants = 1;
    W.giants = reinterpret_cast<Giant*>(take(kMaxGiants * sizeof(Giant)));
    W.gstate = reinterpret_cast<GiantState*>(take(sizeof(GiantState)));
    W.runs = reinterpret_cast<BigRun*>(take((kMaxBigRuns + kMaxHugeRuns) * sizeof(BigRun)));
    W.win = reinterpret_cast<WinItem*>(take(K.win_cap * sizeof(WinItem)));
    W.win_live = reinterpret_cast<uint32_t*>(take(K.win_cap * 4));
    W.win_first = reinterpret_cast<uint32_t*>(take(K.win_cap * 4));
    W.win_cnt = reinterpret_cast<uint32_t*>(take(K.win_cap * 4));
    W.win_cap = K.win_cap;
    W.status = status;
    W.slow_cap = K.slow_cap;
    W.miss_shard_cap = K.miss_shard_cap;
    W.mpool_shard_cap = K.mpool_shard_cap;
    W.pool_cap = K.pool_cap;
    W.scratch_cap = K.scratch_cap;
    return off;
}

// smallest list capacities a batch of n_bytes starts with (they grow from the status word of an overflowing attempt)
Caps initial_caps(uint64_t n_bytes) {
    Caps K;
    K.slow_cap = n_bytes / 256 + 1024;
    K.miss_shard_cap = n_bytes / 2048 + 256;
    K.pool_cap = n_bytes / 8 + 4096;
    K.mpool_shard_cap = n_bytes / 16 / kMissShards + 1024;
    K.scratch_cap = 16ull << 20;
    K.win_cap = n_bytes / 8192 + 1024;
    return K;
}

// A stream placeholder is -2 - e with e < 2^31 - 2: short entries use e = shard * miss_shard_cap + index < kSlowBase,
// slow entries e = kSlowBase + index.  Lists that outgrow that space (tens of GiB of text in one call, or lists grown by
// many overflow retries) cannot be addressed: the caller has to split the batch.
bool caps_addressable(const Caps& K) {
    return (uint64_t)kMissShards * K.miss_shard_cap < (uint64_t)kSlowBase && K.slow_cap < (uint64_t)kSlowBase - 2;
}

int ensure_workspace(l3_ctx* c, uint64_t n_bytes, uint64_t n_docs, Workspace& W) {
    if (!c->w_status.p) {
        // Cleared in stream order in front of the first batch (status_dirty).  NOT with hipMemset here: that runs on the null
        // stream, which the context's non-blocking stream does not wait for -- the first batch of a new context could start
        // on whatever the recycled memory held, or have its flags wiped under it.
        HIPCHK(c, c->w_status.ensure(sizeof(Status)));
        c->status_dirty = true;
    }
    const Caps K0 = initial_caps(n_bytes);
    Caps& K = c->caps;   // grow-only across calls
    if (K.slow_cap < K0.slow_cap) K.slow_cap = K0.slow_cap;
    if (K.miss_shard_cap < K0.miss_shard_cap) K.miss_shard_cap = K0.miss_shard_cap;
    if (K.pool_cap < K0.pool_cap) K.pool_cap = K0.pool_cap;
    if (K.mpool_shard_cap < K0.mpool_shard_cap) K.mpool_shard_cap = K0.mpool_shard_cap;
    if (K.scratch_cap < K0.scratch_cap) K.scratch_cap = K0.scratch_cap;
    if (K.win_cap < K0.win_cap) K.win_cap = K0.win_cap;
    if (!caps_addressable(K)) {
        c->err = "batch too large for one call (deferred pre-token lists exceed 2^30 entries): split it";
        return L3_E_INVALID_ARG;
    }
    const size_t need = carve_workspace(nullptr, n_bytes, n_docs, K, nullptr, W);
    HIPCHK(c, c->w_arena.ensure(need));
    carve_workspace(c->w_arena.as<uint8_t>(), n_bytes, n_docs, K, c->w_status.as<Status>(), W);
    W.bos_id = c->H.bos_id;   // (looked up, not constants: tokenizer.go:226,259)
    W.eos_id = c->H.eos_id;
    W.slow_grid = c->slow_grid;
    return L3_OK;
}

// Orders `stream` behind the last batch of the context (the workspace is shared) ...
int order_behind_last(l3_ctx* c, hipStream_t stream) {
    if (c->order_pending) {
        c->order_pending = false;
        if (!(c->order_on_own && stream == c->stream)) HIPCHK(c, hipStreamWaitEvent(stream, c->ev_order, 0));
    }
    return L3_OK;
}
// ... and marks the end of this one
int mark_batch_end(l3_ctx* c, hipStream_t stream) {
    // always recorded: the next batch may come on another stream than this one, whichever of the two is the context's own
    // (order_behind_last skips the wait only between two batches on the context's own stream: a stream is ordered anyway)
    HIPCHK(c, hipEventRecord(c->ev_order, stream));
    c->order_pending = true;
    c->order_on_own = stream == c->stream;
    return L3_OK;
}

// After L3_E_OUTPUT_CAP and a larger output buffer: only the final scatter again.  The scatter counts the tiles it leaves
// to the generic kernel in the (already cleared) status word, so the word is cleared once more behind it.
int rerun_emit(l3_ctx* c, const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    int rc = order_behind_last(c, stream);
    if (rc) return rc;
    HIPCHK(c, launch_encode(c->T, W, A, stream, nullptr, true));
    HIPCHK(c, launch_status_out(W, c->h_status, stream));
    return mark_batch_end(c, stream);
}

// Milliseconds between the events of one profiled call: [docs, tile, merge, slow, scan, host-driven passes, emit, first start -> last end]; in
// light mode only docs, tile and the total exist (the others are 0).
int event_times(l3_ctx* c, hipEvent_t* ev, bool light, float* ms) {
    for (int i = 0; i < kEvCount; i++) ms[i] = 0;
    if (light) {
        HIPCHK(c, hipEventElapsedTime(&ms[0], ev[kEvStart], ev[kEvDocs]));
        HIPCHK(c, hipEventElapsedTime(&ms[1], ev[kEvDocs], ev[kEvTile]));
    } else {
        for (int i = 0; i + 1 < kEvCount; i++) HIPCHK(c, hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    }
    HIPCHK(c, hipEventElapsedTime(&ms[kEvCount - 1], ev[kEvStart], ev[kEvEmit]));
    return L3_OK;
}

// Waits for the batches queued by l3_encode_submit (they share the workspace with every other call on the context).
int drain_pending(l3_ctx* c) {
    for (auto& p : c->pend)
        if (p.in_use) HIPCHK(c, hipEventSynchronize(p.done));
    return L3_OK;
}

// Pieces too long for one workgroup were listed by slow_kernel (kGiantPending; the kernels behind it have skipped): the
// grid-wide merge (l3_giant.hip) in rounds until its state says done, its tokens to the images, then the tail of the
// pipeline.  Leaves the final status in c->h_status.  `flat`: 24 bytes per symbol (allocated here; the allocation-free
// entry point does not come this way).
int run_giants(l3_ctx* c, const Workspace& W, const EncodeArgs& A, hipStream_t stream, const Status& st) {
    const uint32_t n_giants = st.giant.count < kMaxGiants ? st.giant.count : kMaxGiants;
    HIPCHK(c, launch_giants_order(W, n_giants, stream));
    HIPCHK(c, hipMemcpyAsync(c->h_total, &W.gstate->n, 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    const uint64_t n_syms = *c->h_total & 0xFFFFFFFFull;   // (a sum of 2^31 and more is reported as 0xFFFFFFFF)
    c->counters[7] = n_syms;
    if (n_syms == 0) { c->err = "internal: empty giant list"; return L3_E_HIP; }
    if (n_syms >= (1ull << 31)) { c->err = "pieces of 2^31 and more symbols that no cut divides: split the batch"; return L3_E_INVALID_ARG; }
    HIPCHK(c, c->w_giant.ensure(24 * n_syms + 256));
    HIPCHK(c, launch_giants_init(c->T, W, n_giants, n_syms, c->w_giant.p, stream));
    // Every round applies at least the pair with the smallest key (l3_merge_dev.h: merge_rounds), so n_syms rounds are
    // always enough; ordinary text needs a few dozen, ranks that rise monotonically along a piece one round per merge
    // (slow, but the result stays the reference's).
    bool done = false;
    for (uint64_t it = 0; it < n_syms / 4 + 8 && !done; it++) {
        // rounds per look at the done flag: a finished state makes the remaining launches of a look empty (2 us each, seven
        // per round), a look costs a stream synchronisation; most pieces are through in 8 to 12 rounds
        HIPCHK(c, launch_giants_rounds(c->T, W, n_syms, c->w_giant.p, it < 3 ? 4 : it < 5 ? 8 : 16, stream));
        HIPCHK(c, hipMemcpyAsync(c->h_total, &W.gstate->done, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        done = (*c->h_total & 0xFFFFFFFFull) != 0;
    }
    if (!done) { c->err = "internal: grid-wide merge made no progress"; return L3_E_HIP; }
    HIPCHK(c, launch_giants_store(W, n_syms, c->w_giant.p, stream));
    if (c->profiling && !c->profiling_light) HIPCHK(c, hipEventRecord(c->ev[kEvHost], stream));   // up to here: the host-driven passes
    HIPCHK(c, hipMemsetAsync(&W.status->flags, 0, 8, stream));   // flags (kGiantPending) and generic_tiles
    HIPCHK(c, launch_encode_tail(W, A, stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[kEvEmit], stream));
    HIPCHK(c, launch_status_out(W, c->h_status, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    return L3_OK;
}

// Runs the pipeline on device-resident data.  On L3_OK or L3_E_OUTPUT_CAP *n_tok is the total.
// `W_out` (optional) receives the workspace used, so a caller can re-run the final scatter.
int encode_core(l3_ctx* c, const EncodeArgs& A, hipStream_t stream, uint64_t* n_tok, Workspace* W_out) {
    if (A.n_bytes >= (1ull << 40) || A.n_docs >= 0xFFFFFFFEull) {
        c->err = "batch too large: split it (n_bytes < 2^40, n_docs < 2^32-2)";
        return L3_E_INVALID_ARG;
    }
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    if (n_tiles > 0x7FFFFFFFull) { c->err = "batch too large: split it"; return L3_E_INVALID_ARG; }
    c->counters[4] = 0;
    if (c->n_pending) { int rc0 = drain_pending(c); if (rc0) return rc0; }
    for (int attempt = 0;; attempt++) {
        Workspace W{};
        int rc = ensure_workspace(c, A.n_bytes, A.n_docs, W);
        if (rc) return rc;
        // W.status is all zero here (or status_dirty says otherwise): cleared after every read-back below, so the clear
        // runs while the host is busy between two calls instead of in front of the kernels
        if ((rc = order_behind_last(c, stream))) return rc;
        if (c->status_dirty) { HIPCHK(c, hipMemsetAsync(W.status, 0, sizeof(Status), stream)); c->status_dirty = false; }
        HIPCHK(c, launch_encode(c->T, W, A, stream, c->profiling ? c->ev : nullptr, false, c->profiling_light ? kEvLight : kEvAll));
        HIPCHK(c, launch_status_out(W, c->h_status, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (c->h_status->flags & (kSeqPending | kGiantPending)) c->status_dirty = true;   // until a pass below has handed it out cleared
        c->counters[7] = 0;
        bool redo_tail = false;
        if ((c->h_status->flags & kSeqPending) && !(c->h_status->flags & (kFatalFlags | kRetryFlags))) {
            // long pre-tokens with invalid UTF-8 have been expanded: their windows now, in parallel
            HIPCHK(c, launch_slow_second_pass(c->T, W, A, stream));
            HIPCHK(c, launch_status_out(W, c->h_status, stream, 0));   // (its counts are still needed on the device)
            HIPCHK(c, hipStreamSynchronize(stream));
            redo_tail = true;
        }
        if ((c->h_status->flags & kGiantPending) && !(c->h_status->flags & (kFatalFlags | kRetryFlags))) {
            const Status st0 = *c->h_status;
            if ((rc = run_giants(c, W, A, stream, st0))) return rc;
            redo_tail = false;
            c->status_dirty = false;
        }
        if (redo_tail && !(c->h_status->flags & (kFatalFlags | kRetryFlags))) {
            if (c->profiling && !c->profiling_light) HIPCHK(c, hipEventRecord(c->ev[kEvHost], stream));
            HIPCHK(c, launch_encode_tail(W, A, stream));
            if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[kEvEmit], stream));
            HIPCHK(c, launch_status_out(W, c->h_status, stream));
            HIPCHK(c, hipStreamSynchronize(stream));
            c->status_dirty = false;
        }
        if (c->h_status->generic_tiles && !(c->h_status->flags & (kFatalFlags | kRetryFlags | kOvfOut))) {
            // tiles with empty documents or token floods: the generic scatter, launched only when the fast one asks
            HIPCHK(c, launch_emit_generic(W, A, stream));
            HIPCHK(c, hipStreamSynchronize(stream));
        }
        if ((c->h_status->big_runs | c->h_status->huge_runs) && !(c->h_status->flags & (kFatalFlags | kRetryFlags | kOvfOut))) {
            // pre-tokens of thousands of tokens: copied to the output by the whole grid
            HIPCHK(c, launch_copy_runs(W, A, (long long)c->h_status->big_runs, (long long)c->h_status->huge_runs, stream));
            HIPCHK(c, hipStreamSynchronize(stream));
        }
        if ((rc = mark_batch_end(c, stream))) return rc;
        const Status st = *c->h_status;
        if (st.flags & kBadOffsets) {
            c->err = "doc_off must start at 0, be ascending and end at n_bytes";
            return L3_E_INVALID_ARG;
        }
        if (st.flags & kTooLong) {
            c->err = "a single pre-token of 2 GiB or more is not supported";
            return L3_E_INVALID_ARG;
        }
        bool grow = false;
        if (st.flags & kOvfWin) { c->caps.win_cap = (uint64_t)st.slow_q.count + st.slow_q.count / 4 + 1024; grow = true; }
        if (st.flags & kOvfSlow) { c->caps.slow_cap = (uint64_t)st.slow_count + st.slow_count / 4 + 1024; grow = true; }
        uint64_t miss_total = 0, miss_max = 0, mpool_max = 0, mpool_total = 0;
        for (int i = 0; i < kMissShards; i++) {
            miss_total += st.shard[i].miss_count;
            if (st.shard[i].miss_count > miss_max) miss_max = st.shard[i].miss_count;
            if (st.shard[i].pool_used > mpool_max) mpool_max = st.shard[i].pool_used;
            mpool_total += st.shard[i].pool_used;
        }
        if (st.flags & kOvfMiss) { c->caps.miss_shard_cap = miss_max + miss_max / 4 + 256; grow = true; }
        if (st.flags & kOvfPool) {
            if (st.pool_used > c->caps.pool_cap) c->caps.pool_cap = st.pool_used + st.pool_used / 4 + 4096;
            if (mpool_max > c->caps.mpool_shard_cap) c->caps.mpool_shard_cap = mpool_max + mpool_max / 4 + 1024;
            grow = true;
        }
        if (st.flags & kOvfScratch) { c->caps.scratch_cap = st.scratch_used + st.scratch_used / 4 + (1ull << 20); grow = true; }
        if (grow) {
            if (attempt >= 8) { c->err = "workspace did not converge"; return L3_E_HIP; }
            c->counters[4]++;
            continue;
        }
        if (c->profiling) {
            if ((rc = event_times(c, c->ev, c->profiling_light, c->last_ms))) return rc;
            for (int i = 0; i < kEvCount; i++) c->sum_ms[i] += c->last_ms[i];
            c->sum_calls++;
        }
        c->counters[0] = n_tiles;
        c->counters[1] = st.slow_count;
        c->counters[2] = st.pool_used + mpool_total;
        c->counters[3] = st.scratch_used;
        c->counters[5] = miss_total;
        c->counters[6] = st.slow_q.count;
        c->slow_grid = (st.slow_count > 64 || st.slow_q.count > 32) ? 0u : (uint32_t)L3_IDLE_SLOW_GRID;   // (the slow list feeds the wave pass and slow_kernel)
        *n_tok = st.total_tokens;
        if (W_out) *W_out = W;
        return (st.flags & kOvfOut) ? L3_E_OUTPUT_CAP : L3_OK;
    }
}

int check_offsets_host(l3_ctx* c, const uint64_t* off, uint64_t n) {
    if (off[0] != 0) { c->err = "offsets must start at 0"; return L3_E_INVALID_ARG; }
    for (uint64_t i = 0; i < n; i++)
        if (off[i + 1] < off[i]) { c->err = "offsets must be ascending"; return L3_E_INVALID_ARG; }
    return L3_OK;
}

}  // namespace

static size_t noalloc_need(uint64_t n_bytes, uint64_t n_docs);

extern "C" {

int l3_create(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len, int device,
              l3_ctx** out) {
    return l3_create_ex(vocab_b64, vocab_len, merges_b64, merges_len, nullptr, 0, device, out);
}

int l3_create_ex(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len,
                 const char* const* special_tokens, size_t n_special, int device, l3_ctx** out) {
    if (!out) return L3_E_INVALID_ARG;
    *out = nullptr;
    if (!vocab_b64 || !merges_b64) { g_create_err = "null data pointer"; return L3_E_INVALID_ARG; }
    std::vector<std::string> specials;
    if (special_tokens) {   // WithSpecialTokens (options.go:17-37): "<|...|>", at least five bytes, no duplicates
        for (size_t i = 0; i < n_special; i++) {
            const std::string t = special_tokens[i] ? special_tokens[i] : "";
            bool ok = t.size() >= 5 && t.compare(0, 2, "<|") == 0 && t.compare(t.size() - 2, 2, "|>") == 0;
            for (size_t j = 0; ok && j < specials.size(); j++) ok = specials[j] != t;
            if (!ok) { g_create_err = "special token \"" + t + "\": must be <|...|> and must not repeat"; return L3_E_INVALID_ARG; }
            specials.push_back(t);
        }
    }
    l3_ctx* c = new l3_ctx();
    if (!build_tables(vocab_b64, vocab_len, merges_b64, merges_len, c->H, special_tokens ? &specials : nullptr)) {
        g_create_err = c->H.error;
        delete c;
        return L3_E_DATA;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_err = "no usable HIP device (there is no CPU fallback)";
        delete c;
        return L3_E_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is not gfx950 (MI355X): ") + prop.gcnArchName;
        delete c;
        return L3_E_NO_DEVICE;
    }
    c->device = device;
    auto fail = [&](int rc) {
        g_create_err = c->err;
        l3_destroy(c);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return fail(L3_E_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(L3_E_HIP); }
    for (int i = 0; i < kEvCount; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(L3_E_HIP); }
    c->have_events = true;
    if (hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) != hipSuccess) { c->ev_order = nullptr; c->err = "hipEventCreate failed"; return fail(L3_E_HIP); }
    if (hipHostMalloc((void**)&c->h_status, sizeof(Status), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_total, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        c->err = "hipHostMalloc failed";
        return fail(L3_E_HIP);
    }
    {   // the allocation-free entry point's workspace: its address on THIS device, asked while the device is current
        hipError_t e = scratch_address(device, &c->na_scratch);
        if (e != hipSuccess) { c->err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return fail(L3_E_HIP); }
        if (noalloc_need(L3_NOALLOC_MAX_BYTES, L3_NOALLOC_MAX_DOCS) > kNoAllocScratchBytes) {
            c->err = "internal: L3_NOALLOC_SCRATCH_BYTES does not cover L3_NOALLOC_MAX_BYTES / L3_NOALLOC_MAX_DOCS";
            return fail(L3_E_HIP);
        }
    }
    int rc = upload_tables(c);
    if (rc) return fail(rc);
    for (size_t i = 0; i < c->H.special_names.size(); i++) c->special_by_name[c->H.special_names[i]] = c->H.n_base + (int)i;
    *out = c;
    return L3_OK;
}

void l3_destroy(l3_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->t_cls1, &c->t_clsb, &c->t_clsd, &c->t_bytetok, &c->t_bytepair, &c->t_pairs, &c->t_vslots, &c->t_vblob,
                      &c->t_special, &c->t_tokoff, &c->t_tokbytes, &c->t_prefix2, &c->t_bloom, &c->t_rankm, &c->w_arena, &c->w_status, &c->w_giant, &c->s_text, &c->s_off, &c->s_out, &c->s_outoff, &c->s_kinds, &c->s_ids, &c->s_idoff,
                      &c->s_total, &c->s_decblk};
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < 2; i++) if (c->ev_dec[i]) (void)hipEventDestroy(c->ev_dec[i]);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_total) (void)hipHostFree(c->h_total);
    for (int i = 0; i < kEvCount; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);   // (a failed create leaves the rest null)
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    for (auto& p : c->pend) {
        if (!p.made) continue;
        if (p.in_use) (void)hipEventSynchronize(p.done);
        if (p.h_status) (void)hipHostFree(p.h_status);
        if (p.done) (void)hipEventDestroy(p.done);
        for (int i = 0; i < kEvCount; i++) if (p.ev[i]) (void)hipEventDestroy(p.ev[i]);
    }
    if (c->have_pipe) {
        for (int i = 0; i < 8; i++) { (void)hipEventDestroy(c->ev_in[i]); (void)hipEventDestroy(c->ev_piece[i]); }
        for (int i = 0; i < 2; i++) { (void)hipEventDestroy(c->ev_out[i]); c->p_out[i].release(); c->p_outoff[i].release(); }
        (void)hipStreamDestroy(c->stream_in);
        (void)hipStreamDestroy(c->stream_out);
    }
    if (c->hs_text) (void)hipHostFree(c->hs_text);
    if (c->hs_out) (void)hipHostFree(c->hs_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* l3_last_error(const l3_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// BOS / EOS are only added if the tokenizer has such a token: with a custom special-token list that lacks it
// GetSpecialTokenID fails and Encode adds nothing (tokenizer.go:225-229,258-262)
static uint32_t effective_flags(const l3_ctx* c, uint32_t flags) {
    if (c->H.bos_id < 0) flags &= ~L3_BOS;
    if (c->H.eos_id < 0) flags &= ~L3_EOS;
    return flags;
}
// flags are L3_BOS | L3_EOS | L3_NO_SPECIAL | L3_ONE_PRETOKEN
static_assert(L3_NO_SPECIAL == kNoSpecial && L3_ONE_PRETOKEN == kOnePretoken, "public flag bits are the kernels' flag bits");
static bool bad_flags(l3_ctx* c, uint32_t flags) {
    if (flags & ~(L3_BOS | L3_EOS | L3_NO_SPECIAL | L3_ONE_PRETOKEN)) { c->err = "unknown bits in flags"; return true; }
    return false;
}

int l3_encode_batch_device(l3_ctx* c, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off,
                           uint64_t n_docs, uint32_t flags, int32_t* d_out_tok, uint64_t out_cap,
                           uint64_t* d_out_off, uint64_t* n_tok, void* stream) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!d_doc_off || !d_out_off || !n_tok || (!d_text && n_bytes) || (!d_out_tok && out_cap)) {
        c->err = "null pointer argument";
        return L3_E_INVALID_ARG;
    }
    if (bad_flags(c, flags)) return L3_E_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    EncodeArgs A{};
    A.text = d_text; A.n_bytes = n_bytes; A.doc_off = d_doc_off; A.n_docs = n_docs; A.flags = effective_flags(c, flags);
    A.out_tok = d_out_tok; A.out_cap = out_cap; A.out_off = d_out_off; A.kinds = nullptr;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = encode_core(c, A, s, n_tok, nullptr);
    if (rc == L3_E_OUTPUT_CAP) c->err = "output buffer too small";
    return rc;
}

// ---- stream-ordered calls ------------------------------------------------------------------------------------
int l3_encode_submit(l3_ctx* c, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off, uint64_t n_docs,
                     uint32_t flags, int32_t* d_out_tok, uint64_t out_cap, uint64_t* d_out_off, void* stream, int* ticket) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!ticket || !d_doc_off || !d_out_off || (!d_text && n_bytes) || (!d_out_tok && out_cap)) {
        c->err = "null pointer argument";
        return L3_E_INVALID_ARG;
    }
    *ticket = -1;
    if (bad_flags(c, flags)) return L3_E_INVALID_ARG;
    if (n_bytes >= (1ull << 40) || n_docs >= 0xFFFFFFFEull || num_tiles(n_bytes) > 0x7FFFFFFFull) {
        c->err = "batch too large: split it (n_bytes < 2^40, n_docs < 2^32-2)";
        return L3_E_INVALID_ARG;
    }
    int slot = -1;
    for (int i = 0; i < L3_MAX_PENDING; i++)
        if (!c->pend[i].in_use) { slot = i; break; }
    if (slot < 0) { c->err = "L3_MAX_PENDING batches are pending already: l3_encode_wait for one first"; return L3_E_INVALID_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    l3_ctx::Pending& P = c->pend[slot];
    if (!P.made) {
        HIPCHK(c, hipHostMalloc((void**)&P.h_status, sizeof(Status), hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&P.done, hipEventDisableTiming));
        for (int i = 0; i < kEvCount; i++) HIPCHK(c, hipEventCreate(&P.ev[i]));
        P.made = true;
    }
    // the workspace is shared with the batches still in flight: it may only be moved (grown) once they are through
    {
        Workspace W0{};
        Caps K = c->caps;
        const Caps K0 = initial_caps(n_bytes);
        if (K.slow_cap < K0.slow_cap) K.slow_cap = K0.slow_cap;
        if (K.miss_shard_cap < K0.miss_shard_cap) K.miss_shard_cap = K0.miss_shard_cap;
        if (K.pool_cap < K0.pool_cap) K.pool_cap = K0.pool_cap;
        if (K.mpool_shard_cap < K0.mpool_shard_cap) K.mpool_shard_cap = K0.mpool_shard_cap;
        if (K.scratch_cap < K0.scratch_cap) K.scratch_cap = K0.scratch_cap;
        if (K.win_cap < K0.win_cap) K.win_cap = K0.win_cap;
        if (carve_workspace(nullptr, n_bytes, n_docs, K, nullptr, W0) > c->w_arena.cap && c->n_pending) {
            int rc0 = drain_pending(c);
            if (rc0) return rc0;
        }
    }
    Workspace W{};
    int rc = ensure_workspace(c, n_bytes, n_docs, W);
    if (rc) return rc;
    EncodeArgs A{};
    A.text = d_text; A.n_bytes = n_bytes; A.doc_off = d_doc_off; A.n_docs = n_docs; A.flags = effective_flags(c, flags);
    A.out_tok = d_out_tok; A.out_cap = out_cap; A.out_off = d_out_off; A.kinds = nullptr;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if ((rc = order_behind_last(c, s))) return rc;
    if (c->status_dirty) { HIPCHK(c, hipMemsetAsync(W.status, 0, sizeof(Status), s)); c->status_dirty = false; }
    P.profiled = c->profiling;
    P.light = c->profiling_light;
    HIPCHK(c, launch_encode(c->T, W, A, s, P.profiled ? P.ev : nullptr, false, P.light ? kEvLight : kEvAll));
    // nobody reads the status word in between, so the (rare) tiles the fast scatter flags are looked for every time; the same
    // launch hands the status word out and clears it (a batch that needs a host-driven pass is run again by its wait)
    HIPCHK(c, launch_finish(W, A, P.h_status, s));
    if ((rc = mark_batch_end(c, s))) return rc;
    HIPCHK(c, hipEventRecord(P.done, s));
    P.A = A; P.stream = s; P.in_use = true;
    c->n_pending++;
    *ticket = slot;
    return L3_OK;
}

int l3_encode_wait(l3_ctx* c, int ticket, uint64_t* n_tok) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!n_tok || ticket < 0 || ticket >= L3_MAX_PENDING || !c->pend[ticket].in_use) { c->err = "no such pending batch"; return L3_E_INVALID_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    l3_ctx::Pending& P = c->pend[ticket];
    HIPCHK(c, hipEventSynchronize(P.done));
    P.in_use = false;
    c->n_pending--;
    const Status st = *P.h_status;
    if (st.flags & kBadOffsets) { c->err = "doc_off must start at 0, be ascending and end at n_bytes"; return L3_E_INVALID_ARG; }
    if (st.flags & (kRetryFlags | kSeqPending | kGiantPending | kTooLong)) {
        // the batch needs a larger workspace or the host-driven passes (long pre-tokens with invalid UTF-8, giant pieces):
        // once more, synchronously (encode_core waits for the other pending batches first)
        // (on the context's own stream: the caller's stream handle was only good for the submit -- it may be gone by now --,
        // and the batch's event has been waited for, so no ordering is lost)
        int rc = encode_core(c, P.A, c->stream, n_tok, nullptr);
        if (rc == L3_E_OUTPUT_CAP) c->err = "output buffer too small";
        return rc;
    }
    if (P.profiled) {
        float ms[kEvCount];
        int rc = event_times(c, P.ev, P.light, ms);
        if (rc) return rc;
        for (int i = 0; i < kEvCount; i++) { c->last_ms[i] = ms[i]; c->sum_ms[i] += ms[i]; }
        c->sum_calls++;
    }
    uint64_t miss_total = 0, mpool_total = 0;
    for (int i = 0; i < kMissShards; i++) { miss_total += st.shard[i].miss_count; mpool_total += st.shard[i].pool_used; }
    c->counters[0] = num_tiles(P.A.n_bytes); c->counters[1] = st.slow_count; c->counters[2] = st.pool_used + mpool_total;
    c->counters[3] = st.scratch_used; c->counters[4] = 0; c->counters[5] = miss_total; c->counters[6] = st.slow_q.count; c->counters[7] = 0;
    c->slow_grid = (st.slow_count > 64 || st.slow_q.count > 32) ? 0u : (uint32_t)L3_IDLE_SLOW_GRID;   // (the slow list feeds the wave pass and slow_kernel)
    *n_tok = st.total_tokens;
    if (st.flags & kOvfOut) { c->err = "output buffer too small"; return L3_E_OUTPUT_CAP; }
    return L3_OK;
}

// list capacities of the allocation-free entry point: fixed, sized for its largest batch
static Caps noalloc_caps() {
    Caps K = initial_caps(L3_NOALLOC_MAX_BYTES);
    K.slow_cap *= 2;
    K.scratch_cap = 8ull << 20;
    return K;
}
// bytes of the array a batch of the largest permitted size takes (status word in front)
static size_t noalloc_need(uint64_t n_bytes, uint64_t n_docs) {
    Workspace W{};
    return ((sizeof(Status) + 255) & ~(size_t)255) + carve_workspace(nullptr, n_bytes, n_docs, noalloc_caps(), nullptr, W);
}

int l3_encode_batch_device_noalloc(l3_ctx* c, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off,
                                   uint64_t n_docs, uint32_t flags, int32_t* d_out_tok, uint64_t out_cap,
                                   uint64_t* d_out_off, uint64_t* n_tok, void* stream) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!d_doc_off || !d_out_off || !n_tok || (!d_text && n_bytes) || (!d_out_tok && out_cap)) {
        c->err = "null pointer argument";
        return L3_E_INVALID_ARG;
    }
    if (bad_flags(c, flags)) return L3_E_INVALID_ARG;
    if (n_bytes > (uint64_t)L3_NOALLOC_MAX_BYTES || n_docs > (uint64_t)L3_NOALLOC_MAX_DOCS) {
        c->err = "batch exceeds the static workspace of l3_encode_batch_device_noalloc (L3_NOALLOC_MAX_BYTES / "
                 "L3_NOALLOC_MAX_DOCS): split it or use l3_encode_batch_device";
        return L3_E_INVALID_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    // calls on one GPU share the array (also across contexts): one at a time
    std::lock_guard<std::mutex> dev_lk(g_scratch[c->device].mu);
    const Caps K = noalloc_caps();
    Workspace W{};
    unsigned char* const base = c->na_scratch;                     // the device address, not the array's name
    Status* const status = reinterpret_cast<Status*>(base);        // the status word lives at the front of the array
    const size_t status_bytes = (sizeof(Status) + 255) & ~(size_t)255;
    const size_t need = status_bytes + carve_workspace(base + status_bytes, n_bytes, n_docs, K, status, W);
    if (need > kNoAllocScratchBytes) { c->err = "internal: static workspace too small for its own limits"; return L3_E_INVALID_ARG; }
    W.bos_id = c->H.bos_id;
    W.eos_id = c->H.eos_id;
    W.allow_giants = 0;   // the grid-wide merge of very long cut-free pieces allocates: here one workgroup merges each
    EncodeArgs A{};
    A.text = d_text; A.n_bytes = n_bytes; A.doc_off = d_doc_off; A.n_docs = n_docs; A.flags = effective_flags(c, flags);
    A.out_tok = d_out_tok; A.out_cap = out_cap; A.out_off = d_out_off; A.kinds = nullptr;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // the array keeps the previous call's contents: the status word (counters, flags, queues) must start at zero; every
    // other part of the workspace is written before it is read
    HIPCHK(c, hipMemsetAsync(status, 0, sizeof(Status), s));
    HIPCHK(c, launch_encode(c->T, W, A, s, nullptr, false));
    HIPCHK(c, hipMemcpyAsync(c->h_status, status, sizeof(Status), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const Status st = *c->h_status;
    if (st.flags & kBadOffsets) {
        c->err = "doc_off must start at 0, be ascending and end at n_bytes";
        return L3_E_INVALID_ARG;
    }
    if (st.flags & kTooLong) {
        c->err = "a single pre-token of 2 GiB or more is not supported";
        return L3_E_INVALID_ARG;
    }
    if (st.flags & (kRetryFlags | kGiantPending | kSeqPending)) {
        c->err = "batch needs more workspace than the static array of l3_encode_batch_device_noalloc holds (unusually many "
                 "deferred pre-tokens): use l3_encode_batch_device";
        return L3_E_SCRATCH;
    }
    *n_tok = st.total_tokens;
    if (st.flags & kOvfOut) { c->err = "output buffer too small"; return L3_E_OUTPUT_CAP; }
    if (st.generic_tiles) {
        HIPCHK(c, launch_emit_generic(W, A, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    if (st.big_runs | st.huge_runs) {
        HIPCHK(c, launch_copy_runs(W, A, (long long)st.big_runs, (long long)st.huge_runs, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return L3_OK;
}

// page-locked host memory (hipHostMalloc / hipHostRegister), which the copy engines reach directly
static bool is_pinned(const void* p) {
    hipPointerAttribute_t attr;
    if (!p || hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

static hipError_t ensure_pinned(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap && *p) return hipSuccess;
    if (*p) { hipError_t e = hipHostFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return e; }
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e != hipSuccess) { *p = nullptr; return e; }
    *cap = want;
    return hipSuccess;
}

constexpr int kCopyThreads = 4;   // host threads per direction that move pageable caller memory to / from the staging buffers

// Host-fed batches of >= 16 MiB go through the GPU in up to eight pieces of whole documents: the upload of piece k + 1
// and the download of piece k - 1 run on streams of their own while piece k is encoded, so a call costs about the
// longer of the two PCIe directions instead of upload + kernels + download.  Results are identical to one pass:
// documents are independent, and the per-piece offsets are shifted on the host.
// The copy engines only overlap with page-locked host memory.  A caller's pageable text (pin_in false) is therefore copied
// piece by piece into a page-locked staging buffer of the context by kCopyThreads host threads, each piece uploaded as soon
// as it is staged; pageable output (pin_out false) is downloaded into staging and copied out by kCopyThreads threads as
// the pieces arrive.  (The runtime's own path for pageable memory stages through one thread: 6.0 ms per 64 MiB batch.)
static int encode_batch_pieces(l3_ctx* c, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                               int32_t* out_tok, uint64_t out_cap, uint64_t* out_off, uint64_t* n_tok, bool pin_in, bool pin_out) {
    const uint64_t n_bytes = doc_off[n_docs];
    if (!c->have_pipe) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_in, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_out, hipStreamNonBlocking));
        for (int i = 0; i < 8; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
        for (int i = 0; i < 8; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_piece[i], hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        c->have_pipe = true;
    }
    uint64_t want = n_bytes >> 23;   // pieces of about 8 MiB
    const int S = (int)(want < 2 ? 2 : (want > 8 ? 8 : want));
    uint64_t d_lo[9];
    d_lo[0] = 0;
    for (int k = 1; k < S; k++) {   // first document that starts at or after k / S of the bytes
        const uint64_t target = n_bytes / (uint64_t)S * (uint64_t)k;
        uint64_t lo = d_lo[k - 1], hi = n_docs;
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (doc_off[mid] < target) lo = mid + 1; else hi = mid; }
        d_lo[k] = lo;
    }
    d_lo[S] = n_docs;
    // offsets relative to each piece's first byte
    std::vector<uint64_t> rel((size_t)n_docs + (size_t)S + 1);
    uint64_t rel0[9];
    {
        size_t w = 0;
        for (int k = 0; k < S; k++) {
            rel0[k] = w;
            const uint64_t b0 = doc_off[d_lo[k]];
            for (uint64_t d = d_lo[k]; d <= d_lo[k + 1]; d++) rel[w++] = doc_off[d] - b0;
        }
        rel0[S] = w;
    }
    HIPCHK(c, c->s_text.ensure(n_bytes + 16));
    HIPCHK(c, c->s_off.ensure(rel.size() * 8));
    // threads of the two staging directions; joined on every way out of this function
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (std::thread& t : th) if (t.joinable()) t.join(); }
    } workers;
    std::atomic<int> staged[8];
    for (int k = 0; k < 8; k++) staged[k].store(0);
    std::atomic<int> out_ready{0};          // pieces whose download has been queued (ev_piece[k] recorded), or -1: give up
    uint64_t piece_tok[8] = {0}, piece_base[8] = {0};
    const uint8_t* up_src = text;
    if (!pin_in) {
        HIPCHK(c, ensure_pinned(&c->hs_text, &c->hs_text_cap, n_bytes + 16));
        uint8_t* dst = static_cast<uint8_t*>(c->hs_text);
        up_src = dst;
        for (int t = 0; t < kCopyThreads; t++)
            workers.th.emplace_back([=, &staged] {
                for (int k = 0; k < S; k++) {
                    const uint64_t b0 = doc_off[d_lo[k]], nb = doc_off[d_lo[k + 1]] - b0;
                    const uint64_t lo = nb * (uint64_t)t / kCopyThreads, hi = nb * (uint64_t)(t + 1) / kCopyThreads;
                    if (hi > lo) memcpy(dst + b0 + lo, text + b0 + lo, hi - lo);
                    staged[k].fetch_add(1, std::memory_order_release);
                }
            });
    }
    if (!pin_out) {
        // sized like the device buffers, from the text (the caller's capacity may be far larger than what comes out); a
        // piece that does not fit goes straight to the caller's memory
        uint64_t guess = n_bytes / 3 + 2 * n_docs + 1024;
        if (guess > out_cap) guess = out_cap;
        HIPCHK(c, ensure_pinned(&c->hs_out, &c->hs_out_cap, (size_t)guess * 4 + 16));
        const int32_t* src = static_cast<const int32_t*>(c->hs_out);
        const int device = c->device;
        hipEvent_t* evp = c->ev_piece;
        for (int t = 0; t < kCopyThreads; t++)
            workers.th.emplace_back([=, &out_ready, &piece_tok, &piece_base] {
                (void)hipSetDevice(device);
                for (int k = 0; k < S; k++) {
                    int r;
                    while ((r = out_ready.load(std::memory_order_acquire)) <= k && r >= 0) std::this_thread::yield();
                    if (r < 0) return;
                    const uint64_t n = piece_tok[k];
                    if (!n) continue;
                    if (hipEventSynchronize(evp[k]) != hipSuccess) return;
                    const uint64_t lo = n * (uint64_t)t / kCopyThreads, hi = n * (uint64_t)(t + 1) / kCopyThreads;
                    if (hi > lo) memcpy(out_tok + piece_base[k] + lo, src + piece_base[k] + lo, (hi - lo) * 4);
                }
            });
    }
    struct GiveUp {   // whatever ends the function early: the download threads must not wait for pieces that never come
        std::atomic<int>& r; bool armed = true;
        ~GiveUp() { if (armed) r.store(-1, std::memory_order_release); }
    } give_up{out_ready};
    for (int k = 0; k < S; k++) {
        const uint64_t b0 = doc_off[d_lo[k]], b1 = doc_off[d_lo[k + 1]];
        if (!pin_in) while (staged[k].load(std::memory_order_acquire) < kCopyThreads) std::this_thread::yield();
        if (b1 > b0) HIPCHK(c, hipMemcpyAsync(c->s_text.as<uint8_t>() + b0, up_src + b0, b1 - b0, hipMemcpyHostToDevice, c->stream_in));
        HIPCHK(c, hipMemcpyAsync(c->s_off.as<uint64_t>() + rel0[k], rel.data() + rel0[k], (rel0[k + 1] - rel0[k]) * 8,
                                 hipMemcpyHostToDevice, c->stream_in));
        HIPCHK(c, hipEventRecord(c->ev_in[k], c->stream_in));
    }
    uint64_t total = 0, base[9];
    bool overflow = false;
    uint64_t sum_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < S; k++) {
        const uint64_t nd = d_lo[k + 1] - d_lo[k];
        const uint64_t b0 = doc_off[d_lo[k]], nb = doc_off[d_lo[k + 1]] - b0;
        base[k] = total;
        if (nd == 0) continue;
        DevBuf& ob = c->p_out[k & 1];
        DevBuf& oo = c->p_outoff[k & 1];
        if (k >= 2) HIPCHK(c, hipEventSynchronize(c->ev_out[k & 1]));   // the download of piece k - 2 has left these buffers
        HIPCHK(c, ob.ensure((nb / 3 + 2 * nd + 1024) * 4));
        HIPCHK(c, oo.ensure((nd + 1) * 8));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in[k], 0));
        EncodeArgs A{};
        A.text = c->s_text.as<uint8_t>() + b0; A.n_bytes = nb; A.doc_off = c->s_off.as<uint64_t>() + rel0[k]; A.n_docs = nd;
        A.flags = effective_flags(c, flags); A.out_tok = ob.as<int32_t>(); A.out_cap = ob.cap / 4; A.out_off = oo.as<uint64_t>(); A.kinds = nullptr;
        Workspace W{};
        uint64_t tk = 0;
        int rc = encode_core(c, A, c->stream, &tk, &W);
        if (rc == L3_E_OUTPUT_CAP) {   // our guess was too small: enlarge and redo only the final scatter
            HIPCHK(c, ob.ensure(tk * 4));
            A.out_tok = ob.as<int32_t>();
            A.out_cap = ob.cap / 4;
            rc = rerun_emit(c, W, A, c->stream);
            if (rc == L3_OK && hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "hipStreamSynchronize failed"; rc = L3_E_HIP; }
        }
        if (rc) { (void)hipStreamSynchronize(c->stream_in); (void)hipStreamSynchronize(c->stream_out); return rc; }
        for (int i = 0; i < 8; i++) sum_counters[i] += c->counters[i];
        total += tk;
        if (total > out_cap) overflow = true;
        HIPCHK(c, hipMemcpyAsync(out_off + d_lo[k], oo.p, nd * 8, hipMemcpyDeviceToHost, c->stream_out));
        const bool to_stage = !pin_out && (base[k] + tk) * 4 <= c->hs_out_cap;
        int32_t* dl_dst = to_stage ? static_cast<int32_t*>(c->hs_out) : out_tok;
        if (!overflow && tk) HIPCHK(c, hipMemcpyAsync(dl_dst + base[k], ob.p, tk * 4, hipMemcpyDeviceToHost, c->stream_out));
        HIPCHK(c, hipEventRecord(c->ev_out[k & 1], c->stream_out));
        if (!pin_out) {
            HIPCHK(c, hipEventRecord(c->ev_piece[k], c->stream_out));
            piece_tok[k] = (overflow || !to_stage) ? 0 : tk;
            piece_base[k] = base[k];
            out_ready.store(k + 1, std::memory_order_release);
        }
    }
    if (!pin_out) out_ready.store(S, std::memory_order_release);   // (pieces without documents were skipped above)
    give_up.armed = false;
    HIPCHK(c, hipStreamSynchronize(c->stream_in));
    HIPCHK(c, hipStreamSynchronize(c->stream_out));
    for (std::thread& t : workers.th) if (t.joinable()) t.join();
    for (int k = 1; k < S; k++)
        if (base[k]) for (uint64_t d = d_lo[k]; d < d_lo[k + 1]; d++) out_off[d] += base[k];
    out_off[n_docs] = total;
    for (int i = 0; i < 8; i++) c->counters[i] = sum_counters[i];
    *n_tok = total;
    if (overflow) { c->err = "output buffer too small"; return L3_E_OUTPUT_CAP; }
    return L3_OK;
}

// One host-fed batch in two steps, so that a caller that shards a batch over several contexts (l3_multi_encode_batch) can
// place every shard's ids behind the ids of the shards before it without a second host copy: host_stage_encode uploads and
// encodes and leaves ids and offsets in the context's staging buffers, host_stage_download copies them to their final place.
// The context's mutex is held by the caller across both.
static int host_stage_encode(l3_ctx* c, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                             uint64_t* total) {
    const uint64_t n_bytes = doc_off[n_docs];
    hipStream_t s = c->stream;
    HIPCHK(c, c->s_text.ensure(n_bytes + 16));
    HIPCHK(c, c->s_off.ensure((n_docs + 1) * 8));
    HIPCHK(c, c->s_outoff.ensure((n_docs + 1) * 8));
    uint64_t guess = n_bytes / 3 + 2 * n_docs + 1024;
    HIPCHK(c, c->s_out.ensure(guess * 4));
    if (n_bytes) HIPCHK(c, hipMemcpyAsync(c->s_text.p, text, n_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->s_off.p, doc_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    EncodeArgs A{};
    A.text = c->s_text.as<uint8_t>(); A.n_bytes = n_bytes; A.doc_off = c->s_off.as<uint64_t>(); A.n_docs = n_docs;
    A.flags = effective_flags(c, flags); A.out_tok = c->s_out.as<int32_t>(); A.out_cap = c->s_out.cap / 4;
    A.out_off = c->s_outoff.as<uint64_t>(); A.kinds = nullptr;
    Workspace W{};
    int rc = encode_core(c, A, s, total, &W);
    if (rc == L3_E_OUTPUT_CAP) {  // our guess was too small: enlarge and redo only the final scatter
        HIPCHK(c, c->s_out.ensure(*total * 4));
        A.out_tok = c->s_out.as<int32_t>();
        A.out_cap = c->s_out.cap / 4;
        rc = rerun_emit(c, W, A, s);
    }
    return rc;
}

// n_off offsets (n_docs, or n_docs + 1 with the closing one) and, when `with_tokens`, the ids
static int host_stage_download(l3_ctx* c, uint64_t n_off, uint64_t total, bool with_tokens, int32_t* out_tok, uint64_t* out_off) {
    hipStream_t s = c->stream;
    if (n_off) HIPCHK(c, hipMemcpyAsync(out_off, c->s_outoff.p, n_off * 8, hipMemcpyDeviceToHost, s));
    if (with_tokens && total) HIPCHK(c, hipMemcpyAsync(out_tok, c->s_out.p, total * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return L3_OK;
}

int l3_encode_batch(l3_ctx* c, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                    int32_t* out_tok, uint64_t out_cap, uint64_t* out_off, uint64_t* n_tok) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (bad_flags(c, flags)) return L3_E_INVALID_ARG;
    if (!doc_off || !out_off || !n_tok || (!out_tok && out_cap)) { c->err = "null pointer argument"; return L3_E_INVALID_ARG; }
    int rc = check_offsets_host(c, doc_off, n_docs);
    if (rc) return rc;
    const uint64_t n_bytes = doc_off[n_docs];
    if (!text && n_bytes) { c->err = "null text"; return L3_E_INVALID_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    if (n_bytes >= (16ull << 20) && n_docs >= 16)
        return encode_batch_pieces(c, text, doc_off, n_docs, flags, out_tok, out_cap, out_off, n_tok, is_pinned(text),
                                   out_cap == 0 || is_pinned(out_tok));
    uint64_t total = 0;
    rc = host_stage_encode(c, text, doc_off, n_docs, flags, &total);
    if (rc) return rc;
    *n_tok = total;
    rc = host_stage_download(c, n_docs + 1, total, total <= out_cap, out_tok, out_off);
    if (rc) return rc;
    if (total > out_cap) { c->err = "output buffer too small"; return L3_E_OUTPUT_CAP; }
    return L3_OK;
}

int l3_pretokenize_batch(l3_ctx* c, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                         uint8_t* kinds) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (bad_flags(c, flags)) return L3_E_INVALID_ARG;
    if (!doc_off) { c->err = "null pointer argument"; return L3_E_INVALID_ARG; }
    int rc = check_offsets_host(c, doc_off, n_docs);
    if (rc) return rc;
    const uint64_t n_bytes = doc_off[n_docs];
    if ((!text || !kinds) && n_bytes) { c->err = "null pointer argument"; return L3_E_INVALID_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, c->s_text.ensure(n_bytes + 16));
    HIPCHK(c, c->s_off.ensure((n_docs + 1) * 8));
    HIPCHK(c, c->s_outoff.ensure((n_docs + 1) * 8));
    HIPCHK(c, c->s_kinds.ensure(n_bytes + 16));
    HIPCHK(c, c->s_out.ensure(16));
    if (n_bytes) HIPCHK(c, hipMemcpyAsync(c->s_text.p, text, n_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->s_off.p, doc_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    EncodeArgs A{};
    A.text = c->s_text.as<uint8_t>(); A.n_bytes = n_bytes; A.doc_off = c->s_off.as<uint64_t>(); A.n_docs = n_docs;
    A.flags = flags & L3_NO_SPECIAL; A.out_tok = c->s_out.as<int32_t>(); A.out_cap = 0;  // boundaries only: the scatter is skipped
    A.out_off = c->s_outoff.as<uint64_t>(); A.kinds = c->s_kinds.as<uint8_t>();
    uint64_t total = 0;
    rc = encode_core(c, A, s, &total, nullptr);
    if (rc != L3_OK && rc != L3_E_OUTPUT_CAP) return rc;
    if (n_bytes) HIPCHK(c, hipMemcpyAsync(kinds, c->s_kinds.p, n_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return L3_OK;
}

int l3_decode_batch(l3_ctx* c, const int32_t* ids, const uint64_t* id_off, uint64_t n_seqs, uint8_t* out_bytes,
                    uint64_t out_cap, uint64_t* out_off, uint64_t* n_bytes) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!id_off || !out_off || !n_bytes || (!out_bytes && out_cap)) { c->err = "null pointer argument"; return L3_E_INVALID_ARG; }
    int rc = check_offsets_host(c, id_off, n_seqs);
    if (rc) return rc;
    const uint64_t n_ids = id_off[n_seqs];
    if (!ids && n_ids) { c->err = "null ids"; return L3_E_INVALID_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, c->s_ids.ensure(n_ids * 4 + 16));
    HIPCHK(c, c->s_idoff.ensure((n_seqs + 1) * 8));
    HIPCHK(c, c->s_outoff.ensure((n_seqs + 1) * 8));
    HIPCHK(c, c->s_total.ensure(16));
    HIPCHK(c, c->s_decblk.ensure((decode_blocks(n_ids) + 1) * 8));
    if (c->profiling && !c->ev_dec[0]) for (int i = 0; i < 2; i++) HIPCHK(c, hipEventCreate(&c->ev_dec[i]));
    // every id decodes to at most kMaxTokenBytes bytes; start from a typical size and retry once
    uint64_t guess = n_ids * 6 + 1024;
    HIPCHK(c, c->s_kinds.ensure(guess));
    if (n_ids) HIPCHK(c, hipMemcpyAsync(c->s_ids.p, ids, n_ids * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->s_idoff.p, id_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, s));
    for (int attempt = 0; attempt < 2; attempt++) {
        if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_dec[0], s));
        HIPCHK(c, launch_decode(c->t_tokoff.as<uint32_t>(), c->t_tokbytes.as<uint8_t>(), c->H.n_tokens,
                                c->s_ids.as<int32_t>(), n_ids, c->s_idoff.as<uint64_t>(), n_seqs, c->s_decblk.as<uint64_t>(),
                                c->s_kinds.as<uint8_t>(), c->s_kinds.cap, c->s_outoff.as<uint64_t>(),
                                c->s_total.as<unsigned long long>(), s));
        if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_dec[1], s));
        HIPCHK(c, hipMemcpyAsync(c->h_total, c->s_total.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->profiling) HIPCHK(c, hipEventElapsedTime(&c->last_decode_ms, c->ev_dec[0], c->ev_dec[1]));
        if (*c->h_total <= c->s_kinds.cap) break;
        HIPCHK(c, c->s_kinds.ensure(*c->h_total));
    }
    const uint64_t total = *c->h_total;
    *n_bytes = total;
    HIPCHK(c, hipMemcpyAsync(out_off, c->s_outoff.p, (n_seqs + 1) * 8, hipMemcpyDeviceToHost, s));
    if (total > out_cap) {
        HIPCHK(c, hipStreamSynchronize(s));
        c->err = "output buffer too small";
        return L3_E_OUTPUT_CAP;
    }
    if (total) HIPCHK(c, hipMemcpyAsync(out_bytes, c->s_kinds.p, total, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return L3_OK;
}

int l3_special_token_id(const l3_ctx* c, const char* name, size_t name_len, int32_t* id) {
    if (!c || !name || !id) return L3_E_INVALID_ARG;
    auto it = c->special_by_name.find(std::string(name, name_len));
    if (it == c->special_by_name.end()) return L3_E_NOT_FOUND;
    *id = it->second;
    return L3_OK;
}

int l3_vocab_size(const l3_ctx* c) { return c ? c->H.n_tokens : L3_E_INVALID_ARG; }

int l3_host_alloc(l3_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return L3_E_INVALID_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void* p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    *out = p;
    return L3_OK;
}

int l3_host_free(l3_ctx* c, void* p) {
    if (!c) return L3_E_INVALID_ARG;
    if (p) HIPCHK(c, hipHostFree(p));
    return L3_OK;
}

int l3_token_bytes(const l3_ctx* c, int32_t id, uint8_t* out, size_t cap) {
    if (!c) return L3_E_INVALID_ARG;
    if (id < 0 || id >= c->H.n_tokens) return L3_E_NOT_FOUND;
    uint32_t a = c->H.tok_off[id], b = c->H.tok_off[id + 1];
    size_t l = b - a;
    if (out && cap) memcpy(out, c->H.tok_bytes.data() + a, l < cap ? l : cap);
    return (int)l;
}

int l3_set_profiling(l3_ctx* c, int enabled) {
    if (!c) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profiling = enabled != 0;
    c->profiling_light = enabled == 2;
    for (int i = 0; i < kEvCount; i++) c->sum_ms[i] = 0;
    c->sum_calls = 0;
    return L3_OK;
}

int l3_last_timings(l3_ctx* c, float* ms, int cap) {
    if (!c || !ms) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int n = cap < kEvCount ? cap : kEvCount;
    for (int i = 0; i < n; i++) ms[i] = c->last_ms[i];
    if (cap > kEvCount) ms[n++] = c->last_decode_ms;
    return n;
}

int l3_timing_totals(l3_ctx* c, double* ms, int cap, uint64_t* calls) {
    if (!c || !ms || !calls) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int n = cap < kEvCount ? cap : kEvCount;
    for (int i = 0; i < n; i++) ms[i] = c->sum_ms[i];
    *calls = c->sum_calls;
    return n;
}

int l3_last_counters(l3_ctx* c, uint64_t* out, int cap) {
    if (!c || !out) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int n = cap < 8 ? cap : 8;
    for (int i = 0; i < n; i++) out[i] = c->counters[i];
    return n;
}

// ---- several GPUs behind one handle ---------------------------------------------------------------------------------
// Documents are independent, so a batch is cut into one contiguous range of documents per context, balanced by bytes
// (the rule of tokenizer_amd/shard.py:10-21), every range is encoded on its own GPU by a host thread of its own, and the
// results are concatenated in range order.  Nothing crosses GPUs on the data path.

struct l3_multi {
    std::vector<l3_ctx*> ctx;
    std::mutex mu;   // one batch at a time per handle
    std::string err;
};

int l3_partition_by_bytes(const uint64_t* doc_off, uint64_t n_docs, int world, uint64_t* cuts) {
    if (!doc_off || !cuts || world < 1) return L3_E_INVALID_ARG;
    const uint64_t total = doc_off[n_docs];
    cuts[0] = 0;
    for (int r = 1; r < world; r++) {
        const uint64_t target = (uint64_t)((unsigned __int128)total * (unsigned)r / (unsigned)world);
        uint64_t lo = 0, hi = n_docs + 1;   // first index with doc_off[i] >= target
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (doc_off[mid] < target) lo = mid + 1; else hi = mid; }
        if (lo < cuts[r - 1]) lo = cuts[r - 1];
        cuts[r] = lo < n_docs ? lo : n_docs;
    }
    cuts[world] = n_docs;
    return L3_OK;
}

int l3_multi_create(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len,
                    const char* const* special_tokens, size_t n_special, const int* devices, int n_dev, l3_multi** out) {
    if (!out) return L3_E_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > 64) { g_create_err = "l3_multi_create: 1 to 64 devices"; return L3_E_INVALID_ARG; }
    std::unique_ptr<l3_multi> m(new l3_multi());
    for (int i = 0; i < n_dev; i++) {
        l3_ctx* c = nullptr;
        int rc = l3_create_ex(vocab_b64, vocab_len, merges_b64, merges_len, special_tokens, n_special, devices[i], &c);
        if (rc != L3_OK) {
            for (l3_ctx* x : m->ctx) l3_destroy(x);
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m.release();
    return L3_OK;
}

void l3_multi_destroy(l3_multi* m) {
    if (!m) return;
    for (l3_ctx* c : m->ctx) l3_destroy(c);
    delete m;
}

int l3_multi_size(const l3_multi* m) { return m ? (int)m->ctx.size() : L3_E_INVALID_ARG; }
l3_ctx* l3_multi_context(l3_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }
const char* l3_multi_last_error(const l3_multi* m) { return m ? m->err.c_str() : g_create_err.c_str(); }

int l3_multi_encode_batch(l3_multi* m, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                          int32_t* out_tok, uint64_t out_cap, uint64_t* out_off, uint64_t* n_tok) {
    if (!m) return L3_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (!doc_off || !out_off || !n_tok || (!out_tok && out_cap)) { m->err = "null pointer argument"; return L3_E_INVALID_ARG; }
    l3_ctx* c0 = m->ctx[0];
    {
        std::lock_guard<std::mutex> l0(c0->mu);
        if (bad_flags(c0, flags)) { m->err = c0->err; return L3_E_INVALID_ARG; }
        int rc = check_offsets_host(c0, doc_off, n_docs);
        if (rc) { m->err = c0->err; return rc; }
    }
    if (!text && doc_off[n_docs]) { m->err = "null text"; return L3_E_INVALID_ARG; }
    const int world = (int)m->ctx.size();
    std::vector<uint64_t> cuts((size_t)world + 1);
    l3_partition_by_bytes(doc_off, n_docs, world, cuts.data());

    // every shard: encode, meet the others (all totals known -> every base known), download to the final place
    std::vector<uint64_t> totals((size_t)world, 0);
    std::vector<int> rcs((size_t)world, L3_OK);      // encode step: written before the meeting point, read behind it
    std::vector<int> rcs_dl((size_t)world, L3_OK);   // download step: every thread its own entry, read after the joins
    std::mutex meet_mu;
    std::condition_variable meet_cv;
    int arrived = 0;
    bool fits = false, all_ok = false;               // decided by the last thread to arrive, under meet_mu
    auto shard = [&](int r) {
        l3_ctx* c = m->ctx[(size_t)r];
        const uint64_t d0 = cuts[(size_t)r], d1 = cuts[(size_t)r + 1], nd = d1 - d0, b0 = doc_off[d0];
        std::lock_guard<std::mutex> lc(c->mu);
        int rc = L3_OK;
        std::vector<uint64_t> rel((size_t)nd + 1);
        for (uint64_t d = 0; d <= nd; d++) rel[(size_t)d] = doc_off[d0 + d] - b0;
        if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; rc = L3_E_HIP; }
        if (rc == L3_OK && nd) rc = host_stage_encode(c, text + b0, rel.data(), nd, flags, &totals[(size_t)r]);
        rcs[(size_t)r] = rc;
        uint64_t base = 0;
        {
            std::unique_lock<std::mutex> lm(meet_mu);
            if (++arrived == world) {
                uint64_t sum = 0;
                bool ok = true;
                for (int i = 0; i < world; i++) { sum += totals[(size_t)i]; ok = ok && rcs[(size_t)i] == L3_OK; }
                all_ok = ok;
                fits = ok && sum <= out_cap;
                meet_cv.notify_all();
            } else {
                meet_cv.wait(lm, [&] { return arrived == world; });
            }
            for (int i = 0; i < r; i++) base += totals[(size_t)i];
            if (!all_ok || !nd) return;
        }
        rc = host_stage_download(c, nd, totals[(size_t)r], fits, out_tok + base, out_off + d0);
        rcs_dl[(size_t)r] = rc;
        if (rc == L3_OK && base) for (uint64_t d = d0; d < d1; d++) out_off[d] += base;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < world; r++) th.emplace_back(shard, r);
    shard(0);
    for (std::thread& t : th) t.join();
    uint64_t sum = 0;
    for (int r = 0; r < world; r++) {
        const int rc = rcs[(size_t)r] != L3_OK ? rcs[(size_t)r] : rcs_dl[(size_t)r];
        if (rc != L3_OK) { m->err = "device " + std::to_string(m->ctx[(size_t)r]->device) + ": " + m->ctx[(size_t)r]->err; return rc; }
        sum += totals[(size_t)r];
    }
    out_off[n_docs] = sum;
    *n_tok = sum;
    if (sum > out_cap) { m->err = "output buffer too small"; return L3_E_OUTPUT_CAP; }
    return L3_OK;
}

}  // extern "C"
