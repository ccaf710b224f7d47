// l3_kernels.h -- launch interface of the HIP kernels (l3_pipeline.hip and the stage units l3_tile / l3_merge / l3_slow / l3_emit / l3_decode .hip) used by the C ABI (l3_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "l3_common.h"

namespace l3 {

// ---- geometry of the tile kernel -----------------------------------------------------------------
#ifndef L3_TILE_BYTES
#define L3_TILE_BYTES 3840
#endif
constexpr int kTile = L3_TILE_BYTES;  // bytes of text owned by one workgroup
constexpr int kBack = 64;         // look-behind halo (special token 30 B + 4 runes 16 B + slack)
constexpr int kFwd = 192;         // look-ahead halo
constexpr int kRegion = kBack + kTile + kFwd;  // 4096: one 64-bit mask word per lane of a wave, 16 bytes per thread
constexpr int kEvalSlack = 36;    // positions closer than this to a soft region end cannot be decided
constexpr int kLaneMax = 64;      // longest pre-token (bytes) merged by a single lane in the tile kernel
constexpr int kTileCap = kTile;  // stream items one tile can emit: one per pre-token that starts in it
constexpr int kTileThreads = 256;
constexpr int kSlowThreads = 256;
#ifndef L3_SLOW_LDS
#define L3_SLOW_LDS 2048
#endif
constexpr int kSlowLds = L3_SLOW_LDS;    // symbols of a slow pre-token merged in LDS; longer ones use HBM scratch
#ifndef L3_MID_BYTES
#define L3_MID_BYTES 512
#endif
constexpr int kMidBytes = L3_MID_BYTES;  // slow entries of known length up to this are merged by ONE wave each
constexpr int kMidSyms = L3_SLOW_LDS / 4; // ... in LDS when they have at most this many symbols (four waves share the arena)
#ifndef L3_WIN
#define L3_WIN 1920
#endif
constexpr int kWin = L3_WIN;             // symbols of a long pre-token one window item owns
constexpr int kWinBack = 128;            // ... and looks back (a vocabulary entry has at most kMaxTokenBytes bytes)

// A stream item is a token id (>= 0) or the placeholder -2 - e of a deferred pre-token: e < kSlowBase
// is an index into Workspace::miss_in / miss_res (short: merged one lane each by merge_kernel), e >= kSlowBase is
// kSlowBase + an index into Workspace::slow (long / invalid UTF-8 / open-ended: slow_kernel).
constexpr uint32_t kNoSpecial = 4;  // EncodeArgs::flags: special-token names are ordinary text (bare pre-tokenizer)
constexpr uint32_t kOnePretoken = 8;  // EncodeArgs::flags: every document is one pre-token of raw bytes (BPE.EncodeBPE)
constexpr int32_t kDead = -1;
constexpr uint32_t kSlowBase = 1u << 30;
constexpr uint32_t kNoFirst = 0x7FFFFFFFu, kFirstInv = 0x80000000u;   // Workspace::tile_first

// A pre-token the tile kernel does not finish itself: it contains invalid UTF-8 (its symbols are
// not its bytes), is longer than kLaneMax without being a vocabulary entry, or ends past the halo.
struct SlowEntry {
    uint64_t start;      // global byte offset of its first byte
    uint64_t win0;       // written by the preparation pass: index of its first window item (Workspace::win); bit 63
                         // (kEntrySequential): it holds invalid UTF-8, ONE item works through all its windows
    uint32_t len;        // source bytes, 0 = unknown (ends beyond the tile's look-ahead; the preparation pass fills it in)
    uint32_t tile;       // owning tile
    uint32_t count;      // in (tile kernel): 1 = its bytes are known to be valid UTF-8; out: tokens produced
    uint32_t pad;
    uint64_t pool_off;   // first token in the slow pool
    uint64_t xoff;       // a long pre-token with invalid UTF-8: offset of its symbol array in the scratch
};
constexpr uint64_t kEntrySequential = 1ull << 63;
constexpr uint64_t kEntryExpanded = 1ull << 62;   // ... and its symbols are written out (SlowEntry::xoff): the second pass of slow_kernel
                                                  // takes its windows one item each
constexpr uint64_t kEntryWinMask = ~(kEntrySequential | kEntryExpanded);
// A long pre-token is worked on in windows of kWin symbols, one work item each (slow_kernel).
struct WinItem { uint32_t entry, k; };

// status word written by the kernels (device memory, zeroed per call)
enum : uint32_t {
    kOvfSlow = 1,      // more slow entries than slow_cap
    kOvfPool = 2,      // slow pool too small
    kOvfOut = 4,       // out_cap too small (total_tokens holds the need)
    kBadOffsets = 8,   // doc_off is not ascending / does not start at 0 / does not end at n_bytes
    kOvfScratch = 16,  // giant pre-token scratch too small
    kOvfMiss = 32,     // more short deferred pre-tokens than miss_cap
    kOvfWin = 64,      // more window items than win_cap
    kTooLong = 128,    // a single pre-token of 2 GiB or more
    kSeqPending = 512,    // long pre-tokens with invalid UTF-8 were only expanded: the host runs slow_kernel's second pass over their windows
    kGiantPending = 256,  // pieces were left to the grid-wide merge (Workspace::giants): the host runs it, then the tail of the pipeline
};
constexpr uint32_t kRetryFlags = kOvfSlow | kOvfPool | kOvfScratch | kOvfMiss | kOvfWin;   // the host grows a list and runs the batch again
constexpr uint32_t kFatalFlags = kBadOffsets | kTooLong;
struct Status {
    uint32_t slow_count;        // entries appended so far (may exceed capacity: overflow)
    uint32_t flags;             // kOvf* / kBadOffsets
    uint32_t generic_tiles;     // tiles the fast emit kernel left to the generic one (the host launches it only then)
    uint32_t done_blocks;       // workgroups of finish_kernel that are through (the last one hands the word to the host)
    unsigned long long pool_used;        // tokens reserved in the slow pool (may exceed capacity)
    unsigned long long total_tokens;     // tokens of the whole batch incl. BOS/EOS (written by the wave of the emit kernel that has tile 0)
    unsigned long long scratch_used;     // bytes reserved in the giant-pre-token scratch (may exceed capacity)
    uint32_t big_runs, huge_runs;        // runs of tokens the emit kernel left to copy_runs, by size class (may exceed kMaxBigRuns /
                                         // kMaxHugeRuns: the surplus was copied by the wave that met it)
    // Per shard (shard = tile % kMissShards), each on a cache line of its own: atomics that return a value and
    // share an address (or a line) queue up behind each other, ~10^4 of them cost tens of microseconds.
    struct alignas(128) Shard {
        uint32_t miss_count;            // short entries appended so far (may exceed the shard capacity: overflow)
        uint32_t pad;
        unsigned long long pool_used;   // tokens reserved in the shard's part of the pool (may exceed mpool_shard_cap)
    } shard[64];
    // Work queues, on cache lines of their own.  mid_q: over the slow list; the waves of the pass inside merge_kernel's
    // launch take it a few entries at a time, do the entries of known, moderate length themselves and prepare the others
    // (end, room in the pool, window items).  slow_q: over the window items (count of them); slow_kernel's workgroups
    // take those one at a time.
    struct alignas(128) Queue { uint32_t next, count; } slow_q, mid_q, seq_q, exp_q;
    // pieces left to the grid-wide merge: their number (may exceed kMaxGiants: the surplus was merged by one workgroup each)
    struct alignas(128) GiantCount { uint32_t count, pad; } giant;
};
// A piece of a long pre-token that no cut divides and that is too long for one workgroup (one character repeated a million
// times, a DNA string): merged by the whole GPU, all such pieces of a batch as ONE flat sequence (giant_* kernels).
constexpr uint32_t kGiantMin = 32768;    // symbols from which a piece takes that path
constexpr uint32_t kMaxGiants = 1024;
struct Giant {
    const uint8_t* sp;   // symbol bytes of its pre-token (the text, or the expanded copy in the scratch)
    uint64_t q;          // its first symbol
    uint32_t n;          // its symbols; 0: open end, looked up in win_first by giant_order_kernel
    uint32_t entry;      // slow entry of its pre-token
    uint32_t k, pad;     // window that owns it
    uint64_t flat;       // its place in the flat sequence (giant_order_kernel)
    int32_t* img;        // image of its pre-token (slot of symbol i: img[i])
    uint32_t* live;      // window counters of its pre-token
};
#ifndef L3_GIANT_GRID
#define L3_GIANT_GRID 1024
#endif
constexpr int kGiantGrid = L3_GIANT_GRID, kGiantThreads = 256, kGiantWaves = kGiantGrid * kGiantThreads / 64;
struct GiantSeg { uint32_t cnt, open; };
struct GiantState {
    uint32_t n, n_next, round, done;
    unsigned long long best, best_next, theta;
    uint32_t n_giants, pad;
    uint32_t order[kMaxGiants];          // giants sorted by their place in the flat sequence
    uint32_t wave_live[kGiantWaves];
    GiantSeg wave_seg[kGiantWaves];
};
// A pre-token of thousands of tokens (a long run of ideographs, a megabyte of one letter) is ONE item of ONE tile: its tokens
// are not copied to the output by the wave that emits that tile but listed, and copied by the whole grid (copy_runs).
constexpr uint32_t kBigRun = 4096;       // tokens from which a run is listed
constexpr uint32_t kMaxBigRuns = 16384;
constexpr uint32_t kHugeRun = 65536;     // tokens from which a run is copied in pieces by all workgroups (else: one workgroup)
constexpr uint32_t kMaxHugeRuns = 1024;
struct BigRun { uint64_t src, dst, n; };  // pool offset, output index, tokens
constexpr int kMissShards = 64;
// a short deferred pre-token (input of merge_kernel).  Its result is one word of miss_res: bits 0..7 = n tokens;
// n <= 3: the tokens themselves in bits 8.., 17 bits each; n > 3: bits 8.. = offset of the tokens in the pool
struct alignas(16) MissIn {
    uint32_t tile;    // owning tile
    uint32_t len;     // bytes (<= kLaneMax), valid UTF-8
    uint64_t start;   // global byte offset
    uint32_t w[4];    // its first 16 bytes as little-endian words (bytes past len: unspecified)
};
constexpr uint32_t kMissInline = 3;   // results of up to this many tokens live in the miss_res word itself

// written by block_totals_kernel once a tile's token count is final
struct alignas(32) TileRec {
    uint64_t pre;        // tokens of the tiles of the same block of 1024 that precede the tile
    uint32_t n_items;    // items in the tile's stream
    uint32_t n_extra;    // tokens beyond one per item (deferred pre-tokens)
    uint32_t d_lo, d_hi; // documents that start in the tile: [d_lo, d_hi)
    uint32_t pad[2];
};

struct Workspace {
    int32_t bos_id, eos_id;  // ids of <|begin_of_text|> / <|end_of_text|> in this vocabulary (tokenizer.go:226,259: looked up, not constants)
    int32_t* stream;        // [n_tiles * kTileCap] per-tile token stream incl. slow placeholders
    uint32_t* tile_items;   // [n_tiles] items in the tile's stream
    uint32_t* tile_extra;   // [n_tiles] sum of (tokens - 1) over the tile's deferred pre-tokens (atomic adds)
    uint32_t* tile_digits;  // [n_tiles] 0, or 1 + the number of decimal-digit runes right before the tile start (a tile that had to
                            // count them leaves the figure for the tiles after it: see digits_before_global)
    uint32_t* tile_first;   // [n_tiles] offset (from the tile start) of the first pre-token start the tile owns, kNoFirst: none;
                            // bit 31 (kFirstInv): among the bytes the tile owns one is not valid UTF-8
    uint64_t* tile_base;    // [ceil(n_tiles / 1024)] exclusive scan of the token totals of blocks of 1024 tiles
    TileRec* tile_rec;      // [n_tiles] what the emit kernels need to know about a tile, in one 32-byte load
    uint32_t* tile_flag;    // [n_tiles] 1 = the fast emit kernel left the tile to the generic one
    uint32_t* tile_doc0;    // [n_tiles + 1] first document whose start is >= the tile start
    uint32_t* doc_item;     // [n_docs + 1] stream index (inside its tile) of the document's first item
    SlowEntry* slow;        // [slow_cap]
    MissIn* miss_in;        // [kMissShards * miss_shard_cap]
    uint64_t* miss_res;     // [kMissShards * miss_shard_cap]
    int32_t* pool;          // [kMissShards * mpool_shard_cap + pool_cap]: results of more than kMissInline tokens of the
                            // merge kernel, one part per shard, then the tokens of the slow pre-tokens
    uint8_t* scratch;       // [scratch_cap] symbol arrays of pre-tokens too long for LDS
    uint32_t slow_grid;     // workgroups of slow_kernel (0: the default); the host keeps it small while batches have no long pre-tokens
    uint32_t allow_giants;  // 0: no piece is left to the grid-wide merge (the allocation-free entry point: that path allocates)
    Giant* giants;          // [kMaxGiants]
    GiantState* gstate;
    BigRun* runs;           // [kMaxBigRuns + kMaxHugeRuns]: the big runs, behind them the huge ones
    WinItem* win;           // [win_cap] window items of the long pre-tokens
    uint32_t* win_live;     // [win_cap] tokens in the image slots of each window's range
    uint32_t* win_cnt;      // [win_cap] symbols of the chunk of bytes each window item of a pre-token with invalid UTF-8 expands
    uint32_t* win_first;    // [win_cap] symbol position of the first cut inside each window, 0xFFFFFFFF: none
    Status* status;
    uint64_t slow_cap, miss_shard_cap, mpool_shard_cap, pool_cap, scratch_cap, win_cap;
};

struct EncodeArgs {
    const uint8_t* text;      // device
    uint64_t n_bytes;
    const uint64_t* doc_off;  // device, n_docs + 1 ascending offsets, doc_off[0] = 0, doc_off[n_docs] = n_bytes
    uint64_t n_docs;
    uint32_t flags;           // bit0 BOS, bit1 EOS, bit2 kNoSpecial, bit3 kOnePretoken
    int32_t* out_tok;         // device
    uint64_t out_cap;
    uint64_t* out_off;        // device, n_docs + 1
    uint8_t* kinds;           // optional (pre-tokenize API): one byte per text byte, 1 = pre-token start,
                              // 2 = special-token start, 0 otherwise
};

__host__ __device__ inline uint64_t num_tiles(uint64_t n_bytes) { return n_bytes == 0 ? 1 : (n_bytes + kTile - 1) / kTile; }

// kEvHost: right behind kEvScan in an ordinary batch; a batch with host-driven passes (long pre-tokens with invalid UTF-8, the
// grid-wide merge) records it again in front of the tail of the pipeline it then runs, so that the interval kEvScan -> kEvHost
// holds those passes and kEvHost -> kEvEmit the final scatter alone.
enum { kEvStart = 0, kEvDocs, kEvTile, kEvMerge, kEvSlow, kEvScan, kEvHost, kEvEmit, kEvCount };

// Enqueue the encode pipeline on `stream`; nothing is synchronised.  `ev` (optional, kEvCount
// events) are recorded between the kernels.  `emit_only` re-runs just the final scatter (after
// the caller enlarged the output buffer).
// `ev_mask`: which of the events are recorded (bit i = event i); kEvLight = only what the roofline figures need.
constexpr uint32_t kEvAll = 0xFFFFFFFFu, kEvLight = (1u << kEvStart) | (1u << kEvDocs) | (1u << kEvTile) | (1u << kEvEmit);
hipError_t launch_encode(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream,
                         hipEvent_t* ev, bool emit_only, uint32_t ev_mask = kEvAll);
// The stages launch_encode strings together, each defined beside its kernels (l3_tile.hip, l3_merge.hip, l3_slow.hip,
// l3_emit.hip).  n_slow_blocks / grid: workgroups of the passes over the slow list; pass: 0 first, 1 windows, 2 expansion.
#ifndef L3_SLOW_GRID
#define L3_SLOW_GRID 768   // three workgroups per CU: what slow_kernel's registers and LDS allow (l3_slow.hip, L3_SLOW_OCC)
#endif
hipError_t launch_stage_docs(const Workspace& W, const EncodeArgs& A, hipStream_t stream);
hipError_t launch_stage_tile(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream);
hipError_t launch_stage_merge(const Tables& T, const Workspace& W, const EncodeArgs& A, uint32_t n_slow_blocks, hipStream_t stream);
hipError_t launch_stage_slow(const Tables& T, const Workspace& W, const EncodeArgs& A, uint32_t grid, int pass, hipStream_t stream);
hipError_t launch_stage_totals(const Workspace& W, const EncodeArgs& A, hipStream_t stream);
hipError_t launch_stage_emit(const Workspace& W, const EncodeArgs& A, hipStream_t stream);
// The status word to `host_dst` (page-locked host memory); clear: 0 = it is left as it is, 1 = zeroed unless kSeqPending or
// kGiantPending is set (the host-driven passes still need it), 2 = zeroed.
hipError_t launch_status_out(const Workspace& W, Status* host_dst, hipStream_t stream, int clear = 1);
// Second pass of slow_kernel (clears kSeqPending first): the windows of the long pre-tokens with invalid UTF-8 that the
// first pass expanded.
hipError_t launch_slow_second_pass(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream);
// The tail of the pipeline again (block totals + finish, scan, emit): after the grid-wide merge of the giant pieces.
hipError_t launch_encode_tail(const Workspace& W, const EncodeArgs& A, hipStream_t stream);
// Grid-wide merge of the pieces listed in W.giants (n_giants of them, n_syms symbols in all); `flat` = 24 * n_syms bytes.
// launch_giants_order: ends of the pieces listed with an open end, their places in the flat sequence; GiantState::n = n_syms.
hipError_t launch_giants_order(const Workspace& W, uint32_t n_giants, hipStream_t stream);
hipError_t launch_giants_init(const Tables& T, const Workspace& W, uint32_t n_giants, uint64_t n_syms, void* flat, hipStream_t stream);
hipError_t launch_giants_rounds(const Tables& T, const Workspace& W, uint64_t n_syms, void* flat, int rounds, hipStream_t stream);
hipError_t launch_giants_store(const Workspace& W, uint64_t n_syms, void* flat, hipStream_t stream);
// The second half of the final scatter, for the tiles the fast emit kernel flagged (Status::generic_tiles > 0).
hipError_t launch_emit_generic(const Workspace& W, const EncodeArgs& A, hipStream_t stream);
// launch_emit_generic + launch_status_out(clear = 2) as ONE launch (the end of a stream-ordered batch).
hipError_t launch_finish(const Workspace& W, const EncodeArgs& A, Status* host_dst, hipStream_t stream);
// Copies the runs of tokens the emit kernel listed (Status::big_runs, huge_runs) to the output.  n_big, n_huge: their numbers as
// the host read them from the status word, or -1: the kernel reads them from the status word on the device.
hipError_t launch_copy_runs(const Workspace& W, const EncodeArgs& A, long long n_big, long long n_huge, hipStream_t stream);
// Decode: ids -> bytes (tokenizer.go:339-354) in blocks of ids, whatever the sequences are: block sums, their scan, then the
// bytes and the sequence offsets.  block_base: decode_blocks(n_ids) + 1 words of workspace.
uint64_t decode_blocks(uint64_t n_ids);
hipError_t launch_decode(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens, const int32_t* ids, uint64_t n_ids,
                         const uint64_t* id_off, uint64_t n_seqs, uint64_t* block_base, uint8_t* out, uint64_t out_cap,
                         uint64_t* out_off, unsigned long long* total, hipStream_t stream);

}  // namespace l3
