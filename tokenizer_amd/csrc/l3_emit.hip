// l3_emit.hip -- the output stages: block_totals_kernel (+ finish_window), emit_kernel, emit_generic_kernel, copy_runs,
// finish_kernel, status_out_kernel.  Overview: l3_pipeline.hip.
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_dev.h"

namespace l3 {

namespace {

// The tokens of window item `w` from the pre-token's image (slow_kernel) to their final place, the windows of a pre-token in
// order: by ONE WAVE, 64 consecutive slots at a time -- a ballot gives every live slot its place, no LDS, no barrier, the
// loads of several rows in flight together.  (Round 2 gave a window to a workgroup of 1,024 threads: five dependent loads and
// five barriers per window with two workgroups resident per CU; 16 independent waves per workgroup hide that chain.)
constexpr int kFinishRows = kWin / 64;
constexpr int kFinishU = 6;   // rows loaded together
static_assert(kWin % 64 == 0, "a window is a whole number of rows of 64 slots");
__device__ void finish_window(const Workspace& W, uint32_t w) {
    const uint32_t lane = (uint32_t)lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const WinItem it = W.win[w];
    const uint64_t win0 = W.slow[it.entry].win0;
    const uint32_t* live = W.win_live + (win0 & kEntryWinMask);
    const uint32_t mine = live[it.k];
    if (mine == 0) return;   // (uniform in the wave)
    const uint32_t len = W.slow[it.entry].len, L = W.slow[it.entry].pad;
    const uint64_t slots = (win0 & kEntrySequential) ? 3ull * len : (uint64_t)len;
    int32_t* const fin = W.pool + W.slow[it.entry].pool_off;
    const int32_t* const img = fin + slots;
    // tokens of the windows before this one
    unsigned long long before = 0;
    for (uint32_t j = lane; j < it.k; j += 64) before += live[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    const uint64_t lo = (uint64_t)it.k * (uint64_t)kWin;
    uint64_t o = before;
#pragma unroll 1
    for (int r0 = 0; r0 < kFinishRows; r0 += kFinishU) {
        int32_t v[kFinishU];
#pragma unroll
        for (int u = 0; u < kFinishU; u++) {
            const uint64_t p = lo + (uint64_t)((r0 + u) * 64) + lane;
            v[u] = (r0 + u < kFinishRows && p < (uint64_t)L) ? img[p] : -1;
        }
#pragma unroll
        for (int u = 0; u < kFinishU; u++) {
            const unsigned long long m = __ballot(v[u] >= 0);
            if (v[u] >= 0) fin[o + (uint64_t)__popcll(m & lt)] = v[u];
            o += (uint64_t)__popcll(m);
        }
    }
}

// ---- per-tile output bases: block totals, scan of the block totals -----------------------------------------
constexpr int kScanBlock = 1024;  // tiles per block total
// Workgroups [0, n_blocks): the totals; the workgroups behind them move the tokens of the long pre-tokens from their
// images to their final place (same launch: both only need what the kernels before have left).
// (second launch bound: waves per SIMD.  8 = at most 64 VGPRs, so that TWO workgroups of 1,024 threads fit a CU: the window
//  finish is a chain of short phases per window and wants the residency)
#ifndef L3_TOTALS_OCC
#define L3_TOTALS_OCC 8
#endif
__global__ __launch_bounds__(kScanBlock, L3_TOTALS_OCC) void block_totals_kernel(Workspace W, uint64_t n_tiles, uint64_t n_docs, uint32_t n_blocks) {
    __shared__ unsigned long long wsum[kScanBlock / 64];
    if (W.status->flags & (kFatalFlags | kGiantPending | kSeqPending)) return;
    if (blockIdx.x >= n_blocks) {
        if (W.status->flags & kRetryFlags) return;
        const uint32_t n_items = W.status->slow_q.count;
        const uint32_t n_waves = (gridDim.x - n_blocks) * (uint32_t)(kScanBlock / 64);
        for (uint32_t w = (blockIdx.x - n_blocks) * (uint32_t)(kScanBlock / 64) + (threadIdx.x >> 6); w < n_items; w += n_waves) finish_window(W, w);
        return;
    }
    const uint64_t t = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;   // one tile per thread
    const uint32_t n_items = t < n_tiles ? W.tile_items[t] : 0u, n_extra = t < n_tiles ? W.tile_extra[t] : 0u;
    const uint32_t d_lo = t < n_tiles ? W.tile_doc0[t] : 0u;
    const uint32_t d_hi = t + 1 < n_tiles ? W.tile_doc0[t + 1] : (uint32_t)(n_docs + 1);
    const unsigned long long v = (unsigned long long)n_items + n_extra;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = __shfl_up(inc, d, 64);
        if (lane_id() >= d) inc += o;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; i++) {
        const unsigned long long x = wsum[i];
        if (i < (int)(threadIdx.x >> 6)) pre += x;
        tot += x;
    }
    if (t < n_tiles) {
        TileRec rec;
        rec.pre = pre + inc - v; rec.n_items = n_items; rec.n_extra = n_extra; rec.d_lo = d_lo; rec.d_hi = d_hi;
        rec.pad[0] = rec.pad[1] = 0;
        W.tile_rec[t] = rec;
    }
    if (threadIdx.x == 0) W.tile_base[blockIdx.x] = tot;
}
// Tokens of the blocks of kScanBlock tiles in front of block `blk`, and of all blocks (block_totals_kernel leaves the totals
// in W.tile_base): every wave that needs them adds them up itself -- 18 values for a 64 MiB batch --, which costs less than
// the launch of a kernel that scans them.  All 64 lanes must be active; the results are uniform.
__device__ __forceinline__ void block_prefix(const Workspace& W, uint32_t blk, uint32_t n_blocks, unsigned long long* before,
                                             unsigned long long* all) {
    unsigned long long pre = 0, tot = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 64) {   // (uniform)
        const uint32_t b = b0 + (uint32_t)lane_id();
        const unsigned long long v = b < n_blocks ? W.tile_base[b] : 0ull;
        tot += v;
        if (b < blk) pre += v;
    }
    pre = wave_iscan64(pre);
    tot = wave_iscan64(tot);
    *before = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pre >> 32), 63) << 32) |
              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pre, 63);
    *all = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tot >> 32), 63) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tot, 63);
}

// ---- final scatter -------------------------------------------------------------------------------------
constexpr int kEmitQueue = 128;
// Token count of a deferred pre-token and where its tokens are: *inl = they are the 17-bit fields of *loc
// (merge_kernel results of up to kMissInline tokens), else *loc = their offset in the pool
__device__ __forceinline__ void deferred_result(const Workspace& W, int32_t item, uint64_t* loc, uint32_t* count, bool* inl) {
    const uint32_t e = (uint32_t)(-2 - item);
    if (e >= kSlowBase) {
        *loc = W.slow[e - kSlowBase].pool_off;
        *count = W.slow[e - kSlowBase].count;
        *inl = false;
    } else {
        const uint64_t r = W.miss_res[e];
        *loc = r >> 8;
        *count = (uint32_t)(r & 0xFFu);
        *inl = *count <= kMissInline;
    }
}
__device__ __forceinline__ int32_t deferred_token(const Workspace& W, uint64_t loc, bool inl, uint32_t k) {
    return inl ? (int32_t)((loc >> (17u * k)) & 0x1FFFFu) : W.pool[loc + k];
}
// exclusive scan of two 32-bit counters packed in one 64-bit word; tmp: >= 4 words.  Two barriers.
__device__ __forceinline__ unsigned long long block_escan64(unsigned long long v, unsigned long long* tmp, unsigned long long* total) {
    const unsigned long long inc = wave_iscan64(v);
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kTileThreads / 64; i++) {
        unsigned long long x = tmp[i];
        if (i < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
// ---- final scatter, fast path ---------------------------------------------------------------------------
// One WAVE per tile, no LDS, no workgroup barrier: the 64 lanes take 64 consecutive items of the tile's stream at a time
// (four such steps are in flight together), so the tokens of a step leave as one coalesced store wherever every item is one
// token, and the prefix sums a step needs are wave scans by DPP.  The document starts of the tile (a bit per item) live in
// registers: lane w holds word w of the bit map and the number of starts before it.  Tiles with several documents on one
// item (empty documents) or more than kEmitDocs documents go to the generic kernel.
constexpr int kEmitDocs = 256;
constexpr int kEmitSteps = 4;                       // steps of 64 items in flight
constexpr int kEmitRound = 64 * kEmitSteps;
static_assert((kTileCap + 1 + 63) / 64 <= 64, "one lane per word of the document-start bit map");
#ifndef L3_EMIT_OCC
#define L3_EMIT_OCC 8
#endif
__global__ __launch_bounds__(kTileThreads, L3_EMIT_OCC) void emit_kernel(Workspace W, EncodeArgs A, uint32_t n_tiles) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t tile = (uint64_t)blockIdx.x * (kTileThreads / 64) + (uint32_t)wave_id();
    if (tile >= n_tiles) return;   // (whole waves: nothing below synchronises across waves)
    // every parameter load is requested before the first branch: one memory round trip, not three
    const TileRec rec = W.tile_rec[tile];
    const uint32_t fl = W.status->flags;
    const uint32_t n_blocks = (n_tiles + (uint32_t)kScanBlock - 1u) / (uint32_t)kScanBlock;
    unsigned long long blk_base, total_tokens;
    block_prefix(W, (uint32_t)(tile / kScanBlock), n_blocks, &blk_base, &total_tokens);
    if (fl & (kFatalFlags | kRetryFlags | kGiantPending | kSeqPending)) return;
    total_tokens += (unsigned long long)A.n_docs * ((A.flags & 1u) + ((A.flags >> 1) & 1u));
    if (tile == 0 && lane == 0) {   // tokens of the whole batch incl. BOS / EOS: what the caller's buffer must hold
        W.status->total_tokens = total_tokens;
        if (total_tokens > A.out_cap) atomicOr(&W.status->flags, kOvfOut);
    }
    if (total_tokens > A.out_cap) return;
    const uint32_t n_items = rec.n_items;
    const uint64_t d_lo = rec.d_lo, d_hi = rec.d_hi;
    const uint32_t bos = A.flags & 1u, eos = (A.flags >> 1) & 1u, be = bos + eos;
    const unsigned long long base = blk_base + rec.pre;
    const int32_t* in = W.stream + tile * (uint64_t)kTileCap;
    const uint32_t n_tdocs = (uint32_t)(d_hi - d_lo);
    bool generic = n_tdocs > n_items + 1u || n_tdocs > (uint32_t)kEmitDocs;
    // -- document starts: bit `it` of the map for every document whose first item is `it`
    u64 dm = 0;         // lane w: word w
    if (!generic) {
        for (uint32_t d0 = 0; d0 < n_tdocs; d0 += 64) {
            const uint32_t nd = n_tdocs - d0 < 64u ? n_tdocs - d0 : 64u;
            const uint32_t it_v = lane < nd ? W.doc_item[d_lo + d0 + lane] : 0u;
            for (uint32_t k = 0; k < nd; k++) {   // (uniform)
                const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)it_v, (int)k);
                if (lane == (it >> 6)) dm |= 1ull << (it & 63u);
            }
        }
    }
    const uint32_t dcnt = (uint32_t)__popcll(dm);
    const uint32_t dinc = wave_iscan(dcnt);
    const uint32_t dpre = dinc - dcnt;   // lane w: document starts before word w
    if ((uint32_t)__builtin_amdgcn_readlane((int)dinc, 63) != n_tdocs) generic = true;   // two documents start at one item
    if (generic) {   // (uniform over the wave)
        if (lane == 0) { W.tile_flag[tile] = 1; atomicAdd(&W.status->generic_tiles, 1u); }
        return;
    }
    if (lane == 0) W.tile_flag[tile] = 0;
    // output slot of relative position 0 (may be -1 for the first tile; nothing is written below position 0 then)
    const long long o_base = (long long)base + (long long)d_lo * be - be + bos;
    int32_t* out = A.out_tok;

    uint32_t carry = 0;   // tokens of the items before the current step (uniform)
    const uint32_t rounds = (n_items + 1u + (uint32_t)kEmitRound - 1u) / (uint32_t)kEmitRound;
#pragma unroll 1
    for (uint32_t r = 0; r < rounds; r++) {
        int32_t item[kEmitSteps];
        uint32_t cnt[kEmitSteps];
        uint64_t loc[kEmitSteps];
        bool inl[kEmitSteps];
#pragma unroll
        for (int s = 0; s < kEmitSteps; s++) {
            const uint32_t i = r * (uint32_t)kEmitRound + (uint32_t)s * 64u + lane;
            item[s] = i < n_items ? in[i] : kDead;
        }
#pragma unroll
        for (int s = 0; s < kEmitSteps; s++) {
            cnt[s] = item[s] >= 0 ? 1u : 0u;
            loc[s] = 0;
            inl[s] = false;
            if (item[s] < kDead) deferred_result(W, item[s], &loc[s], &cnt[s], &inl[s]);
        }
#pragma unroll
        for (int s = 0; s < kEmitSteps; s++) {
            const uint32_t w = r * (uint32_t)kEmitSteps + (uint32_t)s;      // word of the bit map this step looks at (uniform)
            const uint32_t i = w * 64u + lane;
            const uint32_t inc = wave_iscan(cnt[s]);
            const uint32_t tp = carry + inc - cnt[s];                         // tokens of the tile before item i
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dm, (int)(w & 63u));
            const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dm >> 32), (int)(w & 63u));
            const uint32_t w_pre = (uint32_t)__builtin_amdgcn_readlane((int)dpre, (int)(w & 63u));
            const u64 word = ((u64)w_hi << 32) | w_lo;
            const bool starts = (word >> lane) & 1ull;
            const uint32_t dp = w_pre + (uint32_t)__popcll(word & ((2ull << lane) - 1ull));   // documents that start at or before item i
            const uint32_t rel = tp + dp * be;
            if (starts && i <= n_items) {   // a document starts here (tokenizer.go:225-229,258-262)
                const uint64_t d = d_lo + dp - 1;
                A.out_off[d] = base + tp + d * be;
                if (bos && d < A.n_docs) out[o_base + (long long)rel - bos] = W.bos_id;
                if (eos && d > 0) out[o_base + (long long)rel - bos - 1] = W.eos_id;
            }
            if (i < n_items) {
                if (item[s] >= 0) {
                    out[o_base + (long long)rel] = item[s];
                } else if (inl[s]) {   // up to three tokens inside the result word: straight-line stores
                    out[o_base + (long long)rel] = (int32_t)(loc[s] & 0x1FFFFu);
                    if (cnt[s] > 1) out[o_base + (long long)rel + 1] = (int32_t)((loc[s] >> 17) & 0x1FFFFu);
                    if (cnt[s] > 2) out[o_base + (long long)rel + 2] = (int32_t)((loc[s] >> 34) & 0x1FFFFu);
                }
            }
            // runs of tokens from the pool: copied by the whole wave, one run after the other
            u64 runs = __ballot(i < n_items && item[s] < kDead && !inl[s] && cnt[s] > 0);
            while (runs) {   // (uniform)
                const int b = __builtin_ctzll(runs);
                runs &= runs - 1ull;
                const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)cnt[s], b);
                const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)rel, b);
                const uint64_t po = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(loc[s] >> 32), b) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)loc[s], b);
                if (cn > kBigRun) {   // listed for copy_runs: other workgroups copy it
                    const bool huge = cn > kHugeRun;
                    uint32_t idx = 0;
                    if (lane == 0) idx = atomicAdd(huge ? &W.status->huge_runs : &W.status->big_runs, 1u);
                    idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
                    if (idx < (huge ? kMaxHugeRuns : kMaxBigRuns)) {
                        if (lane == 0) W.runs[(huge ? kMaxBigRuns : 0u) + idx] = BigRun{po, (uint64_t)(o_base + (long long)rl), cn};
                        continue;
                    }
                }
                const int32_t* src = W.pool + po;
                int32_t* dst = out + (o_base + (long long)rl);
                for (uint32_t k = lane; k < cn; k += 64) dst[k] = src[k];
            }
        }
    }
}

// Tiles the fast kernel leaves (W.tile_flag): several documents starting at one item (empty documents), or more tokens
// than its staging buffer holds.  One workgroup looks after kGenericPer consecutive tiles.
constexpr int kGenericPer = 16;
__device__ void emit_tile_generic(const Workspace& W, const EncodeArgs& A, uint64_t tile, uint64_t n_tiles, uint32_t* docinc,
                                  uint32_t* tokpre, unsigned long long* tmp64, unsigned long long* q_out, int32_t* q_item,
                                  uint32_t* q_n_p) {
    uint32_t& q_n = *q_n_p;
    const int tid = threadIdx.x;
    const bool last_tile = tile + 1 == n_tiles;
    const uint32_t n_items = W.tile_items[tile];
    const uint64_t d_lo = W.tile_doc0[tile];
    const uint64_t d_hi = last_tile ? A.n_docs + 1 : (uint64_t)W.tile_doc0[tile + 1];
    const uint32_t bos = A.flags & 1u, eos = (A.flags >> 1) & 1u, be = bos + eos;
    const int32_t* in = W.stream + tile * (uint64_t)kTileCap;

    for (int i = tid; i <= (int)n_items; i += kTileThreads) docinc[i] = 0;
    if (tid == 0) q_n = 0;
    // tokens before this tile: scanned block base + the tiles of the block that precede it
    unsigned long long blk_base, all_blocks;
    block_prefix(W, (uint32_t)(tile / kScanBlock), (uint32_t)((n_tiles + kScanBlock - 1) / kScanBlock), &blk_base, &all_blocks);
    const unsigned long long base = blk_base + W.tile_rec[tile].pre;
    __syncthreads();   // (the counters are cleared by other threads than those that count: without this barrier a document
                       // start was lost once in some ten thousand heavy tiles, and the tile's tokens landed one slot early)
    for (uint64_t d = d_lo + tid; d < d_hi; d += kTileThreads) atomicAdd(&docinc[W.doc_item[d]], 1u);
    __syncthreads();

    // items in rounds of one per thread (coalesced); running token / document prefixes carried across rounds
    unsigned long long carry = 0;  // low 32: tokens, high 32: documents started
    const uint32_t rounds = (n_items + 1 + kTileThreads - 1) / kTileThreads;
#pragma unroll 1
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = r * kTileThreads + tid;
        const int32_t item = i < n_items ? in[i] : kDead;
        uint32_t cnt = item >= 0 ? 1u : 0u;
        uint64_t poff = 0;
        bool inl = false;
        if (item < kDead) deferred_result(W, item, &poff, &cnt, &inl);
        const uint32_t dinc = i <= n_items ? docinc[i] : 0u;
        unsigned long long tot;
        const unsigned long long ex = carry + block_escan64(((unsigned long long)dinc << 32) | cnt, tmp64, &tot);
        carry += tot;
        const uint32_t tp = (uint32_t)ex;
        const uint32_t dp = (uint32_t)(ex >> 32) + dinc;  // documents that start at or before item i
        if (i <= n_items) tokpre[i] = tp;
        if (i < n_items) {
            const uint64_t D = d_lo + dp - 1;
            const uint64_t o = base + tp + D * be + bos;
            if (item >= 0) {
                A.out_tok[o] = item;
            } else if (cnt <= 64) {
                for (uint32_t k = 0; k < cnt; k++) A.out_tok[o + k] = deferred_token(W, poff, inl, k);
            } else {
                uint32_t qi = atomicAdd(&q_n, 1u);
                if (qi < kEmitQueue) { q_out[qi] = o; q_item[qi] = item; }
                else for (uint32_t k = 0; k < cnt; k++) A.out_tok[o + k] = deferred_token(W, poff, inl, k);   // queue full: the owner copies
            }
        }
    }
    __syncthreads();
    // per-document offsets, BOS / EOS (tokenizer.go:225-229,258-262)
    for (uint64_t d = d_lo + tid; d < d_hi; d += kTileThreads) {
        const uint64_t off = base + tokpre[W.doc_item[d]] + d * be;
        A.out_off[d] = off;
        if (bos && d < A.n_docs) A.out_tok[off] = W.bos_id;
        if (eos && d > 0) A.out_tok[off - 1] = W.eos_id;
    }
    // long runs of tokens from the pool: copied by the whole workgroup.  (A pre-token of more than 64 tokens has at least
    // 22 bytes -- an invalid byte is three symbols --, so a tile can hold 174 of them: what the queue does not take its
    // owner has copied above.)
    const uint32_t qn = q_n < kEmitQueue ? q_n : kEmitQueue;
    for (uint32_t qi = 0; qi < qn; qi++) {
        uint64_t po; uint32_t cn; bool il;
        deferred_result(W, q_item[qi], &po, &cn, &il);
        const uint64_t o = q_out[qi];
        for (uint32_t k = tid; k < cn; k += kTileThreads) A.out_tok[o + k] = W.pool[po + k];
    }
}
__global__ __launch_bounds__(kTileThreads) void emit_generic_kernel(Workspace W, EncodeArgs A, uint64_t n_tiles) {
    __shared__ uint32_t docinc[kTileCap + 1];
    __shared__ uint32_t tokpre[kTileCap + 1];
    __shared__ unsigned long long tmp64[8];
    __shared__ unsigned long long q_out[kEmitQueue];
    __shared__ int32_t q_item[kEmitQueue];
    __shared__ uint32_t q_n;
    const uint32_t fl = W.status->flags;
    if (fl & (kFatalFlags | kRetryFlags | kGiantPending | kSeqPending)) return;
    if (fl & kOvfOut) return;   // (set by the fast kernel, which ran before)
    const uint64_t t0 = (uint64_t)blockIdx.x * kGenericPer;
    for (uint64_t t = t0; t < t0 + kGenericPer && t < n_tiles; t++) {
        if (!W.tile_flag[t]) continue;   // uniform over the workgroup
        emit_tile_generic(W, A, t, n_tiles, docinc, tokpre, tmp64, q_out, q_item, &q_n);
        __syncthreads();
    }
}


// The status word to the host (page-locked, device-visible memory) and, unless a host-driven pass is still to come,
// back to zero for the next batch: one small kernel instead of a copy and a memset between two batches.
// clear: 0 = leave it, 1 = zero it unless kSeqPending / kGiantPending is set, 2 = zero it.
__global__ __launch_bounds__(256) void status_out_kernel(Status* st, Status* host_dst, int clear) {
    uint32_t* src = reinterpret_cast<uint32_t*>(st);
    uint32_t* dst = reinterpret_cast<uint32_t*>(host_dst);
    const uint32_t fl = st->flags;
    constexpr uint32_t n_words = sizeof(Status) / 4;
    for (uint32_t i = threadIdx.x; i < n_words; i += 256) dst[i] = src[i];
    __syncthreads();
    if (clear == 0 || (clear == 1 && (fl & (kSeqPending | kGiantPending)))) return;
    for (uint32_t i = threadIdx.x; i < n_words; i += 256) src[i] = 0;
}
// The listed runs of tokens (BigRun) from the pool to the output: a big run by one workgroup (the runs dealt out over the
// grid in turn), a huge one in pieces of kRunChunk tokens by all workgroups.
constexpr uint32_t kRunChunk = 8192;
__device__ void copy_runs(const Workspace& W, const EncodeArgs& A, uint32_t n_big, uint32_t n_huge, uint32_t block, uint32_t n_blocks) {
    for (uint32_t e = block; e < n_big; e += n_blocks) {   // (uniform)
        const BigRun r = W.runs[e];
        for (uint64_t k = threadIdx.x; k < r.n; k += blockDim.x) A.out_tok[r.dst + k] = W.pool[r.src + k];
    }
    uint64_t g = 0;   // pieces of the huge runs before the current one
    for (uint32_t e = 0; e < n_huge; e++) {   // (uniform)
        const BigRun r = W.runs[kMaxBigRuns + e];
        const uint64_t pieces = (r.n + kRunChunk - 1) / kRunChunk;
        for (uint64_t c = ((uint64_t)block + n_blocks - g % n_blocks) % n_blocks; c < pieces; c += n_blocks) {
            const uint64_t lo = c * kRunChunk, hi = lo + kRunChunk < r.n ? lo + kRunChunk : r.n;
            for (uint64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) A.out_tok[r.dst + k] = W.pool[r.src + k];
        }
        g += pieces;
    }
}
__global__ __launch_bounds__(kTileThreads) void copy_runs_kernel(Workspace W, EncodeArgs A, long long n_big_host, long long n_huge_host) {
    const uint32_t fl = W.status->flags;
    if (n_big_host < 0 && (fl & (kFatalFlags | kRetryFlags | kGiantPending | kSeqPending | kOvfOut))) return;
    const uint32_t nb = n_big_host < 0 ? W.status->big_runs : (uint32_t)n_big_host;
    const uint32_t nh = n_huge_host < 0 ? W.status->huge_runs : (uint32_t)n_huge_host;
    copy_runs(W, A, nb < kMaxBigRuns ? nb : kMaxBigRuns, nh < kMaxHugeRuns ? nh : kMaxHugeRuns, blockIdx.x, gridDim.x);
}
// The two launches that end a stream-ordered batch in one: the tiles the fast emit kernel flagged (looked for by a fixed,
// small grid: normally there are none), then the status word to the host and back to zero by whichever workgroup is
// through last -- every other workgroup has made its last access to the word by then.
constexpr int kFinishBlocks = 256;
__global__ __launch_bounds__(kTileThreads) void finish_kernel(Workspace W, EncodeArgs A, uint64_t n_tiles, Status* host_dst) {
    __shared__ uint32_t docinc[kTileCap + 1];
    __shared__ uint32_t tokpre[kTileCap + 1];
    __shared__ unsigned long long tmp64[8];
    __shared__ unsigned long long q_out[kEmitQueue];
    __shared__ int32_t q_item[kEmitQueue];
    __shared__ uint32_t q_n, n_flagged, s_last;
    __shared__ uint32_t flagged[kTileThreads];
    const uint32_t fl = W.status->flags;
    if (!(fl & (kFatalFlags | kRetryFlags | kGiantPending | kSeqPending | kOvfOut))) {
        for (uint64_t t0 = (uint64_t)blockIdx.x * kTileThreads; t0 < n_tiles; t0 += (uint64_t)gridDim.x * kTileThreads) {   // (uniform)
            if (threadIdx.x == 0) n_flagged = 0;
            __syncthreads();
            const uint64_t t = t0 + threadIdx.x;
            if (t < n_tiles && W.tile_flag[t]) flagged[atomicAdd(&n_flagged, 1u)] = (uint32_t)t;
            __syncthreads();
            const uint32_t nf = n_flagged;
            for (uint32_t k = 0; k < nf; k++) {
                emit_tile_generic(W, A, flagged[k], n_tiles, docinc, tokpre, tmp64, q_out, q_item, &q_n);
                __syncthreads();
            }
        }
        const uint32_t nb = W.status->big_runs, nh = W.status->huge_runs;
        copy_runs(W, A, nb < kMaxBigRuns ? nb : kMaxBigRuns, nh < kMaxHugeRuns ? nh : kMaxHugeRuns, blockIdx.x, gridDim.x);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&W.status->done_blocks, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    uint32_t* src = reinterpret_cast<uint32_t*>(W.status);
    uint32_t* dst = reinterpret_cast<uint32_t*>(host_dst);
    constexpr uint32_t n_words = sizeof(Status) / 4;
    for (uint32_t i = threadIdx.x; i < n_words; i += kTileThreads) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_words; i += kTileThreads) src[i] = 0;
}

}  // namespace

hipError_t launch_emit_generic(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    hipLaunchKernelGGL(emit_generic_kernel, dim3((uint32_t)((n_tiles + kGenericPer - 1) / kGenericPer)), dim3(kTileThreads), 0,
                       stream, W, A, n_tiles);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

#ifndef L3_FINISH_GRID
#define L3_FINISH_GRID 512
#endif
static hipError_t launch_totals_scan(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    const uint64_t n_blocks = (n_tiles + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(block_totals_kernel, dim3((uint32_t)n_blocks + L3_FINISH_GRID), dim3(kScanBlock), 0, stream, W, n_tiles, A.n_docs,
                       (uint32_t)n_blocks);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_copy_runs(const Workspace& W, const EncodeArgs& A, long long n_big, long long n_huge, hipStream_t stream) {
    hipLaunchKernelGGL(copy_runs_kernel, dim3(1024), dim3(kTileThreads), 0, stream, W, A, n_big, n_huge);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_finish(const Workspace& W, const EncodeArgs& A, Status* host_dst, hipStream_t stream) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    uint64_t grid = (n_tiles + kTileThreads - 1) / kTileThreads;
    if (grid > kFinishBlocks) grid = kFinishBlocks;
    hipLaunchKernelGGL(finish_kernel, dim3((uint32_t)grid), dim3(kTileThreads), 0, stream, W, A, n_tiles, host_dst);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_status_out(const Workspace& W, Status* host_dst, hipStream_t stream, int clear) {
    hipLaunchKernelGGL(status_out_kernel, dim3(1), dim3(256), 0, stream, W.status, host_dst, clear);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_stage_totals(const Workspace& W, const EncodeArgs& A, hipStream_t stream) { return launch_totals_scan(W, A, stream); }

hipError_t launch_stage_emit(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    hipLaunchKernelGGL(emit_kernel, dim3((uint32_t)((n_tiles + kTileThreads / 64 - 1) / (kTileThreads / 64))), dim3(kTileThreads), 0, stream, W, A,
                       (uint32_t)n_tiles);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_encode_tail(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    hipError_t e_ = launch_totals_scan(W, A, stream);
    if (e_ != hipSuccess) return e_;
    return launch_stage_emit(W, A, stream);
}

}  // namespace l3
