// l3_merge.hip -- merge_kernel: the short deferred pre-tokens, one lane each; its last workgroups make the wave-per-entry
// pass over the slow list (mid_pass) and cut the long entries into window items (prep_entry).  Overview: l3_pipeline.hip.
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_dev.h"
#include "l3_merge_dev.h"

namespace l3 {

namespace {

// Stores the result of a finished slot: miss_res word (tokens inline or in the pool), tile_extra.  All lanes of
// the wave call it together (`act`: this lane has a result); pool space is one atomic per wave.
__device__ __forceinline__ void lane_finish(const Workspace& W, const uint32_t* sy, bool act, uint64_t ei, uint32_t tile,
                                            uint32_t shard) {
    uint32_t n_out = 0;
    uint64_t inl = 0;
    if (act) {
        uint32_t p = 0;
        while (p != kNone) {
            const uint32_t w = sy[p];
            if (n_out < kMissInline) inl |= (uint64_t)(w & 0x1FFFFu) << (17u * n_out);
            n_out++;
            p = (w >> 17) & 127u;
        }
    }
    const uint32_t need = (act && n_out > kMissInline) ? n_out : 0u;
    const uint32_t inc = wave_iscan(need);
    const uint32_t wtot = __shfl(inc, 63, 64);
    unsigned long long wbase = 0;
    if (wtot) {   // uniform over the wave
        if (lane_id() == 63) wbase = atomicAdd(&W.status->shard[shard].pool_used, (unsigned long long)wtot);
        wbase = __shfl(wbase, 63, 64);
    }
    if (act && need) {
        const unsigned long long loc = wbase + inc - need;   // inside the shard's part of the pool
        const unsigned long long dst = (unsigned long long)shard * W.mpool_shard_cap + loc;
        if (loc + n_out <= W.mpool_shard_cap) {
            uint32_t p = 0, k = 0;
            while (p != kNone) {
                const uint32_t w = sy[p];
                W.pool[dst + k++] = (int32_t)(w & 0x1FFFFu);
                p = (w >> 17) & 127u;
            }
        } else {
            atomicOr(&W.status->flags, kOvfPool);
        }
        inl = dst;
    }
    if (act) W.miss_res[ei] = (inl << 8) | n_out;
    // tile_extra: neighbouring lanes mostly hold pre-tokens of one tile, so the additions of a run of lanes with the
    // same tile are summed in the wave first (segmented scan) and its last lane issues one atomic
    const uint32_t key = act ? tile : 0xFFFFFFFFu;
    uint32_t sum = (act && n_out > 1) ? n_out - 1 : 0u;
    const uint32_t prev_key = __shfl_up(key, 1, 64);
    const bool head = lane_id() == 0 || prev_key != key;
    // distance to the head of the run: max-scan of the head positions
    uint32_t hp = head ? (uint32_t)lane_id() : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(hp, d, 64);
        if (lane_id() >= d) hp = o > hp ? o : hp;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(sum, d, 64);
        if (lane_id() >= d && (uint32_t)(lane_id() - d) >= hp) sum += o;
    }
    const uint32_t next_key = __shfl_down(key, 1, 64);
    const bool tail = lane_id() == 63 || next_key != key;
    if (act && tail && sum) atomicAdd(&W.tile_extra[tile], sum);
}

struct MidLds;
__device__ void mid_pass(const Tables& T, const Workspace& W, const EncodeArgs& A, MidLds* mid, uint32_t block, uint32_t n_blocks);

// Workgroups [0, n_merge_blocks) merge the short entries; the workgroups behind them run mid_pass() on the slow list
// (same launch: a launch of its own would cost every batch 4 us, and this kernel's registers are not what limits it).
__global__ __launch_bounds__(kMergeThreads, 4) void merge_kernel(Tables T, Workspace W, EncodeArgs A, uint32_t n_merge_blocks) {
    __shared__ __align__(16) uint32_t arena[2 * kMergeArena];
    uint32_t* const sy_a = arena;
    uint32_t* const pq_a = arena + kMergeArena;
    __shared__ uint8_t byte_tok[256];
    __shared__ uint32_t tmp[16];
    if (W.status->flags & kBadOffsets) return;
    if (blockIdx.x >= n_merge_blocks) {
        mid_pass(T, W, A, reinterpret_cast<MidLds*>(arena), blockIdx.x - n_merge_blocks, gridDim.x - n_merge_blocks);
        return;
    }
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += kMergeThreads) byte_tok[i] = T.byte_tok[i];
    // Windows of 256 entries over all shards, longest length class first (its windows take longest); workgroup b
    // takes windows b, b + gridDim.x, ...
    __shared__ uint32_t win0[kMissShards + 1];   // first window of shard 63 - i
    if (tid < 64) {
        uint64_t c = W.status->shard[kMissShards - 1 - tid].miss_count;
        if (c > W.miss_shard_cap) c = W.miss_shard_cap;
        const uint32_t nw = (uint32_t)((c + kMergeThreads - 1) / kMergeThreads);
        const uint32_t inc = wave_iscan(nw);
        win0[tid] = inc - nw;
        if (tid == 63) win0[kMissShards] = inc;
    }
    __syncthreads();
    const uint32_t n_win = win0[kMissShards];
#pragma unroll 1
    for (uint32_t w = blockIdx.x; w < n_win; w += n_merge_blocks) {
        uint32_t lo = 0, hi = kMissShards - 1;   // last i with win0[i] <= w (empty shards share a value: take the last)
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (win0[mid] <= w) lo = mid; else hi = mid - 1;
        }
        const uint32_t shard = kMissShards - 1 - lo;
        uint64_t n_ent = W.status->shard[shard].miss_count;
        if (n_ent > W.miss_shard_cap) n_ent = W.miss_shard_cap;
        const uint64_t chunk = (uint64_t)(w - win0[lo]) * kMergeThreads;
        const bool have = chunk + tid < n_ent;
        const uint64_t ei = shard * W.miss_shard_cap + chunk + tid;
        MissIn mi;
        mi.len = 0; mi.tile = 0; mi.start = 0; mi.w[0] = mi.w[1] = mi.w[2] = mi.w[3] = 0;
        if (have) mi = W.miss_in[ei];
        const uint32_t len = mi.len;
        // ---- up to 16 bytes: every lane in its own 16-element slot
        {
            const bool act = have && len >= 1 && len <= 16;
            uint32_t* sy = sy_a + tid * 17;
            uint32_t* pq = pq_a + tid * 17;
            if (act) {
                uint32_t bb[16], pp[15];
#pragma unroll
                for (int j = 0; j < 16; j++) bb[j] = (mi.w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
#pragma unroll
                for (int j = 0; j < 15; j++) pp[j] = (uint32_t)(j + 1) < len ? T.bytepair[(bb[j] << 8) | bb[j + 1]] : kNoPrio;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t nxt = (uint32_t)(j + 1) < len ? (uint32_t)(j + 1) : kNone, prv = j ? (uint32_t)(j - 1) : kNone;
                    sy[j] = (uint32_t)byte_tok[bb[j]] | (nxt << 17) | (prv << 24);
                    pq[j] = (j < 15 && pp[j < 15 ? j : 0] != kNoPrio && (uint32_t)j < len) ? (pp[j < 15 ? j : 0] << 7) | (uint32_t)j : 0xFFFFFFFFu;
                }
                lane_merge<16>(T, sy, pq);
            }
            lane_finish(W, sy, act, ei, mi.tile, shard);
        }
        // ---- 17..64 bytes: groups of 64 lanes, 64-element slots of the same arena
        const bool is_long = have && len > 16;
        const int n_long = __syncthreads_count(is_long);   // also: everyone is done with the short slots
        if (n_long) {
            uint32_t tot;
            const uint32_t ord = block_escan<kMergeWaves>(is_long ? 1u : 0u, tmp, &tot);
#pragma unroll 1
            for (uint32_t g = 0; g * 64u < (uint32_t)n_long; g++) {
                const bool act = is_long && ord / 64u == g;
                uint32_t* sy = sy_a + (ord % 64u) * 65u;
                uint32_t* pq = pq_a + (ord % 64u) * 65u;
                if (act) {
                    const uint8_t* src = A.text + mi.start;
                    uint32_t prev = src[0];
                    for (uint32_t j = 0; j < 64; j++) {
                        if (j < len) {
                            const uint32_t cur = j + 1 < len ? (uint32_t)src[j + 1] : 0u;
                            const uint32_t pr = j + 1 < len ? T.bytepair[(prev << 8) | cur] : kNoPrio;
                            sy[j] = (uint32_t)byte_tok[prev] | ((j + 1 < len ? j + 1 : kNone) << 17) | ((j ? j - 1 : kNone) << 24);
                            pq[j] = pr != kNoPrio ? (pr << 7) | j : 0xFFFFFFFFu;
                            prev = cur;
                        } else {
                            pq[j] = 0xFFFFFFFFu;
                        }
                    }
                    lane_merge<64>(T, sy, pq);
                }
                // all four waves take part (lanes of a group are spread over them)
                lane_finish(W, sy, act, ei, mi.tile, shard);
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t mid_symbols(const Tables& T, const EncodeArgs& A, uint64_t start, uint32_t len) {
    const uint32_t lane = (uint32_t)lane_id();
    GlobalSrc gs{A.text, (int64_t)start, (int64_t)(start + len)};
    uint32_t L = 0;
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t q = base + lane;
        const bool in = q < len;
        const bool inv = in && (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID);
        L += (uint32_t)__popcll(__ballot(in)) + 2u * (uint32_t)__popcll(__ballot(inv));
    }
    return L;
}
// The entry's tokens, by the calling wave alone; L = mid_symbols(); poff = where its tokens go in the slow pool
// (room for L of them), -1: no room.
__device__ void mid_entry(const Tables& T, const Workspace& W, const EncodeArgs& A, uint64_t ei, uint32_t L, long long poff,
                          MidLds& M) {
    const uint32_t lane = (uint32_t)lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint64_t start = W.slow[ei].start;
    const uint32_t len = W.slow[ei].len;
    const bool valid = W.slow[ei].count == 1u;   // the tile kernel saw no invalid byte in it: its symbols are its bytes
    GlobalSrc gs{A.text, (int64_t)start, (int64_t)(start + len)};
    group_sync<true>();  // the arena may still be read by this wave's previous entry
    int32_t *sa = M.sA, *sb = M.sB;
    uint32_t *pa = M.pA, *pb = M.pB;
    if (L > (uint32_t)kMidSyms) {  // many invalid bytes: the arrays go to the HBM scratch
        long long off = 0;
        if (lane == 0) {
            const unsigned long long need = (unsigned long long)L * 16ull;
            const unsigned long long o = atomicAdd(&W.status->scratch_used, need);
            off = (o + need <= W.scratch_cap) ? (long long)o : -1ll;
            if (off < 0) atomicOr(&W.status->flags, kOvfScratch);
        }
        off = __shfl(off, 0, 64);
        if (off < 0) return;
        uint8_t* base = W.scratch + off;
        sa = reinterpret_cast<int32_t*>(base);
        sb = sa + L;
        pa = reinterpret_cast<uint32_t*>(sb + L);
        pb = pa + L;
    }
    uint32_t o = 0;
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t q = base + lane;
        const bool in = q < len;
        const bool inv = in && !valid && (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID);
        const unsigned long long mi = __ballot(in), mv = __ballot(inv);
        const uint32_t pos = o + (uint32_t)__popcll(mi & lt) + 2u * (uint32_t)__popcll(mv & lt);
        if (inv) {
            sa[pos] = 0xEF; sa[pos + 1] = 0xBF; sa[pos + 2] = 0xBD;
        } else if (in) {
            sa[pos] = A.text[start + q];
        }
        o += (uint32_t)__popcll(mi) + 2u * (uint32_t)__popcll(mv);
    }
    group_sync<true>();
    int32_t whole = -1;
    if (L <= (uint32_t)kMaxTokenBytes) {  // bpe.go:27-33
        if (lane == 0) whole = vocab_lookup(T, [&](uint32_t i) { return (uint32_t)sa[i]; }, L);
        whole = __shfl(whole, 0, 64);
    }
    uint32_t n_out;
    if (whole >= 0) {
        if (lane == 0) sa[0] = whole;
        n_out = 1;
        group_sync<true>();
    } else {
        for (uint32_t i = lane; i < L; i += 64)
            pa[i] = i + 1 < L ? T.bytepair[((uint32_t)sa[i] << 8) | (uint32_t)sa[i + 1]] : kNoPrio;
        group_sync<true>();
        for (uint32_t i = lane; i < L; i += 64) sa[i] = T.byte_tok[sa[i]];
        group_sync<true>();
        n_out = merge_rounds<true>(T, sa, pa, sb, pb, L, nullptr, nullptr, nullptr);
        group_sync<true>();
    }
    if (poff < 0) return;
    if (lane == 0) {
        W.slow[ei].count = n_out;
        W.slow[ei].pool_off = (unsigned long long)kMissShards * W.mpool_shard_cap + (unsigned long long)poff;
        if (n_out > 1) atomicAdd(&W.tile_extra[W.slow[ei].tile], n_out - 1);
    }
    int32_t* dst = W.pool + (unsigned long long)kMissShards * W.mpool_shard_cap + poff;
    for (uint32_t i = lane; i < n_out; i += 64) dst[i] = sa[i] & kIdMask;
}

// Preparation of a long slow entry for slow_kernel, by the calling wave: where it ends, whether its symbols are its
// bytes, room for its tokens in the slow pool, its window items.
//   End (state_machine.go:53-79: at the next pre-token start): its tile saw none up to the end of its look-ahead, which
//   lies in the next tile; every tile decides all the positions it owns, so the end is the first pre-token start owned
//   by a later tile (document ends and special tokens are starts), or the end of the text.
//   Room: an IMAGE of the pre-token (one slot per symbol, see slow_kernel) behind the place of its final tokens.
__device__ void prep_entry(const Workspace& W, const EncodeArgs& A, uint32_t e) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t start = W.slow[e].start;
    uint64_t len = W.slow[e].len;
    bool valid = W.slow[e].count == 1u;
    if (len == 0) {
        const uint64_t n_tiles = num_tiles(A.n_bytes);
        const uint64_t tile0 = W.slow[e].tile;
        bool inv = (W.tile_first[tile0] & kFirstInv) != 0;
        uint64_t end = A.n_bytes;
        for (uint64_t t0 = tile0 + 1; t0 < n_tiles; t0 += 64) {
            const uint64_t t = t0 + lane;
            const uint32_t f = t < n_tiles ? W.tile_first[t] : kNoFirst;
            const unsigned long long hm = __ballot((f & ~kFirstInv) != kNoFirst), im = __ballot((f & kFirstInv) != 0);
            if (hm) {
                const int b = __builtin_ctzll(hm);
                if (im & ((2ull << b) - 1ull)) inv = true;
                end = (t0 + (uint64_t)b) * (uint64_t)kTile + (uint64_t)(__shfl(f, b, 64) & ~kFirstInv);
                break;
            }
            if (im) inv = true;
        }
        valid = !inv;
        len = end - start;
    }
    if (len >= (1ull << 31)) {
        if (lane == 0) { atomicOr(&W.status->flags, kTooLong); W.slow[e].len = 0; W.slow[e].count = 0; W.slow[e].win0 = ~0ull; }
        return;
    }
    const uint64_t slots = valid ? len : 3ull * len;                 // symbols (an invalid byte is three: EF BF BD), at most
    const uint32_t n_win = (uint32_t)((slots + (uint64_t)kWin - 1) / (uint64_t)kWin);
    long long poff = -1, wbase = -1;
    if (lane == 0) {
        const unsigned long long off = atomicAdd(&W.status->pool_used, 2ull * slots);
        if (off + 2ull * slots <= W.pool_cap) poff = (long long)off; else atomicOr(&W.status->flags, kOvfPool);
        const uint32_t wb = atomicAdd(&W.status->slow_q.count, n_win);
        if ((uint64_t)wb + n_win <= W.win_cap) wbase = (long long)wb; else atomicOr(&W.status->flags, kOvfWin);
        W.slow[e].len = (uint32_t)len;
        W.slow[e].count = 0;
        W.slow[e].win0 = (uint64_t)(wbase < 0 ? 0 : wbase) | (valid ? 0ull : kEntrySequential);
        W.slow[e].pool_off = (unsigned long long)kMissShards * W.mpool_shard_cap + (unsigned long long)(poff < 0 ? 0 : poff);
    }
    poff = __shfl(poff, 0, 64);
    wbase = __shfl(wbase, 0, 64);
    if (poff < 0 || wbase < 0) return;   // the host grows the lists and runs the batch again
    for (uint32_t k = lane; k < n_win; k += 64) {
        W.win[(uint64_t)wbase + k] = WinItem{e, k};
        W.win_live[(uint64_t)wbase + k] = 0;
    }
}

// Every wave on its own takes a few entries of the slow list at a time, handles those of known, moderate length and
// notes the others for slow_kernel (Status::slow_q).  Room in the slow pool is reserved once per visit, one token
// per symbol of the entries taken (a returning atomic per entry on the one counter queued up: 0.3 ms for 58 k entries).
constexpr int kMidPop = 8;
static_assert(kMergeThreads == kSlowThreads, "mid_pass runs inside merge_kernel's launch");
static_assert(sizeof(MidLds) * (kSlowThreads / 64) <= 2 * kMergeArena * sizeof(uint32_t), "mid_pass uses merge_kernel's arena");
__device__ void mid_pass(const Tables& T, const Workspace& W, const EncodeArgs& A, MidLds* mid, uint32_t block, uint32_t n_blocks) {
    uint64_t n_slow = W.status->slow_count;
    if (n_slow > W.slow_cap) n_slow = W.slow_cap;
    const uint32_t lane = (uint32_t)lane_id();
    MidLds& M = mid[wave_id()];
    // entries per visit to the counter: few enough that every wave of the grid gets some, at most kMidPop
    uint32_t pop = (uint32_t)(n_slow / ((uint64_t)n_blocks * (kSlowThreads / 64)));
    pop = pop < 1u ? 1u : pop > (uint32_t)kMidPop ? (uint32_t)kMidPop : pop;
    // the first visit needs no counter: wave g starts at g * pop (an empty or short list costs no atomics at all)
    const uint32_t n_waves = n_blocks * (uint32_t)(kSlowThreads / 64);
    bool first = true;
#pragma unroll 1
    for (;;) {
        uint32_t e0 = (block * (uint32_t)(kSlowThreads / 64) + (uint32_t)wave_id()) * pop;
        if (!first) {
            if (lane == 0) e0 = n_waves * pop + atomicAdd(&W.status->mid_q.next, pop);
            e0 = __shfl(e0, 0, 64);
        }
        first = false;
        if (e0 >= n_slow) break;
        const uint32_t e = e0 + lane;
        const bool in = lane < pop && e < n_slow;
        const uint32_t len = in ? W.slow[e].len : 1u;
        const uint64_t start = in ? W.slow[e].start : 0ull;
        const bool known_valid = in && W.slow[e].count == 1u;
        const bool big = in && (len == 0 || len > (uint32_t)kMidBytes);
        const unsigned long long bm = __ballot(big);
        for (unsigned long long m = bm; m; m &= m - 1ull)   // not for a single wave: prepared for slow_kernel
            prep_entry(W, A, e0 + (uint32_t)__builtin_ctzll(m));
        const unsigned long long mm = __ballot(in && !big);
        if (!mm) continue;
        uint32_t my_l = 0;  // lane b: symbols of entry e0 + b
        for (unsigned long long m = mm; m; m &= m - 1ull) {
            const int b = __builtin_ctzll(m);
            const uint32_t len_b = __shfl(len, b, 64);
            const uint32_t lb = __shfl(known_valid ? 1 : 0, b, 64) ? len_b : mid_symbols(T, A, __shfl(start, b, 64), len_b);
            if (lane == (uint32_t)b) my_l = lb;
        }
        const uint32_t inc = wave_iscan(my_l);
        const uint32_t total = __shfl(inc, 63, 64);
        long long base = 0;
        if (lane == 0) {
            const unsigned long long off = atomicAdd(&W.status->pool_used, (unsigned long long)total);
            const bool fits = off + total <= W.pool_cap;
            if (!fits) atomicOr(&W.status->flags, kOvfPool);
            base = fits ? (long long)off : -1ll;
        }
        base = __shfl(base, 0, 64);
        for (unsigned long long m = mm; m; m &= m - 1ull) {
            const int b = __builtin_ctzll(m);
            const uint32_t lb = __shfl(my_l, b, 64);
            const uint32_t pre = __shfl(inc, b, 64) - lb;
            mid_entry(T, W, A, e0 + (uint32_t)b, lb, base < 0 ? -1ll : base + (long long)pre, M);
        }
    }
}

}  // namespace

#ifndef L3_MERGE_GRID
#define L3_MERGE_GRID 2048
#endif
// n_slow_blocks: workgroups for the wave-per-entry pass behind the merge workgroups
hipError_t launch_stage_merge(const Tables& T, const Workspace& W, const EncodeArgs& A, uint32_t n_slow_blocks, hipStream_t stream) {
    uint64_t mgrid = ((W.miss_shard_cap + kMergeThreads - 1) / kMergeThreads) * kMissShards;  // windows at most
    if (mgrid > L3_MERGE_GRID) mgrid = L3_MERGE_GRID;
    if (mgrid == 0) mgrid = 1;
    hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)mgrid + n_slow_blocks), dim3(kMergeThreads), 0, stream, T, W, A, (uint32_t)mgrid);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3
