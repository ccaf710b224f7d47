// l3_slow.hip -- slow_kernel: the deferred pre-tokens the merge kernel's passes leave (long, invalid UTF-8, or ending beyond
// the halo) as independent WINDOW items.  Overview: l3_pipeline.hip.
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_dev.h"
#include "l3_merge_dev.h"

namespace l3 {

namespace {

// ---- long pre-tokens piece by piece, window by window ---------------------------------------------------------
// A long pre-token is cut at the positions no vocabulary entry can span (l3_common.h: prefix tables, reach, cuts); the
// pieces are independent BPE problems.  Ideographs, byte soup and most non-Latin text fall into pieces of a few bytes,
// which single lanes merge in LDS slots as merge_kernel does; only a piece of more than kLaneMax symbols (glued words,
// one character repeated) still needs the round-based merge.
// Work items are WINDOWS of kWin symbols (prep_entry made them): an item decides the cuts inside its window -- a match
// is at most kMaxTokenBytes long, so the reaches of the kWinBack symbols in front of the window are all it needs from
// outside -- and handles the pieces that START in it, wherever they end.  Results go into an IMAGE of the pre-token in
// the slow pool: one int32 slot per symbol, a piece's tokens in the slots of its first symbols, -1 in its others.  No
// two pieces share a slot, so windows and pieces can be done in any order by any workgroup; win_live counts the tokens
// in each window's slots, and finish_windows() (in the launch of block_totals_kernel) moves them to their final place.
constexpr int kWinExam = kWin + kWinBack;                 // positions whose match length a window item computes
constexpr int kWinPer = kWinExam / kSlowThreads;          // ... consecutive ones per thread
constexpr int kWinAhead = kMaxTokenBytes;                 // a match looks at most this far ahead
constexpr int kSeqInline = 8;                             // windows up to which a pre-token with invalid UTF-8 is finished by the item that expands it
constexpr int kTailSteps = 4;                             // steps of kSlowThreads positions a window walks on to find the end of its last piece
static_assert(kWinPer * kSlowThreads == kWinExam && kWinPer <= 16 && kWinBack >= kMaxTokenBytes, "window geometry");
static_assert(kWin >= 2 * kLaneMax, "a lane piece reaches at most into the next window's slots");
struct PieceLds {
    uint8_t wraw[kWinExam + kWinAhead + 48];     // the window's symbols (bytes): look-behind, own, look-ahead; loaded in
                                                 // aligned 16-byte pieces, so the first symbol sits at wraw[0 .. 15]
    uint16_t pl[kWin + 2];                       // piece starts (window-local: index into wb), ascending
    uint16_t ml[kWin];                           // indices (into pl) of the pieces of four and more symbols
    uint32_t tmp[16];
    uint32_t vars[8];
};
union SlowLds {
    MergeLds M;                                  // round-based merge of a long piece
    uint32_t arena[2 * kMergeArena];             // lane slots of the short pieces
};

// Round-based merge of the piece [q, q + n) of the symbol bytes `sp`, by the whole workgroup; `whole`: the piece is the
// entire pre-token (then the whole-string probe of bpe.go:27-33 comes first).  Tokens go to img[q ..], -1 behind them.
// Returns the number of tokens (0 when the scratch is too small: the host grows it and runs the batch again).
__device__ __noinline__ uint32_t rounds_piece(const Tables& T, const Workspace& W, SlowLds& S, const uint8_t* sp, uint64_t q, uint32_t n,
                                              bool whole, int32_t* img, uint32_t* tmp, unsigned long long* tmp64, Seg* segtmp,
                                              long long* lvars, uint8_t* xb) {
    const int tid = threadIdx.x;
    int32_t *sa, *sb;
    uint32_t *pa, *pb;
    __syncthreads();   // the union is free: everybody is done with the lane slots
    if (n <= (uint32_t)kSlowLds) {
        sa = S.M.sA; sb = S.M.sB; pa = S.M.pA; pb = S.M.pB;
    } else {
        if (tid == 0) {
            const unsigned long long need = (unsigned long long)n * 16ull;
            const unsigned long long off = atomicAdd(&W.status->scratch_used, need);
            lvars[2] = (off + need <= W.scratch_cap) ? (long long)off : -1ll;
            if (off + need > W.scratch_cap) atomicOr(&W.status->flags, kOvfScratch);
        }
        __syncthreads();
        const long long off = lvars[2];
        if (off < 0) return 0;   // (uniform)
        uint8_t* base = W.scratch + off;
        sa = reinterpret_cast<int32_t*>(base);
        sb = sa + n;
        pa = reinterpret_cast<uint32_t*>(sb + n);
        pb = pa + n;
    }
    for (uint32_t i = (uint32_t)tid; i < n; i += kSlowThreads) sa[i] = sp[q + i];
    __syncthreads();
    int32_t whole_id = -1;
    if (whole && n <= (uint32_t)kMaxTokenBytes) {  // bpe.go:27-33
        if ((uint32_t)tid < n) xb[tid] = (uint8_t)sa[tid];
        __syncthreads();
        if (tid == 0) lvars[3] = vocab_lookup(T, [&](uint32_t i) { return (uint32_t)xb[i]; }, n);
        __syncthreads();
        whole_id = (int32_t)lvars[3];
    }
    uint32_t n_out;
    if (whole_id >= 0) {
        n_out = 1;
        if (tid == 0) sa[0] = whole_id;
        __syncthreads();
    } else {
        for (uint32_t i = (uint32_t)tid; i < n; i += kSlowThreads)
            pa[i] = i + 1 < n ? T.bytepair[((uint32_t)sa[i] << 8) | (uint32_t)sa[i + 1]] : kNoPrio;
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < n; i += kSlowThreads) sa[i] = T.byte_tok[sa[i]];
        __syncthreads();
        n_out = merge_rounds<false>(T, sa, pa, sb, pb, n, tmp, tmp64, segtmp);
    }
    for (uint32_t i = (uint32_t)tid; i < n; i += kSlowThreads) img[q + i] = i < n_out ? (sa[i] & kIdMask) : -1;
    __syncthreads();
    return n_out;
}

// Tokens of a finished lane slot to the image: piece [q, q + len).  Returns their number.
__device__ __forceinline__ uint32_t slot_to_image(const uint32_t* sy, int32_t* img, uint64_t q, uint32_t len) {
    uint32_t p = 0, k = 0;
    while (p != kNone) {
        const uint32_t w = sy[p];
        img[q + k++] = (int32_t)(w & 0x1FFFFu);
        p = (w >> 17) & 127u;
    }
    const uint32_t cnt = k;
    for (; k < len; k++) img[q + k] = -1;
    return cnt;
}

// Shared state of slow_kernel's workgroup (LDS), handed to window_work()
struct SlowShared {
    SlowLds* S;
    PieceLds* P;
    const uint8_t* byte_tok;
    uint32_t* tmp;
    unsigned long long* tmp64;
    Seg* segtmp;
    long long* lvars;
    uint8_t* xb;
};

// One window: symbols [k * kWin, (k + 1) * kWin) of the L symbol bytes `sp` of a pre-token whose image is `img` and
// whose window counters are live[0 ..].  Whole workgroup.  Returns (uniformly) the number of tokens it wrote.
__device__ __forceinline__ uint32_t window_work(const Tables& T, const Workspace& W, const SlowShared& H, const uint8_t* sp, uint32_t L,
                                                uint32_t k, int32_t* img, uint32_t* live, const uint8_t* buf_lo, const uint8_t* buf_hi,
                                                uint32_t entry, uint32_t* first) {
    SlowLds& S = *H.S;
    PieceLds& P = *H.P;
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)lane_id(), wave = (uint32_t)wave_id();
    uint32_t* const sy_a = S.arena;
    uint32_t* const pq_a = S.arena + kMergeArena;
    const uint32_t lo = k * (uint32_t)kWin;
    const uint32_t hi = L - lo < (uint32_t)kWin ? L : lo + (uint32_t)kWin;
    const uint32_t back = lo < (uint32_t)kWinBack ? lo : (uint32_t)kWinBack;
    const uint32_t o0 = lo - back;                       // symbol position of wb[0]
    const uint32_t n_ex = hi - o0;                       // examined positions: wb[0 .. n_ex)
    const uint32_t avail = L - o0 < n_ex + (uint32_t)kWinAhead ? L - o0 : n_ex + (uint32_t)kWinAhead;
    __syncthreads();
    // the bytes, as aligned 16-byte loads wherever a whole piece lies inside the buffer [buf_lo, buf_hi)
    const uint32_t shift = (uint32_t)(((uintptr_t)(sp + o0)) & 15u);
    {
        const uint8_t* const al = sp + o0 - shift;
        const uint32_t n_chunks = (shift + avail + 15u) / 16u;
        for (uint32_t c = (uint32_t)tid; c < n_chunks; c += kSlowThreads) {
            const uint8_t* src = al + 16u * c;
            if (src >= buf_lo && src + 16 <= buf_hi) {
                *reinterpret_cast<uint4*>(&P.wraw[16u * c]) = *reinterpret_cast<const uint4*>(src);
            } else {
                for (uint32_t j = 0; j < 16u; j++) P.wraw[16u * c + j] = (src + j >= buf_lo && src + j < buf_hi) ? src[j] : (uint8_t)0;
            }
        }
    }
    const uint8_t* const wb = P.wraw + shift;   // wb[j] = symbol o0 + j, j < avail (what lies behind is unspecified)
    __syncthreads();
    // Match length at each of this thread's consecutive positions.  A walk is a chain of dependent table probes (is the
    // string one byte longer still a prefix of an entry?).  What the pass is short of is the L2's request rate (2.1e8
    // requests per C3 batch, most of them probes of the prefix filter, which no L1 holds), then issue slots -- not loads
    // in flight.  So: first the exact two-byte table for all positions (8 KB, it lives in the L1s); six positions of ten
    // end there.  Then lengths 3, 4 and 5 of all positions together -- the bytes are in LDS, the hashes of the longer
    // strings cost nothing to form ahead -- where a lane whose walk is over reads word 0 of the filter like all its kind:
    // one line, no request of its own.  Then, a position at a time, four more lengths per step for the lanes that matched
    // all five (a lane that is done sits the loop out: no idle probes).
    const uint32_t p0 = (uint32_t)tid * (uint32_t)kWinPer;
    uint32_t mlen[kWinPer], hh[kWinPer];
    uint32_t pass2 = 0, more = 0;
    {
        uint32_t t2[kWinPer];
#pragma unroll
        for (int j = 0; j < kWinPer; j++) {
            const uint32_t a = wb[p0 + (uint32_t)j], b = wb[p0 + (uint32_t)j + 1u];
            t2[j] = T.prefix2[(a << 3) | (b >> 5)] >> (b & 31u);
        }
#pragma unroll
        for (int j = 0; j < kWinPer; j++) {
            const uint32_t p = p0 + (uint32_t)j;
            const uint32_t room = p < n_ex ? avail - p : 0u;   // bytes from p to the end of what may be matched (>= 1)
            if (room >= 2 && (t2[j] & 1u)) pass2 |= 1u << j;
        }
    }
    {
        uint32_t w3[kWinPer], w4[kWinPer], w5[kWinPer];
#pragma unroll
        for (int j = 0; j < kWinPer; j++) {
            const uint32_t p = p0 + (uint32_t)j;
            const bool on = (pass2 >> j) & 1u;
            const uint32_t h2 = cut_hash2(wb[p], wb[p + 1]), h3 = cut_hash_add(h2, wb[p + 2]), h4 = cut_hash_add(h3, wb[p + 3]),
                           h5 = cut_hash_add(h4, wb[p + 4]);
            hh[j] = h5;
            const uint32_t b3 = cut_hash_bit(h3, T.bloom_log2), b4 = cut_hash_bit(h4, T.bloom_log2), b5 = cut_hash_bit(h5, T.bloom_log2);
            w3[j] = T.prefix_bloom[on ? b3 >> 5 : 0u] >> (b3 & 31u);
            w4[j] = T.prefix_bloom[on ? b4 >> 5 : 0u] >> (b4 & 31u);
            w5[j] = T.prefix_bloom[on ? b5 >> 5 : 0u] >> (b5 & 31u);
        }
#pragma unroll
        for (int j = 0; j < kWinPer; j++) {
            const uint32_t p = p0 + (uint32_t)j;
            const uint32_t room = p < n_ex ? avail - p : 0u;
            uint32_t l = 1;
            if ((pass2 >> j) & 1u) { l = 2; if (room >= 3 && (w3[j] & 1u)) { l = 3; if (room >= 4 && (w4[j] & 1u)) { l = 4; if (room >= 5 && (w5[j] & 1u)) l = 5; } } }
            mlen[j] = l;
            if (l == 5 && room > 5) more |= 1u << j;
        }
    }
#pragma unroll
    for (int j = 0; j < kWinPer; j++) {
        const uint32_t p = p0 + (uint32_t)j;
        bool ok = (more >> j) & 1u;
        while (ok) {   // four more lengths per step
            uint32_t w[4];
            uint32_t h = hh[j];
            const uint32_t at = p + mlen[j];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                h = cut_hash_add(h, wb[at + (uint32_t)t]);
                const uint32_t bit = cut_hash_bit(h, T.bloom_log2);
                w[t] = (T.prefix_bloom[bit >> 5] >> (bit & 31u)) & 1u;
            }
            uint32_t l = mlen[j];
            uint32_t lim = avail - p;
            if (lim > (uint32_t)kMaxTokenBytes) lim = (uint32_t)kMaxTokenBytes;
            int t = 0;
            while (t < 4 && l < lim && w[t]) { l++; t++; }
            mlen[j] = l;
            hh[j] = h;
            ok = t == 4 && l < lim;
        }
    }
    // furthest reach before each position (window-local): own earlier positions, earlier threads
    uint32_t own = 0, pre[kWinPer];
#pragma unroll
    for (int j = 0; j < kWinPer; j++) {
        pre[j] = own;
        const uint32_t p = p0 + (uint32_t)j;
        if (p < n_ex) { const uint32_t r = p + mlen[j]; own = r > own ? r : own; }
    }
    uint32_t inc = own;   // inclusive max-scan over the threads
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc = o > inc ? o : inc;
    }
    if (lane == 63) P.tmp[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) {
        const uint32_t x = P.tmp[w2];
        if (w2 < wave) before = x > before ? x : before;
        all = x > all ? x : all;
    }
    {
        const uint32_t ex = __shfl_up(inc, 1, 64);
        if (lane > 0) before = ex > before ? ex : before;
    }
    uint32_t cuts = 0;   // cuts at OWN positions (the look-behind only supplies reaches)
#pragma unroll
    for (int j = 0; j < kWinPer; j++) {
        const uint32_t p = p0 + (uint32_t)j;
        const uint32_t pm = pre[j] > before ? pre[j] : before;
        if (p >= back && p < n_ex && pm <= p) cuts |= 1u << j;
    }
    uint32_t k_total;
    const uint32_t kbase = block_escan<kSlowThreads / 64>((uint32_t)__popc(cuts), P.tmp + 8, &k_total);
    {
        uint32_t kk = kbase, c = cuts;
        while (c) { const int j = __builtin_ctz(c); c &= c - 1; P.pl[kk++] = (uint16_t)(p0 + (uint32_t)j); }
    }
    if (tid == 0) P.vars[0] = 0;   // length of the list ml (below)
    __syncthreads();
    if (tid == 0) first[k] = k_total ? o0 + (uint32_t)P.pl[0] : 0xFFFFFFFFu;   // (for pieces whose end is looked up later)
    if (k_total == 0) return 0;   // (uniform) the whole window lies inside a piece that started earlier
    // where the last piece ends: at the end of the pre-token, at the window's end if that is a cut, else at the first cut
    // behind it -- found by walking on, one position per thread and step.  After kTailSteps steps without a cut the
    // piece is LISTED for the grid-wide merge with its end left open (the later windows record their first cuts:
    // Workspace::win_first) -- walking a million blanks this way would take a hundred milliseconds.
    uint32_t last_end = n_ex;      // window-local, may lie far beyond the window
    bool last_deferred = false;
    if (hi < L && all > n_ex) {
        uint32_t carry = all;      // furthest reach of all positions before `pos` (window-local)
        uint32_t pos = n_ex;
        for (uint32_t step = 0;; step++) {
            if (step == (uint32_t)kTailSteps && W.allow_giants) {
                __syncthreads();
                if (tid == 0) {
                    const uint32_t g = atomicAdd(&W.status->giant.count, 1u);
                    H.lvars[6] = 0;
                    if (g < kMaxGiants) {
                        Giant G;
                        G.sp = sp; G.q = (uint64_t)o0 + P.pl[k_total - 1u]; G.n = 0; G.entry = entry; G.k = k; G.flat = 0; G.img = img; G.live = live;
                        W.giants[g] = G;
                        atomicOr(&W.status->flags, kGiantPending);
                        H.lvars[6] = 1;
                    }
                }
                __syncthreads();
                if (H.lvars[6]) { last_deferred = true; break; }   // (uniform; a full list: walk on)
            }
            const uint32_t p = pos + (uint32_t)tid;
            const uint32_t pa = o0 + p;   // symbol position
            uint32_t r = 0;
            if (pa < L) r = p + cut_match_len(T.prefix2, T.prefix_bloom, T.bloom_log2, [&](uint32_t i) { return (uint32_t)sp[pa + i]; }, L - pa);
            uint32_t inc2 = r;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc2, d, 64);
                if (lane >= (uint32_t)d) inc2 = o > inc2 ? o : inc2;
            }
            __syncthreads();
            if (lane == 63) P.tmp[wave] = inc2;
            __syncthreads();
            uint32_t bef = carry, al = carry;
            for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) {
                const uint32_t x = P.tmp[w2];
                if (w2 < wave) bef = x > bef ? x : bef;
                al = x > al ? x : al;
            }
            const uint32_t ex = __shfl_up(inc2, 1, 64);
            if (lane > 0) bef = ex > bef ? ex : bef;
            const bool cut = pa < L && bef <= p;
            const unsigned long long cm = __ballot(cut);
            if (lane == 0) P.tmp[8 + wave] = cm ? pos + wave * 64u + (uint32_t)__builtin_ctzll(cm) : 0xFFFFFFFFu;
            __syncthreads();
            uint32_t fc = 0xFFFFFFFFu;
            for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) fc = P.tmp[8 + w2] < fc ? P.tmp[8 + w2] : fc;
            if (fc != 0xFFFFFFFFu) { last_end = fc; break; }
            pos += kSlowThreads;
            if (o0 + pos >= L) { last_end = L - o0; break; }
            carry = al;
        }
        __syncthreads();
    }
    const uint32_t range_hi = lo + (uint32_t)kWin;   // slots from here on belong to the next window's counter
    const bool whole_single = !last_deferred && k_total == 1 && lo == 0 && last_end == L;   // the pre-token is one piece (o0 == 0)
    const uint32_t n_pieces = last_deferred ? k_total - 1u : k_total;   // (the listed one is not this kernel's any more)
    uint32_t own_cnt = 0, nxt_cnt = 0, long_cnt = 0;
    // single symbols need no merge: their token goes straight to the image; the others are listed
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < n_pieces; i0 += kSlowThreads) {
        const uint32_t i = i0 + (uint32_t)tid;
        const bool have = i < n_pieces;
        const uint32_t s = have ? P.pl[i] : 0u;
        const uint32_t e = have ? (i + 1 < k_total ? (uint32_t)P.pl[i + 1] : last_end) : 0u;
        const uint32_t plen = e - s;
        // Pieces of two and three symbols (half of all pieces of ideographic text, which falls into pieces of 1.9 symbols on
        // average) are merged here in registers: one rule for the first merge out of the byte-pair table, one probe for the
        // second.  The same steps lane_merge would take: smallest rank first, the left pair among equals.
        const bool quick = have && !whole_single && (plen == 2 || plen == 3);
        const bool multi = have && !quick && (plen >= 2 || whole_single);
        if (have && !multi && !quick) { img[(uint64_t)o0 + s] = H.byte_tok[wb[s]]; own_cnt++; }
        if (quick) {
            const uint32_t a = wb[s], b = wb[s + 1], c = wb[s + 2];   // (c: inside the LDS array in any case, used for three symbols)
            const uint32_t ta = H.byte_tok[a], tb = H.byte_tok[b], tc = H.byte_tok[c];
            const uint32_t r0 = T.bytepair[(a << 8) | b];
            const uint32_t r1 = plen == 3 ? T.bytepair[(b << 8) | c] : kNoPrio;
            uint32_t t0 = ta, t1 = tb, t2 = tc, cnt = plen;
            if (r0 != kNoPrio || r1 != kNoPrio) {
                const bool left = r0 <= r1;
                const uint32_t m = prio_merged_id(T, left ? r0 : r1);
                if (plen == 2) {
                    t0 = m; cnt = 1;
                } else {
                    const uint32_t x = left ? m : ta, y = left ? tc : m;
                    const uint32_t r2 = pair_prio(T, x, y);
                    if (r2 != kNoPrio) { t0 = prio_merged_id(T, r2); cnt = 1; } else { t0 = x; t1 = y; cnt = 2; }
                }
            }
            const uint64_t q = (uint64_t)o0 + s;   // (< range_hi: the piece starts at one of the window's own positions)
            img[q] = (int32_t)t0;
            img[q + 1] = cnt >= 2 ? (int32_t)t1 : -1;
            if (plen == 3) img[q + 2] = cnt == 3 ? (int32_t)t2 : -1;
            const uint32_t room = (uint32_t)((uint64_t)range_hi - q);
            const uint32_t mine = cnt < room ? cnt : room;
            own_cnt += mine; nxt_cnt += cnt - mine;
        }
        const unsigned long long mm = __ballot(multi);   // the list in any order: one LDS atomic per wave, no barrier
        if (mm) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&P.vars[0], (uint32_t)__popcll(mm));
            at = (uint32_t)__shfl((int)at, 0, 64);
            if (multi) P.ml[at + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const uint32_t n_multi = P.vars[0];
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < n_multi; i0 += kSlowThreads) {
        const bool have = i0 + (uint32_t)tid < n_multi;
        const uint32_t i = have ? P.ml[i0 + (uint32_t)tid] : 0u;
        const uint32_t s = have ? P.pl[i] : 0u;
        const uint32_t e = have ? (i + 1 < k_total ? (uint32_t)P.pl[i + 1] : last_end) : 0u;
        const uint32_t plen = e - s;
        const uint64_t q = (uint64_t)o0 + s;
        const bool is_long = have && (plen > (uint32_t)kLaneMax || whole_single);
        uint32_t cnt = 0;
        __syncthreads();   // the union is free (a rounds_piece of the previous chunk is over)
        // -- 2 .. 16 symbols: a 16-element slot per lane
        {
            const bool act = have && !is_long && plen <= 16;
            uint32_t* sy = sy_a + tid * 17;
            uint32_t* pq = pq_a + tid * 17;
            if (act) {
                uint32_t bb[16], pp[15];
#pragma unroll
                for (int j = 0; j < 16; j++) bb[j] = wb[s + (uint32_t)j];
#pragma unroll
                for (int j = 0; j < 15; j++) pp[j] = (uint32_t)(j + 1) < plen ? T.bytepair[(bb[j] << 8) | bb[j + 1]] : kNoPrio;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t nxt = (uint32_t)(j + 1) < plen ? (uint32_t)(j + 1) : kNone, prv = j ? (uint32_t)(j - 1) : kNone;
                    sy[j] = (uint32_t)H.byte_tok[bb[j]] | (nxt << 17) | (prv << 24);
                    pq[j] = (j < 15 && pp[j < 15 ? j : 0] != kNoPrio && (uint32_t)j < plen) ? (pp[j < 15 ? j : 0] << 7) | (uint32_t)j : 0xFFFFFFFFu;
                }
                lane_merge<16>(T, sy, pq);
                cnt = slot_to_image(sy, img, q, plen);
            }
        }
        // -- 17 .. kLaneMax symbols: groups of 64 lanes, 64-element slots of the same arena
        const bool is_mid = have && !is_long && plen > 16;
        const int n_mid = __syncthreads_count(is_mid);
        if (n_mid) {
            uint32_t tot;
            const uint32_t ord = block_escan<kSlowThreads / 64>(is_mid ? 1u : 0u, P.tmp, &tot);
#pragma unroll 1
            for (uint32_t g = 0; g * 64u < (uint32_t)n_mid; g++) {
                const bool act = is_mid && ord / 64u == g;
                uint32_t* sy = sy_a + (ord % 64u) * 65u;
                uint32_t* pq = pq_a + (ord % 64u) * 65u;
                if (act) {
                    uint32_t prev = wb[s];
                    for (uint32_t j = 0; j < 64; j++) {
                        if (j < plen) {
                            const uint32_t cur = j + 1 < plen ? (uint32_t)wb[s + j + 1] : 0u;
                            const uint32_t pr = j + 1 < plen ? T.bytepair[(prev << 8) | cur] : kNoPrio;
                            sy[j] = (uint32_t)H.byte_tok[prev] | ((j + 1 < plen ? j + 1 : kNone) << 17) | ((j ? j - 1 : kNone) << 24);
                            pq[j] = pr != kNoPrio ? (pr << 7) | j : 0xFFFFFFFFu;
                            prev = cur;
                        } else {
                            pq[j] = 0xFFFFFFFFu;
                        }
                    }
                    lane_merge<64>(T, sy, pq);
                    cnt = slot_to_image(sy, img, q, plen);
                }
                __syncthreads();
            }
        }
        if (cnt) {   // which window's slots hold them
            const uint32_t a = (uint64_t)range_hi > q ? ((uint64_t)range_hi - q < cnt ? (uint32_t)((uint64_t)range_hi - q) : cnt) : 0u;
            own_cnt += a; nxt_cnt += cnt - a;
        }
        // -- longer pieces of this chunk, one after the other by the whole workgroup
        const int n_long = __syncthreads_count(is_long);
        if (n_long) {
            const unsigned long long lm0 = __ballot(is_long);
            if (lane == 0) { P.tmp[wave * 2] = (uint32_t)lm0; P.tmp[wave * 2 + 1] = (uint32_t)(lm0 >> 32); }
            __syncthreads();
#pragma unroll 1
            for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) {
                unsigned long long lm = (unsigned long long)P.tmp[w2 * 2] | ((unsigned long long)P.tmp[w2 * 2 + 1] << 32);
                while (lm) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(lm);
                    lm &= lm - 1ull;
                    const uint32_t pi = P.ml[i0 + w2 * 64u + b];
                    const uint32_t ps = P.pl[pi], pe = pi + 1 < k_total ? (uint32_t)P.pl[pi + 1] : last_end;
                    const uint64_t pq0 = (uint64_t)o0 + ps;
                    if (pe - ps >= kGiantMin && W.allow_giants) {   // too long for one workgroup: listed for the grid-wide merge (if the list has room)
                        __syncthreads();
                        if (tid == 0) {
                            const uint32_t g = atomicAdd(&W.status->giant.count, 1u);
                            H.lvars[6] = 0;
                            if (g < kMaxGiants) {
                                Giant G;
                                G.sp = sp; G.q = pq0; G.n = pe - ps; G.entry = entry; G.k = k; G.flat = 0; G.img = img; G.live = live;
                                W.giants[g] = G;
                                atomicOr(&W.status->flags, kGiantPending);
                                H.lvars[6] = 1;
                            }
                        }
                        __syncthreads();
                        if (H.lvars[6]) continue;   // (uniform)
                    }
                    const uint32_t n_out = rounds_piece(T, W, S, sp, pq0, pe - ps, whole_single, img, H.tmp, H.tmp64, H.segtmp, H.lvars, H.xb);
                    long_cnt += n_out;
                    if (tid == 0) {   // its tokens may lie in the slots of several windows
                        uint64_t a = pq0;
                        const uint64_t end = pq0 + n_out;
                        while (a < end) {
                            const uint64_t r = a / (uint64_t)kWin, rh = (r + 1) * (uint64_t)kWin;
                            const uint64_t z = end < rh ? end : rh;
                            atomicAdd(&live[r], (uint32_t)(z - a));
                            a = z;
                        }
                    }
                }
            }
        }
    }
    // the tokens of this window's short pieces: own slots / the next window's slots
    __syncthreads();
    {
        uint32_t a = own_cnt, b = nxt_cnt;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
        if (lane == 0) { P.tmp[wave] = a; P.tmp[4 + wave] = b; }
    }
    __syncthreads();
    uint32_t a = 0, b = 0;
    for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) { a += P.tmp[w2]; b += P.tmp[4 + w2]; }
    if (tid == 0) {
        if (a) atomicAdd(&live[k], a);
        if (b) atomicAdd(&live[k + 1], b);
    }
    return a + b + long_cnt;
}

// Symbols of the bytes [lo, hi) of the pre-token [start, start + len) of the text: one per byte, EF BF BD for a byte that is
// not valid UTF-8 (state_machine.go:36).  Whole workgroup; every wave takes one contiguous range of the bytes, 64 at a time
// (order by ballots).  expand_count: the number of symbols, and in *wave_off the symbols in front of the calling wave's range;
// expand_write: the symbols themselves to xs[0 ..] (wave_off as expand_count gave it).  tmp: 4 words of LDS.
__device__ __forceinline__ void expand_range(uint32_t lo, uint32_t hi, uint32_t* b0, uint32_t* b1) {
    const uint32_t n = hi - lo, bw = ((n + kSlowThreads - 1) / kSlowThreads) * 64u, wave = (uint32_t)wave_id();
    *b0 = lo + (wave * bw < n ? wave * bw : n);
    *b1 = *b0 + bw < hi ? *b0 + bw : hi;
}
__device__ uint32_t expand_count(const Tables& T, const EncodeArgs& A, uint64_t start, uint32_t len, uint32_t lo, uint32_t hi,
                                 uint32_t* tmp, uint32_t* wave_off) {
    GlobalSrc gs{A.text, (int64_t)start, (int64_t)(start + len)};
    const uint32_t lane = (uint32_t)lane_id(), wave = (uint32_t)wave_id();
    uint32_t b0, b1;
    expand_range(lo, hi, &b0, &b1);
    uint32_t nsym = 0;
    for (uint32_t base = b0; base < b1; base += 64) {
        const uint32_t q = base + lane;
        const bool in = q < b1;
        const bool inv = in && (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID);
        nsym += (uint32_t)__popcll(__ballot(in)) + 2u * (uint32_t)__popcll(__ballot(inv));
    }
    __syncthreads();
    if (lane == 0) tmp[wave] = nsym;
    __syncthreads();
    uint32_t L = 0, so = 0;
    for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) {
        const uint32_t x = tmp[w2];
        if (w2 < wave) so += x;
        L += x;
    }
    *wave_off = so;
    return L;
}
__device__ void expand_write(const Tables& T, const EncodeArgs& A, uint64_t start, uint32_t len, uint32_t lo, uint32_t hi, uint8_t* xs,
                             uint32_t wave_off) {
    GlobalSrc gs{A.text, (int64_t)start, (int64_t)(start + len)};
    const uint32_t lane = (uint32_t)lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t b0, b1;
    expand_range(lo, hi, &b0, &b1);
    uint32_t o = wave_off;
    for (uint32_t base = b0; base < b1; base += 64) {
        const uint32_t q = base + lane;
        const bool in = q < b1;
        const bool inv = in && (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID);
        const unsigned long long mi = __ballot(in), mv = __ballot(inv);
        const uint32_t pos = o + (uint32_t)__popcll(mi & lt) + 2u * (uint32_t)__popcll(mv & lt);
        if (inv) {
            xs[pos] = 0xEF; xs[pos + 1] = 0xBF; xs[pos + 2] = 0xBD;
        } else if (in) {
            xs[pos] = A.text[start + q];
        }
        o += (uint32_t)__popcll(mi) + 2u * (uint32_t)__popcll(mv);
    }
}
constexpr uint32_t kSeqChunk = (uint32_t)kWin / 3u;   // bytes of a pre-token with invalid UTF-8 that one of its window items expands

// Waves per SIMD the register allocation aims at (the second launch bound).  With 1 the kernel took 206 VGPRs: two waves per
// SIMD, two workgroups per CU, whatever the grid (SQ_WAVE_CYCLES: 2,020 waves resident at 512 and at 1,024 workgroups), and
// its chains of dependent probes and barriers had nothing to hide behind.  3 = 168 VGPRs (128 B more scratch per lane) and
// three workgroups per CU, which is also what 46 KB of LDS each allow: C3's slow pass 2.00 -> 1.64 ms.  L3_SLOW_GRID (l3_kernels.h)
// is 3 x 256 accordingly.
#ifndef L3_SLOW_OCC
#define L3_SLOW_OCC 3
#endif
__global__ __launch_bounds__(kSlowThreads, L3_SLOW_OCC) void slow_kernel(Tables T, Workspace W, EncodeArgs A, int pass) {
    __shared__ __align__(16) SlowLds S;
    __shared__ __align__(16) PieceLds P;
    __shared__ uint8_t byte_tok[256];
    __shared__ uint32_t tmp[16];
    __shared__ unsigned long long tmp64[8];
    __shared__ Seg segtmp[4];
    __shared__ long long lvars[8];
    __shared__ int ivars[4];
    __shared__ uint8_t xb[kMaxTokenBytes];

    if (W.status->flags & (kFatalFlags | kOvfWin | kOvfPool)) return;   // (the window list may be incomplete then)
    const int tid = threadIdx.x;
    const uint32_t n_items = W.status->slow_q.count;  // final: the preparation pass has run
    if (blockIdx.x >= n_items) return;
    for (int i = tid; i < 256; i += kSlowThreads) byte_tok[i] = T.byte_tok[i];
    const SlowShared H{&S, &P, byte_tok, tmp, tmp64, segtmp, lvars, xb};
    bool first = true;  // the first item needs no counter: workgroup g starts at g
    // Window items differ in cost by orders of magnitude (a window inside a long piece is nothing, the window that owns
    // that piece merges all of it), so they are handed out one at a time from a counter.
#pragma unroll 1
    for (;;) {
        __syncthreads();
        if (tid == 0) ivars[1] = first ? (int)blockIdx.x : (int)(gridDim.x + atomicAdd(pass == 2 ? &W.status->exp_q.next : pass ? &W.status->seq_q.next : &W.status->slow_q.next, 1u));
        first = false;
        __syncthreads();
        if ((uint32_t)ivars[1] >= n_items) break;
        const WinItem it = W.win[(uint32_t)ivars[1]];
        const uint32_t ei = it.entry;
        const uint64_t start = W.slow[ei].start;
        const uint32_t len = W.slow[ei].len;
        const uint64_t win0 = W.slow[ei].win0;
        const bool sequential = (win0 & kEntrySequential) != 0;
        uint32_t* const live = W.win_live + (win0 & kEntryWinMask);
        uint32_t* const first = W.win_first + (win0 & kEntryWinMask);
        const uint64_t slots = sequential ? 3ull * len : (uint64_t)len;
        int32_t* const img = W.pool + W.slow[ei].pool_off + slots;   // the image lies behind the room for the final tokens
        uint32_t total = 0;
        // A long pre-token with invalid UTF-8 (its symbols are not its bytes) is expanded to a symbol array in the scratch by all
        // its window items, each a chunk of kSeqChunk bytes: counted in the first pass, written in the expansion pass (the
        // host starts it when the status word says kSeqPending), and merged window by window in the second pass.
        const bool spread = sequential && W.allow_giants && (slots + (uint64_t)kWin - 1) / (uint64_t)kWin > (uint64_t)kSeqInline;
        if (pass == 2) {   // expansion pass
            if (!spread) continue;
            uint32_t* const cnt = W.win_cnt + (win0 & kEntryWinMask);
            const uint32_t n_win = (uint32_t)((slots + (uint64_t)kWin - 1) / (uint64_t)kWin);
            unsigned long long before = 0, all = 0;   // symbols of the chunks in front of this one, of all chunks
            for (uint32_t j = (uint32_t)tid; j < n_win; j += kSlowThreads) {
                const uint32_t x = cnt[j];
                all += x;
                if (j < it.k) before += x;
            }
            before = block_sum64(before, tmp64);
            all = block_sum64(all, tmp64);
            const uint32_t lo = it.k * kSeqChunk < len ? it.k * kSeqChunk : len, hi = lo + kSeqChunk < len ? lo + kSeqChunk : len;
            uint32_t wave_off;
            (void)expand_count(T, A, start, len, lo, hi, tmp, &wave_off);
            expand_write(T, A, start, len, lo, hi, W.scratch + W.slow[ei].xoff + before, wave_off);
            if (tid == 0 && it.k == 0) {
                W.slow[ei].pad = (uint32_t)all;
                W.slow[ei].win0 = win0 | kEntryExpanded;
            }
            continue;
        }
        if (pass) {   // second pass: the windows of the pre-tokens that were expanded, one item each
            if (!(win0 & kEntryExpanded)) continue;
            const uint32_t L = W.slow[ei].pad;
            if ((uint64_t)it.k * (uint64_t)kWin >= (uint64_t)L) continue;
            const uint8_t* xs = W.scratch + W.slow[ei].xoff;
            total = window_work(T, W, H, xs, L, it.k, img, live, xs, xs + (((uint64_t)L + 31ull) & ~15ull), ei, first);
            if (tid == 0 && total) {
                atomicAdd(&W.slow[ei].count, total);
                atomicAdd(&W.tile_extra[W.slow[ei].tile], total);   // (the first pass has taken the one off)
            }
            continue;
        }
        if (!sequential) {
            total = window_work(T, W, H, A.text + start, len, it.k, img, live, A.text, A.text + A.n_bytes, ei, first);
            if (tid == 0 && it.k == 0) W.slow[ei].pad = len;   // its symbols
        } else {
            if (spread) {   // every window item counts the symbols of its chunk of the bytes; the first one reserves the scratch
                const uint32_t lo = it.k * kSeqChunk < len ? it.k * kSeqChunk : len, hi = lo + kSeqChunk < len ? lo + kSeqChunk : len;
                uint32_t wave_off;
                const uint32_t n_sym = expand_count(T, A, start, len, lo, hi, tmp, &wave_off);
                if (tid == 0) {
                    W.win_cnt[(win0 & kEntryWinMask) + it.k] = n_sym;
                    if (it.k == 0) {
                        const unsigned long long need = (3ull * len + 31ull) & ~15ull;   // (at most three symbols a byte)
                        const unsigned long long so2 = atomicAdd(&W.status->scratch_used, need);
                        if (so2 + need > W.scratch_cap) atomicOr(&W.status->flags, kOvfScratch);
                        W.slow[ei].xoff = so2;
                        atomicOr(&W.status->flags, kSeqPending);
                        atomicAdd(&W.tile_extra[W.slow[ei].tile], 0u - 1u);   // (the one its stream item stands for; the second pass adds the tokens)
                    }
                }
                continue;
            }
            if (it.k != 0) continue;   // a short one: ONE item expands it and works through its windows
            uint32_t so = 0;
            const uint32_t L = expand_count(T, A, start, len, 0, len, tmp, &so);
            if (tid == 0) {
                const unsigned long long need = ((unsigned long long)L + 31ull) & ~15ull;
                const unsigned long long so2 = atomicAdd(&W.status->scratch_used, need);
                lvars[5] = (so2 + need <= W.scratch_cap) ? (long long)so2 : -1ll;
                if (so2 + need > W.scratch_cap) atomicOr(&W.status->flags, kOvfScratch);
            }
            __syncthreads();
            if (lvars[5] < 0) continue;   // (uniform) the host grows the scratch and runs the batch again
            uint8_t* xs = W.scratch + lvars[5];
            expand_write(T, A, start, len, 0, len, xs, so);
            __syncthreads();
            const uint32_t nw = (L + (uint32_t)kWin - 1) / (uint32_t)kWin;
#pragma unroll 1
            for (uint32_t kk = 0; kk < nw; kk++) total += window_work(T, W, H, xs, L, kk, img, live, xs, xs + (((uint64_t)L + 31ull) & ~15ull), ei, first);
            if (tid == 0) W.slow[ei].pad = L;
        }
        if (tid == 0 && (total || it.k == 0)) {
            atomicAdd(&W.slow[ei].count, total);
            // tile_extra counts the tokens beyond the one the pre-token's stream item stands for (modulo 2^32: the tokens
            // of a giant piece are added later)
            atomicAdd(&W.tile_extra[W.slow[ei].tile], total - ((it.k == 0) ? 1u : 0u));
        }
    }
}

// The tokens of window item `w` from the image to their final place (the windows of a pre-token in order): whole

__global__ void status_and_kernel(Status* st, uint32_t mask) { st->flags &= mask; }

}  // namespace

hipError_t launch_stage_slow(const Tables& T, const Workspace& W, const EncodeArgs& A, uint32_t grid, int pass, hipStream_t stream) {
    hipLaunchKernelGGL(slow_kernel, dim3(grid), dim3(kSlowThreads), 0, stream, T, W, A, pass);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_slow_second_pass(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    hipLaunchKernelGGL(status_and_kernel, dim3(1), dim3(1), 0, stream, W.status, ~(uint32_t)kSeqPending);
    L3_LAUNCH_CHECK();
    uint64_t sgrid = W.slow_cap < L3_SLOW_GRID ? W.slow_cap : L3_SLOW_GRID;   // (the whole grid: the host knows there is work)
    if (sgrid == 0) sgrid = 1;
    hipLaunchKernelGGL(slow_kernel, dim3((uint32_t)sgrid), dim3(kSlowThreads), 0, stream, T, W, A, 2);   // expansion
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(slow_kernel, dim3((uint32_t)sgrid), dim3(kSlowThreads), 0, stream, T, W, A, 1);   // windows
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3
