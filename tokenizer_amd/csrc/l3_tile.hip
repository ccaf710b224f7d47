// l3_tile.hip -- docs_kernel and tile_kernel: the first two stages of the encode pipeline (overview: l3_pipeline.hip).
//
//   docs_kernel    doc_off sanity; first document of every tile; clears the per-tile counters
//   tile_kernel    one workgroup per 3,840 bytes of text (+ halos) staged in LDS:
//                  UTF-8 validity + character class per byte    (state_machine.go:36,310-320)
//                  special-token literals                       (special.go:12,48-83)
//                  position-local start-of-pre-token predicate  (state_machine.go:82-284, l3_rules.h)
//                  whole-pre-token vocabulary probe             (bpe.go:27-33)
//                  one stream item per pre-token: its id, or a placeholder of a deferred pre-token
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_dev.h"

namespace l3 {

namespace {

// number of consecutive Nd runes immediately before global position `pos` (= the start of tile `tile`) inside
// [seg_lo, seg_hi)  (whole workgroup; pathological digit runs longer than the look-behind halo only).
// Every tile of a long digit run asks this question, and answering it by walking back to the start of the run is
// quadratic in the run length.  So a tile leaves its answer in memo[tile] (1 + count; 0 = none), and a walk that
// arrives at the start of an earlier tile takes that tile's answer if it is there already: tiles are dispatched in
// order, so it nearly always is, and a walk is one tile long.  Nobody waits: a missing answer only means walking on.
__device__ uint32_t digits_before_global(const uint8_t* text, int64_t seg_lo, int64_t seg_hi, int64_t pos,
                                         const Tables& T, unsigned long long* tmp /* 8 words */, uint32_t* memo, uint64_t tile) {
    GlobalSrc gs{text, seg_lo, seg_hi};
    uint32_t total = 0;
    int64_t hi = pos;
    for (;;) {
        if (hi <= seg_lo) break;
        int64_t q = hi - 1 - (int64_t)threadIdx.x;
        bool valid = q >= seg_lo;
        uint32_t w = valid ? byte_info(gs, q, T.cls_stage1, T.cls_blocks) : 0u;
        bool rune = valid && (w & I_RUNE);
        bool isn = rune && (w & I_CLS) == C_N;
        bool brk = !valid || (rune && !isn);
        unsigned long long bm = __ballot(brk), nm = __ballot(isn);
        if (lane_id() == 0) { tmp[wave_id()] = bm; tmp[4 + wave_id()] = nm; }
        __syncthreads();
        bool found = false;
        for (int i = 0; i < kTileThreads / 64 && !found; i++) {
            unsigned long long b = tmp[i], m = tmp[4 + i];
            if (b) {
                int fb = __builtin_ctzll(b);
                total += __popcll(m & ((1ull << fb) - 1ull));
                found = true;
            } else {
                total += __popcll(m);
            }
        }
        __syncthreads();
        if (found) break;
        hi -= kTileThreads;
#ifndef L3_NO_DIGIT_MEMO
        if (hi > seg_lo && (pos - hi) % kTile == 0) {  // the start of an earlier tile of the same document
            if (threadIdx.x == 0)
                tmp[0] = __hip_atomic_load(&memo[tile - (uint64_t)((pos - hi) / kTile)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const uint32_t known = (uint32_t)tmp[0];
            __syncthreads();
            if (known) { total += known - 1u; break; }
        }
#endif
    }
    if (threadIdx.x == 0) __hip_atomic_store(&memo[tile], total + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return total;
}

// ---- validate the offsets, first document per tile, clear the per-tile counters ------------------------
// One thread per document offset.  A thread whose own offset is in order writes the first-document entries of the
// tiles that start in (previous offset, offset]; the clamp keeps every write in bounds whatever the other offsets
// are, and all later kernels return at once when kBadOffsets is set.
__global__ void docs_kernel(Workspace W, EncodeArgs A, uint64_t n_tiles) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t t = gid; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) { W.tile_extra[t] = 0; W.tile_digits[t] = 0; }
    const uint64_t d = gid;
    if (d > A.n_docs) return;
    const uint64_t off = A.doc_off[d];
    const uint64_t prev = d ? A.doc_off[d - 1] : 0;
    const bool bad = (d == 0 && off != 0) || (d == A.n_docs && off != A.n_bytes) || off < prev || off > A.n_bytes;
    if (bad) { atomicOr(&W.status->flags, kBadOffsets); return; }
    const uint64_t t_first = d ? prev / kTile + 1 : 0;   // tiles whose start lies in (prev, off]
    uint64_t t_last = off / kTile;
    if (t_last > n_tiles - 1) t_last = n_tiles - 1;
    for (uint64_t t = t_first; t <= t_last; t++) W.tile_doc0[t] = (uint32_t)d;
    if (d == A.n_docs) W.tile_doc0[n_tiles] = (uint32_t)(A.n_docs + 1);
}

// ---- the tile kernel ---------------------------------------------------------------------------------
// Bit-parallel formulation: every property of a byte (rune start, class, contraction letter, ...) is
// one bit in a 64-bit word per 64 bytes of the region; the start-of-pre-token predicate of
// SURVEY.md App. B then is a few dozen 64-bit logic operations per 64 bytes, evaluated by one lane per
// word (one wave per tile), with neighbouring words fetched by wave shuffles.
constexpr int kWords = kRegion / 64;
static_assert(kWords * 64 == kRegion && kWords <= 64, "one wave must hold one mask word per lane");
// rows from B_LIVE on are dead once the start bits exist; the lookup phase reuses them as its item staging array
enum { B_ST = 0, B_SPS, B_INV, B_DS, B_LIVE, B_RS = B_LIVE, B_L, B_N, B_NL, B_SP, B_WS, B_APOS, B_NX, B_CE2, B_CE3,
       B_SPB, B_SPE, B_CONTN, B_COUNT };

// The bitmap rows from B_LIVE on, reused by the lookup phase: a list of the pre-tokens its first stage does not settle
// (16-bit list indices; from the front those that missed their first bucket, from the back the rare kinds) and the notes
// of the deferred ones (32-bit).  A tile that overflows either starts over in the general way (see the kernel).
constexpr int kLiveBytes = (B_COUNT - B_LIVE) * (kWords + 2) * 8;
constexpr int kLstCap = 1920, kDeferCap = 756;
static_assert(kLstCap * 2 + kDeferCap * 4 <= kLiveBytes && (kLstCap * 2) % 4 == 0, "work lists of the lookup phase must fit the dead bitmap rows");
constexpr uint32_t kOpenEnd = 0xFFFFu;   // TileLds::list[n_pre]: the last own pre-token ends beyond what the halo decides
// Short deferred pre-tokens are binned by length: shard = length class * 4 + (tile & 3), class = min(len, 16) - 1.
// The 64 lanes of a merge wave then walk merge chains of about the same length (the chain of a wave is that of
// its longest lane), which cut the merge kernel's time by a third; (tile & 3) spreads a class's atomics over
// four counters.
constexpr int kLenClasses = 16;
static_assert(kLenClasses * 4 == kMissShards, "shard = length class * 4 + (tile & 3)");
__device__ __forceinline__ uint32_t len_class(uint32_t len) { return (len < (uint32_t)kLenClasses ? len : (uint32_t)kLenClasses) - 1u; }
#ifndef L3_LIST_CAP
#define L3_LIST_CAP (kTile + 2)
#endif
struct TileLds {
    uint8_t b[kRegion + 16];
    u64 bm[B_COUNT][kWords + 2];
    union {
        uint16_t list[L3_LIST_CAP];  // compacted own pre-token starts (region-local offsets) + the end of the last one
        uint2 ctab[256];       // before that: class bits of every byte value, see the classification step
    };
    uint16_t wpre[kWords + 1]; // own pre-token starts before each mask word
};
constexpr int kListPer = 16;
static_assert(kListPer * kTileThreads >= kTile && 64 % kListPer == 0 && kBack % kListPer == 0, "list compaction geometry");
enum { TV_NPRE = 0, TV_NEED, TV_MB, TV_NSLOW, TV_SLOW0, TV_NAB, TV_ANYINV, TV_NDEF, TV_OWNINV, TV_ANYSPS, TV_RARE, TV_COUNT = 12 };

struct TileSrc {  // byte source over the tile's LDS image, LOCAL coordinates [0, n)
    const TileLds* R;
    int n;
    bool hard;
    __device__ uint32_t b(int64_t q) const { return R->b[q]; }
    __device__ bool ds(int64_t q) const { return (q == n && hard) || ((R->bm[B_DS][q >> 6] >> (q & 63)) & 1ull); }
    __device__ int64_t lo() const { return 0; }
    __device__ int64_t hi() const { return n; }
};


// Start-of-pre-token bits of the whole region, by the lanes of ONE wave (lane w = mask word w).
// `edge_digits`: number of Nd runes immediately before the tile start when a digit run fills the
// whole look-behind halo (only read when `use_edge`).  Writes bm[B_ST], bm[B_N], bm[B_CONTN], wpre[];
// returns flags: bit0 a digit run reaches the region start (edge count needed), bit1 multi-byte digits, bit2 the tile's
// own bytes hold invalid UTF-8.
__device__ __noinline__ uint32_t compute_starts(TileLds& R, int n, int own_lo, int own_hi, int eval_hi, bool region_at_text_start,
                                                bool use_edge, uint32_t edge_digits, int* n_pre_out) {
    const int w = lane_id();
    const bool act = w < kWords;
#define L3_LD(prop) (act ? R.bm[prop][w] : 0ull)
    const u64 SPB = L3_LD(B_SPB), nspb = ~SPB;
    const u64 RS = L3_LD(B_RS) & nspb;
    const u64 L = L3_LD(B_L) & nspb, N = L3_LD(B_N) & nspb, NL = L3_LD(B_NL) & nspb, SP = L3_LD(B_SP) & nspb,
              WS = L3_LD(B_WS) & nspb;
    const u64 P = RS & ~(L | N | NL | SP | WS);
    const u64 SEG = L3_LD(B_DS) | L3_LD(B_SPE);
    // CE2 / CE3: a letter pattern of a contraction ('s 't 'm 'd | 're 've 'll) ends right before this byte
    const u64 APOS = L3_LD(B_APOS) & nspb, CE2 = L3_LD(B_CE2), CE3 = L3_LD(B_CE3);
    const u64 NX = L3_LD(B_NX) & nspb;
    const u64 cont = ~RS;
    // Does the region hold any byte that is not a rune start (continuation bytes, special-token bodies)?  Pure
    // ASCII text does not, and the three smearing loops below (each three rounds of wave shuffles) have nothing
    // to do: uniform over the wave, so they are skipped.
    bool multi;
    {
        const int lo = w * 64;
        const u64 rng = n <= lo ? 0ull : (n >= lo + 64 ? ~0ull : (1ull << (n - lo)) - 1ull);
        multi = __ballot(act && (cont & rng) != 0) != 0;
    }
    // class of the rune that CONTAINS each byte (classes smeared over continuation bytes)
    u64 sL = L, sN = N, sWS = WS, sP = P;
#pragma unroll
    for (int it = 0; it < 3 && multi; it++) {
        uint32_t t = (uint32_t)(sL >> 63) | ((uint32_t)(sN >> 63) << 1) | ((uint32_t)(sWS >> 63) << 2) | ((uint32_t)(sP >> 63) << 3);
        uint32_t u = up32(t);
        sL |= ((sL << 1) | (u & 1u)) & cont;
        sN |= ((sN << 1) | ((u >> 1) & 1u)) & cont;
        sWS |= ((sWS << 1) | ((u >> 2) & 1u)) & cont;
        sP |= ((sP << 1) | ((u >> 3) & 1u)) & cont;
    }
    // class of the previous rune, at every rune start
    u64 p_L, p_N, p_NL, p_SP, p_WS, p_P;
    {
        uint32_t t = (uint32_t)(sL >> 63) | ((uint32_t)(sN >> 63) << 1) | ((uint32_t)(NL >> 63) << 2) | ((uint32_t)(SP >> 63) << 3) |
                     ((uint32_t)(sWS >> 63) << 4) | ((uint32_t)(sP >> 63) << 5);
        uint32_t u = up32(t);
        p_L = (sL << 1) | (u & 1u);
        p_N = (sN << 1) | ((u >> 1) & 1u);
        p_NL = (NL << 1) | ((u >> 2) & 1u);
        p_SP = (SP << 1) | ((u >> 3) & 1u);
        p_WS = (sWS << 1) | ((u >> 4) & 1u);
        p_P = (sP << 1) | ((u >> 5) & 1u);
    }
    // apostrophe that can start a contraction: at a segment start or after L / N / newline / other white space
    const u64 AS = APOS & (SEG | p_L | p_N | p_NL | p_WS);
    // punctuation rune preceded (in its segment) by punctuation or ' ': a letter after it starts a token
    u64 sQ = P & ~SEG & (p_P | p_SP);
#pragma unroll
    for (int it = 0; it < 3 && multi; it++) {
        uint32_t u = up32((uint32_t)(sQ >> 63));
        sQ |= ((sQ << 1) | (u & 1u)) & cont;
    }
    u64 p_Q;
    u64 AS2, AS3, SEG1, SEG2;
    {
        // top three bits of the previous word of every mask that is shifted by up to 3
        uint32_t t = (uint32_t)(AS >> 61) | ((uint32_t)(SEG >> 61) << 3) | ((uint32_t)(sQ >> 61) << 6);
        uint32_t u = up32(t);
#define L3_SHL(X, k, f) (((X) << (k)) | (u64)((((u) >> (f)) & 7u) >> (3 - (k))))
        AS2 = L3_SHL(AS, 2, 0); AS3 = L3_SHL(AS, 3, 0);
        SEG1 = L3_SHL(SEG, 1, 3); SEG2 = L3_SHL(SEG, 2, 3);
        p_Q = L3_SHL(sQ, 1, 6);
#undef L3_SHL
    }
    const u64 ce2 = AS2 & CE2 & ~SEG1;
    const u64 ce3 = AS3 & CE3 & ~SEG2 & ~SEG1;
    const u64 startL = L & (p_N | p_NL | p_Q | (p_L & (ce2 | ce3)));
    const u64 startP = P & (p_L | p_N | p_WS | p_NL);
    // white space: the look-ahead "followed by a non-space rune of the same segment"
    const u64 Wany = NL | SP | WS;
    u64 B;
    {
        const u64 NWS = RS & ~SEG & ~Wany;
        B = (NWS >> 1) | ((u64)(dn32((uint32_t)NWS) & 1u) << 63);
#pragma unroll
        for (int it = 0; it < 3 && multi; it++) {
            u64 x = B & cont;
            B |= (x >> 1) | ((u64)(dn32((uint32_t)x) & 1u) << 63);
        }
    }
    const u64 pws = p_NL | p_SP | p_WS;
    const u64 startW = Wany & (p_L | p_N | (p_P & ~NL) | (pws & ~NL & (p_NL | B)));
    // digits: groups of three inside every run of Nd runes
    const u64 contN = N & p_N & ~SEG;
    const bool mb = __ballot(NX != 0) != 0;
    uint32_t flags = mb ? 2u : 0u;
    u64 DST = 0;
    if (!mb) {
        u64 c2 = contN;
        if (w == 0 && !region_at_text_start && !(SEG & 1ull)) {
            c2 |= N & 1ull;  // unknown: assume linked
            // The region may begin inside a rune whose first bytes lie before it (up to three continuation bytes):
            // that rune may be a multi-byte digit, so a digit right after those bytes is assumed linked as well --
            // the walk back through global memory then finds out
            int k = 0;
            while (k < 3 && (R.b[k] & 0xC0u) == 0x80u && !((SEG >> (k + 1)) & 1ull)) k++;
            if (k && ((N >> k) & 1ull)) c2 |= (2ull << k) - 1ull;
        }
        Seg sg;
        if (!(N >> 63)) { sg.cnt = 0; sg.open = 0; }
        else {
            u64 t = ~c2;
            if (t == 0) { sg.cnt = 64; sg.open = 1; }
            else { sg.cnt = (uint32_t)__builtin_clzll(t) + 1u; sg.open = 0; }
        }
        Seg c = wave_seg_escan(sg);
        const bool linked0 = (contN & 1ull) != 0;
        const bool from_edge = linked0 && c.open && w >= 1 && !region_at_text_start;
        if (__ballot(from_edge) != 0) flags |= 1u;
        uint32_t carry = linked0 ? c.cnt : 0u;
        if (from_edge && use_edge) carry = edge_digits + 64u * (uint32_t)(w - 1);  // own_lo is bit 0 of word 1
        u64 rem = N;
        while (rem) {
            const int s = __builtin_ctzll(rem);
            const int extra = s == 63 ? 0 : __builtin_ctzll(~(contN >> (s + 1)));
            const int len = 1 + (extra > 63 - s ? 63 - s : extra);
            const uint32_t o = (s == 0 && linked0) ? carry % 3u : 0u;
            const int first = s + (int)((3u - o) % 3u);
            const u64 runmask = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << s;
            if (first < 64) DST |= (0x9249249249249249ull << first) & runmask;
            rem &= ~runmask;
        }
    }
    const u64 startN = N & (~p_N | DST);
    u64 ST = RS & (SEG | startL | startP | startW | startN);
    // keep only decidable positions
    {
        const int lo = w * 64;
        if (eval_hi <= lo) ST = 0;
        else if (eval_hi < lo + 64) ST &= (1ull << (eval_hi - lo)) - 1ull;
    }
    if (act) {
        R.bm[B_ST][w] = ST;
        R.bm[B_N][w] = N;
        R.bm[B_CONTN][w] = contN;
    }
    // own starts before every word
    u64 own = ST;
    {
        const int lo = w * 64;
        if (own_lo >= lo + 64 || own_hi <= lo) own = 0;
        else {
            if (own_lo > lo) own &= ~0ull << (own_lo - lo);
            if (own_hi < lo + 64) own &= (1ull << (own_hi - lo)) - 1ull;
        }
    }
    {   // invalid UTF-8 among the bytes the tile OWNS (the halos cut characters in two: their stray bytes do not count)
        u64 inv = L3_LD(B_INV);
        const int lo = w * 64;
        if (own_lo >= lo + 64 || own_hi <= lo) inv = 0;
        else {
            if (own_lo > lo) inv &= ~0ull << (own_lo - lo);
            if (own_hi < lo + 64) inv &= (1ull << (own_hi - lo)) - 1ull;
        }
        if (__ballot(inv != 0) != 0) flags |= 4u;
    }
    uint32_t cnt = (uint32_t)__popcll(own);
    uint32_t inc = wave_iscan(cnt);
    if (act) R.wpre[w] = (uint16_t)(inc - cnt);
    uint32_t total = __shfl(inc, 63, 64);
    *n_pre_out = (int)total;
#undef L3_LD
    return flags;
}

// kOnePretoken (BPE.EncodeBPE): every document is one pre-token, so the start bits are the document starts and nothing
// else.  Same outputs as compute_starts (bm[B_ST], wpre[], the number of own starts), by the lanes of one wave.
__device__ __noinline__ int document_starts_only(TileLds& R, int own_lo, int own_hi, int eval_hi) {
    const int w = lane_id();
    const bool act = w < kWords;
    const int lo = w * 64;
    u64 ST = act ? R.bm[B_DS][w] : 0ull;
    if (eval_hi <= lo) ST = 0;
    else if (eval_hi < lo + 64) ST &= (1ull << (eval_hi - lo)) - 1ull;
    if (act) R.bm[B_ST][w] = ST;
    u64 own = ST;
    if (own_lo >= lo + 64 || own_hi <= lo) own = 0;
    else {
        if (own_lo > lo) own &= ~0ull << (own_lo - lo);
        if (own_hi < lo + 64) own &= (1ull << (own_hi - lo)) - 1ull;
    }
    const uint32_t cnt = (uint32_t)__popcll(own);
    const uint32_t inc = wave_iscan(cnt);
    if (act) R.wpre[w] = (uint16_t)(inc - cnt);
    return (int)__builtin_amdgcn_readlane((int)inc, 63);
}

// One pre-token [s, e) (e < 0: it ends beyond what the halo decides) the long way: special token, whole-string probe of any
// length (bpe.go:27-33), or deferral.  Returns 0 and the token id in *item, or the kind of deferral (*item untouched): 1 = up
// to kLaneMax valid bytes (a lane of the merge kernel), 2 = slow, 3 = slow and known to be valid UTF-8.
__device__ __forceinline__ uint32_t tile_generic_item(const Tables& T, const TileLds& R, const TileSrc& src, const uint32_t* bw, int s,
                                                      int e, int32_t* item) {
    if ((R.bm[B_SPS][s >> 6] >> (s & 63)) & 1ull) {
        int canon = 0;
        special_match(src, s, &canon);
        *item = T.special_id[canon];
        return 0u;
    }
    const int len = e - s;
    const bool valid = e >= 0 && !any_bit_in(R.bm[B_INV], s, e);
    if (valid && len <= kMaxTokenBytes) {
        const uint32_t* wp = bw + (s >> 2);
        const uint32_t sh = (uint32_t)(s & 3) * 8u;
        const uint32_t nw = ((uint32_t)len + 3u) >> 2;
        const uint32_t tail = (len & 3) ? (1u << ((len & 3) * 8)) - 1u : 0xFFFFFFFFu;
        const int32_t id = vocab_lookup_words(T, [&](uint32_t i) {
            uint32_t x = __funnelshift_r(wp[i], wp[i + 1], sh);
            return i + 1 == nw ? x & tail : x;
        }, (uint32_t)len);
        if (id >= 0) { *item = id; return 0u; }
    }
    if (valid && len <= kLaneMax) return 1u;
    return valid ? 3u : 2u;
}

// Entry record of the deferred pre-token k of a tile (list[k] .. list[k + 1]) in the global lists; returns its stream item.
// slow: entry s0 + ord of the slow list; else entry cm0[c] + ord of the tile's shard of length class c.
__device__ __forceinline__ int32_t tile_deferred_record(const Workspace& W, const TileLds& R, const uint32_t* bw, const int* cm0,
                                                        uint64_t g_lo, uint64_t tile, uint32_t sub, uint32_t s0, uint32_t k, bool slow,
                                                        uint32_t ord, uint32_t c, uint32_t valid) {
    const uint32_t s = R.list[k], ev = R.list[k + 1u];
    uint64_t ei;
    if (slow) {
        ei = (uint64_t)s0 + ord;
        if (ei < W.slow_cap) {
            SlowEntry se;
            se.start = g_lo + (uint64_t)s;
            se.win0 = 0;
            se.len = ev == kOpenEnd ? 0u : ev - s;
            se.tile = (uint32_t)tile;
            se.count = valid;   // in: 1 = known to be valid UTF-8
            se.pad = 0; se.pool_off = 0; se.xoff = 0;
            W.slow[ei] = se;
        }
        ei += kSlowBase;
    } else {
        const uint64_t li = (uint64_t)(uint32_t)cm0[c] + ord;
        ei = (uint64_t)(c * 4u + sub) * W.miss_shard_cap + li;
        if (li < W.miss_shard_cap) {
            MissIn mi;
            mi.start = g_lo + (uint64_t)s;
            mi.tile = (uint32_t)tile;
            mi.len = ev - s;
            {   // the first 16 bytes ride along (LDS image, unaligned start), so the merge kernel reads no text
                const uint32_t* wp = bw + (s >> 2);
                const uint32_t sh = s * 8u;
                const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2], x3 = wp[3], x4 = wp[4];
                mi.w[0] = __funnelshift_r(x0, x1, sh); mi.w[1] = __funnelshift_r(x1, x2, sh);
                mi.w[2] = __funnelshift_r(x2, x3, sh); mi.w[3] = __funnelshift_r(x3, x4, sh);
            }
            W.miss_in[ei] = mi;
        }
    }
    return -2 - (int32_t)(uint32_t)ei;
}

// Class bits of the 256 byte values for the table step of the tile kernel (state_machine.go:310-320 for ASCII: letters, Nd
// digits, unicode.IsSpace), one class per byte of two words: x = ASCII | letter << 8 | digit << 16 | newline << 24,
// y = blank | other white space << 8 | apostrophe << 16 | '<' << 24.  Bytes >= 0x80 have an all-zero entry.  A constant of
// the code object: every tile copies it into LDS with one coalesced load.
struct AsciiCls { uint32_t v[512]; };
constexpr AsciiCls make_ascii_cls() {
    AsciiCls t{};
    for (uint32_t v = 0; v < 256; v++) {
        const uint32_t f = v | 0x20u;
        const uint32_t asc = v < 0x80u ? 1u : 0u;
        const uint32_t cl = (f >= 'a' && f <= 'z') ? asc : 0u;
        const uint32_t cn = (v >= '0' && v <= '9') ? 1u : 0u;
        const uint32_t cnl = (v == '\n' || v == '\r') ? 1u : 0u;
        const uint32_t csp = v == ' ' ? 1u : 0u;
        const uint32_t cws = (v >= 9u && v <= 13u && !cnl) ? 1u : 0u;
        const uint32_t cap = v == '\'' ? 1u : 0u;
        const uint32_t clt = v == '<' ? 1u : 0u;
        t.v[2 * v] = asc | (cl << 8) | (cn << 16) | (cnl << 24);
        t.v[2 * v + 1] = csp | (cws << 8) | (cap << 16) | (clt << 24);
    }
    return t;
}
__device__ const AsciiCls g_ascii_cls = make_ascii_cls();

#ifndef L3_TILE_OCC
#define L3_TILE_OCC 7
#endif
#ifndef L3_LOOK_U
#define L3_LOOK_U 4
#endif
// -DL3_PHASE_PROF (tools/build_variant.sh): wall cycles of every phase of the tile kernel, summed over the workgroups of all
// launches since the last reset, read with l3_debug_phases().  Diagnostic builds only.
#ifdef L3_PHASE_PROF
__device__ unsigned long long g_phase[16];
#define L3_PH_INIT() long long ph_prev = clock64()
#define L3_PH(i)                                                                   \
    do {                                                                           \
        if (threadIdx.x == 0) {                                                    \
            const long long ph_now = clock64();                                    \
            atomicAdd(&g_phase[i], (unsigned long long)(ph_now - ph_prev));        \
            ph_prev = ph_now;                                                      \
        }                                                                          \
    } while (0)
#else
#define L3_PH_INIT() do {} while (0)
#ifdef L3_TILE_EXIT
// -DL3_TILE_EXIT=i (tools/build_variant.sh): the tile kernel ends behind phase i with a well-formed EMPTY tile (no items, every
// document of the tile at item 0), so that the kernels behind it run on.  Differences of SQ_INSTS_* between consecutive
// variants are the instruction counts of the phases (profiles/r03_tile_budget.md).  Diagnostic builds only.
#define L3_PH(i)                                                                                                   \
    do {                                                                                                           \
        if (L3_TILE_EXIT == (i)) {                                                                                 \
            if (threadIdx.x == 0) { W.tile_items[blockIdx.x] = 0; W.tile_first[blockIdx.x] = kNoFirst; }           \
            const uint64_t de_ = blockIdx.x + 1 == gridDim.x ? A.n_docs + 1 : (uint64_t)W.tile_doc0[blockIdx.x + 1]; \
            for (uint64_t d_ = (uint64_t)W.tile_doc0[blockIdx.x] + threadIdx.x; d_ < de_; d_ += kTileThreads) W.doc_item[d_] = 0; \
            return;                                                                                                \
        }                                                                                                          \
    } while (0)
#else
#define L3_PH(i) do {} while (0)
#endif
#endif
__global__ __launch_bounds__(kTileThreads, L3_TILE_OCC) void tile_kernel(Tables T, Workspace W, EncodeArgs A) {
    __shared__ __align__(16) TileLds R;
    __shared__ unsigned long long tmp64[8];
    __shared__ int vars[TV_COUNT];
    __shared__ int ccnt[kLenClasses], cm0[kLenClasses];   // short deferred pre-tokens per length class; their first entry
    __shared__ uint4 mk[kVocabInline + 1];
#ifdef L3_LDS_PAD   // occupancy experiments (tools/build_kvariant.sh): extra LDS per workgroup
    __shared__ uint32_t lds_pad[L3_LDS_PAD / 4];
    asm volatile("" :: "v"(&lds_pad[threadIdx.x]) : "memory");
#endif

    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    L3_PH_INIT();
    const uint64_t tile = blockIdx.x;
    const uint64_t t_lo = tile * (uint64_t)kTile;
    const uint64_t t_hi = t_lo + kTile < A.n_bytes ? t_lo + kTile : A.n_bytes;
    const uint64_t g_lo = t_lo >= (uint64_t)kBack ? t_lo - kBack : 0;
    const uint64_t g_hi = t_lo + kTile + kFwd < A.n_bytes ? t_lo + kTile + kFwd : A.n_bytes;
    const int n = (int)(g_hi - g_lo);
    const int own_lo = (int)(t_lo - g_lo), own_hi = (int)(t_hi - g_lo);
    const bool hard = g_hi == A.n_bytes;
    const int eval_hi = hard ? n : n - kEvalSlack;

    // -- the loads a tile starts with are requested together (its own 16 bytes of text, the documents of the tile, the
    //    status word, the class table entry); the LDS set-up below runs while they are in flight.  Text: g_lo and c0 are
    //    multiples of 16, so with a 16-byte aligned batch every thread makes ONE aligned 16-byte load, also where fewer than
    //    16 bytes remain (an aligned 16-byte piece lies inside one page: the load cannot fault; the surplus is masked below).
    const int c0 = tid * 16;            // region bytes [c0, c0 + 16) are classified by this thread
    const bool cls_thread = tid < kRegion / 16;
    const bool cls_in = cls_thread && c0 < n;
    const bool text_al = (((uintptr_t)A.text) & 15u) == 0;   // uniform
    uint4 xv = make_uint4(0, 0, 0, 0);
    if (cls_in && text_al) xv = *reinterpret_cast<const uint4*>(A.text + g_lo + c0);
    const uint32_t d0 = W.tile_doc0[tile];
    const uint32_t d_end32 = W.tile_doc0[tile + 1];   // (the last tile's entry is n_docs + 1: docs_kernel)
    const uint32_t status_flags = W.status->flags;
    const uint2 cls_entry = reinterpret_cast<const uint2*>(g_ascii_cls.v)[tid];   // (256 threads, 256 entries)

    // -- clear what is accumulated with atomics
    for (int i = tid; i < kWords + 2; i += kTileThreads) {
        R.bm[B_DS][i] = 0; R.bm[B_SPS][i] = 0; R.bm[B_SPB][i] = 0; R.bm[B_SPE][i] = 0; R.bm[B_CE2][i] = 0; R.bm[B_CE3][i] = 0;
    }
    if (tid == 0) { vars[TV_NSLOW] = 0; vars[TV_NAB] = 0; vars[TV_NDEF] = 0; vars[TV_ANYINV] = 0; vars[TV_ANYSPS] = 0; vars[TV_RARE] = 0; }
    if (tid <= kVocabInline) {   // mask of the first `tid` bytes of three words + the length tag of a vocabulary slot
        const uint32_t l = (uint32_t)tid;
        mk[tid] = make_uint4(l >= 4u ? 0xFFFFFFFFu : (1u << (8u * l)) - 1u, l >= 8u ? 0xFFFFFFFFu : (l > 4u ? (1u << (8u * (l - 4u))) - 1u : 0u),
                             l >= 12u ? 0xFFFFFFFFu : (l > 8u ? (1u << (8u * (l - 8u))) - 1u : 0u), l << 17);
    }
    if (tid < kLenClasses) ccnt[tid] = 0;
    // (the loaded values are first looked at here, behind the set-up)
    asm volatile("" : "+v"(xv.x), "+v"(xv.y), "+v"(xv.z), "+v"(xv.w));
    R.ctab[tid] = cls_entry;
    if (status_flags & kBadOffsets) return;   // (uniform; tile_doc0 is only trustworthy without it)
    uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
    if (cls_in && (!text_al || c0 + 16 > n)) {   // rare: a batch that is not 16-byte aligned, the last bytes of the text
        if (!text_al) {
            const uint8_t* srcp = A.text + g_lo + c0;
            for (int k = 0; k < 16 && c0 + k < n; k++) xw[k >> 2] |= (uint32_t)srcp[k] << ((k & 3) * 8);
        } else {
            const int nb = n - c0;   // 1..15 bytes are text
            for (int j = 0; j < 4; j++) {
                const int keep = nb - 4 * j;
                if (keep <= 0) xw[j] = 0;
                else if (keep < 4) xw[j] &= (1u << (8 * keep)) - 1u;
            }
        }
    }
    if (cls_thread) *reinterpret_cast<uint4*>(&R.b[c0]) = make_uint4(xw[0], xw[1], xw[2], xw[3]);
    if (tid == 0) *reinterpret_cast<uint4*>(&R.b[kRegion]) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    L3_PH(0);
    // -- document starts inside the region (one document per thread: d0 - 1 + tid; tiles with more take the loop further
    //    down): the offsets are requested here and marked behind the classification, which runs while they are in flight;
    //    the same offsets serve the per-document items at the end of the kernel
    const uint64_t d_mine = (uint64_t)d0 + tid;
    uint64_t pos_mine = ~0ull, pos_prev = ~0ull;
    if (d_mine <= A.n_docs) pos_mine = A.doc_off[d_mine];
    if (tid == 0 && d0 > 0) pos_prev = A.doc_off[d0 - 1];
    // -- ASCII bytes by table: ctab[v] holds the eight class bits of byte value v, one per byte of two words (x: rune
    //    start, letter, digit, newline; y: blank, other white space, apostrophe, '<').  OR-ing the entry of byte p shifted
    //    left by p mod 8 gathers, for each class, the 8-bit mask of the thread's bytes 0..7 and of its bytes 8..15: one
    //    8-byte LDS read and two shift-ORs per byte (the arithmetic version of this step took 29 ALU operations per byte),
    //    then four byte permutes put the halves together, two classes per word at bits 0 and 16.  Bytes >= 0x80 have an
    //    all-zero entry.
    uint32_t m_rs = 0, m_l = 0, m_n = 0, m_nl = 0, m_sp = 0, m_ws = 0, m_apos = 0, m_lt = 0;
    if (cls_thread) {
        const uint32_t valid = c0 + 16 <= n ? 0xFFFFu : (c0 < n ? (1u << (n - c0)) - 1u : 0u);
        uint32_t lx = 0, ly = 0, hx = 0, hy = 0;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const uint2 e = R.ctab[(xw[p >> 2] >> ((p & 3) * 8)) & 0xFFu];
            lx |= e.x << p; ly |= e.y << p;
            const uint2 f = R.ctab[(xw[2 + (p >> 2)] >> ((p & 3) * 8)) & 0xFFu];
            hx |= f.x << p; hy |= f.y << p;
        }
        const uint32_t ax = __builtin_amdgcn_perm(hx, lx, 0x05010400u), ay = __builtin_amdgcn_perm(hx, lx, 0x07030602u);
        const uint32_t az = __builtin_amdgcn_perm(hy, ly, 0x05010400u), aw = __builtin_amdgcn_perm(hy, ly, 0x07030602u);
        m_rs = ax & valid; m_l = (ax >> 16) & valid;
        m_n = ay & valid; m_nl = (ay >> 16) & valid;
        m_sp = az & valid; m_ws = (az >> 16) & valid;
        m_apos = aw & valid; m_lt = (aw >> 16) & valid;
    }
    if (pos_prev != ~0ull && pos_prev >= g_lo) atomicOr(&R.bm[B_DS][(pos_prev - g_lo) >> 6], 1ull << ((pos_prev - g_lo) & 63));
    if (pos_mine < g_hi) {
        atomicOr(&R.bm[B_DS][(pos_mine - g_lo) >> 6], 1ull << ((pos_mine - g_lo) & 63));
        for (uint64_t d = d_mine + kTileThreads; d <= A.n_docs; d += kTileThreads) {   // (more than 256 documents start in the region)
            const uint64_t pos = A.doc_off[d];
            if (pos >= g_hi) break;
            atomicOr(&R.bm[B_DS][(pos - g_lo) >> 6], 1ull << ((pos - g_lo) & 63));
        }
    }
    // -- contraction letters after an apostrophe (state_machine.go:131-148); the masks of the ASCII bytes.  A thread that
    //    holds bytes outside ASCII or a '<' says so: those take a phase of their own below, which pure ASCII text without
    //    '<' skips together with its barrier
    // kOnePretoken: the bytes are symbols and nothing else -- no UTF-8 decoding, no special tokens (no byte is ever marked
    // invalid or special), and the start bits are the document starts (compute_starts)
    const bool one_pre = (A.flags & kOnePretoken) != 0;   // uniform
    const uint32_t lt_mask = (A.flags & (kNoSpecial | kOnePretoken)) ? 0u : m_lt;
    uint32_t hi_mask = 0;
    if (cls_thread) {
        const uint32_t valid = c0 + 16 <= n ? 0xFFFFu : (c0 < n ? (1u << (n - c0)) - 1u : 0u);
        hi_mask = one_pre ? 0u : ~m_rs & valid;  // bytes >= 0x80
        if (hi_mask | lt_mask) vars[TV_RARE] = 1;   // (every writer stores the same value)
        uint32_t ap = m_apos;
        while (ap) {
            const int q = c0 + __builtin_ctz(ap);
            ap &= ap - 1;
            const uint32_t b1 = (uint32_t)R.b[q + 1] | 0x20u, b2 = (uint32_t)R.b[q + 2] | 0x20u;  // zero padded past n
            if (b1 == 's' || b1 == 't' || b1 == 'm' || b1 == 'd') atomicOr(&R.bm[B_CE2][(q + 2) >> 6], 1ull << ((q + 2) & 63));
            if ((b1 == 'r' && b2 == 'e') || (b1 == 'v' && b2 == 'e') || (b1 == 'l' && b2 == 'l'))
                atomicOr(&R.bm[B_CE3][(q + 3) >> 6], 1ull << ((q + 3) & 63));
        }
        uint16_t* h;
        h = reinterpret_cast<uint16_t*>(R.bm[B_RS]); h[tid] = (uint16_t)m_rs;
        h = reinterpret_cast<uint16_t*>(R.bm[B_L]); h[tid] = (uint16_t)m_l;
        h = reinterpret_cast<uint16_t*>(R.bm[B_N]); h[tid] = (uint16_t)m_n;
        h = reinterpret_cast<uint16_t*>(R.bm[B_NL]); h[tid] = (uint16_t)m_nl;
        h = reinterpret_cast<uint16_t*>(R.bm[B_SP]); h[tid] = (uint16_t)m_sp;
        h = reinterpret_cast<uint16_t*>(R.bm[B_WS]); h[tid] = (uint16_t)m_ws;
        h = reinterpret_cast<uint16_t*>(R.bm[B_APOS]); h[tid] = (uint16_t)m_apos;
        h = reinterpret_cast<uint16_t*>(R.bm[B_INV]); h[tid] = 0;
        h = reinterpret_cast<uint16_t*>(R.bm[B_NX]); h[tid] = 0;
    }
    __syncthreads();
    L3_PH(1);

    // -- bytes outside ASCII (generic UTF-8 decode + class table), special-token literals after '<'
    //    (state_machine.go:36,310-320; special.go:12): only in tiles that hold any (uniform)
    const TileSrc src{&R, n, hard};
    if (vars[TV_RARE]) {
        if (hi_mask) {
            uint32_t m_inv = 0, m_nx = 0;
            uint32_t hi = hi_mask;
            // byte_info() of the thread's bytes, lead bytes first: a valid sequence takes its continuation bytes out of the
            // walk (they start no rune); a continuation byte that no sequence covers is a stray one (a rune of its own,
            // U+FFFD).  This part reads LDS only; the classes of the characters it finds follow below.
            uint32_t m_lead = 0;   // bytes that begin a valid sequence of two to four bytes
            // A thread with no document start from two bytes in front of its bytes to three behind them, all inside the
            // text, needs none of the generic walk's document checks (every byte it looks at is in one document): it
            // works on its 16 bytes and their neighbours in registers, branch-free per lead byte.
            bool plain = false;
#ifndef L3_NO_PLAIN
            if (c0 >= 16 && c0 + 19 <= n) {
                const int q0 = c0 - 2;
                const uint32_t sh = (uint32_t)q0 & 63u;   // 14, 30, 46 or 62
                const u64 d0 = R.bm[B_DS][q0 >> 6], d1 = R.bm[B_DS][(q0 + 20) >> 6];
                const u64 ds_bits = (d0 >> sh) | (((q0 >> 6) != ((q0 + 20) >> 6)) ? d1 << (64u - sh) : 0ull);
                plain = (ds_bits & ((1ull << 21) - 1ull)) == 0;
            }
#endif
            if (plain) {
                const uint4 own = *reinterpret_cast<const uint4*>(&R.b[c0]);
                const uint32_t prv = *reinterpret_cast<const uint32_t*>(&R.b[c0 - 4]), nxt = *reinterpret_cast<const uint32_t*>(&R.b[c0 + 16]);
                const uint32_t ow[4] = {own.x, own.y, own.z, own.w};
                uint32_t contm = 0;   // continuation bytes (0x80 .. 0xBF) among the thread's
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t x = ow[j] & ~(ow[j] << 1) & 0x80808080u;
                    contm |= (((x >> 7) & 1u) | ((x >> 14) & 2u) | ((x >> 21) & 4u) | ((x >> 28) & 8u)) << (4 * j);
                }
                uint32_t cov = 0;     // continuation bytes inside a valid sequence
                if (contm & 1u) {     // the first byte continues something: a sequence that begins one to three bytes in front?
#pragma unroll
                    for (int j = 1; j <= 3; j++) {
                        const uint32_t w = __builtin_amdgcn_alignbyte(own.x, prv, 4 - j);   // bytes c0 - j .. c0 - j + 3
                        const uint32_t sl = seq_len_word(w);
                        if (sl > (uint32_t)j) cov |= (1u << (sl - (uint32_t)j)) - 1u;
                    }
                }
                uint32_t lead = hi_mask & ~contm;
                m_rs |= lead;         // every other byte starts a rune, valid or not
                while (lead) {
                    const uint32_t k = (uint32_t)__builtin_ctz(lead);
                    lead &= lead - 1u;
                    const uint32_t i4 = k >> 2;
                    const uint32_t wa = i4 == 0 ? own.x : i4 == 1 ? own.y : i4 == 2 ? own.z : own.w;
                    const uint32_t wb = i4 == 0 ? own.y : i4 == 1 ? own.z : i4 == 2 ? own.w : nxt;
                    const uint32_t sl = seq_len_word(__builtin_amdgcn_alignbyte(wb, wa, k & 3u));
                    if (sl == 0) { m_inv |= 1u << k; continue; }
                    cov |= ((1u << sl) - 2u) << k;
                    m_lead |= 1u << k;
                }
                const uint32_t stray = hi_mask & contm & ~cov;
                m_rs |= stray;
                m_inv |= stray;
            } else
            // (the generic walk: among the thread's first three bytes a continuation byte's sequence may have begun in front
            // of them, those take the look-back of rune_starts_at)
            while (hi) {
                const int k = __builtin_ctz(hi);
                hi &= hi - 1;
                const int p = c0 + k;
                const uint32_t b0 = R.b[p], bit = 1u << k;
                if (is_cont(b0)) {
                    if (k >= 3 || rune_starts_at(src, p)) { m_rs |= bit; m_inv |= bit; }
                    continue;
                }
                const int sl = seq_len(src, p);
                m_rs |= bit;
                if (sl == 0) { m_inv |= bit; continue; }
                m_lead |= bit;
                hi &= ~(((1u << sl) - 1u) << k);
            }
#ifdef L3_PHASE_PROF
            L3_PH(9);
#endif
            // Classes.  The two-stage table costs a character two loads that depend on one another, and one character after
            // the other a thread of Cyrillic or Greek text waited for sixteen round trips in turn.  Here: ONE load per
            // character from the direct table of planes 0 and 1 (Tables::cls_direct, the same bytes), the loads of all the
            // thread's characters -- at most eight: a sequence has two bytes or more -- in flight together.  Whole blocks of
            // letters need no load; planes 2 to 16 (rare) take the two-stage table afterwards.
            {
                uint32_t vq[8];
                uint32_t shq = 0, ldq = 0, far = 0;   // per slot: cp & 3, "vq holds a table byte"; per byte: a character of planes 2-16
                uint32_t m2 = m_lead;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool on = m2 != 0;
                    const uint32_t k = on ? (uint32_t)__builtin_ctz(m2) : 0u;
                    m2 &= m2 - 1u;   // (0 stays 0)
                    const int p = c0 + (int)k;
                    const uint32_t b0 = R.b[p], b1 = R.b[p + 1], b2 = R.b[p + 2], b3 = R.b[p + 3];   // (inside the LDS image: it is padded)
                    uint32_t cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
                    if (b0 >= 0xE0u) cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
                    if (b0 >= 0xF0u) cp = ((b0 & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
                    const bool tab = on && !letter_block(cp);
                    const bool near = tab && cp < kClsDirectCps;
                    if (tab && !near) far |= 1u << k;
                    vq[j] = near ? (uint32_t)T.cls_direct[cp >> 2] : 0u;
                    shq |= (cp & 3u) << (2 * j);
                    ldq |= (near ? 1u : 0u) << j;
                }
                m2 = m_lead;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool on = m2 != 0;
                    const uint32_t k = on ? (uint32_t)__builtin_ctz(m2) : 0u, bit = on ? 1u << k : 0u;
                    m2 &= m2 - 1u;
                    const uint32_t c = ((ldq >> j) & 1u) ? class_from_block_byte(vq[j], (shq >> (2 * j)) & 3u) : (uint32_t)C_L;
                    if (far & bit) continue;   // (below)
                    if (c == C_L) m_l |= bit;
                    else if (c == C_N) { m_n |= bit; m_nx |= bit; }
                    else if (c == C_WS) m_ws |= bit;
                }
                while (far) {
                    const uint32_t k = (uint32_t)__builtin_ctz(far), bit = 1u << k;
                    far &= far - 1u;
                    const int p = c0 + (int)k;
                    const uint32_t cp = ((R.b[p] & 0x07u) << 18) | ((R.b[p + 1] & 0x3Fu) << 12) | ((R.b[p + 2] & 0x3Fu) << 6) | (R.b[p + 3] & 0x3Fu);
                    const uint32_t c = class_of_cp(cp, T.cls_stage1, T.cls_blocks);
                    if (c == C_L) m_l |= bit;
                    else if (c == C_N) { m_n |= bit; m_nx |= bit; }
                    else if (c == C_WS) m_ws |= bit;
                }
            }
#ifdef L3_PHASE_PROF
            L3_PH(10);
#endif
            if (m_inv) vars[TV_ANYINV] = 1;   // (every writer stores the same value)
            uint16_t* h;   // (a thread's half-words are its own)
            h = reinterpret_cast<uint16_t*>(R.bm[B_RS]); h[tid] = (uint16_t)m_rs;
            h = reinterpret_cast<uint16_t*>(R.bm[B_L]); h[tid] = (uint16_t)m_l;
            h = reinterpret_cast<uint16_t*>(R.bm[B_N]); h[tid] = (uint16_t)m_n;
            h = reinterpret_cast<uint16_t*>(R.bm[B_WS]); h[tid] = (uint16_t)m_ws;
            h = reinterpret_cast<uint16_t*>(R.bm[B_INV]); h[tid] = (uint16_t)m_inv;
            h = reinterpret_cast<uint16_t*>(R.bm[B_NX]); h[tid] = (uint16_t)m_nx;
        }
        uint32_t lt = lt_mask;
        while (lt) {
            const int q = c0 + __builtin_ctz(lt);
            lt &= lt - 1;
            int canon = 0;
            const int len = special_match(src, q, &canon);
            if (len) {
                atomicOr(&R.bm[B_SPE][q >> 6], 1ull << (q & 63));
                atomicOr(&R.bm[B_SPE][(q + len) >> 6], 1ull << ((q + len) & 63));
                if (T.special_id[canon] != 0) {  // tokenizer.go:236: only a piece whose lookup is non-zero is special
                    vars[TV_ANYSPS] = 1;   // (every writer stores the same value)
                    atomicOr(&R.bm[B_SPS][q >> 6], 1ull << (q & 63));
                    for (int k = 1; k < len; k++) atomicOr(&R.bm[B_SPB][(q + k) >> 6], 1ull << ((q + k) & 63));
                }
            }
        }
        __syncthreads();
    }
    L3_PH(2);

    // -- start-of-pre-token bits (one wave), digit runs that need more context than the halo holds
    if (wave == 0) {
        int n_pre = 0;
        uint32_t fl = 0;
        if (one_pre) n_pre = document_starts_only(R, own_lo, own_hi, eval_hi);   // (uniform)
        else fl = compute_starts(R, n, own_lo, own_hi, eval_hi, g_lo == 0, false, 0u, &n_pre);
        if (lane == 0) {
            vars[TV_NPRE] = n_pre;
            vars[TV_MB] = (fl & 2u) ? 1 : 0;
            vars[TV_NEED] = ((fl & 1u) || ((fl & 2u) && g_lo > 0)) ? 1 : 0;
            vars[TV_OWNINV] = (fl & 4u) ? 1 : 0;
        }
    }
    __syncthreads();
    L3_PH(3);
    if (vars[TV_NEED] || vars[TV_MB]) {  // rare: long digit runs / digits outside ASCII
        uint32_t edge = 0;
        const bool need = vars[TV_NEED] != 0, mbd = vars[TV_MB] != 0;
        if (need) {
            // runes right before the tile start, inside the document that contains it
            const bool at_doc_start = (R.bm[B_DS][own_lo >> 6] >> (own_lo & 63)) & 1ull;
            if (!at_doc_start) {
                int64_t seg_lo = (int64_t)A.doc_off[d0 - 1], seg_hi = (int64_t)A.doc_off[d0];
                edge = digits_before_global(A.text, seg_lo, seg_hi, (int64_t)t_lo, T, tmp64, W.tile_digits, tile);
            }
        }
        __syncthreads();
        if (!mbd) {
            if (wave == 0) {
                int n_pre = 0;
                compute_starts(R, n, own_lo, own_hi, eval_hi, g_lo == 0, true, edge, &n_pre);
                if (lane == 0) vars[TV_NPRE] = n_pre;
            }
        } else if (tid == 0) {
            // serial walk over the Nd rune starts; the other start bits are already final
            uint32_t off = 0;
            bool first_own = need;
            int total = 0;
            for (int w = 0; w < kWords; w++) {
                u64 nn = R.bm[B_N][w];
                const u64 cnn = R.bm[B_CONTN][w];
                u64 st = R.bm[B_ST][w] & ~(nn & cnn);   // linked digits: decided here
                while (nn) {
                    const int bpos = __builtin_ctzll(nn);
                    nn &= nn - 1;
                    const int q = w * 64 + bpos;
                    if ((cnn >> bpos) & 1ull) {
                        if (first_own && q >= own_lo) off = edge; else off++;
                        if (off % 3u == 0u && q < eval_hi) st |= 1ull << bpos;
                    } else {
                        off = 0;
                    }
                    if (q >= own_lo) first_own = false;
                }
                R.bm[B_ST][w] = st;
                u64 own = st;
                const int lo = w * 64;
                if (own_lo >= lo + 64 || own_hi <= lo) own = 0;
                else {
                    if (own_lo > lo) own &= ~0ull << (own_lo - lo);
                    if (own_hi < lo + 64) own &= (1ull << (own_hi - lo)) - 1ull;
                }
                R.wpre[w] = (uint16_t)total;
                total += __popcll(own);
            }
            vars[TV_NPRE] = total;
        }
        __syncthreads();
    }
    const uint32_t n_pre = (uint32_t)vars[TV_NPRE];
    const u64* ST = R.bm[B_ST];

    // -- compacted list of own pre-token starts (kListPer byte positions per thread; own_lo is 0 or kBack, so a
    //    thread's positions never straddle a 64-bit mask word).  list[n_pre] = end of the last own pre-token (the others
    //    end where the next one starts), kOpenEnd when it lies beyond what the look-ahead halo decides
    {
        if (tid == kTileThreads - 1) {
            int e = find_next_bit64(ST, own_hi, eval_hi);   // the first start past the own range
            if (e < 0 && hard) e = n;
            R.list[n_pre] = e < 0 ? (uint16_t)kOpenEnd : (uint16_t)e;
        }
        const int pos = own_lo + tid * kListPer;
        if (pos < own_hi) {
            const int wd = pos >> 6, sh = pos & 63;
            const u64 word = ST[wd];
            u64 ownw = word;
            if (own_lo > wd * 64) ownw &= ~0ull << (own_lo - wd * 64);
            uint32_t bits = (uint32_t)(word >> sh) & ((1u << kListPer) - 1u);
            if (own_hi - pos < kListPer) bits &= (1u << (own_hi - pos)) - 1u;
            uint32_t base = R.wpre[wd] + (uint32_t)__popcll(ownw & ((1ull << sh) - 1ull));
            if (A.kinds) {
                const u64 sps = R.bm[B_SPS][wd] >> sh;
                for (int j = 0; j < kListPer && pos + j < own_hi; j++)
                    A.kinds[g_lo + pos + j] = (bits >> j) & 1u ? (((sps >> j) & 1ull) ? 2 : 1) : 0;
            }
            while (bits) {
                int j = __builtin_ctz(bits);
                bits &= bits - 1;
                R.list[base++] = (uint16_t)(pos + j);
            }
        }
    }
    __syncthreads();
    L3_PH(4);

    // The bitmap rows from B_LIVE on are dead by now: they hold the two work lists of the lookup phase.
    uint16_t* lst = reinterpret_cast<uint16_t*>(&R.bm[B_LIVE][0]);          // [kLstCap]: from the front the pre-tokens that
                                                                            // missed their first bucket, from the back the rest
    uint32_t* defer = reinterpret_cast<uint32_t*>(lst + kLstCap);           // [kDeferCap]: the deferred pre-tokens
    int32_t* out0 = W.stream + tile * (uint64_t)kTileCap;
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(R.b);
    // -- per pre-token: special id / whole-string vocabulary probe (bpe.go:27-33) / defer.  One stream item each.
    //    Stage 1, every pre-token: up to 12 bytes, valid UTF-8, no special token = the common case.  Words straight from
    //    the LDS image, masked by a table row per length, the CHEAP first-choice hash, one 16-byte slot load, kLookU
    //    pre-tokens per thread in flight.  A hit is stored at once (the 64 lanes of a wave hold consecutive items: a
    //    coalesced store).  What is not settled is listed: missed first bucket -> front of `lst`, everything else -> back.
    {
        const bool complex_tile = vars[TV_ANYINV] != 0 || vars[TV_ANYSPS] != 0;   // uniform
        constexpr int kLookU = L3_LOOK_U;
#pragma unroll 1
        for (uint32_t base = 0; base < n_pre; base += kTileThreads * kLookU) {
            uint32_t a0[kLookU], a1[kLookU], a2[kLookU], tag[kLookU];
            VocabSlot sl[kLookU];
            uint32_t m_fast = 0, m_a = 0, m_b = 0;   // bit u: probed / goes to the front list / goes to the back list
#pragma unroll
            for (int u = 0; u < kLookU; u++) {
                // straight-line code on a clamped index (lanes past the end redo the last pre-token and are masked at the
                // stores): branches here cost more scalar work than the lanes they save, and keep the loads apart
                const uint32_t k = base + (uint32_t)u * kTileThreads + tid;
                const uint32_t kk = k < n_pre ? k : n_pre - 1u;
                const uint32_t s = R.list[kk], e = R.list[kk + 1u];
                const uint32_t len = e - s;                     // (an open end gives a huge value)
                const uint4 m = mk[len < (uint32_t)kVocabInline ? len : (uint32_t)kVocabInline];
                const uint32_t* wp = bw + (s >> 2);
                const uint32_t sh = s * 8u;                      // v_alignbit takes the low five bits
                const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2], x3 = wp[3];
                a0[u] = __funnelshift_r(x0, x1, sh) & m.x;
                a1[u] = __funnelshift_r(x1, x2, sh) & m.y;
                a2[u] = __funnelshift_r(x2, x3, sh) & m.z;
                tag[u] = m.w;
                bool fast = len - 1u < (uint32_t)kVocabInline;
                if (complex_tile) {   // (uniform branch)
                    const bool special = (R.bm[B_SPS][s >> 6] >> (s & 63u)) & 1ull;
                    const bool inval = fast && any_bit_in(R.bm[B_INV], (int)s, (int)e);
                    fast = fast && !special && !inval;
                }
                const bool in = k < n_pre;
                if (in && fast) m_fast |= 1u << u;
                if (in && !fast) m_b |= 1u << u;
                sl[u] = T.vocab_slots[vocab_hash_short_a(a0[u], a1[u], a2[u], len, T.vocab_log2)];   // (in bounds whatever the lane holds)
            }
#pragma unroll
            for (int u = 0; u < kLookU; u++) {
                const uint32_t k = base + (uint32_t)u * kTileThreads + tid;
                const uint32_t diff = (sl[u].w[0] ^ a0[u]) | (sl[u].w[1] ^ a1[u]) | (sl[u].w[2] ^ a2[u]) | ((sl[u].w[3] ^ tag[u]) & ~0x1FFFFu);
                const bool probed = (m_fast >> u) & 1u, hit = probed && diff == 0;
                if (hit) out0[k] = (int32_t)(sl[u].w[3] & 0x1FFFFu);
                m_a |= (probed && !hit) ? 1u << u : 0u;
            }
            // one scan per wave for both lists (the two counts packed in one word), one LDS atomic per wave
            const uint32_t cnt_a = (uint32_t)__popc(m_a), cnt_b = (uint32_t)__popc(m_b);
            const uint32_t inc = wave_iscan(cnt_a | (cnt_b << 16));
            const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
            if (tot) {   // uniform over the wave
                uint32_t basep = 0;
                if (lane == 63) basep = atomicAdd(reinterpret_cast<uint32_t*>(&vars[TV_NAB]), tot);
                basep = __builtin_amdgcn_readlane(basep, 63);
                uint32_t pa = (basep & 0xFFFFu) + (inc & 0xFFFFu) - cnt_a;
                uint32_t pb = (basep >> 16) + (inc >> 16) - cnt_b;
#pragma unroll
                for (int u = 0; u < kLookU; u++) {
                    const uint32_t k = base + (uint32_t)u * kTileThreads + tid;
                    // (an index beyond the list is clamped: the lists have overflowed then and are not looked at)
                    const uint32_t to_a = (m_a >> u) & 1u, to_b = (m_b >> u) & 1u;
                    if (to_a) lst[pa < (uint32_t)kLstCap ? pa : (uint32_t)kLstCap - 1u] = (uint16_t)k;
                    if (to_b) lst[(uint32_t)kLstCap - 1u - (pb < (uint32_t)kLstCap ? pb : (uint32_t)kLstCap - 1u)] = (uint16_t)k;
                    pa += to_a;
                    pb += to_b;
                }
            }
        }
    }
    __syncthreads();
    L3_PH(5);
    // Stage 2, the listed pre-tokens, densely packed (one rare pre-token among the 64 of a wave would otherwise drag the
    // whole wave through the long path).  Front list: second bucket of the short ones; a miss there is a pre-token for
    // the merge kernel.  Back list (taken by the threads from the back, so that the two run side by side): special
    // tokens, more than 12 bytes, invalid UTF-8, open ends.  Deferred pre-tokens get their ordinal within the tile (per
    // length class / in the slow list) here and are noted in `defer`; their records follow once the tile has its room
    // in the global lists.
    const uint32_t n_a = (uint32_t)vars[TV_NAB] & 0xFFFFu, n_b = (uint32_t)vars[TV_NAB] >> 16;
    const bool lst_ovf = n_a + n_b > (uint32_t)kLstCap;   // uniform
    if (!lst_ovf) {
#pragma unroll 1
        for (uint32_t it = tid; it < n_a; it += kTileThreads) {
            const uint32_t k = lst[it];
            const uint32_t s = R.list[k], len = (uint32_t)R.list[k + 1u] - s;   // 1..12
            const uint4 m = mk[len];
            const uint32_t* wp = bw + (s >> 2);
            const uint32_t sh = s * 8u;
            const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2], x3 = wp[3];
            const uint32_t b0 = __funnelshift_r(x0, x1, sh) & m.x, b1 = __funnelshift_r(x1, x2, sh) & m.y, b2 = __funnelshift_r(x2, x3, sh) & m.z;
            const VocabSlot v = T.vocab_slots[vocab_hash_short_b(b0, b1, b2, len, T.vocab_log2)];
            if (((v.w[0] ^ b0) | (v.w[1] ^ b1) | (v.w[2] ^ b2) | ((v.w[3] ^ m.w) & ~0x1FFFFu)) == 0) {
                out0[k] = (int32_t)(v.w[3] & 0x1FFFFu);
            } else {
                const uint32_t c = len_class(len);
                const uint32_t ord = (uint32_t)atomicAdd(&ccnt[c], 1);
                const uint32_t di = (uint32_t)atomicAdd(&vars[TV_NDEF], 1);
                if (di < (uint32_t)kDeferCap) defer[di] = k | (c << 12) | (ord << 16);
            }
        }
#pragma unroll 1
        for (uint32_t it = (uint32_t)(kTileThreads - 1 - tid); it < n_b; it += kTileThreads) {
            const uint32_t k = lst[kLstCap - 1 - it];
            const int s = R.list[k];
            const uint32_t ev = R.list[k + 1u];
            const int e = ev == kOpenEnd ? -1 : (int)ev;
            const uint32_t kind = tile_generic_item(T, R, src, bw, s, e, &out0[k]);
            if (kind) {   // deferred: 1 = short (a lane of the merge kernel), 2 / 3 = slow (3: known to be valid UTF-8)
                uint32_t d;
                if (kind == 1u) {
                    const uint32_t c = len_class((uint32_t)(e - s));
                    d = k | (c << 12) | ((uint32_t)atomicAdd(&ccnt[c], 1) << 16);
                } else {
                    d = 0x80000000u | (kind == 3u ? 0x40000000u : 0u) | k | ((uint32_t)atomicAdd(&vars[TV_NSLOW], 1) << 12);
                }
                const uint32_t di = (uint32_t)atomicAdd(&vars[TV_NDEF], 1);
                if (di < (uint32_t)kDeferCap) defer[di] = d;
            }
        }
    }
    __syncthreads();
    L3_PH(6);
    // A tile whose lists overflow (thousands of tiny pre-tokens that are all rare: random bytes) starts over in the
    // general way: every pre-token through the long path, placeholders with the ordinal inside the tile straight to the
    // stream, which the last pass below then reads back.
    const bool general = lst_ovf || (uint32_t)vars[TV_NDEF] > (uint32_t)kDeferCap;   // uniform over the workgroup
    if (general) {
        if (tid == 0) vars[TV_NSLOW] = 0;
        if (tid < kLenClasses) ccnt[tid] = 0;
        __syncthreads();
#pragma unroll 1
        for (uint32_t k = tid; k < n_pre; k += kTileThreads) {
            const int s = R.list[k];
            const uint32_t ev = R.list[k + 1u];
            const int e = ev == kOpenEnd ? -1 : (int)ev;
            int32_t item;
            const uint32_t kind = tile_generic_item(T, R, src, bw, s, e, &item);
            if (kind == 1u) {
                const uint32_t c = len_class((uint32_t)(e - s));
                item = -2 - (int32_t)((c << 12) | (uint32_t)atomicAdd(&ccnt[c], 1));
            } else if (kind) {
                item = -2 - (int32_t)(kSlowBase + (uint32_t)atomicAdd(&vars[TV_NSLOW], 1));
            }
            out0[k] = item;
        }
        __syncthreads();
    }
    // space in the entry lists: one returning atomic per length class the tile uses (lanes 0..15) and one for the slow
    // list (lane 16), issued here and first needed after the document loop below, so that the round trip overlaps
    const uint32_t sub = (uint32_t)(tile & 3u);
    uint32_t m0_r = 0, nm_r = 0;
    if (tid < kLenClasses) {
        nm_r = (uint32_t)ccnt[tid];
        m0_r = nm_r ? atomicAdd(&W.status->shard[(uint32_t)tid * 4u + sub].miss_count, nm_r) : 0u;
    } else if (tid == kLenClasses) {
        nm_r = (uint32_t)vars[TV_NSLOW];
        m0_r = nm_r ? atomicAdd(&W.status->slow_count, nm_r) : 0u;
        W.tile_items[tile] = n_pre;
        // first own pre-token start (relative to the tile) and "the bytes it owns hold invalid UTF-8": how the slow kernel finds the
        // end of a pre-token that runs past this tile's look-ahead, and whether its symbols can differ from its bytes
        W.tile_first[tile] = (n_pre ? (uint32_t)(R.list[0] - own_lo) : kNoFirst) | (vars[TV_OWNINV] ? kFirstInv : 0u);
    }
    // -- stream index of the first item of every document that starts in this tile (the first 256 from the offsets loaded
    //    at the start of the kernel)
    {
        const uint64_t d_end = d_end32;
        for (uint64_t d = d_mine; d < d_end; d += kTileThreads) {
            const int pos = (int)((d == d_mine ? pos_mine : A.doc_off[d]) - g_lo);
            uint32_t item;
            if (pos >= own_hi) item = n_pre;
            else {
                const int wd = pos >> 6, sh = pos & 63;
                u64 ownw = ST[wd];
                if (own_lo > wd * 64) ownw &= ~0ull << (own_lo - wd * 64);
                item = R.wpre[wd] + (uint32_t)__popcll(ownw & ((1ull << sh) - 1ull));
            }
            W.doc_item[d] = item;
        }
    }
    if (tid < kLenClasses) {
        if (nm_r && (uint64_t)m0_r + nm_r > W.miss_shard_cap) atomicOr(&W.status->flags, kOvfMiss);
        cm0[tid] = (int)m0_r;
    } else if (tid == kLenClasses) {
        if (nm_r && (uint64_t)m0_r + nm_r > W.slow_cap) atomicOr(&W.status->flags, kOvfSlow);
        vars[TV_SLOW0] = (int)m0_r;
    }
    __syncthreads();
    L3_PH(7);
    // -- the records of the deferred pre-tokens and their stream items (placeholders that name the global entry)
    {
        const uint32_t s0 = (uint32_t)vars[TV_SLOW0];
        if (!general) {
            const uint32_t n_def = (uint32_t)vars[TV_NDEF];
#pragma unroll 1
            for (uint32_t it = tid; it < n_def; it += kTileThreads) {
                const uint32_t d = defer[it];
                const uint32_t k = d & 4095u;
                const bool slow = (d >> 31) != 0;
                const uint32_t ord = slow ? (d >> 12) & 4095u : (d >> 16) & 4095u;
                out0[k] = tile_deferred_record(W, R, bw, cm0, g_lo, tile, sub, s0, k, slow, ord, (d >> 12) & 15u, (d >> 30) & 1u);
            }
        } else {
            // (every thread re-reads its own stores of the general pass above)
#pragma unroll 1
            for (uint32_t k = tid; k < n_pre; k += kTileThreads) {
                const int32_t item = out0[k];
                if (item < -1) {
                    uint32_t ord = (uint32_t)(-2 - item);
                    const bool slow = ord >= kSlowBase;
                    if (slow) ord -= kSlowBase;
                    const uint32_t s = R.list[k], ev = R.list[k + 1u];
                    const uint32_t valid = (ev != kOpenEnd && !any_bit_in(R.bm[B_INV], (int)s, (int)ev)) ? 1u : 0u;
                    out0[k] = tile_deferred_record(W, R, bw, cm0, g_lo, tile, sub, s0, k, slow, ord & 4095u, ord >> 12, valid);
                }
            }
        }
    }
    L3_PH(8);
}

}  // namespace

hipError_t launch_stage_docs(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    const uint32_t dblocks = (uint32_t)((A.n_docs + 1 + 255) / 256);
    hipLaunchKernelGGL(docs_kernel, dim3(dblocks), dim3(256), 0, stream, W, A, num_tiles(A.n_bytes));
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_stage_tile(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    hipLaunchKernelGGL(tile_kernel, dim3((uint32_t)num_tiles(A.n_bytes)), dim3(kTileThreads), 0, stream, T, W, A);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3

#ifdef L3_PHASE_PROF
extern "C" int l3_debug_phases(unsigned long long* out, int n, int reset) {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(l3::g_phase), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 16; i++) out[i] = h[i];
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(l3::g_phase), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

