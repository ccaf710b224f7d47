// l3_kernels.hip -- the encode pipeline on gfx950 (wave64, LDS-staged tiles).
//
// Replaces the loops of llama3.Tokenizer.Encode (reference llama3/tokenizer.go:217-265) for a
// batch of documents held as one concatenated byte array + offsets:
//
//   docs_kernel    doc_off sanity; first document of every tile; clears the per-tile counters
//   tile_kernel    one workgroup per 3,840 bytes of text (+ halos) staged in LDS:
//                  UTF-8 validity + character class per byte    (state_machine.go:36,310-320)
//                  special-token literals                       (special.go:12,48-83)
//                  position-local start-of-pre-token predicate  (state_machine.go:82-284, l3_rules.h)
//                  whole-pre-token vocabulary probe             (bpe.go:27-33)
//                  one stream item per pre-token: its id, or a placeholder of a deferred pre-token
//   merge_kernel   one lane per short deferred pre-token (<= 64 bytes): greedy merge in an LDS slot (bpe.go:36-71);
//                  its last workgroups make the wave-per-entry pass over the slow list (mid_pass)
//                  and cut the long ones into window items (prep_entry)
//   slow_kernel    the remaining deferred pre-tokens (long, invalid UTF-8, or ending beyond the halo) as independent
//                  WINDOW items of 1,920 symbols: cut points where no vocabulary entry can span, every piece merged by
//                  the exact round-based parallel greedy merge, token ids into a sparse per-symbol image; pieces of
//                  32,768 symbols and more are listed for the grid-wide merge (l3_giant.hip, driven by the host)
//   block_totals_kernel   per-tile output bases inside blocks of 1,024 tiles and the totals of the blocks (the waves
//                  of the emit kernels add the block totals in front of their tile up themselves: no scan launch); the
//                  launch also carries the finish_window workgroups that compact the window images
//   emit_kernel    scatter to the final token array, per-document offsets, BOS/EOS (tokenizer.go:225-262);
//   emit_generic_kernel   the tiles the fast one flags
//   status_out_kernel     the status word to page-locked host memory, cleared for the next batch
//   decode_sum / decode_scan / decode_write_kernel   ids -> bytes in blocks of 4,096 ids (tokenizer.go:339-354)
//
// All cross-workgroup data flows through kernel boundaries on one stream; atomics are used only
// for counters / appends that later kernels read.
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"

namespace l3 {

namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- workgroup primitives (256 threads = 4 waves) --------------------------------------------
// Data movement between the lanes of a wave by DPP (one VALU operation, no trip through the LDS crossbar as __shfl makes):
// CTRL 0x111 / 0x112 / 0x114 / 0x118 = row_shr:1/2/4/8 inside rows of 16 lanes, 0x142 / 0x143 = row_bcast:15 / :31 (the last
// lane of a row to the rows behind it: ROWS 0xA / 0xC), 0x138 / 0x130 = wave_shr:1 / wave_shl:1 (previous / next lane of the
// wave).  A lane whose source does not exist gets `idle`.  Checked on the device by tools/probes/dpp_shift.hip.  All 64 lanes
// must be active.
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ uint32_t dpp_move(uint32_t v, uint32_t idle = 0u) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)idle, (int)v, CTRL, ROWS, 0xF, false);
}
// inclusive sum scan over the wave: six v_add_u32_dpp
__device__ __forceinline__ uint32_t wave_iscan(uint32_t v) {
    v += dpp_move<0x111>(v);
    v += dpp_move<0x112>(v);
    v += dpp_move<0x114>(v);
    v += dpp_move<0x118>(v);
    v += dpp_move<0x142, 0xA>(v);
    v += dpp_move<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ unsigned long long wave_iscan64(unsigned long long v) {
#define L3_STEP(CTRL, ROWS) v += ((unsigned long long)dpp_move<CTRL, ROWS>((uint32_t)(v >> 32)) << 32) | dpp_move<CTRL, ROWS>((uint32_t)v)
    L3_STEP(0x111, 0xF); L3_STEP(0x112, 0xF); L3_STEP(0x114, 0xF); L3_STEP(0x118, 0xF); L3_STEP(0x142, 0xA); L3_STEP(0x143, 0xC);
#undef L3_STEP
    return v;
}
// exclusive sum scan over the workgroup; tmp: >= 4 words of LDS.  Contains two barriers.
template <int NW = kTileThreads / 64>
__device__ __forceinline__ uint32_t block_escan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    uint32_t inc = wave_iscan(v);
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t x = tmp[i];
        if (i < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
template <int NW = kTileThreads / 64>
__device__ __forceinline__ unsigned long long block_min64(unsigned long long v, unsigned long long* tmp) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    if (lane_id() == 0) tmp[wave_id()] = v;
    __syncthreads();
    unsigned long long r = tmp[0];
#pragma unroll
    for (int i = 1; i < NW; i++) r = tmp[i] < r ? tmp[i] : r;
    __syncthreads();
    return r;
}
// sum over the workgroup; tmp: >= NW words of LDS.  Two barriers.
template <int NW = kTileThreads / 64>
__device__ __forceinline__ unsigned long long block_sum64(unsigned long long v, unsigned long long* tmp) {
    v = wave_iscan64(v);
    __syncthreads();
    if (lane_id() == 63) tmp[wave_id()] = v;
    __syncthreads();
    unsigned long long r = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) r += tmp[i];
    return r;
}
// segmented counter: `cnt` = length of the run at the end of the span, `open` = the span holds no break
struct Seg { uint32_t cnt, open; };
__device__ __forceinline__ Seg seg_comb(Seg a, Seg b) {
    Seg r;
    r.cnt = b.open ? a.cnt + b.cnt : b.cnt;
    r.open = a.open & b.open;
    return r;
}
// byte source over global memory restricted to one segment [lo, hi)
struct GlobalSrc {
    const uint8_t* t;
    int64_t lo_, hi_;
    __device__ uint32_t b(int64_t q) const { return t[q]; }
    __device__ bool ds(int64_t q) const { return q == lo_ || q == hi_; }
    __device__ int64_t lo() const { return lo_; }
    __device__ int64_t hi() const { return hi_; }
};

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
typedef unsigned long long u64;
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

__device__ __forceinline__ int find_next_bit64(const u64* bits, int from, int limit) {
    if (from >= limit) return -1;
    int w = from >> 6;
    u64 x = bits[w] & (~0ull << (from & 63));
    const int wl = (limit - 1) >> 6;
    for (;;) {
        if (x) {
            int q = (w << 6) + __builtin_ctzll(x);
            return q < limit ? q : -1;
        }
        if (++w > wl) return -1;
        x = bits[w];
    }
}
__device__ __forceinline__ bool any_bit_in(const u64* bits, int from, int to) {  // [from, to), from < to
    int w0 = from >> 6, w1 = (to - 1) >> 6;
    u64 m0 = ~0ull << (from & 63), m1 = ~0ull >> (63 - ((to - 1) & 63));
    if (w0 == w1) return (bits[w0] & m0 & m1) != 0;
    if (bits[w0] & m0) return true;
    for (int w = w0 + 1; w < w1; w++)
        if (bits[w]) return true;
    return (bits[w1] & m1) != 0;
}

__device__ __forceinline__ uint32_t up32(uint32_t v) { return dpp_move<0x138>(v); }  // value of the previous lane (lane 0: 0)
__device__ __forceinline__ uint32_t dn32(uint32_t v) { return dpp_move<0x130>(v); }  // value of the next lane (lane 63: 0)
// exclusive segmented scan over the wave; (0, open) is the identity of seg_comb
__device__ __forceinline__ Seg wave_seg_escan(Seg v) {
    Seg inc = v;
#define L3_STEP(CTRL, ROWS) { Seg t; t.cnt = dpp_move<CTRL, ROWS>(inc.cnt, 0u); t.open = dpp_move<CTRL, ROWS>(inc.open, 1u); inc = seg_comb(t, inc); }
    L3_STEP(0x111, 0xF) L3_STEP(0x112, 0xF) L3_STEP(0x114, 0xF) L3_STEP(0x118, 0xF) L3_STEP(0x142, 0xA) L3_STEP(0x143, 0xC)
#undef L3_STEP
    Seg ex;
    ex.cnt = dpp_move<0x138>(inc.cnt, 0u);
    ex.open = dpp_move<0x138>(inc.open, 1u);
    return ex;
}

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
    const uint32_t lt_mask = (A.flags & kNoSpecial) ? 0u : m_lt;
    uint32_t hi_mask = 0;
    if (cls_thread) {
        const uint32_t valid = c0 + 16 <= n ? 0xFFFFu : (c0 < n ? (1u << (n - c0)) - 1u : 0u);
        hi_mask = ~m_rs & valid;  // bytes >= 0x80
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
            while (hi) {
                const int k = __builtin_ctz(hi);
                hi &= hi - 1;
                const uint32_t wd = byte_info(src, c0 + k, T.cls_stage1, T.cls_blocks);
                if (wd & I_RUNE) {
                    const uint32_t c = wd & I_CLS, bit = 1u << k;
                    m_rs |= bit;
                    if (c == C_L) m_l |= bit;
                    else if (c == C_N) { m_n |= bit; m_nx |= bit; }
                    else if (c == C_WS) m_ws |= bit;
                    if (wd & I_INVALID) m_inv |= bit;
                }
            }
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
        uint32_t fl = compute_starts(R, n, own_lo, own_hi, eval_hi, g_lo == 0, false, 0u, &n_pre);
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

// ---- the merge kernel: short deferred pre-tokens, one lane each -------------------------------------------
// Every lane runs the greedy merge loop (bpe.go:36-71) of one pre-token in its own slot of an LDS arena, so all 64
// lanes of a wave keep independent probe chains in flight.  A slot holds S elements of two words:
//   sy = token id | next live element << 17 | previous live element << 24     (a linked list, 127 = none)
//   pq = prio24 of the pair (element, next live element) << 7 | element index  (0xFFFFFFFF = no such rule)
// so the smallest pq of a slot IS the next merge of the reference order: smallest rank, leftmost among equals
// (bpe.go:126, priority_queue.go:20-25).  Slots are S + 1 words apart: all lanes touch element p of their slot
// in the same instruction, and an odd stride puts them on different LDS banks.
// Pre-tokens of up to 16 bytes (their bytes arrive inside the MissIn record) use 16-element slots, all 256 lanes
// at once; longer ones (<= kLaneMax = 64 bytes, bytes read from the text) follow in groups of 64 lanes with
// 64-element slots of the same arena.
#ifndef L3_MERGE_THREADS
#define L3_MERGE_THREADS 256
#endif
constexpr int kMergeThreads = L3_MERGE_THREADS;
constexpr int kMergeWaves = kMergeThreads / 64;
constexpr int kMergeArena = kMergeThreads * 17;   // words per array; also holds 64 slots of 65
constexpr uint32_t kNone = 127;
static_assert(kMergeArena >= 64 * 65 || kMergeThreads < 256, "arena must hold one group of long slots");

template <int S>
__device__ __forceinline__ void lane_merge(const Tables& T, uint32_t* sy, uint32_t* pq) {
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int p = 0; p < S; p++) best = min(best, pq[p]);
        if (best == 0xFFFFFFFFu) return;
        const uint32_t bp = best & 127u, m = prio_merged_id(T, best >> 7);
        const uint32_t wb = sy[bp];
        const uint32_t q = (wb >> 17) & 127u, l = (wb >> 24) & 127u;
        const uint32_t r = (sy[q] >> 17) & 127u;
        sy[bp] = m | (r << 17) | (l << 24);
        pq[q] = 0xFFFFFFFFu;
        uint32_t idr = 0, idl = 0;
        if (r != kNone) { const uint32_t wr = sy[r]; idr = wr & 0x1FFFFu; sy[r] = (wr & 0x00FFFFFFu) | (bp << 24); }
        if (l != kNone) idl = sy[l] & 0x1FFFFu;
        // both new pairs are probed whatever the neighbours are (a missing neighbour probes the pair (0, 0) and the answer is
        // dropped): ONE path for all lanes, four loads in flight, one memory round trip per merge.  With a branch per case the
        // lanes of a wave took up to three round trips in turn.
        uint32_t pa, pb;
        pair_prio2(T, r != kNone ? m : 0u, idr, idl, l != kNone ? m : 0u, &pa, &pb);
        if (r == kNone) pa = kNoPrio;
        if (l == kNone) pb = kNoPrio;
        pq[bp] = pa == kNoPrio ? 0xFFFFFFFFu : (pa << 7) | bp;
        if (l != kNone) pq[l] = pb == kNoPrio ? 0xFFFFFFFFu : (pb << 7) | l;
    }
}

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

// ---- the slow kernel ---------------------------------------------------------------------------------
struct MergeLds {
    int32_t sA[kSlowLds], sB[kSlowLds];
    uint32_t pA[kSlowLds], pB[kSlowLds];
};
struct MidLds {  // one wave's share of the arena
    int32_t sA[kMidSyms], sB[kMidSyms];
    uint32_t pA[kMidSyms], pB[kMidSyms];
};
constexpr int32_t kIdMask = 0x1FFFF;
constexpr int32_t kSelBit = 1 << 30;
constexpr int32_t kDirtyBit = 1 << 29;

// Exact round-based greedy merge of the n symbols in (sa, pa); (sb, pb) is the other buffer of the pair.
//
// The reference applies one merge at a time, always the pair with the smallest key = (rank, position)
// (bpe.go:49-59, priority_queue.go:20-25).  A round here applies MANY of those steps at once:
//   S     = an independent set of pairs that the sequential order would reach with unchanged operands if
//           nothing new interfered: every strict local minimum of the key, plus, for the currently smallest
//           rank r, the occurrences of r alternating from the left inside runs x x x ... (which overlap)
//   beta  = for s in S: max(key(s) + 1, key of the best pair that merging s creates, probed with the
//           neighbours s will see at its turn: products of adjacent lower-key members of S, else the
//           present symbols).  Nothing with a key >= beta(s) may be applied blindly after s.
//   theta = min(beta over S, keys of pairs outside S that no lower-key member of S overlaps)
// Every member of S with key < theta is applied; by induction over the key order these are exactly the next
// steps of the sequential algorithm (everything with a smaller key is applied, dead, or a new pair with
// key >= theta), so the result is identical.  The pair with the smallest key always qualifies.
// Returns the new n; sa/pa are left pointing at the buffers that hold the result.
// WAVE = false: the whole workgroup works on one pre-token; WAVE = true: the calling wave alone does (no workgroup
// barrier anywhere on that path; tmp / tmp64 / segtmp unused).
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the wave's LDS / memory operations complete in order
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}
template <bool WAVE>
__device__ __forceinline__ unsigned long long group_min64(unsigned long long v, unsigned long long* tmp64) {
    if (WAVE) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            unsigned long long o = __shfl_xor(v, d, 64);
            v = o < v ? o : v;
        }
        return v;
    } else {
        return block_min64(v, tmp64);
    }
}
#ifndef L3_ROUND_U
#define L3_ROUND_U 1
#endif
constexpr int kRoundU = L3_ROUND_U;  // elements per thread and step in the probing sweeps of a round
template <bool WAVE>
__device__ uint32_t merge_rounds(const Tables& T, int32_t*& sa, uint32_t*& pa, int32_t*& sb, uint32_t*& pb, uint32_t n,
                                 uint32_t* tmp, unsigned long long* tmp64, Seg* segtmp) {
    // Thread mapping: order-free sweeps go over i = tid, tid + 256, ...; the two sweeps that need the order (parity
    // inside runs, compaction) give every wave one contiguous range, walked 64 symbols at a time with ballots.
    // Either way the 64 lanes of a wave touch 64 consecutive words: coalesced in HBM, conflict-free in LDS.
    constexpr uint32_t kGroup = WAVE ? 64u : (uint32_t)kSlowThreads;
    const uint32_t lane = (uint32_t)lane_id(), wave = WAVE ? 0u : (uint32_t)wave_id();
    const uint32_t tid = WAVE ? lane : threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long best = ~0ull;  // smallest key of the coming round
    for (uint32_t i = tid; i < n; i += kGroup) {
        const unsigned long long v = ((unsigned long long)pa[i] << 32) | i;
        best = v < best ? v : best;
    }
    best = group_min64<WAVE>(best, tmp64);
    for (;;) {
        if (n < 2) break;
        const uint32_t r = (uint32_t)(best >> 32);
        if (r == kNoPrio) break;
        const uint32_t ipos = (uint32_t)best;
        const bool same = (sa[ipos] & kIdMask) == (sa[ipos + 1] & kIdMask);
        const uint32_t lw = ((n + kGroup - 1) / kGroup) * 64u;  // symbols per wave
        const uint32_t w_lo = wave * lw < n ? wave * lw : n;
        const uint32_t w_hi = w_lo + lw < n ? w_lo + lw : n;
        group_sync<WAVE>();
        // (b) the set S
        {
            uint32_t carry = 0;  // same: length of the run of r that ends right before the current 64 symbols
            if (same) {          // runs of r longer than one exist only when both symbols are equal
                Seg sg; sg.cnt = 0; sg.open = 1;
                for (uint32_t base = w_lo; base < w_hi; base += 64) {
                    const uint32_t i = base + lane;
                    const uint32_t vn = w_hi - base < 64u ? w_hi - base : 64u;
                    const unsigned long long full = vn == 64u ? ~0ull : (1ull << vn) - 1ull;
                    const unsigned long long m = __ballot(i < w_hi && pa[i] == r);
                    if (m == full) {
                        sg.cnt += vn;
                    } else {
                        const unsigned long long z = ~m & full;
                        sg.cnt = vn - 1u - (63u - (uint32_t)__builtin_clzll(z));
                        sg.open = 0;
                    }
                }
                if (!WAVE) {
                    if (lane == 0) segtmp[wave] = sg;
                    __syncthreads();
                    Seg c; c.cnt = 0; c.open = 1;
                    for (uint32_t w2 = 0; w2 < wave; w2++) c = seg_comb(c, segtmp[w2]);
                    carry = c.cnt;
                }
            }
            for (uint32_t base = w_lo; base < w_hi; base += 64) {
                const uint32_t i = base + lane;
                const bool in = i < w_hi;
                const uint32_t p = in ? pa[i] : kNoPrio;
                const bool isr = in && p == r;
                bool sel = isr;
                if (same) {
                    const unsigned long long m = __ballot(isr);
                    const unsigned long long z = ~m & lt;
                    const uint32_t idx = z ? lane - (63u - (uint32_t)__builtin_clzll(z)) - 1u : lane + carry;
                    sel = isr && !(idx & 1u);
                    const unsigned long long zz = ~m;
                    carry = zz ? (uint32_t)__builtin_clzll(zz) : carry + 64u;
                }
                if (in) {
                    if (!isr && p != kNoPrio) {  // strict local minimum of (rank, position); pa[n - 1] is kNoPrio
                        const uint32_t pl = i >= 1 ? pa[i - 1] : kNoPrio, pn = pa[i + 1];
                        sel = pl > p && pn >= p;
                    }
                    sa[i] = (sa[i] & kIdMask) | (sel ? kSelBit : 0);
                }
            }
        }
        group_sync<WAVE>();
        // (c) theta.  Everything an element needs is loaded up front (clamped indices, no branches), then its two
        // probes together.  kRoundU elements per thread and step can be batched that way; measured on C3, 2 and 4 are
        // slower than 1 (registers: 201 VGPRs at 4 halve the occupancy), so 1 it is.
        unsigned long long th = ~0ull;
        for (uint32_t i0 = tid; i0 < n; i0 += kGroup * kRoundU) {
            uint32_t pv[kRoundU], m[kRoundU];
            unsigned long long key[kRoundU], k1[kRoundU], k2[kRoundU];
            bool sel[kRoundU], hasl[kRoundU], hasr[kRoundU];
            PairBucket u1[kRoundU], v1[kRoundU], u2[kRoundU], v2[kRoundU];
#pragma unroll
            for (int k = 0; k < kRoundU; k++) {
                const uint32_t i = i0 + (uint32_t)k * kGroup;
                const bool in = i < n;
                const uint32_t ii = in ? i : n - 1u;
                const uint32_t im1 = ii >= 1u ? ii - 1u : 0u, im2 = ii >= 2u ? ii - 2u : 0u;
                const uint32_t ip1 = ii + 1u < n ? ii + 1u : n - 1u, ip2 = ii + 2u < n ? ii + 2u : n - 1u;
                const uint32_t p = pa[ii], pm1 = pa[im1], pm2 = pa[im2], pp1 = pa[ip1], pp2 = pa[ip2];
                const int32_t s0 = sa[ii], sm1 = sa[im1], sm2 = sa[im2], sp1 = sa[ip1], sp2 = sa[ip2];
                const bool valid = in && p != kNoPrio;   // then i + 1 < n: pa[n - 1] is kNoPrio
                sel[k] = valid && (s0 & kSelBit);
                pv[k] = p;
                key[k] = ((unsigned long long)p << 32) | i;
                const bool covered = (i >= 1u && (sm1 & kSelBit) && pm1 <= p) || ((sp1 & kSelBit) && pp1 < p);
                if (valid && !sel[k] && !covered) th = key[k] < th ? key[k] : th;
                m[k] = prio_merged_id(T, p);
                hasl[k] = sel[k] && i >= 1u;
                hasr[k] = sel[k] && i + 2u < n;
                uint32_t lsym = (uint32_t)(sm1 & kIdMask);
                if (i >= 2u && (sm2 & kSelBit) && pm2 <= p) lsym = prio_merged_id(T, pm2);
                uint32_t rsym = (uint32_t)(sp2 & kIdMask);
                if ((sp2 & kSelBit) && pp2 < p) rsym = prio_merged_id(T, pp2);
                const uint32_t a1 = hasl[k] ? lsym : 0u, b1 = hasl[k] ? m[k] : 0u;   // idle lanes probe (0, 0)
                const uint32_t a2 = hasr[k] ? m[k] : 0u, b2 = hasr[k] ? rsym : 0u;
                k1[k] = pair_key(a1, b1);
                k2[k] = pair_key(a2, b2);
                uint32_t h1, h2, h3, h4;
                pair_hash2(a1, b1, T.pair_log2, &h1, &h2);
                pair_hash2(a2, b2, T.pair_log2, &h3, &h4);
                u1[k] = T.pair_buckets[h1]; v1[k] = T.pair_buckets[h2];
                u2[k] = T.pair_buckets[h3]; v2[k] = T.pair_buckets[h4];
            }
#pragma unroll
            for (int k = 0; k < kRoundU; k++) {
                const uint32_t r1 = hasl[k] ? pair_pick(k1[k], u1[k], v1[k]) : kNoPrio;
                const uint32_t r2 = hasr[k] ? pair_pick(k2[k], u2[k], v2[k]) : kNoPrio;
                const uint32_t v = r2 < r1 ? r2 : r1;
                unsigned long long beta = v == kNoPrio ? ~0ull : ((unsigned long long)v << 32);
                if (beta < key[k] + 1) beta = key[k] + 1;
                if (sel[k]) th = beta < th ? beta : th;
            }
        }
        th = group_min64<WAVE>(th, tmp64);
        // (d) apply the members of S below theta, compacting into the other buffer
        uint32_t cnt = 0;
        for (uint32_t base = w_lo; base < w_hi; base += 64) {
            const uint32_t i = base + lane;
            const bool in = i < w_hi;
            const bool dead = in && i >= 1 && (sa[i - 1] & kSelBit) && (((unsigned long long)pa[i - 1] << 32) | (i - 1)) < th;
            cnt += (uint32_t)__popcll(__ballot(in && !dead));
        }
        uint32_t j0 = 0, n2 = cnt;
        if (!WAVE) {
            if (lane == 0) tmp[wave] = cnt;
            __syncthreads();
            n2 = 0;
            for (uint32_t w2 = 0; w2 < (uint32_t)(kSlowThreads / 64); w2++) {
                const uint32_t x = tmp[w2];
                if (w2 < wave) j0 += x;
                n2 += x;
            }
        }
        for (uint32_t base = w_lo; base < w_hi; base += 64) {
            const uint32_t i = base + lane;
            const bool in = i < w_hi;
            const bool dead = in && i >= 1 && (sa[i - 1] & kSelBit) && (((unsigned long long)pa[i - 1] << 32) | (i - 1)) < th;
            const bool live = in && !dead;
            const unsigned long long lm = __ballot(live);
            if (live) {
                const uint32_t j = j0 + (uint32_t)__popcll(lm & lt);
                const uint32_t p = pa[i];
                const int32_t x = sa[i];
                const bool applied = (x & kSelBit) && (((unsigned long long)p << 32) | i) < th;
                sb[j] = applied ? (int32_t)(prio_merged_id(T, p) | (uint32_t)kDirtyBit) : (x & kIdMask);
                pb[j] = p;
            }
            j0 += (uint32_t)__popcll(lm);
        }
        group_sync<WAVE>();
        // (e) re-probe the pairs that touch a merged symbol (batched like (c)); the smallest key of the next round
        // on the way
        best = ~0ull;
        for (uint32_t q0 = tid; q0 < n2; q0 += kGroup * kRoundU) {
            uint32_t pv[kRoundU];
            unsigned long long kk[kRoundU];
            bool in[kRoundU], dirty[kRoundU], probe[kRoundU];
            PairBucket u[kRoundU], v[kRoundU];
#pragma unroll
            for (int k = 0; k < kRoundU; k++) {
                const uint32_t q = q0 + (uint32_t)k * kGroup;
                in[k] = q < n2;
                const uint32_t qq = in[k] ? q : n2 - 1u;
                const bool nxt = qq + 1u < n2;
                const int32_t x = sb[qq];
                const int32_t y = nxt ? sb[qq + 1u] : 0;
                pv[k] = pb[qq];
                dirty[k] = in[k] && ((x | y) & kDirtyBit);
                probe[k] = dirty[k] && nxt;
                const uint32_t a = probe[k] ? (uint32_t)(x & kIdMask) : 0u, b = probe[k] ? (uint32_t)(y & kIdMask) : 0u;
                kk[k] = pair_key(a, b);
                uint32_t h1, h2;
                pair_hash2(a, b, T.pair_log2, &h1, &h2);
                u[k] = T.pair_buckets[h1];
                v[k] = T.pair_buckets[h2];
            }
#pragma unroll
            for (int k = 0; k < kRoundU; k++) {
                const uint32_t q = q0 + (uint32_t)k * kGroup;
                uint32_t p = pv[k];
                if (dirty[k]) {
                    p = probe[k] ? pair_pick(kk[k], u[k], v[k]) : kNoPrio;
                    pb[q] = p;
                }
                const unsigned long long val = ((unsigned long long)p << 32) | q;
                if (in[k]) best = val < best ? val : best;
            }
        }
        best = group_min64<WAVE>(best, tmp64);
        { int32_t* t = sa; sa = sb; sb = t; }
        { uint32_t* t = pa; pa = pb; pb = t; }
        n = n2;
    }
    return n;
}

// One slow entry of known, moderate length (<= kMidBytes source bytes), handled by the calling wave alone: a
// workgroup per entry spends most of its time in barriers and probe round trips at these sizes, four independent
// waves keep four entries in flight instead.  Same steps as the workgroup path below, wave-wide.
// symbols of the entry: one per byte, three (EF BF BD) for a byte that is not valid UTF-8 (state_machine.go:36)
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
    uint16_t ml[kWin];                           // indices (into pl) of the pieces of two and more symbols
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
    // string one byte longer still a prefix of an entry?), and a wave waits for its longest chain; so a step probes
    // FOUR lengths at once -- the bytes are in LDS, the hashes of the longer strings cost nothing to form ahead, and
    // what counts is the first length that fails -- for kWalkPos positions together: a step is one memory round trip.
    const uint32_t p0 = (uint32_t)tid * (uint32_t)kWinPer;
    uint32_t mlen[kWinPer];
    constexpr int kWalkPos = 4;
    static_assert(kWinPer % kWalkPos == 0, "positions are walked in groups");
#pragma unroll
    for (int g = 0; g < kWinPer; g += kWalkPos) {
        uint32_t hh[kWalkPos], ok[kWalkPos];
        uint32_t alive = 0;
        {   // lengths 2 (exact table) and 3, 4, 5
            uint32_t w2[kWalkPos], w3[kWalkPos], w4[kWalkPos], w5[kWalkPos];
#pragma unroll
            for (int j = 0; j < kWalkPos; j++) {
                const uint32_t p = p0 + (uint32_t)(g + j);
                const uint32_t a = wb[p], b = wb[p + 1];
                const uint32_t h2 = cut_hash2(a, b), h3 = cut_hash_add(h2, wb[p + 2]), h4 = cut_hash_add(h3, wb[p + 3]),
                               h5 = cut_hash_add(h4, wb[p + 4]);
                hh[j] = h5;
                const uint32_t b3 = cut_hash_bit(h3, T.bloom_log2), b4 = cut_hash_bit(h4, T.bloom_log2), b5 = cut_hash_bit(h5, T.bloom_log2);
                w2[j] = (T.prefix2[(a << 3) | (b >> 5)] >> (b & 31u)) & 1u;
                w3[j] = (T.prefix_bloom[b3 >> 5] >> (b3 & 31u)) & 1u;
                w4[j] = (T.prefix_bloom[b4 >> 5] >> (b4 & 31u)) & 1u;
                w5[j] = (T.prefix_bloom[b5 >> 5] >> (b5 & 31u)) & 1u;
            }
#pragma unroll
            for (int j = 0; j < kWalkPos; j++) {
                const uint32_t p = p0 + (uint32_t)(g + j);
                const uint32_t room = p < n_ex ? avail - p : 0u;   // bytes from p to the end of what may be matched (>= 1)
                uint32_t l = 1;
                if (room >= 2 && w2[j]) { l = 2; if (room >= 3 && w3[j]) { l = 3; if (room >= 4 && w4[j]) { l = 4; if (room >= 5 && w5[j]) l = 5; } } }
                mlen[g + j] = l;
                ok[j] = l == 5 && room > 5;
                if (ok[j]) alive |= 1u << j;
            }
        }
        while (alive) {   // four more lengths per step
            uint32_t w[kWalkPos][4], hn[kWalkPos];
#pragma unroll
            for (int j = 0; j < kWalkPos; j++) {
                const uint32_t p = p0 + (uint32_t)(g + j);
                const uint32_t at = ok[j] ? p + mlen[g + j] : 0u;   // (idle lanes probe wb[0 ..]: in bounds, ignored)
                uint32_t h = hh[j];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    h = cut_hash_add(h, wb[at + (uint32_t)t]);
                    const uint32_t bit = cut_hash_bit(h, T.bloom_log2);
                    w[j][t] = (T.prefix_bloom[bit >> 5] >> (bit & 31u)) & 1u;
                }
                hn[j] = h;
            }
#pragma unroll
            for (int j = 0; j < kWalkPos; j++) {
                if (!ok[j]) continue;
                const uint32_t p = p0 + (uint32_t)(g + j);
                uint32_t l = mlen[g + j];
                uint32_t lim = avail - p;
                if (lim > (uint32_t)kMaxTokenBytes) lim = (uint32_t)kMaxTokenBytes;
                int t = 0;
                while (t < 4 && l < lim && w[j][t]) { l++; t++; }
                mlen[g + j] = l;
                hh[j] = hn[j];
                if (!(t == 4 && l < lim)) { ok[j] = 0; alive &= ~(1u << j); }
            }
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
    uint32_t n_multi = 0;
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < n_pieces; i0 += kSlowThreads) {
        const uint32_t i = i0 + (uint32_t)tid;
        const bool have = i < n_pieces;
        const uint32_t s = have ? P.pl[i] : 0u;
        const uint32_t e = have ? (i + 1 < k_total ? (uint32_t)P.pl[i + 1] : last_end) : 0u;
        const bool multi = have && (e - s >= 2 || whole_single);
        if (have && !multi) { img[(uint64_t)o0 + s] = H.byte_tok[wb[s]]; own_cnt++; }
        uint32_t tot;
        const uint32_t ex = block_escan<kSlowThreads / 64>(multi ? 1u : 0u, P.tmp, &tot);
        if (multi) P.ml[n_multi + ex] = (uint16_t)i;
        n_multi += tot;
    }
    __syncthreads();
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

#ifndef L3_SLOW_OCC
#define L3_SLOW_OCC 1
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
// workgroup of kScanBlock threads.  Runs in the launch of block_totals_kernel, i.e. after slow_kernel has finished.
constexpr int kFinishThreads = 1024;
constexpr int kFinishPer = (kWin + kFinishThreads - 1) / kFinishThreads;
__device__ void finish_window(const Workspace& W, uint32_t w, unsigned long long* wsum) {
    const int tid = threadIdx.x;
    const WinItem it = W.win[w];
    const uint64_t win0 = W.slow[it.entry].win0;
    const uint32_t* live = W.win_live + (win0 & kEntryWinMask);
    const uint32_t mine = live[it.k];
    if (mine == 0) return;   // (uniform)
    const uint32_t len = W.slow[it.entry].len, L = W.slow[it.entry].pad;
    const uint64_t slots = (win0 & kEntrySequential) ? 3ull * len : (uint64_t)len;
    int32_t* const fin = W.pool + W.slow[it.entry].pool_off;
    const int32_t* const img = fin + slots;
    // tokens of the windows before this one
    unsigned long long before = 0;
    for (uint32_t j = (uint32_t)tid; j < it.k; j += kFinishThreads) before += live[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    __syncthreads();
    if (lane_id() == 0) wsum[tid >> 6] = before;
    __syncthreads();
    before = 0;
    for (int i = 0; i < kFinishThreads / 64; i++) before += wsum[i];
    __syncthreads();
    // compaction of the window's slots: kFinishPer consecutive slots per thread
    const uint64_t lo = (uint64_t)it.k * (uint64_t)kWin;
    int32_t v[kFinishPer];
    uint32_t n_live = 0;
#pragma unroll
    for (int j = 0; j < kFinishPer; j++) {
        const uint32_t o = (uint32_t)tid * (uint32_t)kFinishPer + (uint32_t)j;
        const uint64_t p = lo + o;
        v[j] = (o < (uint32_t)kWin && p < (uint64_t)L) ? img[p] : -1;
        if (v[j] >= 0) n_live++;
    }
    uint32_t inc = n_live;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane_id() >= d) inc += o;
    }
    if (lane_id() == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int i = 0; i < (tid >> 6); i++) pre += (uint32_t)wsum[i];
    __syncthreads();
    uint64_t o = before + pre + inc - n_live;
#pragma unroll
    for (int j = 0; j < kFinishPer; j++)
        if (v[j] >= 0) fin[o++] = v[j];
}

// ---- per-tile output bases: block totals, scan of the block totals -----------------------------------------
constexpr int kScanThreads = 1024;
constexpr int kScanBlock = 1024;  // tiles per block total
static_assert(kFinishThreads == kScanBlock, "finish_window runs in block_totals_kernel's workgroups");
// Workgroups [0, n_blocks): the totals; the workgroups behind them move the tokens of the long pre-tokens from their
// images to their final place (same launch: both only need what the kernels before have left).
__global__ __launch_bounds__(kScanBlock) void block_totals_kernel(Workspace W, uint64_t n_tiles, uint64_t n_docs, uint32_t n_blocks) {
    __shared__ unsigned long long wsum[kScanBlock / 64];
    if (W.status->flags & (kFatalFlags | kGiantPending | kSeqPending)) return;
    if (blockIdx.x >= n_blocks) {
        if (W.status->flags & kRetryFlags) return;
        const uint32_t n_items = W.status->slow_q.count;
        for (uint32_t w = blockIdx.x - n_blocks; w < n_items; w += gridDim.x - n_blocks) { finish_window(W, w, wsum); __syncthreads(); }
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


// ---- decode (tokenizer.go:339-354): ids -> bytes, out-of-range ids skipped ------------------------------
// Sequence borders play no part in the bytes: the output of the batch is the concatenation of the bytes of all ids in
// order, and sequence s starts at the byte position of id id_off[s].  So the work is cut into blocks of kDecBlock ids
// whatever the sequences are (one 4 MiB document and 65,536 short ones cost the same):
//   decode_sum_kernel    bytes of every block of ids
//   decode_scan_kernel   exclusive scan of the block sums (one workgroup; 3,400 blocks for 14 M ids), total
//   decode_write_kernel  per block: positions by a workgroup scan, bytes staged in LDS and stored as aligned dwords,
//                        out_off[s] for the sequences that start inside the block
constexpr int kDecThreads = 256, kDecPer = 4, kDecRound = kDecThreads * kDecPer, kDecRounds = 4, kDecBlock = kDecRound * kDecRounds;
constexpr int kDecStage = 16384;   // bytes of one round staged in LDS (1,024 ids: 4.5 KB typical, 128 KB at the very most)

__device__ __forceinline__ uint32_t dec_len(const uint32_t* tok_off, int n_tokens, int32_t id, uint32_t* src) {
    if (id < 0 || id >= n_tokens) { *src = 0; return 0; }
    const uint32_t a = tok_off[id];
    *src = a;
    return tok_off[id + 1] - a;
}

__global__ __launch_bounds__(kDecThreads) void decode_sum_kernel(const uint32_t* tok_off, int n_tokens, const int32_t* ids,
                                                                 uint64_t n_ids, uint64_t* block_base) {
    __shared__ uint32_t tmp[16];
    const uint64_t lo = (uint64_t)blockIdx.x * kDecBlock;
    uint32_t mine = 0, src;
#pragma unroll 4
    for (int k = 0; k < kDecBlock / kDecThreads; k++) {
        const uint64_t i = lo + (uint64_t)k * kDecThreads + threadIdx.x;
        if (i < n_ids) mine += dec_len(tok_off, n_tokens, ids[i], &src);
    }
    uint32_t tot;
    (void)block_escan<kDecThreads / 64>(mine, tmp, &tot);
    if (threadIdx.x == 0) block_base[blockIdx.x + 1] = tot;   // decode_scan_kernel turns these into positions
}

__global__ __launch_bounds__(kScanThreads) void decode_scan_kernel(uint64_t* out_off, uint64_t n_seqs, unsigned long long* total) {
    __shared__ unsigned long long wsum[kScanThreads / 64];
    __shared__ unsigned long long carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_s = 0; out_off[0] = 0; }
    __syncthreads();
    for (uint64_t base = 0; base < n_seqs; base += kScanThreads) {
        uint64_t s = base + tid;
        unsigned long long v = s < n_seqs ? out_off[s + 1] : 0ull, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(inc, d, 64);
            if (lane_id() >= d) inc += o;
        }
        if (lane_id() == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned long long pre = carry_s, tot = 0;
        for (int i = 0; i < kScanThreads / 64; i++) {
            if (i < (tid >> 6)) pre += wsum[i];
            tot += wsum[i];
        }
        if (s < n_seqs) out_off[s + 1] = pre + inc;
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
}

__global__ __launch_bounds__(kDecThreads) void decode_write_kernel(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens,
                                                                   const int32_t* ids, uint64_t n_ids, const uint64_t* id_off,
                                                                   uint64_t n_seqs, const uint64_t* block_base, uint8_t* out,
                                                                   uint64_t out_cap, uint64_t* out_off,
                                                                   const unsigned long long* total) {
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t pos[kDecRound];          // byte position of every id of the round, relative to the round
    __shared__ __align__(16) uint8_t stage[kDecStage];
    __shared__ unsigned long long s_first;
    const int tid = threadIdx.x;
    const bool store = *total <= out_cap;        // too small a buffer: offsets only (the caller learns the size and returns)
    const uint64_t lo = (uint64_t)blockIdx.x * kDecBlock;
    const uint64_t hi = lo + kDecBlock < n_ids ? lo + kDecBlock : n_ids;
    if (tid == 0) {   // first sequence that starts at or after this block's first id
        uint64_t a = 0, b = n_seqs + 1;
        while (a < b) { const uint64_t m = (a + b) / 2; if (id_off[m] < lo) a = m + 1; else b = m; }
        s_first = a;
    }
    __syncthreads();
    uint64_t s_cur = s_first;                    // the same in every thread from here on
    uint64_t run = block_base[blockIdx.x];       // byte position of the round's first id
    for (uint64_t rlo = lo; rlo < hi; rlo += kDecRound) {
        const uint64_t rhi = rlo + kDecRound < hi ? rlo + kDecRound : hi;
        const uint64_t i0 = rlo + (uint64_t)tid * kDecPer;
        uint32_t len[kDecPer], src[kDecPer], tsum = 0;
#pragma unroll
        for (int k = 0; k < kDecPer; k++) {
            len[k] = (i0 + k < rhi) ? dec_len(tok_off, n_tokens, ids[i0 + k], &src[k]) : (src[k] = 0, 0u);
            tsum += len[k];
        }
        uint32_t tot;
        uint32_t p = block_escan<kDecThreads / 64>(tsum, tmp, &tot);
        const uint32_t p0 = p;
#pragma unroll
        for (int k = 0; k < kDecPer; k++) { pos[tid * kDecPer + k] = p; p += len[k]; }
        const bool staged = tot <= (uint32_t)kDecStage;
        if (store) {
            uint32_t q = p0;
            if (staged) {
#pragma unroll
                for (int k = 0; k < kDecPer; k++) {
                    for (uint32_t j = 0; j < len[k]; j++) stage[q + j] = tok_bytes[src[k] + j];
                    q += len[k];
                }
            } else {   // a round of very long vocabulary entries: straight to memory
                uint8_t* dst = out + run + q;
#pragma unroll
                for (int k = 0; k < kDecPer; k++) {
                    for (uint32_t j = 0; j < len[k]; j++) dst[j] = tok_bytes[src[k] + j];
                    dst += len[k];
                }
            }
        }
        __syncthreads();
        // sequences that start inside the round (ascending offsets: the hits are a prefix of what is left)
        for (;;) {
            const uint64_t s = s_cur + (uint64_t)tid;
            uint64_t v = ~0ull;
            if (tid < 64 && s <= n_seqs) v = id_off[s];
            const bool hit = v < rhi;
            if (hit) out_off[s] = run + pos[v - rlo];
            const int cnt = __syncthreads_count(hit);
            s_cur += (uint64_t)cnt;
            if (cnt < 64) break;
        }
        if (store && staged && tot) {
            // aligned dwords of the output: dword j covers stage bytes [4 j - a, 4 j - a + 4), a = misalignment of the start
            uint8_t* dst0 = out + run;
            const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(dst0) & 3u);
            const uint32_t n_dw = (a + tot + 3u) >> 2;
            uint32_t* dw = reinterpret_cast<uint32_t*>(dst0 - a);
            for (uint32_t j = tid; j < n_dw; j += kDecThreads) {
                const int32_t b0 = (int32_t)(4u * j) - (int32_t)a;
                if (b0 >= 0 && (uint32_t)b0 + 4u <= tot) {
                    dw[j] = (uint32_t)stage[b0] | ((uint32_t)stage[b0 + 1] << 8) | ((uint32_t)stage[b0 + 2] << 16) |
                            ((uint32_t)stage[b0 + 3] << 24);
                } else {
                    for (int k = 0; k < 4; k++) {
                        const int32_t bk = b0 + k;
                        if (bk >= 0 && (uint32_t)bk < tot) dst0[bk] = stage[bk];
                    }
                }
            }
        }
        run += tot;
        __syncthreads();
    }
    // sequences that start at the very end (empty ones behind the last id, and the closing offset)
    if (blockIdx.x == gridDim.x - 1)
        for (uint64_t s = s_cur + (uint64_t)tid; s <= n_seqs; s += kDecThreads) out_off[s] = run;
}

}  // namespace

#define L3_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e_ = hipGetLastError();        \
        if (e_ != hipSuccess) return e_;          \
    } while (0)
#define L3_EV(i)                                                          \
    do {                                                                  \
        if (ev && ((ev_mask >> (i)) & 1u)) {                              \
            hipError_t e_ = hipEventRecord(ev[i], stream);                \
            if (e_ != hipSuccess) return e_;                              \
        }                                                                 \
    } while (0)

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
#ifndef L3_SLOW_GRID
#define L3_SLOW_GRID 512
#endif
static hipError_t launch_totals_scan(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    const uint64_t n_blocks = (n_tiles + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(block_totals_kernel, dim3((uint32_t)n_blocks + L3_FINISH_GRID), dim3(kScanBlock), 0, stream, W, n_tiles, A.n_docs,
                       (uint32_t)n_blocks);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

__global__ void status_and_kernel(Status* st, uint32_t mask) { st->flags &= mask; }

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
hipError_t launch_copy_runs(const Workspace& W, const EncodeArgs& A, long long n_big, long long n_huge, hipStream_t stream) {
    hipLaunchKernelGGL(copy_runs_kernel, dim3(1024), dim3(kTileThreads), 0, stream, W, A, n_big, n_huge);
    L3_LAUNCH_CHECK();
    return hipSuccess;
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

hipError_t launch_encode_tail(const Workspace& W, const EncodeArgs& A, hipStream_t stream) {
    hipError_t e_ = launch_totals_scan(W, A, stream);
    if (e_ != hipSuccess) return e_;
    hipLaunchKernelGGL(emit_kernel, dim3((uint32_t)((num_tiles(A.n_bytes) + kTileThreads / 64 - 1) / (kTileThreads / 64))), dim3(kTileThreads), 0, stream, W, A,
                       (uint32_t)num_tiles(A.n_bytes));
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_encode(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream, hipEvent_t* ev,
                         bool emit_only, uint32_t ev_mask) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    if (n_tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (!emit_only) {
        L3_EV(kEvStart);
        const uint32_t dblocks = (uint32_t)((A.n_docs + 1 + 255) / 256);
        hipLaunchKernelGGL(docs_kernel, dim3(dblocks), dim3(256), 0, stream, W, A, n_tiles);
        L3_LAUNCH_CHECK();
        L3_EV(kEvDocs);
        hipLaunchKernelGGL(tile_kernel, dim3((uint32_t)n_tiles), dim3(kTileThreads), 0, stream, T, W, A);
        L3_LAUNCH_CHECK();
        L3_EV(kEvTile);
        // slow_kernel's workgroups take entries from a queue: no more of them than the GPU holds at once (an empty list
        // then costs a short launch); the wave-per-entry pass inside merge_kernel's launch gets the same number
#ifndef L3_SLOW_GRID
#define L3_SLOW_GRID 512
#endif
        uint64_t sgrid = W.slow_cap < L3_SLOW_GRID ? W.slow_cap : L3_SLOW_GRID;
        if (W.slow_grid && W.slow_grid < sgrid) sgrid = W.slow_grid;
        if (sgrid == 0) sgrid = 1;
        {
            uint64_t mgrid = ((W.miss_shard_cap + kMergeThreads - 1) / kMergeThreads) * kMissShards;  // windows at most
#ifndef L3_MERGE_GRID
#define L3_MERGE_GRID 2048
#endif
            if (mgrid > L3_MERGE_GRID) mgrid = L3_MERGE_GRID;
            if (mgrid == 0) mgrid = 1;
            hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)(mgrid + sgrid)), dim3(kMergeThreads), 0, stream, T, W, A,
                               (uint32_t)mgrid);
            L3_LAUNCH_CHECK();
        }
        L3_EV(kEvMerge);
        hipLaunchKernelGGL(slow_kernel, dim3((uint32_t)sgrid), dim3(kSlowThreads), 0, stream, T, W, A, 0);
        L3_LAUNCH_CHECK();
        L3_EV(kEvSlow);
        hipError_t e2_ = launch_totals_scan(W, A, stream);
        if (e2_ != hipSuccess) return e2_;
        L3_EV(kEvScan);
        L3_EV(kEvHost);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((uint32_t)((n_tiles + kTileThreads / 64 - 1) / (kTileThreads / 64))), dim3(kTileThreads), 0, stream, W, A, (uint32_t)n_tiles);
    L3_LAUNCH_CHECK();
    if (emit_only) {   // the re-run after the caller enlarged its buffer: the flags of the first run are gone
        hipError_t e_ = launch_emit_generic(W, A, stream);
        if (e_ != hipSuccess) return e_;
        e_ = launch_copy_runs(W, A, -1, -1, stream);
        if (e_ != hipSuccess) return e_;
    }
    L3_EV(kEvEmit);
    return hipSuccess;
}

uint64_t decode_blocks(uint64_t n_ids) { return n_ids ? (n_ids + kDecBlock - 1) / kDecBlock : 1; }

hipError_t launch_decode(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens, const int32_t* ids, uint64_t n_ids,
                         const uint64_t* id_off, uint64_t n_seqs, uint64_t* block_base, uint8_t* out, uint64_t out_cap,
                         uint64_t* out_off, unsigned long long* total, hipStream_t stream) {
    const uint64_t n_blocks = decode_blocks(n_ids);   // block_base: n_blocks + 1 entries
    if (n_blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_sum_kernel, dim3((uint32_t)n_blocks), dim3(kDecThreads), 0, stream, tok_off, n_tokens, ids, n_ids,
                       block_base);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, block_base, n_blocks, total);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_write_kernel, dim3((uint32_t)n_blocks), dim3(kDecThreads), 0, stream, tok_off, tok_bytes, n_tokens,
                       ids, n_ids, id_off, n_seqs, block_base, out, out_cap, out_off, total);
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
