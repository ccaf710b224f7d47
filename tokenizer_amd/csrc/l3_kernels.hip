// l3_kernels.hip -- the encode pipeline on gfx950 (wave64, LDS-staged tiles).
//
// Replaces the loops of llama3.Tokenizer.Encode (reference llama3/tokenizer.go:217-265) for a
// batch of documents held as one concatenated byte array + offsets:
//
//   validate/docs  doc_off sanity; first document of every tile
//   tile_kernel    one workgroup per 4 KiB of text (+ halos) staged in LDS:
//                  UTF-8 decode + character class per byte      (state_machine.go:36,310-320)
//                  special-token literals                       (special.go:12,48-83)
//                  position-local start-of-pre-token predicate  (state_machine.go:82-284, l3_rules.h)
//                  whole-pre-token vocabulary probe             (bpe.go:27-33)
//                  in-LDS greedy merge, one lane per pre-token  (bpe.go:36-71)
//                  compaction of the surviving ids into the tile's token stream
//   slow_kernel    one workgroup per pre-token the tile kernel deferred (invalid UTF-8, long, or
//                  ending beyond the halo): exact round-based parallel greedy merge
//   scan_kernel    exclusive scan of per-tile token totals
//   emit_kernel    scatter to the final token array, per-document offsets, BOS/EOS (tokenizer.go:225-262)
//
// All cross-workgroup data flows through kernel boundaries on one stream; atomics are used only
// for counters that later kernels read.
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"

namespace l3 {

namespace {

constexpr int kPer = kRegion / kTileThreads;  // consecutive region positions owned by one thread
static_assert(kPer * kTileThreads == kRegion, "region must split evenly over the workgroup");
static_assert(kPer <= 32, "per-thread masks are 32 bit");
constexpr int kBitWords = kRegion / 32 + 2;
constexpr uint32_t I_DOFF_SHIFT = 12;  // bits 12..13 of an info word: offset inside a digit run, mod 3

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- workgroup primitives (256 threads = 4 waves) --------------------------------------------
__device__ __forceinline__ uint32_t wave_iscan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
// exclusive sum scan over the workgroup; tmp: >= 4 words of LDS.  Contains two barriers.
__device__ __forceinline__ uint32_t block_escan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    uint32_t inc = wave_iscan(v);
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kTileThreads / 64; i++) {
        uint32_t x = tmp[i];
        if (i < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
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
    for (int i = 1; i < kTileThreads / 64; i++) r = tmp[i] < r ? tmp[i] : r;
    __syncthreads();
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
__device__ __forceinline__ Seg block_seg_escan(Seg v, Seg* tmp) {
    Seg inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Seg t;
        t.cnt = __shfl_up(inc.cnt, d, 64);
        t.open = __shfl_up(inc.open, d, 64);
        if (lane_id() >= d) inc = seg_comb(t, inc);
    }
    Seg ex;
    ex.cnt = __shfl_up(inc.cnt, 1, 64);
    ex.open = __shfl_up(inc.open, 1, 64);
    if (lane_id() == 0) { ex.cnt = 0; ex.open = 1; }
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    Seg base; base.cnt = 0; base.open = 1;
    for (int i = 0; i < wave_id(); i++) base = seg_comb(base, tmp[i]);
    __syncthreads();
    return seg_comb(base, ex);
}

// ---- the LDS image of one region of text --------------------------------------------------------
struct RegionLds {
    uint8_t b[kRegion + 16];     // raw bytes (16-byte aligned for vector stores)
    uint32_t a[kRegion + 8];     // info word per byte; later: token id / kDead / kSlowMark / kMissMark
    uint32_t p[kRegion + 8];     // special-token id; later: pair priority / deferred length
    uint32_t dsb[kBitWords];     // document-start bits
    uint32_t stb[kBitWords];     // pre-token-start bits
    uint16_t list[kTile];        // compacted local start offsets (pre-tokens, then merge candidates)
};
enum { V_NPRE = 0, V_FIRST, V_ELAST, V_NEED, V_FB, V_SLOW0, V_NMISS, V_END, V_COUNT = 16 };

// byte source + info source over the region, in LOCAL coordinates [0, n)
struct LdsSrc {
    const RegionLds* R;
    int n;
    bool hard;  // position n is the end of the text / document (a hard barrier)
    __device__ uint32_t b(int64_t q) const { return R->b[q]; }
    __device__ bool ds(int64_t q) const { return (q == n && hard) || ((R->dsb[q >> 5] >> (q & 31)) & 1u); }
    __device__ int64_t lo() const { return 0; }
    __device__ int64_t hi() const { return n; }
    __device__ uint32_t info(int64_t q) const { return R->a[q]; }
};
// byte source over global memory restricted to one segment [lo, hi)
struct GlobalSrc {
    const uint8_t* t;
    int64_t lo_, hi_;
    __device__ uint32_t b(int64_t q) const { return t[q]; }
    __device__ bool ds(int64_t q) const { return q == lo_ || q == hi_; }
    __device__ int64_t lo() const { return lo_; }
    __device__ int64_t hi() const { return hi_; }
};

__device__ __forceinline__ void region_load(RegionLds& R, const uint8_t* text, uint64_t g_lo, int n) {
    const uint8_t* src = text + g_lo;
    const bool aligned = (((uintptr_t)src) & 15u) == 0;
    for (int g = threadIdx.x; g * 16 < n + 16; g += blockDim.x) {
        int o = g * 16;
        if (aligned && o + 16 <= n) {
            *reinterpret_cast<uint4*>(&R.b[o]) = *reinterpret_cast<const uint4*>(src + o);
        } else {
            for (int k = 0; k < 16; k++)
                if (o + k < kRegion + 16) R.b[o + k] = (o + k < n) ? src[o + k] : (uint8_t)0;
        }
    }
}

// info words (class, rune start, validity) + special-token overlay.  dsb must be final; contains barriers.
__device__ __forceinline__ void region_classify(RegionLds& R, const LdsSrc& src, const Tables& T) {
    const int c0 = threadIdx.x * kPer;
    const int n = src.n;
#pragma unroll 1
    for (int j = 0; j < kPer; j++) {
        int q = c0 + j;
        if (q < n) R.a[q] = byte_info(src, q, T.cls_stage1, T.cls_blocks);
        else if (q < kRegion + 8) R.a[q] = I_SEG | I_RUNE;
    }
    if (threadIdx.x < 8) R.a[kRegion + threadIdx.x] = I_SEG | I_RUNE;
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < kPer; j++) {
        int q = c0 + j;
        if (q >= n || R.b[q] != '<') continue;
        int canon = 0;
        int len = special_match(src, q, &canon);
        if (!len) continue;
        int32_t id = T.special_id[canon];
        atomicOr(&R.a[q], I_SEG);
        atomicOr(&R.a[q + len], I_SEG);
        if (id != 0) {  // tokenizer.go:236: only a piece whose lookup is non-zero is emitted as special
            atomicOr(&R.a[q], I_SPECIAL);
            R.p[q] = (uint32_t)id;
            for (int k = 1; k < len; k++) R.a[q + k] = I_SPBODY;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int find_next_bit(const uint32_t* bits, int from, int limit) {
    if (from >= limit) return -1;
    int w = from >> 5;
    uint32_t x = bits[w] & (0xFFFFFFFFu << (from & 31));
    const int wl = (limit - 1) >> 5;
    for (;;) {
        if (x) {
            int q = (w << 5) + __builtin_ctz(x);
            return q < limit ? q : -1;
        }
        if (++w > wl) return -1;
        x = bits[w];
    }
}

// number of consecutive Nd runes immediately before global position `pos` inside [seg_lo, seg_hi)
// (whole workgroup; pathological digit runs longer than the look-behind halo only)
__device__ uint32_t digits_before_global(const uint8_t* text, int64_t seg_lo, int64_t seg_hi, int64_t pos,
                                         const Tables& T, unsigned long long* tmp /* 8 words */) {
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
    }
    return total;
}

// ---- validate + first document per tile -----------------------------------------------------------
__global__ void validate_kernel(Workspace W, EncodeArgs A) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > A.n_docs) return;
    uint64_t off = A.doc_off[d];
    uint64_t prev = d ? A.doc_off[d - 1] : 0;
    bool bad = (d == 0 && off != 0) || (d == A.n_docs && off != A.n_bytes) || off < prev || off > A.n_bytes;
    if (bad) atomicOr(&W.status->flags, kBadOffsets);
}

__global__ void docs_kernel(Workspace W, EncodeArgs A, uint64_t n_tiles) {
    if (W.status->flags & kBadOffsets) return;
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > A.n_docs) return;
    uint64_t off = A.doc_off[d];
    uint64_t t_first = d ? A.doc_off[d - 1] / kTile + 1 : 0;   // tiles whose start lies in (prev, off]
    uint64_t t_last = off / kTile;
    if (t_last > n_tiles - 1) t_last = n_tiles - 1;
    for (uint64_t t = t_first; t <= t_last; t++) W.tile_doc0[t] = (uint32_t)d;
    if (d == A.n_docs) W.tile_doc0[n_tiles] = (uint32_t)(A.n_docs + 1);
}

// ---- the tile kernel ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kTileThreads) void tile_kernel(Tables T, Workspace W, EncodeArgs A) {
    __shared__ __align__(16) RegionLds R;
    __shared__ uint32_t tmp[16];
    __shared__ unsigned long long tmp64[8];
    __shared__ Seg segtmp[4];
    __shared__ int vars[V_COUNT];

    if (W.status->flags & kBadOffsets) return;
    const int tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint64_t t_lo = tile * (uint64_t)kTile;
    const uint64_t t_hi = t_lo + kTile < A.n_bytes ? t_lo + kTile : A.n_bytes;
    const uint64_t g_lo = t_lo >= (uint64_t)kBack ? t_lo - kBack : 0;
    const uint64_t g_hi = t_lo + kTile + kFwd < A.n_bytes ? t_lo + kTile + kFwd : A.n_bytes;
    const int n = (int)(g_hi - g_lo);
    const int own_lo = (int)(t_lo - g_lo), own_hi = (int)(t_hi - g_lo);
    const bool hard = g_hi == A.n_bytes;
    const int eval_hi = hard ? n : n - kEvalSlack;
    const int c0 = tid * kPer;
    const uint32_t d0 = W.tile_doc0[tile];

    // -- stage bytes, document-start bits
    for (int i = tid; i < kBitWords; i += kTileThreads) { R.dsb[i] = 0; R.stb[i] = 0; }
    region_load(R, A.text, g_lo, n);
    __syncthreads();
    if (tid == 0 && d0 > 0) {
        uint64_t pos = A.doc_off[d0 - 1];
        if (pos >= g_lo) atomicOr(&R.dsb[(pos - g_lo) >> 5], 1u << ((pos - g_lo) & 31));
    }
    for (uint64_t d = (uint64_t)d0 + tid; d <= A.n_docs; d += kTileThreads) {
        uint64_t pos = A.doc_off[d];
        if (pos >= g_hi) break;
        atomicOr(&R.dsb[(pos - g_lo) >> 5], 1u << ((pos - g_lo) & 31));
    }
    __syncthreads();

    LdsSrc src{&R, n, hard};
    region_classify(R, src, T);

    // -- offset of every Nd rune inside its digit run, mod 3 (state_machine.go:178-191: groups of <= 3)
    {
        Seg sg; sg.cnt = 0; sg.open = 1;
        int first_break = kRegion + 1;
#pragma unroll 1
        for (int j = 0; j < kPer; j++) {
            int q = c0 + j;
            if (q >= n) break;
            uint32_t w = R.a[q];
            if (!(w & I_RUNE)) continue;
            bool isn = (w & I_CLS) == C_N;
            if ((w & I_SEG) || !isn) {
                sg.cnt = 0; sg.open = 0;
                if (first_break > kRegion) first_break = q;
            }
            if (isn) { R.a[q] = w | ((sg.cnt % 3u) << I_DOFF_SHIFT); sg.cnt++; }
        }
        Seg carry = block_seg_escan(sg, segtmp);
        if (carry.cnt % 3u) {
#pragma unroll 1
            for (int j = 0; j < kPer; j++) {
                int q = c0 + j;
                if (q >= n || q >= first_break) break;
                uint32_t w = R.a[q];
                if (!(w & I_RUNE) || (w & I_CLS) != C_N) continue;
                uint32_t off = (((w >> I_DOFF_SHIFT) & 3u) + carry.cnt) % 3u;
                R.a[q] = (w & ~(3u << I_DOFF_SHIFT)) | (off << I_DOFF_SHIFT);
            }
        }
        __syncthreads();
        // a digit run that fills the whole look-behind halo: count it in global memory
        if (tid == 0) {
            int need = 0;
            if (g_lo > 0 && own_lo < n) {
                uint32_t w = R.a[own_lo];
                if ((w & I_RUNE) && (w & I_CLS) == C_N && !(w & I_SEG)) {
                    need = 1;
                    for (int q = own_lo - 1; q >= 8; q--) {
                        uint32_t w2 = R.a[q];
                        if (w2 & I_SPBODY) { need = 0; break; }
                        if (!(w2 & I_RUNE)) continue;
                        if ((w2 & I_CLS) != C_N || (w2 & I_SEG)) { need = 0; break; }
                    }
                }
            }
            vars[V_NEED] = need;
            vars[V_FB] = n;
        }
        __syncthreads();
        if (vars[V_NEED]) {
            int64_t seg_lo = (int64_t)A.doc_off[d0 - 1], seg_hi = (int64_t)A.doc_off[d0];
            uint32_t cnt = digits_before_global(A.text, seg_lo, seg_hi, (int64_t)t_lo, T, tmp64);
            uint32_t old = (R.a[own_lo] >> I_DOFF_SHIFT) & 3u;
            uint32_t adj = (cnt % 3u + 3u - old) % 3u;
            // first break at or after own_lo ends the run
#pragma unroll 1
            for (int j = 0; j < kPer; j++) {
                int q = c0 + j;
                if (q >= n) break;
                if (q <= own_lo) continue;
                uint32_t w = R.a[q];
                if (w & I_SPBODY) { atomicMin(&vars[V_FB], q); break; }
                if (!(w & I_RUNE)) continue;
                if ((w & I_CLS) != C_N || (w & I_SEG)) { atomicMin(&vars[V_FB], q); break; }
            }
            __syncthreads();
            const int fb = vars[V_FB];
            if (adj) {
#pragma unroll 1
                for (int j = 0; j < kPer; j++) {
                    int q = c0 + j;
                    if (q < own_lo || q >= fb || q >= n) continue;
                    uint32_t w = R.a[q];
                    if (!(w & I_RUNE)) continue;
                    uint32_t off = (((w >> I_DOFF_SHIFT) & 3u) + adj) % 3u;
                    R.a[q] = (w & ~(3u << I_DOFF_SHIFT)) | (off << I_DOFF_SHIFT);
                }
            }
            __syncthreads();
        }
    }

    // -- start-of-pre-token predicate on every decidable position from the tile start on
    uint32_t mymask = 0, ownmask = 0;
#pragma unroll 1
    for (int j = 0; j < kPer; j++) {
        int q = c0 + j;
        if (q < own_lo || q >= eval_hi) continue;
        uint32_t w = R.a[q];
        if (!(w & I_RUNE) || (w & I_SPBODY)) continue;
        if ((A.flags & 0x400u) ? ((w & I_CLS) >= C_SP || (w & I_SEG)) : starts_pretoken(src, q, (int)((w >> I_DOFF_SHIFT) & 3u))) {
            mymask |= 1u << j;
            if (q < own_hi) ownmask |= 1u << j;
        }
    }
    if (mymask) {
        int wd = c0 >> 5, sh = c0 & 31;
        atomicOr(&R.stb[wd], mymask << sh);
        if (sh + kPer > 32) atomicOr(&R.stb[wd + 1], mymask >> (32 - sh));
    }
    if (A.kinds) {
        for (int j = 0; j < kPer; j++) {
            int q = c0 + j;
            if (q < own_lo || q >= own_hi) continue;
            A.kinds[g_lo + q] = (ownmask >> j) & 1u ? ((R.a[q] & I_SPECIAL) ? 2 : 1) : 0;
        }
    }
    uint32_t n_pre;
    {
        uint32_t ex = block_escan(__popc(ownmask), tmp, &n_pre);
        uint32_t m = ownmask;
        while (m) {
            int j = __builtin_ctz(m);
            m &= m - 1;
            R.list[ex++] = (uint16_t)(c0 + j);
        }
    }
    __syncthreads();
    if (tid == 0) {
        vars[V_FIRST] = n_pre ? R.list[0] : 0;
        if (!n_pre) vars[V_ELAST] = 0;
    }

    // -- per pre-token: special id / whole-string vocabulary probe / defer
#pragma unroll 1
    for (uint32_t k = tid; k < n_pre; k += kTileThreads) {
        const int s = R.list[k];
        int e = find_next_bit(R.stb, s + 1, eval_hi);
        if (e < 0 && hard) e = n;
        const uint32_t w = R.a[s];
        if (k == n_pre - 1) vars[V_ELAST] = e < 0 ? s + 1 : e;
        if (e < 0) {  // ends beyond what this tile can see
            R.a[s] = (uint32_t)kSlowMark; R.p[s] = 0;
            continue;
        }
        const int len = e - s;
        if (w & I_SPECIAL) {
            R.a[s] = R.p[s];
            for (int q = s + 1; q < e; q++) R.a[q] = (uint32_t)kDead;
            continue;
        }
        uint32_t inval = 0;
        for (int q = s; q < e; q++) inval |= R.a[q] & I_INVALID;
        int32_t id = -1;
        if (A.flags & 0x200u) id = 1;
        else if (!inval && len <= kMaxTokenBytes) id = vocab_lookup(T, [&](uint32_t i) { return (uint32_t)R.b[s + i]; }, (uint32_t)len);
        if (id >= 0) {
            R.a[s] = (uint32_t)id;
            for (int q = s + 1; q < e; q++) R.a[q] = (uint32_t)kDead;
        } else if (!inval && len <= kLaneMax) {
            R.a[s] = (uint32_t)kMissMark; R.p[s] = (uint32_t)len;
        } else {
            R.a[s] = (uint32_t)kSlowMark; R.p[s] = (uint32_t)len;
            for (int q = s + 1; q < e; q++) R.a[q] = (uint32_t)kDead;
        }
    }
    __syncthreads();
    const int first = vars[V_FIRST], e_last = vars[V_ELAST];

    // -- gather the pre-tokens that need merging, then merge each in place (one lane per pre-token)
    uint32_t n_miss;
    {
        uint32_t cnt = 0;
        for (int j = 0; j < kPer; j++) {
            int q = c0 + j;
            if (q >= first && q < e_last && (int32_t)R.a[q] == kMissMark) cnt++;
        }
        uint32_t ex = block_escan(cnt, tmp, &n_miss);
        if (cnt) {
            for (int j = 0; j < kPer; j++) {
                int q = c0 + j;
                if (q >= first && q < e_last && (int32_t)R.a[q] == kMissMark) R.list[ex++] = (uint16_t)q;
            }
        }
    }
    __syncthreads();
    {
        int32_t* sym = reinterpret_cast<int32_t*>(R.a);
        uint32_t* pr = R.p;
#pragma unroll 1
        for (uint32_t k = tid; k < n_miss; k += kTileThreads) {
            const int s = R.list[k];
            const int e = s + (int)R.p[s];
            for (int q = s; q < e; q++) {
                uint32_t b0 = R.b[q];
                sym[q] = T.byte_tok[b0];
                pr[q] = (q + 1 < e) ? T.bytepair[(b0 << 8) | R.b[q + 1]] : kNoPrio;
            }
            if (!(A.flags & 0x100u)) merge_span(T, sym, pr, s, e);
        }
    }
    __syncthreads();

    // -- compact the surviving ids (and slow placeholders) into the tile's stream
    uint32_t cnt = 0;  // low 16: items, high 16: slow placeholders
    for (int j = 0; j < kPer; j++) {
        int q = c0 + j;
        if (q < first || q >= e_last) continue;
        int32_t v = (int32_t)R.a[q];
        if (v >= 0) cnt += 1u;
        else if (v == kSlowMark) cnt += 0x10001u;
    }
    uint32_t tot;
    uint32_t ex = block_escan(cnt, tmp, &tot);
    const uint32_t n_items = tot & 0xFFFFu, n_slow = tot >> 16;
    if (tid == 0) {
        uint32_t s0 = n_slow ? atomicAdd(&W.status->slow_count, n_slow) : 0u;
        vars[V_SLOW0] = (int)s0;
        W.tile_items[tile] = n_items;
        W.tile_nslow[tile] = n_slow;
        W.tile_slow0[tile] = s0;
        if (n_slow && (uint64_t)s0 + n_slow > W.slow_cap) atomicOr(&W.status->flags, kOvfSlow);
    }
    // chunk prefixes for the per-document item index (list[] is free now)
    R.list[tid] = (uint16_t)(ex & 0xFFFFu);
    __syncthreads();
    {
        const uint32_t slow0 = (uint32_t)vars[V_SLOW0];
        int32_t* out = W.stream + tile * (uint64_t)kTileCap;
        uint32_t ii = ex & 0xFFFFu, si = ex >> 16;
        for (int j = 0; j < kPer; j++) {
            int q = c0 + j;
            if (q < first || q >= e_last) continue;
            int32_t v = (int32_t)R.a[q];
            if (v >= 0) {
                out[ii++] = v;
            } else if (v == kSlowMark) {
                out[ii++] = kSlowMark;
                uint64_t ei = (uint64_t)slow0 + si++;
                if (ei < W.slow_cap) {
                    SlowEntry se;
                    se.start = g_lo + (uint64_t)q;
                    se.scan_from = g_lo + (uint64_t)eval_hi;
                    se.len = R.p[q];
                    se.tile = (uint32_t)tile;
                    se.count = 0; se.pad = 0; se.pool_off = 0;
                    W.slow[ei] = se;
                }
            }
        }
    }
    // -- stream index of the first item of every document that starts in this tile
    {
        const bool last_tile = tile + 1 == gridDim.x;
        const uint64_t d_end = last_tile ? A.n_docs + 1 : (uint64_t)W.tile_doc0[tile + 1];
        for (uint64_t d = (uint64_t)d0 + tid; d < d_end; d += kTileThreads) {
            int pos = (int)(A.doc_off[d] - g_lo);
            uint32_t item;
            if (pos <= first) item = 0;
            else if (pos >= e_last) item = n_items;
            else {
                int ch = pos / kPer;
                item = R.list[ch];
                for (int q = ch * kPer; q < pos; q++) {
                    if (q < first) continue;
                    int32_t v = (int32_t)R.a[q];
                    if (v >= 0 || v == kSlowMark) item++;
                }
            }
            W.doc_item[d] = item;
        }
    }
}

// ---- the slow kernel ---------------------------------------------------------------------------------
struct MergeLds {
    int32_t sA[kSlowLds], sB[kSlowLds];
    uint32_t pA[kSlowLds], pB[kSlowLds];
};
union SlowLds {
    RegionLds R;
    MergeLds M;
};
constexpr int32_t kIdMask = 0x1FFFF;
constexpr int32_t kSelBit = 1 << 30;
constexpr int32_t kDirtyBit = 1 << 29;

// Exact round-based greedy merge of the n symbols in (sa, pa); (sb, pb) is the other buffer of the
// pair.  Every round applies, left to right, the non-overlapping occurrences of the currently
// smallest priority for as long as no merge creates a pair that out-ranks it (then the sequential
// order of bpe.go:49-59 would leave this priority, so the round stops there).  Returns the buffer
// that holds the result and the new n.
__device__ uint32_t merge_rounds(const Tables& T, int32_t*& sa, uint32_t*& pa, int32_t*& sb, uint32_t*& pb, uint32_t n,
                                 uint32_t* tmp, unsigned long long* tmp64, Seg* segtmp) {
    const uint32_t tid = threadIdx.x;
    for (;;) {
        if (n < 2) break;
        const uint32_t chunk = (n + kSlowThreads - 1) / kSlowThreads;
        const uint32_t i0 = tid * chunk < n ? tid * chunk : n;
        const uint32_t i1 = i0 + chunk < n ? i0 + chunk : n;
        // (a) smallest priority and where it first occurs
        unsigned long long best = ~0ull;
        for (uint32_t i = i0; i < i1; i++) {
            unsigned long long v = ((unsigned long long)pa[i] << 32) | i;
            best = v < best ? v : best;
        }
        best = block_min64(best, tmp64);
        const uint32_t r = (uint32_t)(best >> 32);
        if (r == kNoPrio) break;
        const uint32_t ipos = (uint32_t)best;
        const int32_t m = (int32_t)prio_merged_id(r);
        const bool same = (sa[ipos] & kIdMask) == (sa[ipos + 1] & kIdMask);
        __syncthreads();
        // (b) select occurrences; overlapping ones (x x x ...) alternate starting from the left
        {
            Seg sg; sg.cnt = 0; sg.open = 1;
            uint32_t first_break = n;
            for (uint32_t i = i0; i < i1; i++) {
                bool cand = pa[i] == r;
                int32_t v = sa[i] & kIdMask;
                if (!cand) {
                    sg.cnt = 0; sg.open = 0;
                    if (first_break == n) first_break = i;
                } else {
                    if (!(sg.cnt & 1u)) v |= kSelBit;
                    sg.cnt++;
                }
                sa[i] = v;
            }
            if (same) {  // chains can cross chunk borders only when both symbols are equal
                Seg carry = block_seg_escan(sg, segtmp);
                if (carry.cnt & 1u) {
                    uint32_t lim = first_break < i1 ? first_break : i1;
                    for (uint32_t i = i0; i < lim; i++) sa[i] ^= kSelBit;
                }
            }
        }
        __syncthreads();
        // (c) first selected merge after which the sequential order would leave priority r
        unsigned long long vmin = ~0ull;
        for (uint32_t i = i0; i < i1; i++) {
            if (!(sa[i] & kSelBit)) continue;
            bool viol = false;
            if (i >= 1) {
                int32_t lp = (i >= 2 && (sa[i - 2] & kSelBit)) ? m : (sa[i - 1] & kIdMask);
                viol = pair_prio(T, (uint32_t)lp, (uint32_t)m) < r;
            }
            if (!viol && i + 2 < n) viol = pair_prio(T, (uint32_t)m, (uint32_t)(sa[i + 2] & kIdMask)) < r;
            if (viol) { vmin = i; break; }
        }
        vmin = block_min64(vmin, tmp64);
        // (d) apply the selected merges up to vmin, compacting into the other buffer
        uint32_t live = 0;
        for (uint32_t i = i0; i < i1; i++) {
            bool dead = i >= 1 && (sa[i - 1] & kSelBit) && (unsigned long long)(i - 1) <= vmin;
            live += dead ? 0u : 1u;
        }
        uint32_t n2;
        uint32_t j = block_escan(live, tmp, &n2);
        for (uint32_t i = i0; i < i1; i++) {
            bool dead = i >= 1 && (sa[i - 1] & kSelBit) && (unsigned long long)(i - 1) <= vmin;
            if (dead) continue;
            bool applied = (sa[i] & kSelBit) && (unsigned long long)i <= vmin;
            sb[j] = applied ? (m | kDirtyBit) : (sa[i] & kIdMask);
            pb[j] = pa[i];
            j++;
        }
        __syncthreads();
        // (e) re-probe the pairs that touch a merged symbol
        {
            const uint32_t chunk2 = (n2 + kSlowThreads - 1) / kSlowThreads;
            const uint32_t j0 = tid * chunk2 < n2 ? tid * chunk2 : n2;
            const uint32_t j1 = j0 + chunk2 < n2 ? j0 + chunk2 : n2;
            for (uint32_t q = j0; q < j1; q++) {
                int32_t x = sb[q];
                int32_t y = q + 1 < n2 ? sb[q + 1] : 0;
                if (!((x | y) & kDirtyBit)) continue;
                pb[q] = q + 1 < n2 ? pair_prio(T, (uint32_t)(x & kIdMask), (uint32_t)(y & kIdMask)) : kNoPrio;
            }
        }
        __syncthreads();
        { int32_t* t = sa; sa = sb; sb = t; }
        { uint32_t* t = pa; pa = pb; pb = t; }
        n = n2;
    }
    return n;
}

__global__ __launch_bounds__(kSlowThreads) void slow_kernel(Tables T, Workspace W, EncodeArgs A) {
    __shared__ __align__(16) SlowLds S;
    __shared__ uint32_t tmp[16];
    __shared__ unsigned long long tmp64[8];
    __shared__ Seg segtmp[4];
    __shared__ long long lvars[8];
    __shared__ int ivars[4];
    __shared__ uint8_t xb[kMaxTokenBytes];

    if (W.status->flags & kBadOffsets) return;
    const int tid = threadIdx.x;
    uint64_t n_slow = W.status->slow_count;
    if (n_slow > W.slow_cap) n_slow = W.slow_cap;
#pragma unroll 1
    for (uint64_t ei = blockIdx.x; ei < n_slow; ei += gridDim.x) {
        const uint64_t start = W.slow[ei].start;
        uint64_t len = W.slow[ei].len;
        __syncthreads();
        // ---- 1. where does it end? (state_machine.go:53-79 continues until the next matcher start)
        if (len == 0) {
            RegionLds& R = S.R;
            if (tid == 0) {  // end of its document: first offset > start
                uint64_t lo = 1, hi = A.n_docs;
                while (lo < hi) {
                    uint64_t mid = (lo + hi) / 2;
                    if (A.doc_off[mid] > start) hi = mid; else lo = mid + 1;
                }
                lvars[0] = (long long)A.doc_off[lo];
            }
            __syncthreads();
            const uint64_t doc_end = (uint64_t)lvars[0];
            uint64_t w0 = W.slow[ei].scan_from;
            uint64_t end = doc_end;
            for (;;) {
                const uint64_t g_lo = w0 - kBack;
                const uint64_t g_hi = g_lo + kRegion < doc_end ? g_lo + kRegion : doc_end;
                const int n = (int)(g_hi - g_lo);
                const bool hard = g_hi == doc_end;
                const int eval_hi = hard ? n : n - kEvalSlack;
                for (int i = tid; i < kBitWords; i += kSlowThreads) { R.dsb[i] = 0; }
                if (tid == 0) ivars[0] = n;
                region_load(R, A.text, g_lo, n);
                __syncthreads();
                LdsSrc src{&R, n, hard};
                region_classify(R, src, T);
                const int c0 = tid * kPer;
                int found = n;
                for (int j = 0; j < kPer; j++) {
                    int q = c0 + j;
                    if (q < kBack || q >= eval_hi) continue;
                    uint32_t w = R.a[q];
                    if (!(w & I_RUNE) || (w & I_SPBODY)) continue;
                    // a pre-token this long holds no digit, so no digit-run offset is needed
                    if (starts_pretoken(src, q, 0)) { found = q; break; }
                }
                if (found < n) atomicMin(&ivars[0], found);
                __syncthreads();
                const int f = ivars[0];
                __syncthreads();
                if (f < n) { end = g_lo + (uint64_t)f; break; }
                if (hard) { end = doc_end; break; }
                w0 = g_lo + (uint64_t)eval_hi;
            }
            len = end - start;
        }
        // ---- 2. symbols: one per byte, EF BF BD for a byte that is not valid UTF-8 (state_machine.go:36)
        GlobalSrc gs{A.text, (int64_t)start, (int64_t)(start + len)};
        const uint64_t bchunk = (len + kSlowThreads - 1) / kSlowThreads;
        const uint64_t b0 = tid * bchunk < len ? tid * bchunk : len;
        const uint64_t b1 = b0 + bchunk < len ? b0 + bchunk : len;
        uint32_t nsym = 0;
        for (uint64_t q = b0; q < b1; q++)
            nsym += (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID) ? 3u : 1u;
        uint32_t L;
        uint32_t so = block_escan(nsym, tmp, &L);
        int32_t *sa, *sb;
        uint32_t *pa, *pb;
        bool ok = true;
        if (L <= (uint32_t)kSlowLds) {
            sa = S.M.sA; sb = S.M.sB; pa = S.M.pA; pb = S.M.pB;
        } else {
            if (tid == 0) {
                unsigned long long need = (unsigned long long)L * 16ull;
                unsigned long long off = atomicAdd(&W.status->scratch_used, need);
                lvars[2] = (off + need <= W.scratch_cap) ? (long long)off : -1ll;
                if (off + need > W.scratch_cap) atomicOr(&W.status->flags, kOvfScratch);
            }
            __syncthreads();
            long long off = lvars[2];
            ok = off >= 0;
            uint8_t* base = W.scratch + (ok ? off : 0);
            sa = reinterpret_cast<int32_t*>(base);
            sb = sa + L;
            pa = reinterpret_cast<uint32_t*>(sb + L);
            pb = pa + L;
        }
        uint32_t n_out = 0;
        if (ok) {
            // symbol bytes first, then pair priorities, then ids
            uint32_t o = so;
            for (uint64_t q = b0; q < b1; q++) {
                if (byte_info(gs, (int64_t)(start + q), T.cls_stage1, T.cls_blocks) & I_INVALID) {
                    sa[o++] = 0xEF; sa[o++] = 0xBF; sa[o++] = 0xBD;
                } else {
                    sa[o++] = A.text[start + q];
                }
            }
            __syncthreads();
            int32_t whole = -1;
            if (L <= (uint32_t)kMaxTokenBytes) {  // bpe.go:27-33
                if ((uint32_t)tid < L) xb[tid] = (uint8_t)sa[tid];
                __syncthreads();
                if (tid == 0) lvars[3] = vocab_lookup(T, [&](uint32_t i) { return (uint32_t)xb[i]; }, L);
                __syncthreads();
                whole = (int32_t)lvars[3];
            }
            if (whole >= 0) {
                if (tid == 0) sa[0] = whole;
                n_out = 1;
                __syncthreads();
            } else {
                const uint32_t chunk = (L + kSlowThreads - 1) / kSlowThreads;
                const uint32_t i0 = (uint32_t)tid * chunk < L ? tid * chunk : L;
                const uint32_t i1 = i0 + chunk < L ? i0 + chunk : L;
                for (uint32_t i = i0; i < i1; i++)
                    pa[i] = i + 1 < L ? T.bytepair[((uint32_t)sa[i] << 8) | (uint32_t)sa[i + 1]] : kNoPrio;
                __syncthreads();
                for (uint32_t i = i0; i < i1; i++) sa[i] = T.byte_tok[sa[i]];
                __syncthreads();
                n_out = merge_rounds(T, sa, pa, sb, pb, L, tmp, tmp64, segtmp);
            }
            // ---- 3. tokens to the pool
            if (tid == 0) {
                unsigned long long off = atomicAdd(&W.status->pool_used, (unsigned long long)n_out);
                bool fits = off + n_out <= W.pool_cap;
                if (!fits) atomicOr(&W.status->flags, kOvfPool);
                lvars[4] = fits ? (long long)off : -1ll;
                W.slow[ei].count = n_out;
                W.slow[ei].pool_off = off;
                atomicAdd(&W.tile_extra[W.slow[ei].tile], n_out);
            }
            __syncthreads();
            if (lvars[4] >= 0) {
                int32_t* dst = W.pool + lvars[4];
                for (uint32_t i = tid; i < n_out; i += kSlowThreads) dst[i] = sa[i] & kIdMask;
            }
        }
        __syncthreads();
    }
}

// ---- scan of per-tile totals ----------------------------------------------------------------------
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void scan_kernel(Workspace W, EncodeArgs A, uint64_t n_tiles) {
    __shared__ unsigned long long wsum[kScanThreads / 64];
    __shared__ unsigned long long carry_s;
    if (W.status->flags & kBadOffsets) return;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += kScanThreads) {
        uint64_t t = base + tid;
        unsigned long long v = 0;
        if (t < n_tiles) v = (unsigned long long)W.tile_items[t] - W.tile_nslow[t] + W.tile_extra[t];
        unsigned long long inc = v;
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
        if (t < n_tiles) W.tile_base[t] = pre + inc - v;
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) {
        unsigned long long total = carry_s;
        W.tile_base[n_tiles] = total;
        uint32_t be = (A.flags & 1u) + ((A.flags >> 1) & 1u);
        total += (unsigned long long)A.n_docs * be;
        W.status->total_tokens = total;
        if (total > A.out_cap) atomicOr(&W.status->flags, kOvfOut);
    }
}

// ---- final scatter -------------------------------------------------------------------------------------
constexpr int kEmitPer = (kTileCap + 1 + kTileThreads - 1) / kTileThreads;  // items per thread
constexpr int kEmitQueue = 256;
__global__ __launch_bounds__(kTileThreads) void emit_kernel(Workspace W, EncodeArgs A) {
    __shared__ uint32_t docinc[kEmitPer * kTileThreads + 1];
    __shared__ uint32_t tokpre[kEmitPer * kTileThreads + 1];
    __shared__ uint32_t tmp[16];
    __shared__ unsigned long long q_out[kEmitQueue];
    __shared__ uint32_t q_slow[kEmitQueue];
    __shared__ uint32_t q_n;

    const uint32_t fl = W.status->flags;
    if (fl & (kBadOffsets | kOvfSlow | kOvfPool | kOvfScratch)) return;
    if (W.status->total_tokens > A.out_cap) return;
    const int tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const bool last_tile = tile + 1 == gridDim.x;
    const uint32_t n_items = W.tile_items[tile];
    const uint32_t slow0 = W.tile_slow0[tile];
    const uint64_t d_lo = W.tile_doc0[tile];
    const uint64_t d_hi = last_tile ? A.n_docs + 1 : (uint64_t)W.tile_doc0[tile + 1];
    const uint32_t bos = A.flags & 1u, eos = (A.flags >> 1) & 1u, be = bos + eos;
    const uint64_t base = W.tile_base[tile];
    const int32_t* in = W.stream + tile * (uint64_t)kTileCap;

    for (int i = tid; i < kEmitPer * kTileThreads + 1; i += kTileThreads) docinc[i] = 0;
    if (tid == 0) q_n = 0;
    __syncthreads();
    for (uint64_t d = d_lo + tid; d < d_hi; d += kTileThreads) atomicAdd(&docinc[W.doc_item[d]], 1u);
    __syncthreads();

    const int c0 = tid * kEmitPer;
    int32_t item[kEmitPer];
    uint32_t ntok = 0, nsl = 0, ndoc = 0;
#pragma unroll
    for (int j = 0; j < kEmitPer; j++) {
        uint32_t i = c0 + j;
        item[j] = i < n_items ? in[i] : kDead;
        if (item[j] == kSlowMark) nsl++;
        if (i <= n_items) ndoc += docinc[i];
    }
    uint32_t t_sl, t_doc, t_tok;
    uint32_t sl = block_escan(nsl, tmp, &t_sl);
    uint32_t cnts[kEmitPer];
#pragma unroll
    for (int j = 0; j < kEmitPer; j++) {
        cnts[j] = item[j] >= 0 ? 1u : 0u;
        if (item[j] == kSlowMark) cnts[j] = W.slow[slow0 + sl++].count;
        ntok += cnts[j];
    }
    sl -= nsl;
    uint32_t tp = block_escan(ntok, tmp, &t_tok);
    uint32_t dp = block_escan(ndoc, tmp, &t_doc);
#pragma unroll
    for (int j = 0; j < kEmitPer; j++) {
        uint32_t i = c0 + j;
        if (i > n_items) break;
        dp += docinc[i];  // documents that start at or before item i
        tokpre[i] = tp;
        if (i == n_items) break;
        const uint64_t D = d_lo + dp - 1;
        const uint64_t o = base + tp + D * be + bos;
        if (item[j] >= 0) {
            A.out_tok[o] = item[j];
        } else if (item[j] == kSlowMark) {
            const SlowEntry& se = W.slow[slow0 + sl];
            if (cnts[j] <= 64) {
                for (uint32_t k = 0; k < cnts[j]; k++) A.out_tok[o + k] = W.pool[se.pool_off + k];
            } else {
                uint32_t qi = atomicAdd(&q_n, 1u);
                if (qi < kEmitQueue) { q_out[qi] = o; q_slow[qi] = slow0 + sl; }
            }
            sl++;
        }
        tp += cnts[j];
    }
    __syncthreads();
    // per-document offsets, BOS / EOS (tokenizer.go:225-229,258-262)
    for (uint64_t d = d_lo + tid; d < d_hi; d += kTileThreads) {
        const uint64_t off = base + tokpre[W.doc_item[d]] + d * be;
        A.out_off[d] = off;
        if (bos && d < A.n_docs) A.out_tok[off] = kBosId;
        if (eos && d > 0) A.out_tok[off - 1] = kEosId;
    }
    // long runs of tokens from the slow pool: copied by the whole workgroup
    const uint32_t qn = q_n < kEmitQueue ? q_n : kEmitQueue;
    for (uint32_t qi = 0; qi < qn; qi++) {
        const SlowEntry& se = W.slow[q_slow[qi]];
        const uint64_t o = q_out[qi];
        for (uint32_t k = tid; k < se.count; k += kTileThreads) A.out_tok[o + k] = W.pool[se.pool_off + k];
    }
}

// ---- decode (tokenizer.go:339-354): ids -> bytes, out-of-range ids skipped ------------------------------
__global__ void decode_len_kernel(const uint32_t* tok_off, int n_tokens, const int32_t* ids, const uint64_t* id_off,
                                  uint64_t n_seqs, uint64_t* out_off) {
    // one workgroup-stride loop per sequence: byte length of every sequence
    __shared__ unsigned long long acc;
    for (uint64_t s = blockIdx.x; s < n_seqs; s += gridDim.x) {
        if (threadIdx.x == 0) acc = 0;
        __syncthreads();
        unsigned long long mine = 0;
        for (uint64_t i = id_off[s] + threadIdx.x; i < id_off[s + 1]; i += blockDim.x) {
            int32_t id = ids[i];
            if (id >= 0 && id < n_tokens) mine += tok_off[id + 1] - tok_off[id];
        }
        atomicAdd(&acc, mine);
        __syncthreads();
        if (threadIdx.x == 0) out_off[s + 1] = acc;
        __syncthreads();
    }
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
constexpr int kDecChunk = 256;  // ids per workgroup pass
__global__ __launch_bounds__(256) void decode_copy_kernel(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens,
                                                          const int32_t* ids, const uint64_t* id_off, uint64_t n_seqs,
                                                          uint8_t* out, uint64_t out_cap, const uint64_t* out_off,
                                                          const unsigned long long* total) {
    __shared__ uint32_t tmp[16];
    __shared__ unsigned long long run;
    if (*total > out_cap) return;
    for (uint64_t s = blockIdx.x; s < n_seqs; s += gridDim.x) {
        if (threadIdx.x == 0) run = out_off[s];
        __syncthreads();
        for (uint64_t base = id_off[s]; base < id_off[s + 1]; base += kDecChunk) {
            uint64_t i = base + threadIdx.x;
            uint32_t len = 0, src = 0;
            if (i < id_off[s + 1]) {
                int32_t id = ids[i];
                if (id >= 0 && id < n_tokens) { src = tok_off[id]; len = tok_off[id + 1] - src; }
            }
            uint32_t tot;
            uint32_t ex = block_escan(len, tmp, &tot);
            uint8_t* dst = out + run + ex;
            for (uint32_t k = 0; k < len; k++) dst[k] = tok_bytes[src + k];
            __syncthreads();
            if (threadIdx.x == 0) run += tot;
            __syncthreads();
        }
    }
}

}  // namespace

#define L3_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e_ = hipGetLastError();        \
        if (e_ != hipSuccess) return e_;          \
    } while (0)
#define L3_EV(i)                                                          \
    do {                                                                  \
        if (ev) {                                                         \
            hipError_t e_ = hipEventRecord(ev[i], stream);                \
            if (e_ != hipSuccess) return e_;                              \
        }                                                                 \
    } while (0)

hipError_t launch_encode(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream, hipEvent_t* ev,
                         bool emit_only) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    if (n_tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (!emit_only) {
        L3_EV(kEvStart);
        const uint32_t dblocks = (uint32_t)((A.n_docs + 1 + 255) / 256);
        hipLaunchKernelGGL(validate_kernel, dim3(dblocks), dim3(256), 0, stream, W, A);
        L3_LAUNCH_CHECK();
        hipLaunchKernelGGL(docs_kernel, dim3(dblocks), dim3(256), 0, stream, W, A, n_tiles);
        L3_LAUNCH_CHECK();
        L3_EV(kEvDocs);
        hipLaunchKernelGGL(tile_kernel, dim3((uint32_t)n_tiles), dim3(kTileThreads), 0, stream, T, W, A);
        L3_LAUNCH_CHECK();
        L3_EV(kEvTile);
        uint64_t sgrid = W.slow_cap < 2048 ? W.slow_cap : 2048;
        if (sgrid == 0) sgrid = 1;
        hipLaunchKernelGGL(slow_kernel, dim3((uint32_t)sgrid), dim3(kSlowThreads), 0, stream, T, W, A);
        L3_LAUNCH_CHECK();
        L3_EV(kEvSlow);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, W, A, n_tiles);
        L3_LAUNCH_CHECK();
        L3_EV(kEvScan);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((uint32_t)n_tiles), dim3(kTileThreads), 0, stream, W, A);
    L3_LAUNCH_CHECK();
    L3_EV(kEvEmit);
    return hipSuccess;
}

hipError_t launch_decode(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens, const int32_t* ids,
                         const uint64_t* id_off, uint64_t n_seqs, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                         unsigned long long* total, hipStream_t stream) {
    uint32_t grid = (uint32_t)(n_seqs < 4096 ? (n_seqs ? n_seqs : 1) : 4096);
    hipLaunchKernelGGL(decode_len_kernel, dim3(grid), dim3(256), 0, stream, tok_off, n_tokens, ids, id_off, n_seqs, out_off);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, out_off, n_seqs, total);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_copy_kernel, dim3(grid), dim3(256), 0, stream, tok_off, tok_bytes, n_tokens, ids, id_off,
                       n_seqs, out, out_cap, out_off, total);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3
