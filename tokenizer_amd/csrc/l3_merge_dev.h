// l3_merge_dev.h -- the two merge engines shared by merge_kernel (l3_merge.hip) and slow_kernel (l3_slow.hip): the greedy
// merge of one pre-token by one lane in an LDS slot (lane_merge) and the exact round-based merge by a wave or a workgroup
// (merge_rounds).  Device code only.
#pragma once
#include "l3_bpe.h"
#include "l3_dev.h"

namespace l3 {

namespace {

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

}  // namespace

}  // namespace l3
