// l3_bpe.h -- table probes and the in-place merge loop for one pre-token (host- and device-compilable).
//
// Restates bpe.Processor.PerformBPE (reference llama3/internal/bpe/bpe.go:18-214) on integer ids:
//   * whole pre-token vocabulary hit first (bpe.go:27-33)            -> vocab_lookup()
//   * one id per byte (bpe.go:74-84; ids 0..255 are the 256 byte-level runes)
//   * repeatedly apply the adjacent pair with the smallest (rank, original position)
//     (bpe.go:49-59,114-130; priority_queue.go:22-25)                -> merge_span()
// rank order is carried by the 24-bit prio (see l3_common.h), which also yields the merged id, so a
// merge costs two independent table probes and no string work.
#pragma once
#include "l3_common.h"

namespace l3 {

// (idA,idB) -> prio or kNoPrio, given the key's two candidate buckets
L3_HD uint32_t pair_pick(uint64_t key, const PairBucket& u, const PairBucket& v) {
    uint32_t r = kNoPrio;
    if ((u.s[0] >> 24) == key) r = (uint32_t)(u.s[0] & 0xFFFFFFu);
    if ((u.s[1] >> 24) == key) r = (uint32_t)(u.s[1] & 0xFFFFFFu);
    if ((v.s[0] >> 24) == key) r = (uint32_t)(v.s[0] & 0xFFFFFFu);
    if ((v.s[1] >> 24) == key) r = (uint32_t)(v.s[1] & 0xFFFFFFu);
    return r;
}
L3_HD uint32_t pair_prio(const Tables& t, uint32_t a, uint32_t b) {
    const uint64_t key = pair_key(a, b);
    uint32_t b1, b2;
    pair_hash2(a, b, t.pair_log2, &b1, &b2);
    const PairBucket u = t.pair_buckets[b1];
    const PairBucket v = t.pair_buckets[b2];
    return pair_pick(key, u, v);
}
// two independent probes: four loads in flight together
L3_HD void pair_prio2(const Tables& t, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t* r0, uint32_t* r1) {
    const uint64_t k0 = pair_key(a0, b0), k1 = pair_key(a1, b1);
    uint32_t p0, q0, p1, q1;
    pair_hash2(a0, b0, t.pair_log2, &p0, &q0);
    pair_hash2(a1, b1, t.pair_log2, &p1, &q1);
    const PairBucket u0 = t.pair_buckets[p0];
    const PairBucket v0 = t.pair_buckets[q0];
    const PairBucket u1 = t.pair_buckets[p1];
    const PairBucket v1 = t.pair_buckets[q1];
    *r0 = pair_pick(k0, u0, v0);
    *r1 = pair_pick(k1, u1, v1);
}

// Whole-string probe for a string of <= 12 bytes given as three zero-padded words.  First-choice bucket first; the
// second load is issued only by the lanes that did not find their string there (tokens that lost their first
// choice at build time, and strings that are no token).
L3_HD int32_t vocab_lookup_short(const Tables& t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len) {
    uint32_t b1, b2;
    vocab_hash_short2(w0, w1, w2, len, t.vocab_log2, &b1, &b2);
    const uint32_t tag_lo = len << 17;
    const VocabSlot s0 = t.vocab_slots[b1];
    if ((s0.w[3] & ~0x1FFFFu) == tag_lo && s0.w[0] == w0 && s0.w[1] == w1 && s0.w[2] == w2) return (int32_t)(s0.w[3] & 0x1FFFFu);
    const VocabSlot s1 = t.vocab_slots[b2];
    if ((s1.w[3] & ~0x1FFFFu) == tag_lo && s1.w[0] == w0 && s1.w[1] == w1 && s1.w[2] == w2) return (int32_t)(s1.w[3] & 0x1FFFFu);
    return -1;
}

// Whole-string lookup by words.  `getw(i)` returns little-endian word i of the (already expanded) pre-token,
// 0 <= i < (len + 3) / 4, the last one zero padded.  Returns the token id or -1.
template <class GetW>
L3_HD int32_t vocab_lookup_words(const Tables& t, GetW getw, uint32_t len) {
    if (len == 0 || len > (uint32_t)kMaxTokenBytes) return -1;
    const uint32_t nw = (len + 3u) >> 2;
    if (len <= (uint32_t)kVocabInline) return vocab_lookup_short(t, getw(0), nw > 1 ? getw(1) : 0u, nw > 2 ? getw(2) : 0u, len);
    StrHash h = strhash_init();
    for (uint32_t i = 0; i < nw; i++) strhash_add(h, getw(i));
    uint32_t b[2];
    strhash_buckets(h, len, t.vocab_log2, &b[0], &b[1]);
    // both candidate slots at once, then the blob words of whichever carries the right length and second hash:
    // two memory round trips, not four
    const VocabSlot v0 = t.vocab_slots[b[0]];
    const VocabSlot v1 = t.vocab_slots[b[1]];
    const bool c0 = v0.w[3] != kVocabEmpty && (v0.w[3] >> 17) == len && v0.w[0] == h.h2;
    const bool c1 = v1.w[3] != kVocabEmpty && (v1.w[3] >> 17) == len && v1.w[0] == h.h2;
    if (c0) {
        const uint32_t* blob = t.vocab_blob + v0.w[1];
        bool eq = true;
        for (uint32_t i = 0; i < nw; i++) eq = eq & (blob[i] == getw(i));
        if (eq) return (int32_t)(v0.w[3] & 0x1FFFFu);
    }
    if (c1) {
        const uint32_t* blob = t.vocab_blob + v1.w[1];
        bool eq = true;
        for (uint32_t i = 0; i < nw; i++) eq = eq & (blob[i] == getw(i));
        if (eq) return (int32_t)(v1.w[3] & 0x1FFFFu);
    }
    return -1;
}
// Same with a byte source: `get(i)` returns byte i, 0 <= i < len.
template <class Get>
L3_HD int32_t vocab_lookup(const Tables& t, Get get, uint32_t len) {
    return vocab_lookup_words(t, [&](uint32_t i) {
        uint32_t w = 0;
        for (uint32_t k = 0; k < 4u && 4u * i + k < len; k++) w |= get(4u * i + k) << (8u * k);
        return w;
    }, len);
}

// In-place greedy merge of the symbols at positions [s, e) of `sym` (id >= 0 live, -1 dead) with the
// pair priorities cached in `pr` (pr[p] = prio of (sym[p], next live symbol), kNoPrio if none).
// On return the live entries of sym[s..e) are the pre-token's token ids in order.
template <class Sym, class Pr>
L3_HD void merge_span(const Tables& t, Sym& sym, Pr& pr, int s, int e) {
    for (;;) {
        uint32_t best = kNoPrio;
        int bp = -1;
        for (int p = s; p < e; p++) {
            uint32_t v = pr[p];
            if (v < best) { best = v; bp = p; }  // strict <: leftmost among equals (bpe.go:126)
        }
        if (bp < 0) return;
        int q = bp + 1;
        while (sym[q] < 0) q++;            // right element of the pair
        int r = q + 1;
        while (r < e && sym[r] < 0) r++;   // next live symbol after the pair (r == e: none)
        int l = bp - 1;
        while (l >= s && sym[l] < 0) l--;  // previous live symbol (l < s: none)
        const uint32_t m = prio_merged_id(t, best);
        sym[bp] = (int32_t)m;
        sym[q] = -1;
        pr[q] = kNoPrio;
        if (r < e && l >= s) {
            uint32_t pa, pb;
            pair_prio2(t, m, (uint32_t)sym[r], (uint32_t)sym[l], m, &pa, &pb);
            pr[bp] = pa;
            pr[l] = pb;
        } else {
            pr[bp] = (r < e) ? pair_prio(t, m, (uint32_t)sym[r]) : kNoPrio;
            if (l >= s) pr[l] = pair_prio(t, (uint32_t)sym[l], m);
        }
    }
}

}  // namespace l3
