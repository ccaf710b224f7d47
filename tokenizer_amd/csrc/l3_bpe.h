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

// (idA,idB) -> prio or kNoPrio, continuing from an already loaded first bucket
L3_HD uint32_t pair_resolve(const Tables& t, uint64_t key, uint32_t b, PairBucket v) {
    const uint32_t mask = (1u << t.pair_log2) - 1u;
    for (;;) {
        if ((v.s[0] >> 24) == key) return (uint32_t)(v.s[0] & 0xFFFFFFu);
        if ((v.s[1] >> 24) == key) return (uint32_t)(v.s[1] & 0xFFFFFFu);
        if (v.s[1] == kPairEmpty) return kNoPrio;  // bucket never overflowed
        b = (b + 1u) & mask;
        v = t.pair_buckets[b];
    }
}
L3_HD uint32_t pair_prio(const Tables& t, uint32_t a, uint32_t b) {
    const uint64_t key = pair_key(a, b);
    const uint32_t h = pair_hash(key, t.pair_log2);
    return pair_resolve(t, key, h, t.pair_buckets[h]);
}
// two independent probes with both first-bucket loads in flight together
L3_HD void pair_prio2(const Tables& t, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t* r0, uint32_t* r1) {
    const uint64_t k0 = pair_key(a0, b0), k1 = pair_key(a1, b1);
    const uint32_t h0 = pair_hash(k0, t.pair_log2), h1 = pair_hash(k1, t.pair_log2);
    const PairBucket v0 = t.pair_buckets[h0];
    const PairBucket v1 = t.pair_buckets[h1];
    *r0 = pair_resolve(t, k0, h0, v0);
    *r1 = pair_resolve(t, k1, h1, v1);
}

// Probe of one candidate slot chain for a string of <= 12 bytes given as three zero-padded words.
L3_HD int32_t vocab_lookup_short(const Tables& t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len) {
    const uint32_t mask = (1u << t.vocab_log2) - 1u;
    uint32_t s = vocab_hash_short(w0, w1, w2, len, t.vocab_log2);
    for (;;) {
        VocabSlot v = t.vocab_slots[s];
        if (v.w[3] == kVocabEmpty) return -1;
        if ((v.w[3] >> 17) == len && v.w[0] == w0 && v.w[1] == w1 && v.w[2] == w2) return (int32_t)(v.w[3] & 0x1FFFFu);
        s = (s + 1u) & mask;
    }
}

// Whole-string lookup.  `get(i)` returns byte i of the (already expanded) pre-token, 0 <= i < len.
// Returns the token id or -1.
template <class Get>
L3_HD int32_t vocab_lookup(const Tables& t, Get get, uint32_t len) {
    if (len == 0 || len > (uint32_t)kMaxTokenBytes) return -1;
    if (len <= (uint32_t)kVocabInline) {
        uint32_t w[3] = {0, 0, 0};
        for (uint32_t i = 0; i < len; i++) w[i >> 2] |= get(i) << ((i & 3) * 8);
        return vocab_lookup_short(t, w[0], w[1], w[2], len);
    }
    StrHash h = strhash_init();
    for (uint32_t i = 0; i < len; i++) strhash_add(h, get(i));
    const uint32_t mask = (1u << t.vocab_log2) - 1u;
    uint32_t s = strhash_slot(h, len, t.vocab_log2);
    for (;;) {
        VocabSlot v = t.vocab_slots[s];
        if (v.w[3] == kVocabEmpty) return -1;
        if ((v.w[3] >> 17) == len && v.w[0] == h.h2) {
            const uint8_t* blob = t.vocab_blob + v.w[1];
            bool eq = true;
            for (uint32_t i = 0; i < len && eq; i++) eq = blob[i] == get(i);
            if (eq) return (int32_t)(v.w[3] & 0x1FFFFu);
        }
        s = (s + 1u) & mask;
    }
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
        const uint32_t m = prio_merged_id(best);
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
