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

// (idA,idB) -> prio or kNoPrio
L3_HD uint32_t pair_prio(const Tables& t, uint32_t a, uint32_t b) {
    const uint64_t key = pair_key(a, b);
    const uint32_t mask = (1u << t.pair_log2) - 1u;
    uint32_t s = pair_hash(key, t.pair_log2);
    for (;;) {
        uint64_t v = t.pair_slots[s];
        if (v == kPairEmpty) return kNoPrio;
        if ((v >> 24) == key) return (uint32_t)(v & 0xFFFFFFu);
        s = (s + 1u) & mask;
    }
}

// Whole-string lookup.  `get(i)` returns byte i of the (already expanded) pre-token, 0 <= i < len.
// Returns the token id or -1.
template <class Get>
L3_HD int32_t vocab_lookup(const Tables& t, Get get, uint32_t len) {
    if (len == 0 || len > (uint32_t)kMaxTokenBytes) return -1;
    StrHash h = strhash_init();
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t c = get(i);
        strhash_add(h, c);
        if (i < 4) w0 |= c << (i * 8);
        else if (i < 8) w1 |= c << ((i - 4) * 8);
        else if (i < 12) w2 |= c << ((i - 8) * 8);
    }
    const uint32_t mask = (1u << t.vocab_log2) - 1u;
    uint32_t s = strhash_slot(h, len, t.vocab_log2);
    for (;;) {
        VocabSlot v = t.vocab_slots[s];
        if (v.w[3] == kVocabEmpty) return -1;
        if ((v.w[3] >> 17) == len) {
            if (len <= (uint32_t)kVocabInline) {
                if (v.w[0] == w0 && v.w[1] == w1 && v.w[2] == w2) return (int32_t)(v.w[3] & 0x1FFFFu);
            } else if (v.w[0] == h.h2) {
                const uint8_t* blob = t.vocab_blob + v.w[1];
                bool eq = true;
                for (uint32_t i = 0; i < len && eq; i++) eq = blob[i] == get(i);
                if (eq) return (int32_t)(v.w[3] & 0x1FFFFu);
            }
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
        pr[bp] = (r < e) ? pair_prio(t, m, (uint32_t)sym[r]) : kNoPrio;
        if (l >= s) pr[l] = pair_prio(t, (uint32_t)sym[l], m);
    }
}

}  // namespace l3
