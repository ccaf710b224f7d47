// l3_common.h -- constants, device-table layout and hash functions shared by the host table
// builder (l3_tables.cpp) and the HIP kernels (l3_kernels.hip).  Host- and device-compilable.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define L3_HD __host__ __device__ __forceinline__
#else
#define L3_HD inline
#endif

namespace l3 {

// ---- vocabulary facts (reference: llama3/constants.go:6-10, internal/tokens/special.go:19-40)
constexpr int kBaseVocab = 128000;
constexpr int kNumSpecial = 256;
constexpr int kMaxSpecialLen = 30;  // "<|reserved_special_token_247|>"
constexpr int kMaxTokenBytes = 128; // longest vocabulary entry (raw bytes)

// ---- merge priority ---------------------------------------------------------------------------
// The reference orders candidate merges by rank = 1-based index of the pair in the merge file
// (internal/vocabulary/decoder.go:67, bpe.go:126).  In the shipped file all pairs producing the same
// token are consecutive and product ids never decrease, so
//     prio = (merged_id << 7) | index_of_pair_within_its_product      (24 bits)
// is order-isomorphic to rank AND carries the merged id: one probe per new pair, nothing else to look up.  The table
// builder checks both facts.  Merge data for which they do not hold (any pair list is valid for the reference: rank is just
// the index, decoder.go:41-71) is served in RANK MODE: prio = the pair's index among the kept rules (< 2^24), and the merged
// id comes from Tables::rank_merged[prio] -- one more (cached) load per applied merge.
constexpr uint32_t kNoPrio = 0xFFFFFFFFu;

// ---- pair table: (idA,idB) -> prio, two-choice (cuckoo) hashing over 16-byte buckets of two slots ---------
// slot = key34 << 24 | prio24 ; empty = ~0.  key34 = idA << 17 | idB.  Every key lives in one of its two
// candidate buckets, so a probe is two independent 16-byte loads and never a dependent chain: what a
// wave pays is one memory latency, whatever its 64 lanes look up.
constexpr uint64_t kPairEmpty = ~0ull;
struct alignas(16) PairBucket { uint64_t s[2]; };
L3_HD uint64_t pair_key(uint32_t a, uint32_t b) { return ((uint64_t)a << 17) | (uint64_t)b; }
// low 32 bits of the product of the low 24 bits of a and b: one full-rate v_mul_u32_u24 on the GPU (a 32 x 32
// multiply costs four of them)
L3_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu));
#endif
}
// two bucket indices of the pair (idA, idB), ids < 2^17.  Six ALU operations each.
L3_HD void pair_hash2(uint32_t a, uint32_t b, uint32_t log2_buckets, uint32_t* b1, uint32_t* b2) {
    uint32_t x = mul24(a, 0x9E3779u) + mul24(b, 0x85EBCBu);
    uint32_t y = mul24(a, 0xC2B2AFu) + mul24(b, 0x27D4EBu) + 0x165667B1u;
    x ^= x >> 11;
    y ^= y >> 13;
    *b1 = mul24(x, 0xD6E8FFu) >> (32 - log2_buckets);
    *b2 = mul24(y, 0x7FEB35u) >> (32 - log2_buckets);
}

// ---- vocabulary table: raw token bytes -> id, two-choice (cuckoo) hashing, one 16-byte slot per bucket ----
// slot: w[3] = id | len << 17 ; len <= 12: w[0..2] hold the bytes (little endian, zero padded);
// len > 12: w[0] = second hash, w[1] = WORD offset of the bytes in the blob (every entry starts on a
// 4-byte boundary, its last word zero padded), w[2] = 0.  empty: w[3] = 0xFFFFFFFF.
struct alignas(16) VocabSlot { uint32_t w[4]; };
constexpr uint32_t kVocabEmpty = 0xFFFFFFFFu;
constexpr int kVocabInline = 12;
L3_HD uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
L3_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// short strings (<= 12 bytes): two independent hashes of their three little-endian words (zero padded) and their length.
// The FIRST choice is the one nearly every probe ends with (tokens are placed in id order and keep their first-choice
// bucket when it is free), so it is made of as few instructions as a usable hash allows: two rotates, one three-input
// xor, one multiply-add, one shift-xor, one 24-bit multiply, one shift = 8 full-rate operations on gfx950 (the hash it
// replaced took 18).  Parameters chosen by counting first-choice collisions over the 123,039 short entries of the Llama-3
// vocabulary: 13,621 (338 among ids < 20,000) against 14,436 expected of an ideal hash and 15,166 (554) of the old one.
// The SECOND choice is only computed for the strings that were not found in their first bucket (a few per cent).
L3_HD uint32_t vocab_hash_short_a(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len, uint32_t log2_buckets) {
    uint32_t t = w0 ^ rotl32(w1, 13) ^ rotl32(w2, 26);
    t += mul24(len, 0x9E3779u);
    t ^= t >> 12;
    return mul24(t, 0xB5297Bu) >> (32 - log2_buckets);
}
L3_HD uint32_t vocab_hash_short_b(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len, uint32_t log2_buckets) {
    uint32_t y = mul24(w0 ^ (w0 >> 9), 0x7FEB35u) + mul24(w1 ^ (w1 >> 9), 0x846CA7u) + mul24(w2 ^ (w2 >> 9), 0xD6E8FFu) +
                 mul24(len, 0x165667u);
    y ^= y >> 14;
    return mul24(y, 0x68E31Du) >> (32 - log2_buckets);
}
L3_HD void vocab_hash_short2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len, uint32_t log2_buckets, uint32_t* b1,
                             uint32_t* b2) {
    *b1 = vocab_hash_short_a(w0, w1, w2, len, log2_buckets);
    *b2 = vocab_hash_short_b(w0, w1, w2, len, log2_buckets);
}
// longer strings: two running hashes over the little-endian words of the (expanded) pre-token, the last
// word zero padded
struct StrHash { uint32_t h1, h2; };
L3_HD StrHash strhash_init() { return StrHash{2166136261u, 0x9747B28Cu}; }
L3_HD void strhash_add(StrHash& h, uint32_t word) {
    h.h1 = (h.h1 ^ word) * 0x9E3779B1u; h.h1 ^= h.h1 >> 15;
    h.h2 = (h.h2 + word) * 0x85EBCA6Bu; h.h2 ^= h.h2 >> 13;
}
L3_HD void strhash_buckets(const StrHash& h, uint32_t len, uint32_t log2_buckets, uint32_t* b1, uint32_t* b2) {
    *b1 = mix32(h.h1 ^ (len * 0x9E3779B1u)) >> (32 - log2_buckets);
    *b2 = mix32(h.h2 + len * 0x85EBCA6Bu + 0x68E31DA4u) >> (32 - log2_buckets);
}

// ---- cut points inside long pre-tokens --------------------------------------------------------------------------
// Every symbol BPE ever creates is a vocabulary entry (a merge rule's product is one), and a symbol's bytes are a
// substring of the pre-token.  So if NO vocabulary entry equals a substring x[a..b) with a < p < b, no merge can ever join
// the symbols left and right of byte position p: the greedy merge loop (bpe.go:49-59) of the pre-token is the merge loop
// of x[0..p) and of x[p..n) run independently (merges on one side never change a pair on the other, and the pair across
// p never has a rule), and their results concatenated.  Such a p is a CUT.
//   reach(s) = s + (length of the longest vocabulary entry that is a prefix of x[s..n))   (>= s + 1: single bytes)
//   p is a cut  <=>  max over s < p of reach(s) <= p.
// The tables answer "is x[s..s+k) a PREFIX of some vocabulary entry": exactly for k = 2 (65,536 bits), by a one-hash
// Bloom filter for k >= 3.  A false positive only lengthens a match: reach is then an over-estimate, cuts get fewer,
// and every cut found is still a true cut.  (Whole pre-token shortcut, bpe.go:27-33: a pre-token with a cut is no
// vocabulary entry, so only cut-free pre-tokens are probed as a whole.)
constexpr int kPrefix2Words = 65536 / 32;
L3_HD uint32_t cut_hash2(uint32_t a, uint32_t b) { return ((a << 8) | b) * 0x9E3779B1u + 0x7F4A7C15u; }   // state after two bytes
L3_HD uint32_t cut_hash_add(uint32_t h, uint32_t byte) { h = (h ^ (byte + 1u)) * 0x85EBCA6Bu; return h ^ (h >> 15); }
L3_HD uint32_t cut_hash_bit(uint32_t h, uint32_t log2_bits) { return (h * 0xC2B2AE35u) >> (32 - log2_bits); }
// length of the longest string x[0..k), k <= n, that the tables accept as a prefix of a vocabulary entry (k >= 1);
// get(i) = byte i.  n = bytes available (up to the end of the pre-token).
template <class Get>
L3_HD uint32_t cut_match_len(const uint32_t* prefix2, const uint32_t* bloom, uint32_t bloom_log2, Get get, uint32_t n) {
    if (n < 2) return n;
    const uint32_t a = get(0), b = get(1);
    if (!((prefix2[(a << 3) | (b >> 5)] >> (b & 31u)) & 1u)) return 1;
    uint32_t len = 2, h = cut_hash2(a, b);
    while (len < n && len < (uint32_t)kMaxTokenBytes) {
        const uint32_t h2 = cut_hash_add(h, get(len));
        const uint32_t bit = cut_hash_bit(h2, bloom_log2);
        if (!((bloom[bit >> 5] >> (bit & 31u)) & 1u)) break;
        h = h2;
        len++;
    }
    return len;
}

// ---- character classes ------------------------------------------------------------------------
// two-stage table: stage1[cp >> 8] -> block; block = 64 bytes, 2 bits per code point:
// 0 other, 1 letter (L*), 2 decimal digit (Nd), 3 white space (unicode.IsSpace)
constexpr int kClsStage1 = 0x110000 >> 8;  // 4352
constexpr uint32_t kClsDirectCps = 0x20000;  // planes 0 and 1 (32 KB of table)

// device-resident tables (all pointers are device pointers when used by kernels)
struct Tables {
    const uint16_t* cls_stage1;   // [kClsStage1]
    const uint8_t* cls_blocks;    // [n_blocks][64]
    const uint8_t* cls_direct;    // [kClsDirectCps / 4] the same bytes for the code points below kClsDirectCps, without the first stage: one load, not two dependent ones (the tile kernel)
    const uint8_t* byte_tok;      // [256] token id of each single byte (ids 0..255)
    const uint32_t* bytepair;     // [65536] prio of (byte a, byte b) indexed a<<8|b, kNoPrio if no rule
    const PairBucket* pair_buckets;  // [1 << pair_log2]
    const VocabSlot* vocab_slots; // [1 << vocab_log2]
    const uint32_t* vocab_blob;   // raw bytes of tokens longer than kVocabInline, as zero-padded words
    const int32_t* special_id;    // [256] canonical special index -> token id (0 = not a token)
    const uint32_t* prefix2;      // [kPrefix2Words] bit (a << 8 | b): some vocabulary entry starts with the bytes a b
    const uint32_t* prefix_bloom; // [1 << (bloom_log2 - 5)] one-hash Bloom filter of the prefixes of 3 and more bytes
    const uint32_t* rank_merged;  // rank mode only (else nullptr): merged id of every prio
    uint32_t pair_log2;
    uint32_t vocab_log2;
    uint32_t bloom_log2;
    uint32_t rank_count;          // rank mode: entries of rank_merged
};

// token id a merge of priority `prio` produces.  Callers may pass kNoPrio (lanes without a pair compute along and drop
// the answer): packed mode is plain arithmetic, rank mode clamps the index so that the load stays inside the table.
L3_HD uint32_t prio_merged_id(const Tables& t, uint32_t prio) {
    return t.rank_merged ? t.rank_merged[prio < t.rank_count ? prio : 0u] : prio >> 7;
}

// canonical special index: 0..7 = the named tokens in regex order (special.go:12), 8+k = reserved_k
constexpr int kSpecialNamed = 8;

}  // namespace l3
