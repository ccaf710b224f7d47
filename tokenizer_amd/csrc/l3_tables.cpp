// l3_tables.cpp -- see l3_tables.h.  Host only (plain C++17).
#include "l3_tables.h"

#include <string.h>

#include <algorithm>
#include <unordered_map>

#include "unicode15_ranges.h"

namespace l3 {

namespace {

bool b64_decode(const uint8_t* s, size_t n, std::vector<uint8_t>& out) {
    int8_t lut[256];
    memset(lut, -1, sizeof lut);
    const char* alpha = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    for (int i = 0; i < 64; i++) lut[(uint8_t)alpha[i]] = (int8_t)i;
    out.clear();
    out.reserve(n / 4 * 3 + 3);
    uint32_t acc = 0;
    int bits = 0;
    bool pad = false;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = s[i];
        if (c == '\n' || c == '\r') continue;  // Go's decoder skips line breaks
        if (c == '=') { pad = true; continue; }
        if (pad || lut[c] < 0) return false;
        acc = (acc << 6) | (uint32_t)lut[c];
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}

// Inverse of the GPT-2 byte <-> printable-rune map (reference encoding.go:35-80): code point -> byte.
void build_unicode_to_byte(int16_t u2b[512], uint16_t b2u[256]) {
    for (int i = 0; i < 512; i++) u2b[i] = -1;
    int next = 256;
    for (int b = 0; b < 256; b++) {
        bool printable = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE);
        int cp = printable ? b : next++;
        b2u[b] = (uint16_t)cp;
        u2b[cp] = (int16_t)b;
    }
}

struct BitReader {  // big-endian bit stream (merges file: 17-bit ids, decoder.go:75-126)
    const uint8_t* p; size_t nbits; size_t pos = 0;
    bool has(int k) const { return pos + (size_t)k <= nbits; }
    uint32_t get(int k) {
        uint32_t v = 0;
        for (int i = 0; i < k; i++, pos++) v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1u);
        return v;
    }
};

int cls_of(uint32_t cp) {
    auto in = [&](const unsigned (*t)[2], unsigned n) {
        unsigned lo = 0, hi = n;
        while (lo < hi) {
            unsigned mid = (lo + hi) / 2;
            if (cp < t[mid][0]) hi = mid; else if (cp > t[mid][1]) lo = mid + 1; else return true;
        }
        return false;
    };
    if (in(L3_UNI_LETTER, L3_UNI_LETTER_COUNT)) return 1;
    if (in(L3_UNI_DIGIT, L3_UNI_DIGIT_COUNT)) return 2;
    if (in(L3_UNI_SPACE, L3_UNI_SPACE_COUNT)) return 3;
    return 0;
}

void build_class_tables(HostTables& T) {
    T.cls_stage1.assign(kClsStage1, 0);
    T.cls_blocks.clear();
    std::unordered_map<std::string, uint16_t> seen;
    for (int blk = 0; blk < kClsStage1; blk++) {
        std::string b(64, '\0');
        for (int i = 0; i < 256; i++) {
            int c = cls_of((uint32_t)blk * 256 + (uint32_t)i);
            b[i >> 2] = (char)((uint8_t)b[i >> 2] | (c << ((i & 3) * 2)));
        }
        auto it = seen.find(b);
        if (it == seen.end()) {
            uint16_t idx = (uint16_t)(T.cls_blocks.size() / 64);
            T.cls_blocks.insert(T.cls_blocks.end(), b.begin(), b.end());
            it = seen.emplace(b, idx).first;
        }
        T.cls_stage1[blk] = it->second;
    }
    T.cls_direct.resize(kClsDirectCps / 4);
    for (uint32_t cp = 0; cp < kClsDirectCps; cp += 4) T.cls_direct[cp >> 2] = T.cls_blocks[(size_t)T.cls_stage1[cp >> 8] * 64u + ((cp & 0xFFu) >> 2)];
}

}  // namespace

std::string special_name(int i) {
    static const char* named[kSpecialNamed] = {"begin_of_text", "end_of_text", "start_header_id", "end_header_id",
                                               "eot_id", "eom_id", "python_tag", "finetune_right_pad_id"};
    if (i < kSpecialNamed) return std::string("<|") + named[i] + "|>";
    return "<|reserved_special_token_" + std::to_string(i - kSpecialNamed) + "|>";
}

// ids follow the order of GetDefaultSpecialTokens (special.go:19-40): the list index + 128000
int special_default_id(int i) {
    static const int named_id[kSpecialNamed] = {128000, 128001, 128006, 128007, 128009, 128008, 128010, 128004};
    if (i < kSpecialNamed) return named_id[i];
    int k = i - kSpecialNamed;  // reserved_special_token_k
    if (k == 0) return 128002;
    if (k == 1) return 128003;
    if (k == 2) return 128005;
    return 128008 + k;
}

bool build_tables(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len, HostTables& T,
                  const std::vector<std::string>* specials) {
    std::vector<uint8_t> vtxt, mbin;
    if (!vocab_b64 || !b64_decode((const uint8_t*)vocab_b64, vocab_len, vtxt) || vtxt.empty()) {
        T.error = "decode vocabulary base64"; return false;
    }
    if (!merges_b64 || !b64_decode((const uint8_t*)merges_b64, merges_len, mbin)) {
        T.error = "decode merges base64"; return false;
    }
    int16_t u2b[512]; uint16_t b2u[256];
    build_unicode_to_byte(u2b, b2u);

    // ---- vocabulary: one token per non-empty line, id = line index; text is in the byte-level alphabet
    T.tok_bytes.clear(); T.tok_off.clear();
    std::unordered_map<std::string, int> by_bytes;
    by_bytes.reserve(1 << 18);
    size_t i = 0;
    while (i <= vtxt.size()) {
        size_t e = i;
        while (e < vtxt.size() && vtxt[e] != '\n') e++;
        if (e > i) {
            T.tok_off.push_back((uint32_t)T.tok_bytes.size());
            size_t start = T.tok_bytes.size();
            for (size_t k = i; k < e;) {  // UTF-8 decode of code points < 0x200
                uint32_t cp; uint8_t c = vtxt[k];
                if (c < 0x80) { cp = c; k += 1; }
                else if ((c & 0xE0) == 0xC0 && k + 1 < e) { cp = ((uint32_t)(c & 0x1F) << 6) | (vtxt[k + 1] & 0x3F); k += 2; }
                else { T.error = "vocabulary entry outside the byte-level alphabet"; return false; }
                if (cp >= 512 || u2b[cp] < 0) { T.error = "vocabulary entry outside the byte-level alphabet"; return false; }
                T.tok_bytes.push_back((uint8_t)u2b[cp]);
            }
            std::string key((const char*)T.tok_bytes.data() + start, T.tok_bytes.size() - start);
            by_bytes[key] = (int)T.tok_off.size() - 1;  // later duplicates win, as in tokenizer.go:201-204
        }
        i = e + 1;
    }
    int n_base = (int)T.tok_off.size();
    // special tokens are appended after the base vocabulary (tokenizer.go:194-198): the defaults in the order of
    // GetDefaultSpecialTokens (special.go:19-40), or the caller's list in its place
    T.special_names.clear();
    if (specials) {
        T.special_names = *specials;
    } else {
        T.special_names.resize(kNumSpecial);
        for (int c = 0; c < kNumSpecial; c++) T.special_names[special_default_id(c) - 128000] = special_name(c);
    }
    const int n_special = (int)T.special_names.size();
    if (n_base < 256 || n_base + n_special > (1 << 17)) { T.error = "unsupported vocabulary size"; return false; }
    T.n_base = n_base;
    std::unordered_map<std::string, int> special_lookup;
    for (int k = 0; k < n_special; k++) {
        T.tok_off.push_back((uint32_t)T.tok_bytes.size());
        T.tok_bytes.insert(T.tok_bytes.end(), T.special_names[k].begin(), T.special_names[k].end());
        special_lookup[T.special_names[k]] = n_base + k;   // (later duplicates win, as in tokenizer.go:201-204)
    }
    // tokenLookup[name] for a name (tokenizer.go:236,357-368): the special list first (its ids are the later ones), then a
    // vocabulary entry that happens to spell it; 0 / -1 if there is none
    auto lookup_name = [&](const std::string& name) {
        auto it = special_lookup.find(name);
        if (it != special_lookup.end()) return it->second;
        auto ib = by_bytes.find(name);
        return ib != by_bytes.end() ? ib->second : -1;
    };
    for (int c = 0; c < kNumSpecial; c++) { const int id = lookup_name(special_name(c)); T.special_id[c] = id > 0 ? id : 0; }
    T.bos_id = lookup_name("<|begin_of_text|>");
    T.eos_id = lookup_name("<|end_of_text|>");
    T.tok_off.push_back((uint32_t)T.tok_bytes.size());
    T.n_tokens = n_base + n_special;
    auto tok = [&](int id) { return std::string((const char*)T.tok_bytes.data() + T.tok_off[id], T.tok_off[id + 1] - T.tok_off[id]); };

    // ---- single-byte tokens
    for (int b = 0; b < 256; b++) {
        auto it = by_bytes.find(std::string(1, (char)b));
        if (it == by_bytes.end() || it->second > 255) { T.error = "single-byte tokens must have ids 0..255"; return false; }
        T.byte_tok[b] = (uint8_t)it->second;
    }

    // ---- merges: 17-bit id pairs, rank = pair index + 1 (decoder.go:41-71: any pair list is valid, a later duplicate of a pair
    //      overwrites the earlier one's rank)
    BitReader br{mbin.data(), mbin.size() * 8};
    struct Rule { uint32_t a, b, merged, prio; };
    std::vector<Rule> rules;
    rules.reserve(mbin.size() * 8 / 34 + 1);
    bool grouped = true;   // all pairs of one product consecutive (<= 128 of them), product ids never decreasing: the packed priority
    {
        std::unordered_map<uint64_t, size_t> seen;   // pair -> index in `rules`
        seen.reserve(1 << 19);
        bool tail = false;   // a rule whose product is not in the vocabulary has been seen
        while (br.has(34)) {
            uint32_t a = br.get(17), b = br.get(17);
            // decoder.go:61-63 skips a pair that names an id beyond the token list (base vocabulary + special tokens); a pair
            // that names a special token is kept there but can never apply: symbol lists hold base tokens only
            if ((int)a >= n_base || (int)b >= n_base) continue;
            auto it = by_bytes.find(tok((int)a) + tok((int)b));
            if (it == by_bytes.end()) {
                // The reference keeps such a rule; when it fires the two operands are frozen and nothing is produced
                // (bpe.go:157-161).  With every such rule ranked after every productive one -- a truncated vocabulary with
                // the full merge file -- they fire only once no productive rule applies any more and change no output: dropped.
                tail = true;
                continue;
            }
            if (tail) { T.error = "a merge rule whose product is not in the vocabulary precedes productive rules (unsupported)"; return false; }
            if (it->second >= n_base) { T.error = "merge rule whose product is a special token"; return false; }
            const uint32_t m = (uint32_t)it->second;
            auto ins = seen.emplace(pair_key(a, b), rules.size());
            if (!ins.second) {   // the pair again: the later rank counts.  The earlier rule becomes a hole in the list
                rules[ins.first->second].merged = 0xFFFFFFFFu;
                ins.first->second = rules.size();
                grouped = false;
            }
            rules.push_back(Rule{a, b, m, 0});
        }
    }
    {
        std::vector<Rule> kept;
        kept.reserve(rules.size());
        for (const Rule& r : rules) if (r.merged != 0xFFFFFFFFu) kept.push_back(r);
        rules.swap(kept);
    }
    {
        uint32_t last_merged = 0, sub = 0;
        for (size_t k = 0; k < rules.size() && grouped; k++) {
            const uint32_t m = rules[k].merged;
            if (k && m < last_merged) grouped = false;
            sub = (k && m == last_merged) ? sub + 1 : 0;
            if (sub >= 128) grouped = false;
            last_merged = m;
            rules[k].prio = (m << 7) | sub;
        }
    }
    T.rank_merged.clear();
    if (!grouped) {   // rank mode (l3_common.h): prio = index among the kept rules, merged id by table
        if (rules.size() >= (1u << 24)) { T.error = "more than 2^24 merge rules"; return false; }
        T.rank_merged.resize(rules.size());
        for (size_t k = 0; k < rules.size(); k++) { rules[k].prio = (uint32_t)k; T.rank_merged[k] = rules[k].merged; }
    }
    T.n_merges = (int)rules.size();

    // direct byte-pair table
    T.tok_flags.assign((size_t)T.n_tokens, 0);
    for (const Rule& r : rules) { T.tok_flags[r.a] |= 1; T.tok_flags[r.b] |= 2; }
    T.bytepair.assign(65536, kNoPrio);
    uint8_t tok_byte[256];  // id -> byte for ids 0..255
    for (int b = 0; b < 256; b++) tok_byte[T.byte_tok[b]] = (uint8_t)b;
    for (const Rule& r : rules)
        if (r.a < 256 && r.b < 256) T.bytepair[((uint32_t)tok_byte[r.a] << 8) | tok_byte[r.b]] = r.prio;

    // ---- pair table: cuckoo placement into two-slot buckets
    uint64_t rng = 0x243F6A8885A308D3ull;
    auto next_rand = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (T.pair_log2 = 15;; T.pair_log2++) {
        if ((2ull << T.pair_log2) * 5 < rules.size() * 9) continue;  // load factor <= 5/9
        T.pair_buckets.assign(1ull << T.pair_log2, PairBucket{{kPairEmpty, kPairEmpty}});
        bool ok = true;
        for (const Rule& r : rules) {
            uint64_t item = (pair_key(r.a, r.b) << 24) | r.prio;
            bool placed = false;
            for (int kick = 0; kick < 1000 && !placed; kick++) {
                uint32_t b[2];
                pair_hash2((uint32_t)(item >> 41), (uint32_t)((item >> 24) & 0x1FFFFu), T.pair_log2, &b[0], &b[1]);
                for (int k = 0; k < 4 && !placed; k++) {
                    uint64_t& slot = T.pair_buckets[b[k >> 1]].s[k & 1];
                    if (slot == kPairEmpty) { slot = item; placed = true; }
                }
                if (!placed) {  // evict a random resident of one of the two buckets and re-place it
                    uint64_t rr = next_rand();
                    uint64_t& slot = T.pair_buckets[b[rr & 1]].s[(rr >> 1) & 1];
                    std::swap(slot, item);
                }
            }
            if (!placed) { ok = false; break; }
        }
        if (ok) break;
        if (T.pair_log2 > 24) { T.error = "pair table placement failed"; return false; }
    }

    // ---- whole-string vocabulary table (bpe.go:27-33 looks the whole pre-token up first): two-choice cuckoo as well
    T.vocab_blob.clear();
    std::vector<VocabSlot> items;
    items.reserve(by_bytes.size());
    for (auto& kv : by_bytes) {
        const std::string& s = kv.first;
        uint32_t len = (uint32_t)s.size();
        if (len == 0 || len > (uint32_t)kMaxTokenBytes) { T.error = "vocabulary entry longer than 128 bytes"; return false; }
        VocabSlot slot{{0, 0, 0, (uint32_t)kv.second | (len << 17)}};
        if (len <= (uint32_t)kVocabInline) {
            for (uint32_t k = 0; k < len; k++) slot.w[k >> 2] |= (uint32_t)(uint8_t)s[k] << ((k & 3) * 8);
        } else {
            StrHash h = strhash_init();
            slot.w[1] = (uint32_t)T.vocab_blob.size();
            for (uint32_t i = 0; i < len; i += 4) {
                uint32_t w = 0;
                for (uint32_t k = 0; k < 4 && i + k < len; k++) w |= (uint32_t)(uint8_t)s[i + k] << (8 * k);
                strhash_add(h, w);
                T.vocab_blob.push_back(w);
            }
            slot.w[0] = h.h2;
        }
        items.push_back(slot);
    }
    while (T.vocab_blob.size() % 4) T.vocab_blob.push_back(0);
    std::sort(items.begin(), items.end(), [](const VocabSlot& a, const VocabSlot& b) { return (a.w[3] & 0x1FFFFu) < (b.w[3] & 0x1FFFFu); });
    auto vbuckets = [&](const VocabSlot& v, uint32_t* b) {
        uint32_t len = v.w[3] >> 17;
        if (len <= (uint32_t)kVocabInline) {
            vocab_hash_short2(v.w[0], v.w[1], v.w[2], len, T.vocab_log2, &b[0], &b[1]);
        } else {
            StrHash h = strhash_init();
            for (uint32_t k = 0; k < (len + 3) / 4; k++) strhash_add(h, T.vocab_blob[v.w[1] + k]);
            strhash_buckets(h, len, T.vocab_log2, &b[0], &b[1]);
        }
    };
    // one slot per bucket, load factor <= 1/4; tokens are placed in id order (roughly by falling frequency) and take
    // their first-choice bucket when it is free, so the probe of a frequent token ends after one 16-byte load
    for (T.vocab_log2 = 14;; T.vocab_log2++) {
        if ((1ull << T.vocab_log2) < items.size() * 4) continue;
        T.vocab_slots.assign(1ull << T.vocab_log2, VocabSlot{{0, 0, 0, kVocabEmpty}});
        bool ok = true;
        for (VocabSlot item : items) {
            bool placed = false;
            uint32_t from = 0xFFFFFFFFu;  // bucket the item in hand was evicted from
            for (int kick = 0; kick < 1000 && !placed; kick++) {
                uint32_t b[2];
                vbuckets(item, b);
                for (int k = 0; k < 2 && !placed; k++) {
                    VocabSlot& slot = T.vocab_slots[b[k]];
                    if (slot.w[3] == kVocabEmpty) { slot = item; placed = true; }
                }
                if (!placed) {
                    uint32_t victim = b[0] == from ? b[1] : (b[1] == from ? b[0] : b[next_rand() & 1]);
                    std::swap(T.vocab_slots[victim], item);
                    from = victim;
                }
            }
            if (!placed) { ok = false; break; }
        }
        if (ok) break;
        if (T.vocab_log2 > 24) { T.error = "vocabulary table placement failed"; return false; }
    }

    // ---- prefix tables for the cut points of long pre-tokens (l3_common.h): every prefix of 2 and more bytes of every
    //      entry of the base vocabulary (special-token names are no BPE symbols)
    T.prefix2.assign(kPrefix2Words, 0);
    {
        size_t n_long = 0;
        for (int id = 0; id < n_base; id++) { uint32_t l = T.tok_off[id + 1] - T.tok_off[id]; if (l > 2) n_long += l - 2; }
        T.bloom_log2 = 20;
        while ((1ull << T.bloom_log2) < n_long * 24 && T.bloom_log2 < 26) T.bloom_log2++;   // <= 1/24 of the bits set: ~4 % false positives
        T.prefix_bloom.assign((size_t)1 << (T.bloom_log2 - 5), 0);
        for (int id = 0; id < n_base; id++) {
            const uint8_t* b = T.tok_bytes.data() + T.tok_off[id];
            const uint32_t l = T.tok_off[id + 1] - T.tok_off[id];
            if (l < 2) continue;
            T.prefix2[((uint32_t)b[0] << 3) | (b[1] >> 5)] |= 1u << (b[1] & 31u);
            uint32_t h = cut_hash2(b[0], b[1]);
            for (uint32_t k = 2; k < l; k++) {
                h = cut_hash_add(h, b[k]);
                const uint32_t bit = cut_hash_bit(h, T.bloom_log2);
                T.prefix_bloom[bit >> 5] |= 1u << (bit & 31u);
            }
        }
    }

    build_class_tables(T);
    return true;
}

}  // namespace l3
