// l3_hosttest.cpp -- host build of the host/device-shared headers (l3_rules.h, l3_bpe.h) and of the
// table builder, exported for the CPU test-suite (tests/test_rules_host.py).  It lets the position-local
// pre-tokenizer rules, the table probes and the merge loop be fuzzed against the oracle without a
// GPU.  It is a unit-test shim: nothing in the product calls it and it is not a CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_tables.h"

using namespace l3;

namespace {
struct Handle {
    HostTables H;
    Tables T;
};
struct DocBytes {  // byte source over one whole document
    const uint8_t* t; int64_t n;
    uint32_t b(int64_t q) const { return t[q]; }
    bool ds(int64_t q) const { return q == 0 || q == n; }
    int64_t lo() const { return 0; }
    int64_t hi() const { return n; }
};
struct DocInfo {
    const uint16_t* w; int64_t n;
    uint32_t info(int64_t q) const { return w[q]; }
    int64_t lo() const { return 0; }
    int64_t hi() const { return n; }
};
}  // namespace

extern "C" {

void* l3t_create(const void* vocab, size_t vlen, const void* merges, size_t mlen) {
    Handle* h = new Handle();
    if (!build_tables(vocab, vlen, merges, mlen, h->H)) { delete h; return nullptr; }
    h->T.cls_stage1 = h->H.cls_stage1.data();
    h->T.cls_blocks = h->H.cls_blocks.data();
    h->T.cls_direct = h->H.cls_direct.data();
    h->T.byte_tok = h->H.byte_tok;
    h->T.bytepair = h->H.bytepair.data();
    h->T.pair_buckets = h->H.pair_buckets.data();
    h->T.vocab_slots = h->H.vocab_slots.data();
    h->T.vocab_blob = h->H.vocab_blob.data();
    h->T.special_id = h->H.special_id;
    h->T.pair_log2 = h->H.pair_log2;
    h->T.vocab_log2 = h->H.vocab_log2;
    h->T.prefix2 = h->H.prefix2.data();
    h->T.prefix_bloom = h->H.prefix_bloom.data();
    h->T.rank_merged = h->H.rank_merged.empty() ? nullptr : h->H.rank_merged.data();
    h->T.rank_count = (uint32_t)h->H.rank_merged.size();
    h->T.bloom_log2 = h->H.bloom_log2;
    return h;
}
void l3t_destroy(void* p) { delete (Handle*)p; }

// table statistics: [n_tokens, n_merges, pair_log2, vocab_log2, class blocks, vocab blob bytes]
void l3t_stats(void* p, int64_t* out) {
    Handle* h = (Handle*)p;
    out[0] = h->H.n_tokens; out[1] = h->H.n_merges; out[2] = h->H.pair_log2; out[3] = h->H.vocab_log2;
    out[4] = (int64_t)h->H.cls_blocks.size() / 64; out[5] = (int64_t)h->H.vocab_blob.size() * 4;
}

// seq_len_word() against seq_len() over every lead and second byte with third and fourth bytes from the edges of the byte
// classes: returns the number of disagreements (0 expected).
int64_t l3t_seq_len_word_check() {
    static const uint8_t edge[] = {0x00, 0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC2, 0xE0, 0xED, 0xF0, 0xF4, 0xFF};
    int64_t bad = 0;
    for (uint32_t b0 = 0x80; b0 < 0x100; b0++)
        for (uint32_t b1 = 0; b1 < 0x100; b1++)
            for (uint8_t b2 : edge)
                for (uint8_t b3 : edge) {
                    const uint8_t t[5] = {(uint8_t)b0, (uint8_t)b1, b2, b3, 0x41};
                    DocBytes a{t, 5};
                    const int want = is_cont(b0) ? 0 : seq_len(a, 0);
                    const uint32_t got = seq_len_word(b0 | (b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24));
                    if ((uint32_t)want != got) bad++;
                }
    return bad;
}

// the direct class table of planes 0 and 1 against the two-stage table: number of code points that differ (0 expected)
int64_t l3t_class_direct_check(void* p) {
    Handle* h = (Handle*)p;
    int64_t bad = 0;
    if (h->H.cls_direct.size() != kClsDirectCps / 4) return -1;
    for (uint32_t cp = 0x80; cp < kClsDirectCps; cp++)
        if (class_from_block_byte(h->T.cls_direct[cp >> 2], cp) != class_from_block_byte(h->T.cls_blocks[class_block_index(h->T.cls_stage1[cp >> 8], cp)], cp)) bad++;
    return bad;
}

int l3t_class_of(void* p, uint32_t cp) {
    Handle* h = (Handle*)p;
    return (int)class_of_cp(cp, h->T.cls_stage1, h->T.cls_blocks);
}

// per-byte info words of one document (special-token overlay applied), as the tile kernel builds them
static void doc_info(Handle* h, const uint8_t* text, int64_t n, std::vector<uint16_t>& w, std::vector<int32_t>& spid) {
    DocBytes a{text, n};
    w.assign((size_t)n + 1, 0);
    spid.assign((size_t)n + 1, 0);
    for (int64_t p = 0; p < n; p++) w[p] = (uint16_t)byte_info(a, p, h->T.cls_stage1, h->T.cls_blocks);
    w[n] = I_SEG | I_RUNE;
    for (int64_t p = 0; p < n; p++) {
        if (text[p] != '<') continue;
        int canon = 0;
        int len = special_match(a, p, &canon);
        if (!len) continue;
        int32_t id = h->T.special_id[canon];
        w[p] |= I_SEG;
        w[p + len] |= I_SEG;
        if (id != 0) {
            w[p] |= I_SPECIAL;
            spid[p] = id;
            for (int k = 1; k < len; k++) w[p + k] = I_SPBODY;
        }
    }
}

int64_t l3t_split_document(void* hp, const uint8_t* text, int64_t n, uint32_t* starts, uint8_t* kinds, int64_t cap) {
    Handle* h = (Handle*)hp;
    std::vector<uint16_t> w; std::vector<int32_t> spid;
    doc_info(h, text, n, w, spid);
    DocInfo s{w.data(), n};
    int64_t cnt = 0;
    for (int64_t p = 0; p < n; p++) {
        uint32_t x = w[p];
        if (!(x & I_RUNE) || (x & I_SPBODY)) continue;
        int doff = 0;
        if ((x & I_CLS) == C_N && !(x & (I_SEG | I_SPECIAL))) doff = digits_before(s, p);
        if (!starts_pretoken(s, p, doff)) continue;
        if (cnt < cap) { starts[cnt] = (uint32_t)p; kinds[cnt] = (x & I_SPECIAL) ? 1 : 0; }
        cnt++;
    }
    return cnt;
}

int64_t l3t_pair_prio(void* hp, uint32_t a, uint32_t b) {
    uint32_t v = pair_prio(((Handle*)hp)->T, a, b);
    return v == kNoPrio ? -1 : (int64_t)v;
}

int32_t l3t_vocab_lookup(void* hp, const uint8_t* bytes, uint32_t n) {
    return vocab_lookup(((Handle*)hp)->T, [&](uint32_t i) { return (uint32_t)bytes[i]; }, n);
}

// PerformBPE on the raw (already U+FFFD-expanded) bytes of one pre-token, the way the kernels do it.
int64_t l3t_bpe(void* hp, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap) {
    Handle* h = (Handle*)hp;
    int32_t id = vocab_lookup(h->T, [&](uint32_t i) { return (uint32_t)bytes[i]; }, (uint32_t)n);
    if (id >= 0) { if (cap > 0) out[0] = id; return 1; }
    std::vector<int32_t> sym((size_t)n);
    std::vector<uint32_t> pr((size_t)n, kNoPrio);
    for (int64_t i = 0; i < n; i++) sym[i] = h->T.byte_tok[bytes[i]];
    for (int64_t i = 0; i + 1 < n; i++) pr[i] = h->T.bytepair[((uint32_t)bytes[i] << 8) | bytes[i + 1]];
    merge_span(h->T, sym, pr, 0, (int)n);
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++)
        if (sym[i] >= 0) { if (k < cap) out[k] = sym[i]; k++; }
    return k;
}

// Probe statistics of the greedy merge of one pre-token: out = [probes, probes whose operands both occur on that
// side of some rule (tok_flags), probes that find a rule, merges]
void l3t_probe_stats(void* hp, const uint8_t* bytes, int64_t n, int64_t* out) {
    Handle* h = (Handle*)hp;
    const Tables& T = h->T;
    std::vector<int32_t> sym((size_t)n);
    std::vector<uint32_t> pr((size_t)n, kNoPrio);
    for (int64_t i = 0; i < n; i++) sym[i] = T.byte_tok[bytes[i]];
    for (int64_t i = 0; i + 1 < n; i++) pr[i] = T.bytepair[((uint32_t)bytes[i] << 8) | bytes[i + 1]];
    auto probe = [&](uint32_t a, uint32_t b) {
        out[0]++;
        if ((h->H.tok_flags[a] & 1) && (h->H.tok_flags[b] & 2)) out[1]++;
        uint32_t p = pair_prio(T, a, b);
        if (p != kNoPrio) out[2]++;
        return p;
    };
    for (;;) {
        uint32_t best = kNoPrio; int bp = -1;
        for (int p = 0; p < (int)n; p++) if (pr[p] < best) { best = pr[p]; bp = p; }
        if (bp < 0) break;
        out[3]++;
        int q = bp + 1; while (sym[q] < 0) q++;
        int r = q + 1; while (r < (int)n && sym[r] < 0) r++;
        int l = bp - 1; while (l >= 0 && sym[l] < 0) l--;
        uint32_t m = prio_merged_id(T, best);
        sym[bp] = (int32_t)m; sym[q] = -1; pr[q] = kNoPrio;
        pr[bp] = r < (int)n ? probe(m, (uint32_t)sym[r]) : kNoPrio;
        if (l >= 0) pr[l] = probe((uint32_t)sym[l], m);
    }
}

// Sequential emulation of the ROUND-BASED merge of the slow kernel (l3_kernels.hip merge_rounds): same set S,
// same beta / theta rule, same compaction; lets the CPU suite fuzz the algorithm against the oracle.
// Returns the number of tokens; *rounds receives the number of rounds it took.
int64_t l3t_bpe_rounds(void* hp, const uint8_t* bytes, int64_t n0, int32_t* out, int64_t cap, int64_t* rounds) {
    Handle* h = (Handle*)hp;
    const Tables& T = h->T;
    std::vector<int32_t> sa((size_t)n0), sb;
    std::vector<uint32_t> pa((size_t)n0, kNoPrio), pb;
    for (int64_t i = 0; i < n0; i++) sa[i] = T.byte_tok[bytes[i]];
    for (int64_t i = 0; i + 1 < n0; i++) pa[i] = T.bytepair[((uint32_t)bytes[i] << 8) | bytes[i + 1]];
    int64_t nr = 0;
    size_t n = (size_t)n0;
    for (;;) {
        if (n < 2) break;
        uint32_t r = kNoPrio;
        for (size_t i = 0; i < n; i++) r = pa[i] < r ? pa[i] : r;
        if (r == kNoPrio) break;
        nr++;
        std::vector<uint8_t> sel(n + 2, 0);
        uint32_t run = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t p = pa[i];
            if (p != r) {
                run = 0;
                if (p != kNoPrio) {
                    uint32_t pl = i >= 1 ? pa[i - 1] : kNoPrio, pn = pa[i + 1];
                    if (pl > p && pn >= p) sel[i] = 1;
                }
            } else {
                if (!(run & 1u)) sel[i] = 1;
                run++;
            }
        }
        unsigned long long th = ~0ull;
        for (size_t i = 0; i < n; i++) {
            uint32_t p = pa[i];
            if (p == kNoPrio) continue;
            unsigned long long key = ((unsigned long long)p << 32) | i;
            if (!sel[i]) {
                bool covered = (i >= 1 && sel[i - 1] && pa[i - 1] <= p) || (sel[i + 1] && pa[i + 1] < p);
                if (!covered && key < th) th = key;
                continue;
            }
            uint32_t m = prio_merged_id(T, p), v = kNoPrio;
            if (i >= 1) {
                uint32_t lsym = (uint32_t)sa[i - 1];
                if (i >= 2 && sel[i - 2] && pa[i - 2] <= p) lsym = prio_merged_id(T, pa[i - 2]);
                v = pair_prio(T, lsym, m);
            }
            if (i + 2 < n) {
                uint32_t rsym = (uint32_t)sa[i + 2];
                if (sel[i + 2] && pa[i + 2] < p) rsym = prio_merged_id(T, pa[i + 2]);
                uint32_t v2 = pair_prio(T, m, rsym);
                v = v2 < v ? v2 : v;
            }
            unsigned long long beta = v == kNoPrio ? ~0ull : ((unsigned long long)v << 32);
            if (beta < key + 1) beta = key + 1;
            if (beta < th) th = beta;
        }
        sb.clear(); pb.clear();
        std::vector<uint8_t> dirty;
        for (size_t i = 0; i < n; i++) {
            bool dead = i >= 1 && sel[i - 1] && ((((unsigned long long)pa[i - 1]) << 32) | (i - 1)) < th;
            if (dead) continue;
            bool applied = sel[i] && ((((unsigned long long)pa[i]) << 32) | i) < th;
            sb.push_back(applied ? (int32_t)prio_merged_id(T, pa[i]) : sa[i]);
            pb.push_back(pa[i]);
            dirty.push_back(applied ? 1 : 0);
        }
        size_t n2 = sb.size();
        for (size_t q = 0; q < n2; q++) {
            bool d = dirty[q] || (q + 1 < n2 && dirty[q + 1]);
            if (!d) continue;
            pb[q] = q + 1 < n2 ? pair_prio(T, (uint32_t)sb[q], (uint32_t)sb[q + 1]) : kNoPrio;
        }
        if (getenv("L3T_TRACE")) {
            size_t cand = 0;
            for (size_t q = 0; q < n2; q++) cand += pb[q] != kNoPrio;
            fprintf(stderr, "round %lld: n %zu -> %zu, candidates left %zu\n", (long long)nr, n, n2, cand);
        }
        sa.swap(sb); pa.swap(pb);
        n = n2;
    }
    if (rounds) *rounds = nr;
    for (size_t i = 0; i < n && (int64_t)i < cap; i++) out[i] = sa[i];
    return (int64_t)n;
}

// PerformBPE of one pre-token (raw, already U+FFFD-expanded bytes) PIECE BY PIECE: the pre-token is cut at the positions
// no vocabulary entry can span (l3_common.h, cut_match_len) and every piece is merged on its own; a pre-token without a
// cut goes through the whole-string probe first, as l3t_bpe does.  This is what the slow kernel does with long
// pre-tokens; the CPU suite fuzzes it against the oracle.  *n_pieces receives the number of pieces.
int64_t l3t_bpe_pieces(void* hp, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap, int64_t* n_pieces) {
    Handle* h = (Handle*)hp;
    const Tables& T = h->T;
    std::vector<int64_t> starts;
    int64_t reach = 0;
    for (int64_t s = 0; s < n; s++) {
        if (reach <= s) starts.push_back(s);    // (s == 0 included)
        const uint32_t m = cut_match_len(T.prefix2, T.prefix_bloom, T.bloom_log2, [&](uint32_t i) { return (uint32_t)bytes[s + i]; },
                                         (uint32_t)(n - s < 1000 ? n - s : 1000));
        if (s + m > reach) reach = s + m;
    }
    starts.push_back(n);
    if (n_pieces) *n_pieces = (int64_t)starts.size() - 1;
    if (starts.size() <= 2) return l3t_bpe(hp, bytes, n, out, cap);
    int64_t k = 0;
    for (size_t pi = 0; pi + 1 < starts.size(); pi++) {
        const int64_t a = starts[pi], len = starts[pi + 1] - a;
        std::vector<int32_t> sym((size_t)len);
        std::vector<uint32_t> pr((size_t)len, kNoPrio);
        for (int64_t i = 0; i < len; i++) sym[i] = T.byte_tok[bytes[a + i]];
        for (int64_t i = 0; i + 1 < len; i++) pr[i] = T.bytepair[((uint32_t)bytes[a + i] << 8) | bytes[a + i + 1]];
        merge_span(T, sym, pr, 0, (int)len);
        for (int64_t i = 0; i < len; i++)
            if (sym[i] >= 0) { if (k < cap) out[k] = sym[i]; k++; }
    }
    return k;
}

// Decode table access for tests: raw bytes of token id (tokenizer.go:339-354)
int64_t l3t_token_bytes(void* hp, int32_t id, uint8_t* out, int64_t cap) {
    Handle* h = (Handle*)hp;
    if (id < 0 || id >= h->H.n_tokens) return -1;
    int64_t l = h->H.tok_off[id + 1] - h->H.tok_off[id];
    memcpy(out, h->H.tok_bytes.data() + h->H.tok_off[id], (size_t)(l < cap ? l : cap));
    return l;
}

}  // extern "C"
