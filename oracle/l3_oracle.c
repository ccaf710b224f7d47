/* l3_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline), never part of the product.
 *
 * A CPU restatement, in plain C, of the reference's Llama-3 encode path.  Every function names the
 * reference lines it restates (paths relative to /root/reference).  It deliberately keeps the
 * reference's data structures (string-keyed maps, rune slices, heap + linked list with Deleted
 * flags) rather than the GPU build's tables, so that the two implementations are independent.
 *
 * Pinned by the reference's own known-answer tests -- see tests/test_oracle_kat.py.
 */
#include "l3_oracle.h"
#include "unicode15_ranges.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small utilities: growable buffers, bytes -> int hash map (stands in for Go's map[string]int) */

typedef struct { uint32_t off, len, hash; int32_t val; } bm_slot;
typedef struct {
    bm_slot* slots; size_t cap, count;         /* cap is a power of two; len==UINT32_MAX marks empty */
    uint8_t* pool; size_t pool_len, pool_cap;
} bmap;

static uint32_t fnv1a(const uint8_t* p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
    return h;
}
static void bmap_init(bmap* m, size_t cap_pow2) {
    m->cap = cap_pow2; m->count = 0;
    m->slots = (bm_slot*)malloc(m->cap * sizeof(bm_slot));
    for (size_t i = 0; i < m->cap; i++) m->slots[i].len = UINT32_MAX;
    m->pool_cap = 1 << 16; m->pool_len = 0; m->pool = (uint8_t*)malloc(m->pool_cap);
}
static void bmap_free(bmap* m) { free(m->slots); free(m->pool); m->slots = NULL; m->pool = NULL; }
static bm_slot* bmap_find(const bmap* m, const uint8_t* k, size_t n, uint32_t h) {
    size_t i = h & (m->cap - 1);
    for (;;) {
        bm_slot* s = &m->slots[i];
        if (s->len == UINT32_MAX) return s;
        if (s->hash == h && s->len == n && memcmp(m->pool + s->off, k, n) == 0) return s;
        i = (i + 1) & (m->cap - 1);
    }
}
static int bmap_get(const bmap* m, const uint8_t* k, size_t n, int32_t* val) {
    bm_slot* s = bmap_find(m, k, n, fnv1a(k, n));
    if (s->len == UINT32_MAX) return 0;
    *val = s->val; return 1;
}
static void bmap_grow(bmap* m) {
    bm_slot* old = m->slots; size_t oc = m->cap;
    m->cap *= 2;
    m->slots = (bm_slot*)malloc(m->cap * sizeof(bm_slot));
    for (size_t i = 0; i < m->cap; i++) m->slots[i].len = UINT32_MAX;
    for (size_t i = 0; i < oc; i++) if (old[i].len != UINT32_MAX) {
        size_t j = old[i].hash & (m->cap - 1);
        while (m->slots[j].len != UINT32_MAX) j = (j + 1) & (m->cap - 1);
        m->slots[j] = old[i];
    }
    free(old);
}
/* map[k] = v (overwrites, like Go map assignment) */
static void bmap_put(bmap* m, const uint8_t* k, size_t n, int32_t v) {
    if ((m->count + 1) * 2 > m->cap) bmap_grow(m);
    uint32_t h = fnv1a(k, n);
    bm_slot* s = bmap_find(m, k, n, h);
    if (s->len != UINT32_MAX) { s->val = v; return; }
    if (m->pool_len + n > m->pool_cap) {
        while (m->pool_len + n > m->pool_cap) m->pool_cap *= 2;
        m->pool = (uint8_t*)realloc(m->pool, m->pool_cap);
    }
    memcpy(m->pool + m->pool_len, k, n);
    s->off = (uint32_t)m->pool_len; s->len = (uint32_t)n; s->hash = h; s->val = v;
    m->pool_len += n; m->count++;
}

typedef struct { int32_t* v; size_t n, cap; } ivec;
static void ivec_push(ivec* a, int32_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 64; a->v = (int32_t*)realloc(a->v, a->cap * sizeof(int32_t)); }
    a->v[a->n++] = x;
}

/* ------------------------------------------------------------------------------------------ */
/* base64 (Go encoding/base64 StdEncoding.DecodeString: '\r' and '\n' are ignored)              */

static int b64_decode(const char* s, size_t n, uint8_t** out, size_t* out_len) {
    static int8_t T[256]; static int init = 0;
    if (!init) {
        memset(T, -1, sizeof T);
        const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; i++) T[(uint8_t)A[i]] = (int8_t)i;
        init = 1;
    }
    uint8_t* o = (uint8_t*)malloc(n / 4 * 3 + 4); size_t k = 0;
    uint32_t acc = 0; int bits = 0, pad = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = (uint8_t)s[i];
        if (c == '\r' || c == '\n') continue;
        if (c == '=') { pad++; continue; }
        if (pad || T[c] < 0) { free(o); return -1; }
        acc = (acc << 6) | (uint32_t)T[c]; bits += 6;
        if (bits >= 8) { bits -= 8; o[k++] = (uint8_t)(acc >> bits); }
    }
    *out = o; *out_len = k; return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* UTF-8 with Go semantics                                                                      */

/* Go's utf8.DecodeRune as used by []rune(string) (state_machine.go:36): a byte that does not
 * start a valid sequence yields U+FFFD and a width of 1. */
static uint32_t decode_rune(const uint8_t* p, size_t n, int* width) {
    uint8_t b0 = p[0];
    *width = 1;
    if (b0 < 0x80) return b0;
    if (b0 < 0xC2 || b0 > 0xF4) return 0xFFFD;
    if (b0 < 0xE0) {
        if (n < 2 || (p[1] & 0xC0) != 0x80) return 0xFFFD;
        *width = 2; return ((uint32_t)(b0 & 0x1F) << 6) | (p[1] & 0x3F);
    }
    if (b0 < 0xF0) {
        if (n < 3) return 0xFFFD;
        uint8_t lo = 0x80, hi = 0xBF;
        if (b0 == 0xE0) lo = 0xA0;
        if (b0 == 0xED) hi = 0x9F;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80) return 0xFFFD;
        *width = 3; return ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    }
    if (n < 4) return 0xFFFD;
    {
        uint8_t lo = 0x80, hi = 0xBF;
        if (b0 == 0xF0) lo = 0x90;
        if (b0 == 0xF4) hi = 0x8F;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) return 0xFFFD;
    }
    *width = 4;
    return ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
}
/* utf8.EncodeRune / string(rune) */
static int encode_rune(uint32_t r, uint8_t* o) {
    if (r < 0x80) { o[0] = (uint8_t)r; return 1; }
    if (r < 0x800) { o[0] = 0xC0 | (r >> 6); o[1] = 0x80 | (r & 0x3F); return 2; }
    if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
    if (r < 0x10000) { o[0] = 0xE0 | (r >> 12); o[1] = 0x80 | ((r >> 6) & 0x3F); o[2] = 0x80 | (r & 0x3F); return 3; }
    o[0] = 0xF0 | (r >> 18); o[1] = 0x80 | ((r >> 12) & 0x3F); o[2] = 0x80 | ((r >> 6) & 0x3F); o[3] = 0x80 | (r & 0x3F);
    return 4;
}

/* ------------------------------------------------------------------------------------------ */
/* character classes: state_machine.go:310-320 -> unicode.IsLetter / IsDigit / IsSpace (15.0.0) */

static int in_ranges(const unsigned int (*t)[2], unsigned int n, uint32_t r) {
    unsigned int lo = 0, hi = n;
    while (lo < hi) {
        unsigned int mid = (lo + hi) / 2;
        if (r < t[mid][0]) hi = mid; else if (r > t[mid][1]) lo = mid + 1; else return 1;
    }
    return 0;
}
static int is_letter(uint32_t r) { return in_ranges(L3_UNI_LETTER, L3_UNI_LETTER_COUNT, r); }
static int is_number(uint32_t r) { return in_ranges(L3_UNI_DIGIT, L3_UNI_DIGIT_COUNT, r); }   /* Nd only */
static int is_whitespace(uint32_t r) { return in_ranges(L3_UNI_SPACE, L3_UNI_SPACE_COUNT, r); }
int l3o_rune_class(uint32_t r) { return is_letter(r) | (is_number(r) << 1) | (is_whitespace(r) << 2); }

/* unicode.ToLower restricted to what matchesAt (state_machine.go:287-307) can observe: the
 * pattern runes are ' s t r e v m l d; the only runes whose simple lower-case mapping lands on
 * one of those are the ASCII letters themselves (U+017F and U+212A do not). */
static uint32_t to_lower(uint32_t r) { return (r >= 'A' && r <= 'Z') ? r + 32 : r; }

/* ------------------------------------------------------------------------------------------ */
/* pre-tokenizer: state_machine.go:82-284                                                       */

typedef struct { const uint32_t* in; size_t n, pos; } sm_t;

static int sm_matches_at(const sm_t* sm, const char* s) {            /* :287-307, caseInsensitive */
    size_t len = strlen(s);
    if (sm->pos + len > sm->n) return 0;
    for (size_t i = 0; i < len; i++)
        if (to_lower(sm->in[sm->pos + i]) != to_lower((uint8_t)s[i])) return 0;
    return 1;
}
static size_t sm_try_contraction(sm_t* sm) {                         /* :131-148 */
    static const char* C[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    if (sm->pos >= sm->n || sm->in[sm->pos] != '\'') return 0;
    for (int i = 0; i < 7; i++)
        if (sm_matches_at(sm, C[i])) { size_t l = strlen(C[i]); sm->pos += l; return l; }
    return 0;
}
static size_t sm_try_word(sm_t* sm) {                                /* :151-175 */
    size_t start = sm->pos;
    if (sm->pos < sm->n) {
        uint32_t ch = sm->in[sm->pos];
        if (!is_letter(ch) && !is_number(ch) && ch != '\r' && ch != '\n') sm->pos++;
    }
    if (sm->pos >= sm->n || !is_letter(sm->in[sm->pos])) { sm->pos = start; return 0; }
    while (sm->pos < sm->n && is_letter(sm->in[sm->pos])) sm->pos++;
    return sm->pos - start;
}
static size_t sm_try_numbers(sm_t* sm) {                             /* :178-191 */
    if (sm->pos >= sm->n || !is_number(sm->in[sm->pos])) return 0;
    size_t start = sm->pos; int count = 0;
    while (sm->pos < sm->n && is_number(sm->in[sm->pos]) && count < 3) { sm->pos++; count++; }
    return sm->pos - start;
}
static size_t sm_try_punct(sm_t* sm) {                               /* :194-231 */
    size_t start = sm->pos;
    if (sm->pos < sm->n && sm->in[sm->pos] == ' ') sm->pos++;
    if (sm->pos >= sm->n || is_whitespace(sm->in[sm->pos]) || is_letter(sm->in[sm->pos]) || is_number(sm->in[sm->pos])) {
        sm->pos = start; return 0;
    }
    while (sm->pos < sm->n) {
        uint32_t ch = sm->in[sm->pos];
        if (is_whitespace(ch) || is_letter(ch) || is_number(ch)) break;
        sm->pos++;
    }
    while (sm->pos < sm->n && (sm->in[sm->pos] == '\r' || sm->in[sm->pos] == '\n')) sm->pos++;
    return sm->pos - start;
}
static size_t sm_try_newlines(sm_t* sm) {                            /* :234-259 */
    size_t start = sm->pos;
    while (sm->pos < sm->n && is_whitespace(sm->in[sm->pos])) {
        if (sm->in[sm->pos] == '\r' || sm->in[sm->pos] == '\n') break;
        sm->pos++;
    }
    int found = 0;
    while (sm->pos < sm->n && (sm->in[sm->pos] == '\r' || sm->in[sm->pos] == '\n')) { sm->pos++; found = 1; }
    if (!found) { sm->pos = start; return 0; }
    return sm->pos - start;
}
static size_t sm_try_whitespace(sm_t* sm) {                          /* :262-284 */
    if (sm->pos >= sm->n || !is_whitespace(sm->in[sm->pos])) return 0;
    size_t start = sm->pos;
    while (sm->pos < sm->n && is_whitespace(sm->in[sm->pos])) sm->pos++;
    if (sm->pos < sm->n && !is_whitespace(sm->in[sm->pos])) {
        if (sm->pos > start + 1) sm->pos--;
    }
    return sm->pos - start;
}
/* matchNext (:82-128): returns the rune length of the next pre-token and advances. */
static size_t sm_match_next(sm_t* sm) {
    size_t l;
    if ((l = sm_try_contraction(sm))) return l;
    if ((l = sm_try_word(sm))) return l;
    if ((l = sm_try_numbers(sm))) return l;
    if ((l = sm_try_punct(sm))) return l;
    if ((l = sm_try_newlines(sm))) return l;
    if ((l = sm_try_whitespace(sm))) return l;
    sm->pos++; return 1;                                             /* fallback :126-127 */
}

/* []rune(text) with the source offset of every rune kept alongside (state_machine.go:36) */
typedef struct { uint32_t* r; uint32_t* off; size_t n; } runes_t;
static runes_t to_runes(const uint8_t* text, size_t n) {
    runes_t R; R.r = (uint32_t*)malloc((n + 1) * sizeof(uint32_t)); R.off = (uint32_t*)malloc((n + 1) * sizeof(uint32_t)); R.n = 0;
    size_t i = 0;
    while (i < n) { int w; R.r[R.n] = decode_rune(text + i, n - i, &w); R.off[R.n] = (uint32_t)i; R.n++; i += (size_t)w; }
    R.off[R.n] = (uint32_t)n;
    return R;
}

int64_t l3o_pretokenize(const uint8_t* text, size_t n, uint32_t* starts, size_t cap) {
    runes_t R = to_runes(text, n);
    sm_t sm = {R.r, R.n, 0};
    int64_t cnt = 0;
    while (sm.pos < sm.n) {                                          /* Tokenize :61-63 */
        size_t s = sm.pos;
        sm_match_next(&sm);
        if ((size_t)cnt < cap) starts[cnt] = R.off[s];
        cnt++;
    }
    free(R.r); free(R.off);
    return cnt;
}

int64_t l3o_pretokenize_strings(const uint8_t* text, size_t n, uint8_t* buf, size_t buf_cap, uint32_t* ends, size_t cap) {
    runes_t R = to_runes(text, n);
    sm_t sm = {R.r, R.n, 0};
    int64_t cnt = 0; size_t k = 0; int overflow = 0;
    while (sm.pos < sm.n) {
        size_t s = sm.pos;
        sm_match_next(&sm);
        for (size_t i = s; i < sm.pos; i++) {                        /* string(sm.input[a:b]) */
            uint8_t tmp[4]; int w = encode_rune(R.r[i], tmp);
            if (k + (size_t)w > buf_cap) { overflow = 1; break; }
            memcpy(buf + k, tmp, (size_t)w); k += (size_t)w;
        }
        if ((size_t)cnt < cap) ends[cnt] = (uint32_t)k; else overflow = 1;
        cnt++;
        if (overflow) break;
    }
    free(R.r); free(R.off);
    return overflow ? -1 : cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* tokenizer tables: tokenizer.go:96-104,142-213; vocab.go; vocabulary/decoder.go              */

typedef struct { bmap m; } pcache;                                   /* bpe/cache.go SimpleCache: pretoken -> slot in ids pool */

struct l3o_tokenizer {
    uint8_t* tok_pool; uint32_t* tok_off; int n_tokens;              /* tokens []string */
    bmap token_lookup;                                               /* tokenLookup map[string]int */
    bmap merge_rules;                                                /* mergeRules map[string]int ("A B" -> rank) */
    uint32_t b2u[256];                                               /* encoding.BytesToUnicode */
    int16_t u2b[512];                                                /* encoding.UnicodeToBytes (all values < 0x144) */
    int cache_enabled;
    bmap cache_keys; ivec cache_ids; ivec cache_spans;               /* shared cache of the single-call API */
};

static const uint8_t* tok_str(const l3o_tokenizer* t, int id, size_t* len) {
    *len = t->tok_off[id + 1] - t->tok_off[id];
    return t->tok_pool + t->tok_off[id];
}

/* encoding.CreateByteMappings (encoding.go:35-80) */
static void create_byte_mappings(l3o_tokenizer* t) {
    int bs[256], cs[256], nb = 0;
    for (int i = '!'; i <= '~'; i++) { bs[nb] = i; cs[nb] = i; nb++; }
    for (int i = 0xA1; i <= 0xAC; i++) { bs[nb] = i; cs[nb] = i; nb++; }
    for (int i = 0xAE; i <= 0xFF; i++) { bs[nb] = i; cs[nb] = i; nb++; }
    int n = 0;
    for (int b = 0; b < 256; b++) {
        int found = 0;
        for (int j = 0; j < nb; j++) if (bs[j] == b) { found = 1; break; }
        if (!found) { bs[nb] = b; cs[nb] = 256 + n; nb++; n++; }
    }
    for (int i = 0; i < 512; i++) t->u2b[i] = -1;
    for (int i = 0; i < nb; i++) { t->b2u[bs[i]] = (uint32_t)cs[i]; t->u2b[cs[i]] = (int16_t)bs[i]; }
}

/* vocabulary.unpackTokenPairIDs (decoder.go:75-126), same 3-byte window / mask / shift arithmetic */
static int* unpack_token_pair_ids(const uint8_t* data, size_t len, size_t* count) {
    *count = 0;
    if (len == 0) return NULL;
    size_t max_bits = len * 8, first_pad = max_bits - (max_bits % 17);
    int* ids = (int*)malloc((first_pad / 17 + 1) * sizeof(int));
    static const int mask[8] = {0x1FFFF, 0x3FFFF, 0x7FFFF, 0xFFFFF, 0x1FFFFF, 0x3FFFFF, 0x7FFFFF, 0xFFFFFF};
    for (size_t bit = 0; bit < first_pad; bit += 17) {
        size_t bi = bit / 8;
        if (bi + 2 >= len) break;
        int v = ((int)data[bi] << 16) + ((int)data[bi + 1] << 8) + (int)data[bi + 2];
        int off = (int)(bit % 8);
        v &= mask[8 - off - 1];
        int rs = 8 - (17 - 8 - (8 - off));
        if (rs > 0 && rs < 24) v >>= rs;
        ids[(*count)++] = v;
    }
    return ids;
}

/* tokens.GetDefaultSpecialTokens (special.go:19-40) */
static int special_token_name(int idx, char* out) {
    static const char* named[11] = {
        "<|begin_of_text|>", "<|end_of_text|>", "<|reserved_special_token_0|>", "<|reserved_special_token_1|>",
        "<|finetune_right_pad_id|>", "<|reserved_special_token_2|>", "<|start_header_id|>", "<|end_header_id|>",
        "<|eom_id|>", "<|eot_id|>", "<|python_tag|>"};
    if (idx < 11) { strcpy(out, named[idx]); return (int)strlen(out); }
    /* for i := 3; i <= 256-9; i++ */
    int i = idx - 11 + 3;
    int n = 0; const char* pre = "<|reserved_special_token_";
    strcpy(out, pre); n = (int)strlen(pre);
    char num[8]; int k = 0; int v = i; do { num[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) out[n++] = num[--k];
    out[n++] = '|'; out[n++] = '>'; out[n] = 0;
    return n;
}

l3o_tokenizer* l3o_new(const char* vocab_b64, size_t vocab_len, const char* merges_b64, size_t merges_len) {
    return l3o_new_special(vocab_b64, vocab_len, merges_b64, merges_len, NULL, 0);
}

/* New(WithSpecialTokens(list)) (options.go:17-37): config.specialTokens replaces getDefaultSpecialTokens() in
 * tokenizer.go:194-198; everything else -- the split regex of the 256 default names included (tokenizer.go:27,232) --
 * stays as it is.  specials == NULL: the defaults. */
l3o_tokenizer* l3o_new_special(const char* vocab_b64, size_t vocab_len, const char* merges_b64, size_t merges_len,
                               const char* const* specials, size_t n_specials) {
    uint8_t *vd = NULL, *md = NULL; size_t vn = 0, mn = 0;
    if (b64_decode(vocab_b64, vocab_len, &vd, &vn) != 0) return NULL;
    if (b64_decode(merges_b64, merges_len, &md, &mn) != 0) { free(vd); return NULL; }
    l3o_tokenizer* t = (l3o_tokenizer*)calloc(1, sizeof *t);
    create_byte_mappings(t);

    /* DecodeVocabulary (decoder.go:16-36): split on '\n', drop empty strings; ID = index */
    size_t n_lines = 0;
    for (size_t i = 0; i < vn; i++) if (vd[i] == '\n') n_lines++;
    size_t n_sp = specials ? n_specials : 256, sp_bytes = 256 * 40;
    if (specials) { sp_bytes = 0; for (size_t k = 0; k < n_specials; k++) sp_bytes += strlen(specials[k]); }
    t->tok_pool = (uint8_t*)malloc(vn + sp_bytes + 16);
    t->tok_off = (uint32_t*)malloc((n_lines + 2 + n_sp + 1) * sizeof(uint32_t));
    size_t pool = 0; int nt = 0; size_t s = 0;
    for (size_t i = 0; i <= vn; i++) {
        if (i == vn || vd[i] == '\n') {
            if (i > s) { t->tok_off[nt] = (uint32_t)pool; memcpy(t->tok_pool + pool, vd + s, i - s); pool += i - s; nt++; }
            s = i + 1;
        }
    }
    /* special tokens appended (tokenizer.go:194-198) */
    for (size_t k = 0; k < n_sp; k++) {
        char dflt[48]; const char* name = dflt; size_t l;
        if (specials) { name = specials[k]; l = strlen(name); } else l = (size_t)special_token_name((int)k, dflt);
        t->tok_off[nt] = (uint32_t)pool; memcpy(t->tok_pool + pool, name, l); pool += l; nt++;
    }
    t->tok_off[nt] = (uint32_t)pool;
    t->n_tokens = nt;
    /* tokenLookup (tokenizer.go:201-204): later ids overwrite earlier ones on duplicate strings */
    bmap_init(&t->token_lookup, 1 << 19);
    for (int id = 0; id < nt; id++) { size_t l; const uint8_t* p = tok_str(t, id, &l); bmap_put(&t->token_lookup, p, l, id); }

    /* DecompressMergeRules (decoder.go:41-71) with getMergeIdentifier = tokens[a]+" "+tokens[b] (tokenizer.go:425-427) */
    size_t n_ids = 0; int* ids = unpack_token_pair_ids(md, mn, &n_ids);
    bmap_init(&t->merge_rules, 1 << 20);
    uint8_t* key = (uint8_t*)malloc(1024);
    for (size_t i = 0; i + 1 < n_ids; i += 2) {
        int a = ids[i], b = ids[i + 1];
        if (a >= nt || b >= nt) continue;
        size_t la, lb; const uint8_t* pa = tok_str(t, a, &la); const uint8_t* pb = tok_str(t, b, &lb);
        memcpy(key, pa, la); key[la] = ' '; memcpy(key + la + 1, pb, lb);
        bmap_put(&t->merge_rules, key, la + 1 + lb, (int32_t)(i / 2 + 1));
    }
    free(key); free(ids); free(vd); free(md);
    t->cache_enabled = 1;
    bmap_init(&t->cache_keys, 1 << 12);
    return t;
}

void l3o_free(l3o_tokenizer* t) {
    if (!t) return;
    free(t->tok_pool); free(t->tok_off);
    bmap_free(&t->token_lookup); bmap_free(&t->merge_rules); bmap_free(&t->cache_keys);
    free(t->cache_ids.v); free(t->cache_spans.v);
    free(t);
}
void l3o_set_cache(l3o_tokenizer* t, int enabled) { t->cache_enabled = enabled; }
int l3o_vocab_size(l3o_tokenizer* t) { return t->n_tokens; }

/* ------------------------------------------------------------------------------------------ */
/* BPE: bpe.go:18-214, priority_queue.go                                                        */

typedef struct node {
    int orig_pos, token_id; double prio;
    uint8_t* merge_to; uint32_t merge_to_len;                        /* MergeToString */
    struct node *prev, *next; int deleted;
} node;

typedef struct { node** a; size_t n, cap; } heap_t;                  /* container/heap min-heap on prio */
static void heap_push(heap_t* h, node* x) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (node**)realloc(h->a, h->cap * sizeof(node*)); }
    size_t i = h->n++; h->a[i] = x;
    while (i > 0) { size_t p = (i - 1) / 2; if (!(h->a[i]->prio < h->a[p]->prio)) break; node* tmp = h->a[i]; h->a[i] = h->a[p]; h->a[p] = tmp; i = p; }
}
static node* heap_pop(heap_t* h) {
    node* top = h->a[0]; h->n--;
    if (h->n) {
        h->a[0] = h->a[h->n]; size_t i = 0;
        for (;;) {
            size_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < h->n && h->a[l]->prio < h->a[m]->prio) m = l;
            if (r < h->n && h->a[r]->prio < h->a[m]->prio) m = r;
            if (m == i) break;
            node* tmp = h->a[i]; h->a[i] = h->a[m]; h->a[m] = tmp; i = m;
        }
    }
    return top;
}

typedef struct arena_blk { struct arena_blk* next; size_t used, cap; } arena_blk;
typedef struct { arena_blk* head; } arena;
static void* arena_alloc(arena* A, size_t n) {
    n = (n + 15) & ~(size_t)15;
    if (!A->head || A->head->used + n > A->head->cap) {
        size_t cap = n > (1 << 20) ? n : (1 << 20);
        arena_blk* b = (arena_blk*)malloc(sizeof(arena_blk) + cap);
        b->next = A->head; b->used = 0; b->cap = cap; A->head = b;
    }
    void* p = (uint8_t*)(A->head + 1) + A->head->used; A->head->used += n; return p;
}
static void arena_free(arena* A) { while (A->head) { arena_blk* n = A->head->next; free(A->head); A->head = n; } }

typedef struct { const l3o_tokenizer* t; arena A; heap_t pq; int pretoken_len; } bpe_ctx;

/* addToMergeQueue (bpe.go:114-130) */
static void add_to_merge_queue(bpe_ctx* c, node* left) {
    if (!left->next) return;
    const l3o_tokenizer* t = c->t;
    int a = left->token_id, b = left->next->token_id;
    if (a >= t->n_tokens || b >= t->n_tokens) return;                /* getMergeIdentifier returns "" -> no rule */
    size_t la, lb; const uint8_t* pa = tok_str(t, a, &la); const uint8_t* pb = tok_str(t, b, &lb);
    uint8_t* key = (uint8_t*)arena_alloc(&c->A, la + lb + 1);
    memcpy(key, pa, la); key[la] = ' '; memcpy(key + la + 1, pb, lb);
    int32_t rank;
    if (!bmap_get(&t->merge_rules, key, la + 1 + lb, &rank)) return;
    left->prio = (double)rank + (double)left->orig_pos / (double)c->pretoken_len;
    /* strings.ReplaceAll(mergeIdentifier, " ", "") -- vocabulary strings never contain ' ' */
    memmove(key + la, key + la + 1, lb);
    left->merge_to = key; left->merge_to_len = (uint32_t)(la + lb);
    heap_push(&c->pq, left);
}
static node* new_node(bpe_ctx* c, int orig_pos, int token_id, node* prev, node* next) {
    node* n = (node*)arena_alloc(&c->A, sizeof(node));
    memset(n, 0, sizeof *n);
    n->orig_pos = orig_pos; n->token_id = token_id; n->prev = prev; n->next = next;
    return n;
}
/* performMerge + updatePreviousNode + updateMergeLinks (bpe.go:146-214) */
static node* perform_merge(bpe_ctx* c, node* left, node* first) {
    left->deleted = 1; left->next->deleted = 1;
    if (left->prev) {                                                /* updatePreviousNode */
        node* old_prev = left->prev; old_prev->deleted = 1;
        node* np = new_node(c, old_prev->orig_pos, old_prev->token_id, old_prev->prev, old_prev->next);
        left->prev = np;
        if (np->prev) np->prev->next = np; else first = np;
    }
    int32_t merged;
    if (!bmap_get(&c->t->token_lookup, left->merge_to, left->merge_to_len, &merged)) return first;
    node* res = new_node(c, left->orig_pos, merged, left->prev, left->next->next);
    if (res->prev) { res->prev->next = res; add_to_merge_queue(c, res->prev); } else first = res;
    if (res->next) { res->next->prev = res; add_to_merge_queue(c, res); }
    return first;
}

/* PerformBPE without the cache probe (bpe.go:26-66). `pt` is the byte-level (GPT-2 alphabet) string. */
static void perform_bpe_nocache(const l3o_tokenizer* t, const uint8_t* pt, size_t n, ivec* out) {
    int32_t id;
    if (bmap_get(&t->token_lookup, pt, n, &id)) { ivec_push(out, id); return; }   /* :27-33 */
    /* pretokenToIDs (:74-84): one id per rune of the byte-level string; unknown runes skipped */
    ivec ids = {0};
    for (size_t i = 0; i < n;) {
        int w; decode_rune(pt + i, n - i, &w);
        if (bmap_get(&t->token_lookup, pt + i, (size_t)w, &id)) ivec_push(&ids, id);
        i += (size_t)w;
    }
    if (ids.n <= 1) { for (size_t i = 0; i < ids.n; i++) ivec_push(out, ids.v[i]); free(ids.v); return; }
    bpe_ctx c; memset(&c, 0, sizeof c); c.t = t; c.pretoken_len = (int)n;   /* len(pretoken) in BYTES (:46) */
    /* buildMergeList (:88-111) */
    node* first = new_node(&c, 0, ids.v[0], NULL, NULL);
    node* prev = first;
    for (size_t i = 1; i < ids.n; i++) {
        node* cur = new_node(&c, (int)i, ids.v[i], prev, NULL);
        prev->next = cur;
        add_to_merge_queue(&c, prev);
        prev = cur;
    }
    while (c.pq.n > 0) {                                             /* :49-59 */
        node* l = heap_pop(&c.pq);
        if (!(l && !l->deleted && l->next && !l->next->deleted)) continue;   /* isValidMerge :141-143 */
        first = perform_merge(&c, l, first);
    }
    for (node* p = first; p; p = p->next) ivec_push(out, p->token_id);
    free(ids.v); free(c.pq.a); arena_free(&c.A);
}

/* PerformBPE with the SimpleCache memo (bpe.go:18-24,67-69; cache.go:79-104) */
typedef struct { bmap* keys; ivec* ids; ivec* spans; int enabled; } cache_ref;
static void perform_bpe(const l3o_tokenizer* t, cache_ref* cr, const uint8_t* pt, size_t n, ivec* out) {
    if (cr && cr->enabled) {
        int32_t slot;
        if (bmap_get(cr->keys, pt, n, &slot)) {
            int32_t a = cr->spans->v[2 * slot], b = cr->spans->v[2 * slot + 1];
            for (int32_t i = a; i < b; i++) ivec_push(out, cr->ids->v[i]);
            return;
        }
    }
    size_t before = out->n;
    perform_bpe_nocache(t, pt, n, out);
    if (cr && cr->enabled) {
        int32_t slot = (int32_t)(cr->spans->n / 2);
        ivec_push(cr->spans, (int32_t)cr->ids->n);
        for (size_t i = before; i < out->n; i++) ivec_push(cr->ids, out->v[i]);
        ivec_push(cr->spans, (int32_t)cr->ids->n);
        bmap_put(cr->keys, pt, n, slot);
    }
}

/* encoding.EncodeBytes (encoding.go:85-96) */
static size_t encode_bytes(const l3o_tokenizer* t, const uint8_t* data, size_t n, uint8_t* out) {
    size_t k = 0;
    for (size_t i = 0; i < n; i++) k += (size_t)encode_rune(t->b2u[data[i]], out + k);
    return k;
}

int64_t l3o_bpe_raw(l3o_tokenizer* t, const uint8_t* raw, size_t n, int32_t* out, size_t cap) {
    uint8_t* bl = (uint8_t*)malloc(2 * n + 4);
    size_t bn = encode_bytes(t, raw, n, bl);
    ivec o = {0};
    perform_bpe_nocache(t, bl, bn, &o);
    for (size_t i = 0; i < o.n && i < cap; i++) out[i] = o.v[i];
    int64_t r = (int64_t)o.n; free(o.v); free(bl); return r;
}

/* ------------------------------------------------------------------------------------------ */
/* special tokens: tokens/special.go:12,48-83                                                   */

/* Length of a match of SpecialTokenRegex starting exactly at p (0 if none).  The regex is an
 * alternation of literals, so "match at p" is a literal prefix test; FindAllStringIndex's
 * leftmost-first, non-overlapping scan is reproduced by the caller's left-to-right walk. */
static size_t special_match_at(const uint8_t* p, size_t n) {
    static const char* names[8] = {"begin_of_text", "end_of_text", "start_header_id", "end_header_id",
                                   "eot_id", "eom_id", "python_tag", "finetune_right_pad_id"};
    if (n < 2 || p[0] != '<' || p[1] != '|') return 0;
    for (int i = 0; i < 8; i++) {
        size_t l = strlen(names[i]);
        if (n >= 2 + l + 2 && memcmp(p + 2, names[i], l) == 0 && p[2 + l] == '|' && p[3 + l] == '>') return l + 4;
    }
    const char* pre = "reserved_special_token_"; size_t pl = strlen(pre);
    if (n < 2 + pl + 1 || memcmp(p + 2, pre, pl) != 0) return 0;
    /* (?:[0-9]|[1-9][0-9]|1[0-9][0-9]|2[0-3][0-9]|24[0-7]) then "|>" : RE2 leftmost-first picks the
     * first alternative that lets the whole pattern match; they have distinct lengths, so try each. */
    const uint8_t* d = p + 2 + pl; size_t rem = n - 2 - pl;
    for (size_t dl = 1; dl <= 3; dl++) {
        if (rem < dl + 2) break;
        int ok = 1; int v = 0;
        for (size_t i = 0; i < dl; i++) { if (d[i] < '0' || d[i] > '9') ok = 0; else v = v * 10 + (d[i] - '0'); }
        if (!ok) break;
        if (dl > 1 && d[0] == '0') ok = 0;
        if (v > 247) ok = 0;
        if (ok && d[dl] == '|' && d[dl + 1] == '>') return 2 + pl + dl + 2;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Encode: tokenizer.go:217-265 (+ pretokenize :490-501)                                        */

static void encode_plain_segment(const l3o_tokenizer* t, cache_ref* cr, const uint8_t* text, size_t n, ivec* out) {
    runes_t R = to_runes(text, n);
    sm_t sm = {R.r, R.n, 0};
    uint8_t* part = (uint8_t*)malloc(4 * 64 + 16); size_t part_cap = 4 * 64;
    uint8_t* enc = (uint8_t*)malloc(8 * 64 + 16);
    while (sm.pos < sm.n) {
        size_t s = sm.pos;
        sm_match_next(&sm);
        size_t nr = sm.pos - s;
        if (nr * 4 > part_cap) { part_cap = nr * 4; part = (uint8_t*)realloc(part, part_cap + 16); enc = (uint8_t*)realloc(enc, part_cap * 2 + 16); }
        size_t k = 0;
        for (size_t i = s; i < sm.pos; i++) k += (size_t)encode_rune(R.r[i], part + k);   /* string(runes) */
        size_t en = encode_bytes(t, part, k, enc);                    /* encodeBytes([]byte(part)) :497 */
        if (en == 0) continue;                                        /* pretoken == "" :247-249 */
        perform_bpe(t, cr, enc, en, out);
    }
    free(part); free(enc); free(R.r); free(R.off);
}

static void encode_impl(const l3o_tokenizer* t, cache_ref* cr, const uint8_t* text, size_t n, int bos, int eos, ivec* out) {
    /* GetSpecialTokenID (tokenizer.go:226,259,357-368): a lookup, so the ids follow the size of the base vocabulary */
    int32_t bos_id = 0, eos_id = 0;
    if (bos && bmap_get(&t->token_lookup, (const uint8_t*)"<|begin_of_text|>", 17, &bos_id)) ivec_push(out, bos_id);
    /* SplitBySpecialTokens (special.go:48-83) fused with the loop at tokenizer.go:234-255 */
    size_t last = 0, i = 0;
    while (i < n) {
        size_t ml = (text[i] == '<') ? special_match_at(text + i, n - i) : 0;
        if (!ml) { i++; continue; }
        if (i > last) encode_plain_segment(t, cr, text + last, i - last, out);
        int32_t id = 0;
        if (bmap_get(&t->token_lookup, text + i, ml, &id) && id != 0) ivec_push(out, id);   /* :236-239 */
        else encode_plain_segment(t, cr, text + i, ml, out);
        i += ml; last = i;
    }
    if (last < n) encode_plain_segment(t, cr, text + last, n - last, out);
    if (eos && bmap_get(&t->token_lookup, (const uint8_t*)"<|end_of_text|>", 15, &eos_id)) ivec_push(out, eos_id);
}

int64_t l3o_encode(l3o_tokenizer* t, const uint8_t* text, size_t n, int bos, int eos, int32_t* out, size_t cap) {
    cache_ref cr = {&t->cache_keys, &t->cache_ids, &t->cache_spans, t->cache_enabled};
    ivec o = {0};
    encode_impl(t, &cr, text, n, bos, eos, &o);
    for (size_t i = 0; i < o.n && i < cap; i++) out[i] = o.v[i];
    int64_t r = (int64_t)o.n; free(o.v); return r;
}

/* Pre-token boundaries of a whole document as Encode sees them (tokenizer.go:232-255): special
 * tokens split the text, each plain piece is pre-tokenised on its own.  starts[] gets source byte
 * offsets, kinds[] 0 for a pre-token and 1 for a special token.  Returns the count. */
int64_t l3o_split_document(const uint8_t* text, size_t n, uint32_t* starts, uint8_t* kinds, size_t cap) {
    int64_t cnt = 0; size_t last = 0, i = 0;
    uint32_t* tmp = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
    while (i <= n) {
        size_t ml = (i < n && text[i] == '<') ? special_match_at(text + i, n - i) : 0;
        if (i < n && !ml) { i++; continue; }
        if (i > last) {
            int64_t k = l3o_pretokenize(text + last, i - last, tmp, n + 1);
            for (int64_t j = 0; j < k; j++) { if ((size_t)cnt < cap) { starts[cnt] = (uint32_t)last + tmp[j]; kinds[cnt] = 0; } cnt++; }
        }
        if (i == n) break;
        if ((size_t)cnt < cap) { starts[cnt] = (uint32_t)i; kinds[cnt] = 1; }
        cnt++;
        i += ml; last = i;
    }
    free(tmp);
    return cnt;
}

/* DecodeBytes (tokenizer.go:339-354) + encoding.DecodeTokenBytes (encoding.go:101-111) */
int64_t l3o_decode(l3o_tokenizer* t, const int32_t* ids, size_t n, uint8_t* out, size_t cap) {
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= t->n_tokens) continue;
        size_t l; const uint8_t* p = tok_str(t, ids[i], &l);
        for (size_t j = 0; j < l;) {
            int w; uint32_t r = decode_rune(p + j, l - j, &w); j += (size_t)w;
            if (r < 512 && t->u2b[r] >= 0) { if (k < cap) out[k] = (uint8_t)t->u2b[r]; k++; }
        }
    }
    return (int64_t)k;
}

/* ------------------------------------------------------------------------------------------ */
/* batch driver (CPU baseline): document-per-task over a shared atomic cursor                  */

typedef struct {
    const l3o_tokenizer* t; const uint8_t* text; const uint64_t* doc_off; uint64_t n_docs; int bos, eos;
    uint64_t* cursor; ivec* per_doc; int cache;
} batch_job;

static void* batch_worker(void* arg) {
    batch_job* j = (batch_job*)arg;
    bmap keys; ivec ids = {0}, spans = {0};
    bmap_init(&keys, 1 << 12);
    cache_ref cr = {&keys, &ids, &spans, j->cache};
    for (;;) {
        uint64_t d = __atomic_fetch_add(j->cursor, 16, __ATOMIC_RELAXED);
        if (d >= j->n_docs) break;
        uint64_t e = d + 16 < j->n_docs ? d + 16 : j->n_docs;
        for (; d < e; d++)
            encode_impl(j->t, &cr, j->text + j->doc_off[d], (size_t)(j->doc_off[d + 1] - j->doc_off[d]), j->bos, j->eos, &j->per_doc[d]);
    }
    bmap_free(&keys); free(ids.v); free(spans.v);
    return NULL;
}

int64_t l3o_encode_batch(l3o_tokenizer* t, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs,
                         int bos, int eos, int32_t* out, uint64_t out_cap, uint64_t* out_off, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    ivec* per_doc = (ivec*)calloc(n_docs ? n_docs : 1, sizeof(ivec));
    uint64_t cursor = 0;
    batch_job job = {t, text, doc_off, n_docs, bos, eos, &cursor, per_doc, t->cache_enabled};
    pthread_t th[256];
    for (int i = 1; i < threads; i++) pthread_create(&th[i], NULL, batch_worker, &job);
    batch_worker(&job);
    for (int i = 1; i < threads; i++) pthread_join(th[i], NULL);
    uint64_t total = 0; int overflow = 0;
    for (uint64_t d = 0; d < n_docs; d++) {
        out_off[d] = total;
        if (total + per_doc[d].n > out_cap) overflow = 1;
        else if (per_doc[d].n) memcpy(out + total, per_doc[d].v, per_doc[d].n * sizeof(int32_t));
        total += per_doc[d].n;
        free(per_doc[d].v);
    }
    out_off[n_docs] = total;
    free(per_doc);
    return overflow ? -1 : (int64_t)total;
}
