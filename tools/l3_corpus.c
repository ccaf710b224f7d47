/* l3_corpus.c -- seeded synthetic corpora for bench.py and the parity tests.
 * Measurement / test support only: nothing in the product links it.
 *
 *   C2  "English": documents of exactly doc_bytes bytes drawn from a Zipf(1.0) word list
 *   C3  multilingual: CJK runs (some very long without punctuation), Cyrillic words, emoji sequences
 */
#include <stdint.h>
#include <string.h>

static inline uint64_t sm64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define L3C_SEED_ENGLISH 0x4C6C616D61330001ull
#define L3C_SEED_MULTI 0x4C6C616D61330003ull

/* words: concatenated word bytes, word_off[n_words+1]; cdf[n_words]: ascending thresholds scaled to 2^64 */
static uint32_t pick(const uint64_t* cdf, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (u <= cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

static void english_doc(uint64_t seed, uint32_t doc_bytes, const uint8_t* words, const uint32_t* word_off,
                        const uint64_t* cdf, uint32_t n_words, uint8_t* out) {
    uint64_t s = seed;
    uint8_t buf[64];
    uint32_t pos = 0;
    while (pos < doc_bytes) {
        uint32_t n = 0;
        if (sm64(&s) % 100 < 1) {
            uint32_t nd = 1 + (uint32_t)(sm64(&s) % 6);
            for (uint32_t i = 0; i < nd; i++) buf[n++] = (uint8_t)('0' + sm64(&s) % 10);
        } else {
            uint32_t w = pick(cdf, n_words, sm64(&s));
            uint32_t l = word_off[w + 1] - word_off[w];
            if (l > 40) l = 40;
            memcpy(buf, words + word_off[w], l);
            n = l;
            if (sm64(&s) % 10 == 0 && buf[0] >= 'a' && buf[0] <= 'z') buf[0] = (uint8_t)(buf[0] - 32);
        }
        uint32_t r = (uint32_t)(sm64(&s) % 100);
        const char* sep = r < 86 ? " " : r < 91 ? ", " : r < 96 ? ". " : r < 98 ? "\n" : r < 99 ? "'s " : "n't ";
        size_t sl = strlen(sep);
        memcpy(buf + n, sep, sl);
        n += (uint32_t)sl;
        if (n > doc_bytes - pos) n = doc_bytes - pos;
        memcpy(out + pos, buf, n);
        pos += n;
    }
}

void l3c_gen_english(uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes, const uint8_t* words,
                     const uint32_t* word_off, const uint64_t* cdf, uint32_t n_words, uint8_t* out) {
    for (uint64_t d = 0; d < n_docs; d++)
        english_doc(L3C_SEED_ENGLISH + first_doc + d, doc_bytes, words, word_off, cdf, n_words, out + d * (uint64_t)doc_bytes);
}

static uint32_t put_utf8(uint8_t* o, uint32_t cp) {
    if (cp < 0x80) { o[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { o[0] = (uint8_t)(0xC0 | (cp >> 6)); o[1] = (uint8_t)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) {
        o[0] = (uint8_t)(0xE0 | (cp >> 12)); o[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); o[2] = (uint8_t)(0x80 | (cp & 0x3F));
        return 3;
    }
    o[0] = (uint8_t)(0xF0 | (cp >> 18)); o[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F));
    o[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (uint8_t)(0x80 | (cp & 0x3F));
    return 4;
}

/* One multilingual document of exactly n bytes (padded with spaces at the end). max_run caps the CJK run
 * length in characters (SURVEY: 4096). */
void l3c_gen_multilingual(uint64_t doc_index, uint64_t n, uint32_t max_run, uint8_t* out) {
    uint64_t s = L3C_SEED_MULTI + doc_index;
    uint64_t pos = 0;
    uint8_t tmp[8];
    uint32_t bits = 0;
    while ((1u << (bits + 1)) <= max_run) bits++;
    while (pos + 4 <= n) {
        uint32_t kind = (uint32_t)(sm64(&s) % 8);
        if (kind < 3) {  /* CJK ideographs, run length log-uniform in [1, max_run] */
            uint32_t e = (uint32_t)(sm64(&s) % (bits + 1));
            uint32_t len = (1u << e) + (uint32_t)(sm64(&s) % (1u << e));
            if (len > max_run) len = max_run;
            for (uint32_t i = 0; i < len && pos + 4 <= n; i++)
                pos += put_utf8(out + pos, 0x4E00 + (uint32_t)(sm64(&s) % (0x9FA5 - 0x4E00 + 1)));
            if (sm64(&s) % 8 != 0 && pos + 4 <= n) pos += put_utf8(out + pos, (sm64(&s) & 1) ? 0xFF0C : 0x3002);
        } else if (kind < 6) {  /* Cyrillic words */
            uint32_t nw = 1 + (uint32_t)(sm64(&s) % 40);
            for (uint32_t w = 0; w < nw && pos + 32 <= n; w++) {
                uint32_t l = 2 + (uint32_t)(sm64(&s) % 13);
                for (uint32_t i = 0; i < l; i++) pos += put_utf8(out + pos, 0x0410 + (uint32_t)(sm64(&s) % 0x40));
                out[pos++] = ' ';
            }
        } else if (kind == 6) {  /* emoji, some joined by ZWJ, some followed by VS16 */
            uint32_t ne = 1 + (uint32_t)(sm64(&s) % 8);
            for (uint32_t i = 0; i < ne && pos + 16 <= n; i++) {
                pos += put_utf8(out + pos, 0x1F300 + (uint32_t)(sm64(&s) % 0x400));
                uint32_t r = (uint32_t)(sm64(&s) % 4);
                if (r == 0) pos += put_utf8(out + pos, 0x200D);
                else if (r == 1) pos += put_utf8(out + pos, 0xFE0F);
                else if (r == 2) out[pos++] = ' ';
            }
        } else {  /* a little ASCII */
            static const char* w[] = {"the ", "model ", "tokenizer ", "2024 ", "GPU, ", "is\n", "fast. ", "don't "};
            uint32_t nw = 1 + (uint32_t)(sm64(&s) % 12);
            for (uint32_t i = 0; i < nw; i++) {
                const char* x = w[sm64(&s) % 8];
                size_t l = strlen(x);
                if (pos + l + 4 > n) break;
                memcpy(out + pos, x, l);
                pos += l;
            }
        }
        (void)tmp;
    }
    while (pos < n) out[pos++] = ' ';
}
