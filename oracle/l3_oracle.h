/* l3_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread per call) of the reference's Llama-3 encode path,
 * used as the parity oracle by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under tokenizer_amd/ may include, link or load it: the product path is HIP only.
 *
 * Pinned against the reference's own known-answer tests (tests/test_oracle_kat.py):
 *   llama3/tokenizer_test.go:22-125,162-186,320-336; README.md:113-114;
 *   llama3/internal/pretokenizer/state_machine_test.go:17-684,736-866,1130-1150.
 * The reference itself (Go) cannot be built or run in this image (no Go toolchain).
 */
#ifndef L3_ORACLE_H
#define L3_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct l3o_tokenizer l3o_tokenizer;

/* llama3.New() with the embedded data files' contents (tokenizer.go:142-213). NULL on bad data. */
l3o_tokenizer* l3o_new(const char* vocab_b64, size_t vocab_len, const char* merges_b64, size_t merges_len);
/* llama3.New(llama3.WithSpecialTokens(list)) (options.go:17-37; tokenizer.go:194-198): the list replaces the 256 default
 * special tokens; specials == NULL: the defaults. */
l3o_tokenizer* l3o_new_special(const char* vocab_b64, size_t vocab_len, const char* merges_b64, size_t merges_len,
                               const char* const* specials, size_t n_specials);
void l3o_free(l3o_tokenizer* t);
/* Enable/disable the pre-token -> ids memo (bpe/cache.go SimpleCache; on by default as in New()). */
void l3o_set_cache(l3o_tokenizer* t, int enabled);

/* Tokenizer.Encode (tokenizer.go:217-265). Returns the number of ids; writes at most cap. */
int64_t l3o_encode(l3o_tokenizer* t, const uint8_t* text, size_t n, int bos, int eos,
                   int32_t* out, size_t cap);

/* pretokenizer.Tokenize (state_machine.go:53-79) on one plain segment: writes the SOURCE byte
 * offset of each pre-token start (ascending, first is 0 when n>0). Returns the count. */
int64_t l3o_pretokenize(const uint8_t* text, size_t n, uint32_t* starts, size_t cap);
/* Same, but returns the re-encoded strings ([]rune -> string, invalid bytes become EF BF BD)
 * concatenated into buf with their end offsets in ends[]. Returns count, or -1 if buf too small. */
int64_t l3o_pretokenize_strings(const uint8_t* text, size_t n, uint8_t* buf, size_t buf_cap,
                                uint32_t* ends, size_t cap);

/* Boundaries of a whole document as Encode sees them (special tokens split the text first,
 * tokenizer.go:232-255). kinds[i]: 0 pre-token, 1 special token. Returns the count. */
int64_t l3o_split_document(const uint8_t* text, size_t n, uint32_t* starts, uint8_t* kinds, size_t cap);

/* Tokenizer.DecodeBytes (tokenizer.go:339-354). Returns byte count; writes at most cap. */
int64_t l3o_decode(l3o_tokenizer* t, const int32_t* ids, size_t n, uint8_t* out, size_t cap);

/* Character classes (state_machine.go:310-320): bit0 letter, bit1 digit(Nd), bit2 space. */
int l3o_rune_class(uint32_t r);
int l3o_vocab_size(l3o_tokenizer* t);
/* PerformBPE on one byte-level-encoded pre-token given as RAW bytes (bpe.go:18-71). */
int64_t l3o_bpe_raw(l3o_tokenizer* t, const uint8_t* raw, size_t n, int32_t* out, size_t cap);

/* Encode a batch with `threads` worker threads (document-per-thread work queue, one shared
 * tokenizer with per-thread caches). out_off has n_docs+1 entries. Returns total ids or -1 when
 * out_cap is too small. Used only for the CPU baseline and for batch parity checks. */
int64_t l3o_encode_batch(l3o_tokenizer* t, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs,
                         int bos, int eos, int32_t* out, uint64_t out_cap, uint64_t* out_off, int threads);

#ifdef __cplusplus
}
#endif
#endif
