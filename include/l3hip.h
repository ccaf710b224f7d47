/* l3hip.h -- C ABI of libl3hip.so: the Llama-3 encode path on AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary for the batch-encode hot path of agentstation/tokenizer.  Every entry point
 * names the reference interface it stands for (paths relative to the reference repository).
 * Plain C types only; the library never keeps a caller pointer after a call returns, so it can be
 * bound from cgo without pinning (see INTEGRATION.md for the Go stubs).
 *
 * Threading: calls on one context are serialised internally; use one context per GPU and per
 * concurrent stream of work.  There is no CPU fallback: if no gfx950 device is usable the calls fail.
 */
#ifndef L3HIP_H
#define L3HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct l3_ctx l3_ctx;

/* return codes */
#define L3_OK 0
#define L3_E_INVALID_ARG (-1) /* null pointer, offsets not ascending / not covering the text, ... */
#define L3_E_DATA (-2)        /* vocabulary / merge data malformed  (reference: *DataError, errors.go) */
#define L3_E_HIP (-3)         /* HIP runtime failure; l3_last_error() has the text */
#define L3_E_OUTPUT_CAP (-4)  /* output buffer too small; *n_tok / *n_bytes holds the size needed */
#define L3_E_NO_DEVICE (-5)   /* no usable gfx950 device */
#define L3_E_NOT_FOUND (-6)   /* reference: ErrTokenNotFound / ErrInvalidToken (tokenizer.go:357-368) */
#define L3_E_SCRATCH (-7)     /* l3_encode_batch_device_noalloc: the batch needs more workspace than the static array holds */

/* EncodeOptions{BOS, EOS} (tokenizer.go:107-112); nil options == L3_BOS|L3_EOS (tokenizer.go:218-220) */
#define L3_BOS 1u
#define L3_EOS 2u
/* special-token names in the text are ordinary text (the bare pre-tokenizer, pretokenizer.Tokenize) */
#define L3_NO_SPECIAL 4u
/* BPE.EncodeBPE (tokenizer.go:80-84,478-488 -> bpe.go:18-71): every document is ONE pre-token, handed over as its raw bytes
 * (the caller has inverted the byte-level alphabet of the reference's pre-token strings, include/llama3.hpp does).  No
 * pre-tokenizer, no special-token matching, no UTF-8 interpretation: the whole-string probe (bpe.go:27-33), then the merge
 * loop over the bytes.  An empty document gives no token.  Combine with L3_BOS / L3_EOS as usual (EncodeBPE itself adds none). */
#define L3_ONE_PRETOKEN 8u

/* llama3.New() (tokenizer.go:142-213): builds the tables from the contents of the reference's two
 * embedded data files (internal/vocabulary/embed.go:13-20: vocab_base64.txt, merges_binary.txt) and
 * uploads them to GPU `device`.  The 256 default special tokens (internal/tokens/special.go:19-40)
 * are appended as the reference does.  On failure *out is NULL and l3_last_error(NULL) has the reason. */
int l3_create(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len, int device,
              l3_ctx** out);
/* llama3.New(llama3.WithSpecialTokens(list)) (options.go:17-37, tokenizer.go:194-198): the list REPLACES the 256 defaults;
 * its tokens get the ids behind the vocabulary in list order.  As in the reference the text is still split at the 256
 * DEFAULT names (tokenizer.go:27,232), and a name that is not in the list is ordinary text (tokenizer.go:236); BOS / EOS
 * are added only if the list holds <|begin_of_text|> / <|end_of_text|> (tokenizer.go:225-229,258-262).  special_tokens ==
 * NULL: the defaults (l3_create).  A malformed or repeated entry: L3_E_INVALID_ARG (reference: *ConfigError). */
int l3_create_ex(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len,
                 const char* const* special_tokens, size_t n_special, int device, l3_ctx** out);
void l3_destroy(l3_ctx* ctx);
/* text of the last error on this context (ctx == NULL: of the last failed l3_create on this thread) */
const char* l3_last_error(const l3_ctx* ctx);

/* Tokenizer.Encode / EncodeBytes / AppendTokens (tokenizer.go:217-330) for a BATCH of documents.
 * Document i is text[doc_off[i] .. doc_off[i+1]); doc_off has n_docs+1 ascending entries, doc_off[0] = 0.
 * Document i's ids are out_tok[out_off[i] .. out_off[i+1]); *n_tok = out_off[n_docs].
 * Bytes may be arbitrary (invalid UTF-8 is handled as Go's []rune conversion does, state_machine.go:36).
 * Host pointers.  If out_cap is too small L3_E_OUTPUT_CAP is returned with *n_tok = the size needed; out_off is still
 * filled in, the contents of out_tok are unspecified.  Batches of 16 MiB and more are moved in pieces of whole documents, so
 * that uploads, kernels and downloads overlap: directly when text / out_tok lie in page-locked memory (l3_host_alloc),
 * else through page-locked staging buffers of the context that a few host threads fill and drain. */
int l3_encode_batch(l3_ctx* ctx, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                    int32_t* out_tok, uint64_t out_cap, uint64_t* out_off, uint64_t* n_tok);

/* Same with DEVICE pointers (inputs already resident in HBM; outputs stay there).  `stream` is a
 * hipStream_t (NULL: the context's own stream).  The call returns after the work has finished. */
int l3_encode_batch_device(l3_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off,
                           uint64_t n_docs, uint32_t flags, int32_t* d_out_tok, uint64_t out_cap,
                           uint64_t* d_out_off, uint64_t* n_tok, void* stream);

/* l3_encode_batch_device as two stream-ordered halves (NO COUNTERPART IN THE REFERENCE, whose Encode is a plain function
 * call): l3_encode_submit queues the whole pipeline of one batch on `stream` and returns without waiting; l3_encode_wait
 * waits for that batch and returns what l3_encode_batch_device returns (*n_tok, L3_E_OUTPUT_CAP with the size needed, ...).
 * A caller that submits batch k + 1 before it waits for batch k keeps the GPU busy while the host looks at a result: the
 * loop of a corpus job (bench.py).  Up to L3_MAX_PENDING batches may be pending on a context; they are executed in
 * submission order (they share the context's workspace), each needs its own output buffers, and all buffers of a batch
 * must stay valid until its wait has returned.  A batch that turns out to need a larger workspace or one of the
 * host-driven passes (very long pre-tokens) is run again, synchronously, inside its l3_encode_wait. */
#define L3_MAX_PENDING 4
int l3_encode_submit(l3_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off, uint64_t n_docs,
                     uint32_t flags, int32_t* d_out_tok, uint64_t out_cap, uint64_t* d_out_off, void* stream, int* ticket);
int l3_encode_wait(l3_ctx* ctx, int ticket, uint64_t* n_tok);

/* l3_encode_batch_device for callers that must not allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot call hipMalloc -- tight serving loops, harnesses that forbid allocation.  (NOT for use under HIP stream capture:
 * the call synchronises the stream and reads a status word on the host, which a capture does not allow.)  Same
 * arguments, results and error codes as l3_encode_batch_device, but its workspace (intermediate items, deferred pre-token
 * lists, counters) is a file-scope __device__ array of L3_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds
 * until the process ends; the call itself performs no hipMalloc / hipFree / allocating library call.
 * Limits: n_bytes <= L3_NOALLOC_MAX_BYTES and n_docs <= L3_NOALLOC_MAX_DOCS, else L3_E_INVALID_ARG; a batch within those
 * limits whose deferred pre-tokens outgrow the fixed lists (pathological text) returns L3_E_SCRATCH -- use
 * l3_encode_batch_device for it.  All calls on one GPU share the array, whatever the context: the library serialises them
 * inside one process; two PROCESSES each have their own copy. */
#define L3_NOALLOC_MAX_BYTES (4u << 20)
#define L3_NOALLOC_MAX_DOCS 65536u
#define L3_NOALLOC_SCRATCH_BYTES (40u << 20)
int l3_encode_batch_device_noalloc(l3_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const uint64_t* d_doc_off,
                                   uint64_t n_docs, uint32_t flags, int32_t* d_out_tok, uint64_t out_cap,
                                   uint64_t* d_out_off, uint64_t* n_tok, void* stream);

/* Pre-token boundaries: kinds[i] = 1 if a pre-token starts at byte i, 2 if a special token starts there, else 0.
 * flags = L3_NO_SPECIAL: Tokenizer.PreTokenize (tokenizer.go:490-506 -> pretokenizer.Tokenize), which knows no
 * special tokens; flags = 0: the split Encode makes, tokens.SplitBySpecialTokens (special.go:48-83) first and the
 * pre-tokenizer on the text between.  Host pointers; kinds has doc_off[n_docs] entries. */
int l3_pretokenize_batch(l3_ctx* ctx, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                         uint8_t* kinds);

/* Tokenizer.DecodeBytes (tokenizer.go:339-354) for a batch of id sequences: ids outside
 * [0, VocabSize) are skipped.  Host pointers; same capacity protocol as l3_encode_batch. */
int l3_decode_batch(l3_ctx* ctx, const int32_t* ids, const uint64_t* id_off, uint64_t n_seqs, uint8_t* out_bytes,
                    uint64_t out_cap, uint64_t* out_off, uint64_t* n_bytes);

/* Tokenizer.GetSpecialTokenID (tokenizer.go:357-368): name like "<|eot_id|>". */
int l3_special_token_id(const l3_ctx* ctx, const char* name, size_t name_len, int32_t* id);
/* Tokenizer.VocabSize (tokenizer.go:420-422) */
int l3_vocab_size(const l3_ctx* ctx);
/* raw bytes of one token (what Decode appends for it); returns its length or L3_E_NOT_FOUND */
int l3_token_bytes(const l3_ctx* ctx, int32_t id, uint8_t* out, size_t cap);

/* Page-locked host memory for batch buffers (no counterpart in the reference: Go slices live on the Go heap, which is
 * pageable).  l3_encode_batch moves a batch whose text and out_tok both lie in memory from l3_host_alloc in overlapped
 * pieces (about 3.5 times the host-fed rate of pageable buffers); a cgo shim allocates its batch text and token
 * buffers here once and reuses them.  The memory belongs to the caller until l3_host_free; the context is only needed
 * to select the device. */
int l3_host_alloc(l3_ctx* ctx, size_t bytes, void** out);
int l3_host_free(l3_ctx* ctx, void* p);

/* Measurement support (no counterpart in the reference): when enabled, HIP events are recorded on the
 * launch stream between the kernels of every encode call; l3_last_timings() returns the milliseconds of
 * the last call: [0] validate+docs, [1] tile kernel, [2] merge kernel, [3] slow kernel, [4] block totals, [5] the host-driven
 * passes of a batch with very long pre-tokens (0 otherwise), [6] emit, [7] first kernel start -> last kernel end, [8] (cap >= 9)
 * the kernels of the last l3_decode_batch.  Returns the number of values written.  enabled = 2: only the events the roofline
 * figures need (around the tile kernel, and first start / last end): [0], [1] and [7] are filled in, the others 0;
 * every event between two kernels keeps the second from starting early, four of them cost a 64 MiB batch about 3 %. */
int l3_set_profiling(l3_ctx* ctx, int enabled);
int l3_last_timings(l3_ctx* ctx, float* ms, int cap);
/* the same eight values summed over the *calls profiled encode calls since l3_set_profiling() was last called (every
 * call of it resets the sums), so a benchmark loop need not ask after every step */
int l3_timing_totals(l3_ctx* ctx, double* ms, int cap, uint64_t* calls);
/* counters of the last encode call: [0] tiles, [1] deferred (slow) pre-tokens, [2] tokens of all
 * deferred pre-tokens, [3] scratch bytes, [4] retries after a workspace grew, [5] short deferred pre-tokens,
 * [6] window items of the long deferred pre-tokens, [7] symbols merged by the grid-wide path (pieces of 32,768 symbols
 * and more that no cut divides) */
int l3_last_counters(l3_ctx* ctx, uint64_t* out, int cap);

/* ---- several GPUs behind one handle --------------------------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE (its only concurrent workload is goroutines calling Encode on one Tokenizer,
 * benchmark_test.go:151-174).  One context per entry of devices[] (a device may be named more than once); a batch is cut
 * into one contiguous range of documents per context, balanced by bytes, each range is encoded on its GPU by a host thread
 * of its own, and the results come back concatenated in document order -- identical to l3_encode_batch on one context.
 * special_tokens / n_special as in l3_create_ex (NULL: the defaults). */
typedef struct l3_multi l3_multi;
int l3_multi_create(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len,
                    const char* const* special_tokens, size_t n_special, const int* devices, int n_dev, l3_multi** out);
void l3_multi_destroy(l3_multi* m);
int l3_multi_size(const l3_multi* m);
/* context i, for everything that is not sharded (l3_special_token_id, l3_vocab_size, l3_decode_batch, counters, ...);
 * owned by the handle */
l3_ctx* l3_multi_context(l3_multi* m, int i);
const char* l3_multi_last_error(const l3_multi* m);
/* Same arguments, results and capacity protocol as l3_encode_batch. */
int l3_multi_encode_batch(l3_multi* m, const uint8_t* text, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                          int32_t* out_tok, uint64_t out_cap, uint64_t* out_off, uint64_t* n_tok);
/* The partition l3_multi_encode_batch uses (host arithmetic only): cuts[0..world], shard r = documents
 * [cuts[r], cuts[r+1]); cuts[r] is the first document that starts at or after r/world of the bytes. */
int l3_partition_by_bytes(const uint64_t* doc_off, uint64_t n_docs, int world, uint64_t* cuts);

#ifdef __cplusplus
}
#endif
#endif
