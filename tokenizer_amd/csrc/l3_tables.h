// l3_tables.h -- host-side table builder: turns the reference's two data files into the tables the
// HIP kernels read.  Stands where llama3.New() builds tokens/tokenLookup/mergeRules
// (reference llama3/tokenizer.go:142-213, internal/vocabulary/decoder.go:16-126).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "l3_common.h"

namespace l3 {

struct HostTables {
    std::vector<uint16_t> cls_stage1;
    std::vector<uint8_t> cls_blocks;
    std::vector<uint8_t> cls_direct;
    uint8_t byte_tok[256];
    std::vector<uint32_t> bytepair;
    std::vector<uint8_t> tok_flags;   // per token id: bit0 = left operand of some merge rule, bit1 = right operand
    std::vector<PairBucket> pair_buckets;  // 1 << pair_log2 buckets
    uint32_t pair_log2 = 0;
    std::vector<VocabSlot> vocab_slots;
    uint32_t vocab_log2 = 0;
    std::vector<uint32_t> vocab_blob;  // words
    int32_t special_id[256];
    std::vector<uint32_t> prefix2;       // kPrefix2Words
    std::vector<uint32_t> prefix_bloom;  // 1 << (bloom_log2 - 5) words
    uint32_t bloom_log2 = 0;
    std::vector<uint32_t> rank_merged;   // rank mode only (merge data whose pairs are not grouped by ascending product): merged id of every prio
    // raw bytes of every token id (0..128255) for Decode (tokenizer.go:339-354)
    std::vector<uint8_t> tok_bytes;
    std::vector<uint32_t> tok_off;  // n_tokens + 1
    int n_tokens = 0;
    int n_merges = 0;
    int n_base = 0;                          // tokens of the vocabulary file; the special tokens follow (tokenizer.go:194-198)
    std::vector<std::string> special_names;  // ... in id order: the 256 defaults, or the list of WithSpecialTokens
    int32_t bos_id = -1, eos_id = -1;        // ids of <|begin_of_text|> / <|end_of_text|>, -1: not in the list (then Encode adds none,
                                             // tokenizer.go:225-229,258-262: GetSpecialTokenID fails)
    std::string error;
};

// Returns false and sets out.error on malformed or unsupported data.
// `specials` (optional): the special-token list of WithSpecialTokens (options.go:17-37), which REPLACES the 256 defaults;
// ids follow the vocabulary in list order.  The split of the text stays at the 256 default names (tokenizer.go:27,232); a
// default name that is not in the list is ordinary text (its lookup is 0, tokenizer.go:236).
bool build_tables(const void* vocab_b64, size_t vocab_len, const void* merges_b64, size_t merges_len,
                  HostTables& out, const std::vector<std::string>* specials = nullptr);

// Name of the special token with canonical index i (0..255) and its default id.
std::string special_name(int canonical_index);
int special_default_id(int canonical_index);

}  // namespace l3
