/* sanitize_main.c -- runs the CPU oracle (l3_oracle.c) over a file of inputs in an AddressSanitizer + UBSan build and
 * writes what it computed, so that tests/test_sanitizers.py can (a) see a clean sanitizer run and (b) compare the output
 * with the ordinary build's.  Test infrastructure, like everything under oracle/.
 *
 *   usage: sanitize_oracle <vocab_base64.txt> <merges_binary.txt> <inputs.bin> <outputs.bin>
 *   inputs.bin:  u32 count, then per record u32 length + bytes
 *   outputs.bin: per record: i64 n_ids, ids (i32) of encode(BOS, EOS); i64 n_starts, starts (u32) of pretokenize;
 *                i64 n_split of split_document; i64 n_bytes, bytes of decode(ids)
 *                then once: i64 n, ids of one 4-thread encode_batch over all records */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "l3_oracle.h"

static uint8_t* read_file(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* p = (uint8_t*)malloc((size_t)sz + 1);
    if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    *n = (size_t)sz;
    return p;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s vocab merges inputs outputs\n", argv[0]); return 2; }
    size_t vn, mn, in_n;
    uint8_t* v = read_file(argv[1], &vn);
    uint8_t* m = read_file(argv[2], &mn);
    uint8_t* in = read_file(argv[3], &in_n);
    l3o_tokenizer* t = l3o_new((const char*)v, vn, (const char*)m, mn);
    if (!t) { fprintf(stderr, "l3o_new failed\n"); return 2; }
    FILE* out = fopen(argv[4], "wb");
    if (!out) return 2;
    uint32_t count;
    memcpy(&count, in, 4);
    size_t pos = 4, total = 0;
    uint64_t* off = (uint64_t*)calloc((size_t)count + 1, 8);
    uint8_t* all = (uint8_t*)malloc(in_n + 1);
    for (uint32_t r = 0; r < count; r++) {
        uint32_t len;
        memcpy(&len, in + pos, 4);
        pos += 4;
        const uint8_t* text = in + pos;
        pos += len;
        memcpy(all + total, text, len);
        total += len;
        off[r + 1] = total;
        /* exact-size buffers: an overrun by one element is an ASan report, not a silent success */
        size_t cap = (size_t)len * 3 + 8;
        int32_t* ids = (int32_t*)malloc(cap * 4);
        int64_t n = l3o_encode(t, text, len, 1, 1, ids, cap);
        fwrite(&n, 8, 1, out);
        fwrite(ids, 4, (size_t)n, out);
        uint32_t* st = (uint32_t*)malloc(((size_t)len + 1) * 4);
        int64_t ns = l3o_pretokenize(text, len, st, (size_t)len + 1);
        fwrite(&ns, 8, 1, out);
        fwrite(st, 4, (size_t)ns, out);
        uint8_t* kd = (uint8_t*)malloc((size_t)len + 1);
        int64_t nk = l3o_split_document(text, len, st, kd, (size_t)len + 1);
        fwrite(&nk, 8, 1, out);
        int64_t nb = l3o_decode(t, ids, (size_t)n, NULL, 0);      /* size first, then an exact buffer */
        uint8_t* dec = (uint8_t*)malloc((size_t)nb + 1);
        int64_t nb2 = l3o_decode(t, ids, (size_t)n, dec, (size_t)nb);
        if (nb2 != nb) { fprintf(stderr, "decode size changed\n"); return 3; }
        fwrite(&nb, 8, 1, out);
        fwrite(dec, 1, (size_t)nb, out);
        free(ids); free(st); free(kd); free(dec);
    }
    size_t cap = total * 3 + 2 * (size_t)count + 8;
    int32_t* ids = (int32_t*)malloc(cap * 4);
    uint64_t* ooff = (uint64_t*)malloc(((size_t)count + 1) * 8);
    int64_t n = l3o_encode_batch(t, all, off, count, 1, 1, ids, cap, ooff, 4);
    fwrite(&n, 8, 1, out);
    fwrite(ids, 4, (size_t)(n > 0 ? n : 0), out);
    fclose(out);
    free(ids); free(ooff); free(all); free(off); free(in); free(v); free(m);
    l3o_free(t);
    return 0;
}
