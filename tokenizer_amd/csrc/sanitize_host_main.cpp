// sanitize_host_main.cpp -- runs the table builder (l3_tables.cpp) and the host build of the shared rule / merge headers
// (l3_hosttest.cpp) over a file of inputs in an AddressSanitizer + UBSan build, and writes what it computed so that
// tests/test_sanitizers.py can compare it with the ordinary build's.  Test support only.
//
//   usage: sanitize_host <vocab_base64.txt> <merges_binary.txt> <inputs.bin> <outputs.bin>
//   inputs.bin:  u32 count, then per record u32 length + bytes (each record doubles as a document and as one raw pre-token)
//   outputs.bin: per record: i64 n + starts (u32) + kinds (u8) of l3t_split_document; i64 n + ids of l3t_bpe;
//                i64 n + ids of l3t_bpe_rounds; i64 n + ids of l3t_bpe_pieces
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

extern "C" {
void* l3t_create(const void* vocab, size_t vlen, const void* merges, size_t mlen);
void l3t_destroy(void* p);
int64_t l3t_split_document(void* hp, const uint8_t* text, int64_t n, uint32_t* starts, uint8_t* kinds, int64_t cap);
int64_t l3t_bpe(void* hp, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap);
int64_t l3t_bpe_rounds(void* hp, const uint8_t* bytes, int64_t n0, int32_t* out, int64_t cap, int64_t* rounds);
int64_t l3t_bpe_pieces(void* hp, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap, int64_t* n_pieces);
}

static std::vector<uint8_t> read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::vector<uint8_t> v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s vocab merges inputs outputs\n", argv[0]); return 2; }
    std::vector<uint8_t> v = read_file(argv[1]), m = read_file(argv[2]), in = read_file(argv[3]);
    void* h = l3t_create(v.data(), v.size(), m.data(), m.size());
    if (!h) { fprintf(stderr, "l3t_create failed\n"); return 2; }
    FILE* out = fopen(argv[4], "wb");
    if (!out) return 2;
    uint32_t count;
    memcpy(&count, in.data(), 4);
    size_t pos = 4;
    for (uint32_t r = 0; r < count; r++) {
        uint32_t len;
        memcpy(&len, in.data() + pos, 4);
        pos += 4;
        // a heap copy of exactly `len` bytes: reads past the end of the text are reports, not luck
        std::vector<uint8_t> text(in.begin() + (long)pos, in.begin() + (long)(pos + len));
        pos += len;
        std::vector<uint32_t> st((size_t)len + 1);
        std::vector<uint8_t> kd((size_t)len + 1);
        int64_t n = l3t_split_document(h, text.data(), (int64_t)len, st.data(), kd.data(), (int64_t)len + 1);
        fwrite(&n, 8, 1, out);
        fwrite(st.data(), 4, (size_t)n, out);
        fwrite(kd.data(), 1, (size_t)n, out);
        std::vector<int32_t> ids((size_t)len + 8);
        int64_t rounds = 0, pieces = 0;
        n = len ? l3t_bpe(h, text.data(), (int64_t)len, ids.data(), (int64_t)ids.size()) : 0;
        fwrite(&n, 8, 1, out);
        fwrite(ids.data(), 4, (size_t)n, out);
        n = len ? l3t_bpe_rounds(h, text.data(), (int64_t)len, ids.data(), (int64_t)ids.size(), &rounds) : 0;
        fwrite(&n, 8, 1, out);
        fwrite(ids.data(), 4, (size_t)n, out);
        n = len ? l3t_bpe_pieces(h, text.data(), (int64_t)len, ids.data(), (int64_t)ids.size(), &pieces) : 0;
        fwrite(&n, 8, 1, out);
        fwrite(ids.data(), 4, (size_t)n, out);
    }
    fclose(out);
    l3t_destroy(h);
    return 0;
}
