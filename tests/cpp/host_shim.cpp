// host_shim.cpp -- C entry points over include/llama3.hpp (the C++ mirror of the reference's Go API) so that the
// pytest suite can drive it through ctypes.  Test support only.
#include <vector>

#include "../../include/llama3.hpp"

using namespace llama3;

namespace {
// scanner_test.go:458-489 controlledReader: hands out the data in reads of the given sizes, then in reads as large
// as the caller allows; EOF arrives on the read after the last byte.  A negative size = a read error at that point
// (scanner_test.go:427-433 errorReader).
struct ControlledReader : Reader {
    const uint8_t* d; size_t n, pos = 0;
    const int* sizes; int n_sizes, k = 0;
    ControlledReader(const uint8_t* data, size_t len, const int* s, int ns) : d(data), n(len), sizes(s), n_sizes(ns) {}
    int Read(uint8_t* p, int cap, bool* eof) override {
        if (k < n_sizes && sizes[k] < 0) { k++; return -1; }
        if (pos >= n) { *eof = true; return 0; }
        int want = k < n_sizes ? sizes[k] : cap;
        k++;
        if (want < 0) return -1;
        if (want > cap) want = cap;
        size_t m = n - pos < (size_t)want ? n - pos : (size_t)want;
        memcpy(p, d + pos, m);
        pos += m;
        return (int)m;
    }
};
// A VocabularyDataLoader (vocab.go:14-22) over the two data files: what a caller with its own storage would write.
// Decodes the files itself (base64, '\n'-separated tokens; 17-bit id pairs -> "left right" -> rank) so that New() sees
// nothing but token strings and a merge map.  rank_step spreads the priorities (1, 1 + step, ...): only their order counts.
struct FileBackedLoader : VocabularyDataLoader {
    std::string dir;
    int rank_step;
    bool fail_merges;
    std::vector<std::string> toks;
    FileBackedLoader(const std::string& d, int step, bool fm) : dir(d), rank_step(step), fail_merges(fm) {}
    static bool unb64(const std::string& in, std::string* out) {
        int val[256];
        for (int i = 0; i < 256; i++) val[i] = -1;
        const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; i++) val[(unsigned char)A[i]] = i;
        uint32_t acc = 0;
        int n = 0;
        for (unsigned char c : in) {
            if (c == '=' || c == '\n' || c == '\r') continue;
            if (val[c] < 0) return false;
            acc = (acc << 6) | (uint32_t)val[c];
            n += 6;
            if (n >= 8) { out->push_back((char)((acc >> (n - 8)) & 0xFF)); n -= 8; }
        }
        return true;
    }
    bool LoadVocabulary(std::vector<std::string>* tokens, Error* err) override {
        std::string file, raw;
        if (!ReadFile(dir + "/vocab_base64.txt", &file) || !unb64(file, &raw)) {
            if (err) { err->kind = ErrKind::DataNotFound; err->op = "loader"; err->detail = dir; }
            return false;
        }
        size_t s = 0;
        for (size_t i = 0; i <= raw.size(); i++)
            if (i == raw.size() || raw[i] == '\n') { if (i > s) toks.push_back(raw.substr(s, i - s)); s = i + 1; }
        *tokens = toks;
        return true;
    }
    bool LoadMerges(std::unordered_map<std::string, int>* merges, Error* err) override {
        std::string file, raw;
        if (fail_merges || !ReadFile(dir + "/merges_binary.txt", &file) || !unb64(file, &raw)) {
            if (err) { err->kind = ErrKind::DataNotFound; err->op = "loader merges"; err->detail = dir; }
            return false;
        }
        std::vector<uint32_t> ids;
        uint64_t acc = 0;
        int n = 0;
        for (unsigned char c : raw) {
            acc = (acc << 8) | c;
            n += 8;
            if (n >= 17) { ids.push_back((uint32_t)((acc >> (n - 17)) & 0x1FFFF)); n -= 17; }
        }
        int rank = 1;
        for (size_t k = 0; k + 1 < ids.size(); k += 2) {
            if (ids[k] >= toks.size() || ids[k + 1] >= toks.size()) continue;
            (*merges)[toks[ids[k]] + " " + toks[ids[k + 1]]] = rank;
            rank += rank_step;
        }
        return true;
    }
};
struct VecWriter : Writer {
    std::vector<uint8_t> v;
    bool Write(const uint8_t* p, size_t n) override { v.insert(v.end(), p, p + n); return true; }
};
}  // namespace

extern "C" {

// chunk ends (cumulative byte offsets) the scanner would encode and whether the scanner knew it was at EOF (an empty
// chunk can only come last: the empty read at EOF); returns the count, -1 on a reader error
long long l3h_chunks(const uint8_t* data, long long n, const int* sizes, int n_sizes, int buf_size, int max_buffer,
                     long long* ends, uint8_t* lasts, long long cap) {
    ControlledReader r(data, (size_t)n, sizes, n_sizes);
    scanner::Options o;
    if (buf_size > 0) o.bufSize = buf_size;
    if (max_buffer > 0) o.maxBuffer = max_buffer;
    scanner::Chunker ch(&r, o);
    std::string chunk;
    bool last = false;
    Error e;
    long long k = 0, total = 0;
    while (ch.Next(&chunk, &last, &e)) {
        total += (long long)chunk.size();
        if (k < cap) { ends[k] = total; lasts[k] = last; }
        k++;
        if (last) break;
    }
    return e ? -1 : k;
}

void* l3h_new(int device, const char* data_dir, int* err_kind) {
    Error e;
    std::vector<Option> opts{WithDevice(device)};
    if (data_dir) opts.push_back(WithDataFiles(std::string(data_dir) + "/vocab_base64.txt", std::string(data_dir) + "/merges_binary.txt"));
    std::unique_ptr<Tokenizer> t = New(opts, &e);
    if (err_kind) *err_kind = (int)e.kind;
    return t.release();
}
// New(WithSpecialTokens(list)): names = n NUL-terminated strings
void* l3h_new_special(int device, const char* data_dir, const char* const* names, int n, int* err_kind) {
    Error e;
    std::vector<Option> opts{WithDevice(device)};
    if (data_dir) opts.push_back(WithDataFiles(std::string(data_dir) + "/vocab_base64.txt", std::string(data_dir) + "/merges_binary.txt"));
    std::vector<std::string> list;
    for (int i = 0; i < n; i++) list.push_back(names[i]);
    opts.push_back(WithSpecialTokens(list));
    std::unique_ptr<Tokenizer> t = New(opts, &e);
    if (err_kind) *err_kind = (int)e.kind;
    return t.release();
}
// New(WithDevices(devices))
void* l3h_new_devices(const int* devices, int n, const char* data_dir, int* err_kind) {
    Error e;
    std::vector<Option> opts{WithDevices(std::vector<int>(devices, devices + n))};
    if (data_dir) opts.push_back(WithDataFiles(std::string(data_dir) + "/vocab_base64.txt", std::string(data_dir) + "/merges_binary.txt"));
    std::unique_ptr<Tokenizer> t = New(opts, &e);
    if (err_kind) *err_kind = (int)e.kind;
    return t.release();
}
// New(WithDataLoader(loader)) with a loader over the files in data_dir; null_loader: WithDataLoader(nil)
void* l3h_new_loader(int device, const char* data_dir, int rank_step, int fail_merges, int null_loader, int* err_kind) {
    Error e;
    FileBackedLoader loader(data_dir, rank_step, fail_merges != 0);
    std::unique_ptr<Tokenizer> t = New({WithDevice(device), WithDataLoader(null_loader ? nullptr : &loader)}, &e);
    if (err_kind) *err_kind = (int)e.kind;
    return t.release();
}
// PackLoaderData on that loader (host arithmetic only): the two files it produces; returns 0, or the error kind
int l3h_pack_loader(const char* data_dir, int rank_step, char* vocab, long long vocab_cap, long long* vocab_len, char* merges,
                    long long merges_cap, long long* merges_len) {
    Error e;
    FileBackedLoader loader(data_dir, rank_step, false);
    std::string v, m;
    if (!PackLoaderData(&loader, &v, &m, &e)) return (int)e.kind ? (int)e.kind : -1;
    *vocab_len = (long long)v.size();
    *merges_len = (long long)m.size();
    if ((long long)v.size() <= vocab_cap) memcpy(vocab, v.data(), v.size());
    if ((long long)m.size() <= merges_cap) memcpy(merges, m.data(), m.size());
    return 0;
}
void l3h_free(void* h) { delete (Tokenizer*)h; }

static long long copy_out(const std::vector<int>& v, int* out, long long cap) {
    for (size_t i = 0; i < v.size() && (long long)i < cap; i++) out[i] = v[i];
    return (long long)v.size();
}
// opts_mode: 0 = nil options, 1 = use bos/eos
long long l3h_encode(void* h, const char* text, long long n, int opts_mode, int bos, int eos, int* out, long long cap) {
    EncodeOptions o{bos != 0, eos != 0};
    return copy_out(((Tokenizer*)h)->Encode(std::string(text, (size_t)n), opts_mode ? &o : nullptr), out, cap);
}
long long l3h_append(void* h, const int* dst, long long n_dst, const char* text, long long n, int bos, int eos, int* out, long long cap) {
    EncodeOptions o{bos != 0, eos != 0};
    std::vector<int> v(dst, dst + n_dst);
    ((Tokenizer*)h)->AppendTokens(v, std::string(text, (size_t)n), &o);
    return copy_out(v, out, cap);
}
long long l3h_decode(void* h, const int* ids, long long n, char* out, long long cap) {
    std::string s = ((Tokenizer*)h)->Decode(std::vector<int>(ids, ids + n));
    memcpy(out, s.data(), s.size() < (size_t)cap ? s.size() : (size_t)cap);
    return (long long)s.size();
}
// BPE.EncodeBPE: `pretoken` in the byte-level alphabet (UTF-8)
long long l3h_encode_bpe(void* h, const char* pretoken, long long n, int* out, long long cap) {
    return copy_out(((Tokenizer*)h)->EncodeBPE(std::string(pretoken, (size_t)n)), out, cap);
}
int l3h_special(void* h, const char* name, int* err_kind) {
    Error e;
    int id = ((Tokenizer*)h)->GetSpecialTokenID(name, &e);
    *err_kind = (int)e.kind;
    return id;
}
int l3h_vocab_size(void* h) { return ((Tokenizer*)h)->VocabSize(); }
int l3h_optimistic_count(void* h, const char* text, long long n) { return ((Tokenizer*)h)->OptimisticCount(std::string(text, (size_t)n)); }
// pieces concatenated into buf with their cumulative ends; raw = before the byte-level mapping
long long l3h_pretokenize(void* h, const char* text, long long n, int raw, uint8_t* buf, long long buf_cap, long long* ends,
                          long long cap) {
    Tokenizer* t = (Tokenizer*)h;
    std::vector<std::string> p = raw ? t->PreTokenizeRaw(std::string(text, (size_t)n)) : t->PreTokenize(std::string(text, (size_t)n));
    long long tot = 0;
    for (size_t i = 0; i < p.size(); i++) {
        if (tot + (long long)p[i].size() <= buf_cap) memcpy(buf + tot, p[i].data(), p[i].size());
        tot += (long long)p[i].size();
        if ((long long)i < cap) ends[i] = tot;
    }
    return (long long)p.size();
}
// Scanner over a controlled reader; returns the token count, -1 on a reader error
long long l3h_scan(void* h, const uint8_t* data, long long n, const int* sizes, int n_sizes, int buf_size, int max_buffer,
                   int bos, int eos, int* out, long long cap) {
    ControlledReader r(data, (size_t)n, sizes, n_sizes);
    scanner::Options o;
    if (buf_size > 0) o.bufSize = buf_size;
    if (max_buffer > 0) o.maxBuffer = max_buffer;
    o.enc.BOS = bos != 0; o.enc.EOS = eos != 0;
    scanner::Scanner sc = ((Tokenizer*)h)->NewScanner(&r, o);
    long long k = 0;
    while (sc.Scan()) { if (k < cap) out[k] = sc.Token(); k++; }
    return sc.Err() ? -1 : k;
}
long long l3h_process(void* h, const uint8_t* data, long long n, int bos, int eos, uint8_t* out, long long cap) {
    BytesReader r(data, (size_t)n);
    VecWriter w;
    scanner::Options o;
    o.enc.BOS = bos != 0; o.enc.EOS = eos != 0;
    long long cnt = ((Tokenizer*)h)->Process(&r, &w, o);
    memcpy(out, w.v.data(), w.v.size() < (size_t)cap ? w.v.size() : (size_t)cap);
    return cnt;
}
// TokenStream over a controlled reader: the consumer takes `take` tokens (all when negative) and then drops the
// stream; returns the count taken, -1 on a reader error seen after the channel closed
long long l3h_token_stream(void* h, const uint8_t* data, long long n, const int* sizes, int n_sizes, long long take, int* out,
                           long long cap) {
    ControlledReader r(data, (size_t)n, sizes, n_sizes);
    std::unique_ptr<TokenChannel> ch = ((Tokenizer*)h)->TokenStream(&r);
    long long k = 0;
    int tok = 0;
    while ((take < 0 || k < take) && ch->Next(&tok)) { if (k < cap) out[k] = tok; k++; }
    if (take >= 0 && k == take) return k;  // dropped early: the destructor unblocks and joins the producer
    return ch->Err() ? -1 : k;
}
// many streams through one batch call; out = concatenated ids, out_off = n_streams + 1 offsets
long long l3h_streams(void* h, const uint8_t* data, const long long* off, int n_streams, int buf_size, int bos, int eos,
                      int* out, long long cap, long long* out_off) {
    std::vector<BytesReader> rs;
    rs.reserve((size_t)n_streams);
    for (int i = 0; i < n_streams; i++) rs.emplace_back(data + off[i], (size_t)(off[i + 1] - off[i]));
    std::vector<Reader*> ps;
    for (auto& r : rs) ps.push_back(&r);
    scanner::Options o;
    if (buf_size > 0) o.bufSize = buf_size;
    o.enc.BOS = bos != 0; o.enc.EOS = eos != 0;
    std::vector<std::vector<int>> res;
    if (Error e = EncodeStreams(*(Tokenizer*)h, ps, o, &res)) {
        fprintf(stderr, "l3h_streams: %s\n", e.Message().c_str());
        return -1;
    }
    long long k = 0;
    out_off[0] = 0;
    for (int i = 0; i < n_streams; i++) {
        for (int v : res[(size_t)i]) { if (k < cap) out[k] = v; k++; }
        out_off[i + 1] = k;
    }
    return k;
}

}  // extern "C"
