// llama3.hpp -- host-side mirror of the reference's Go API for the encode path, in C++ (the reference is
// compiled Go; no Go toolchain exists in the authoring image, see INTEGRATION.md for the cgo stub).
// Header-only; everything goes through the C ABI of include/l3hip.h, so this file is exactly what the Go
// shim would be, with the same names, argument meanings and error behaviour:
//
//   llama3::New(opts...)                      llama3/tokenizer.go:142-213, options.go:17-74
//   Tokenizer::Encode / EncodeBytes           tokenizer.go:217-271  (nil options = BOS and EOS)
//   Tokenizer::AppendTokens                   tokenizer.go:276-330
//   Tokenizer::Decode / DecodeBytes           tokenizer.go:333-354
//   Tokenizer::GetSpecialTokenID              tokenizer.go:357-368  (ErrInvalidToken / ErrTokenNotFound)
//   Tokenizer::OptimisticCount                tokenizer.go:370-417
//   Tokenizer::VocabSize / PreTokenize        tokenizer.go:420-422,490-504
//   Tokenizer::EncodeBatch                    new, additive: the batched hot path
//   WithDevices                               new, additive: EncodeBatch shards its documents over several GPUs (l3_multi_*)
//   scanner::Tokenizer, scanner::New/NewWithOptions, Scanner::Scan/Token/Text/Err
//                                             llama3/scanner/scanner.go:12-111,241-333 (chunking rules :124-181,337-478)
//   Tokenizer::NewScanner / Process           llama3/scanner.go:67-102 (Process writes uint32 little-endian)
//   Tokenizer::TokenStream                    llama3/scanner.go:107-132 (a bounded channel of 100 fed by a thread)
//   WithDataLoader / VocabularyDataLoader     options.go:52-60, vocab.go:14-22 (tokens + merge map from any source; the shim
//                                             packs them into the two file formats the C ABI takes)
//   WithSpecialTokens                         options.go:17-37 (the list replaces the 256 defaults; ConfigError on a
//                                             malformed or duplicated name)
//   EncodeStreams                             new, additive: many streams, the scanner's chunks, ONE batch call
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "l3hip.h"

namespace llama3 {

// ---- errors (llama3/errors.go) ------------------------------------------------------------------------
enum class ErrKind { None, DataNotFound, InvalidToken, TokenNotFound, Device, Reader, Config };
struct Error {
    ErrKind kind = ErrKind::None;
    std::string op, detail;
    explicit operator bool() const { return kind != ErrKind::None; }
    std::string Message() const { return op + ": " + detail; }
};

struct EncodeOptions {  // tokenizer.go:107-112
    bool BOS = true;
    bool EOS = true;
};

// llama3/vocab.go:14-22: a source of vocabulary data.  LoadVocabulary: the token strings in id order, in the byte-level
// alphabet the reference's vocabulary file uses (internal/encoding/encoding.go:26-79).  LoadMerges: merge identifier
// "left right" (tokenizer.go:425-427) -> priority, smaller = earlier.  Either returns false (and may fill *err) on failure.
struct VocabularyDataLoader {
    virtual bool LoadVocabulary(std::vector<std::string>* tokens, Error* err) = 0;
    virtual bool LoadMerges(std::unordered_map<std::string, int>* merges, Error* err) = 0;
    virtual ~VocabularyDataLoader() {}
};

// ---- options (llama3/options.go) ----------------------------------------------------------------------
struct Config {
    VocabularyDataLoader* dataLoader = nullptr;  // WithDataLoader (not owned; used inside New only)
    std::string vocabPath, mergesPath;  // WithDataFiles
    int device = 0;                     // new: which GPU holds the tables
    std::vector<int> devices;           // new: WithDevices -- batches are sharded over these GPUs
    bool haveSpecials = false;          // WithSpecialTokens with a non-nil list (an empty one counts: no special tokens)
    std::vector<std::string> specialTokens;
    Error err;                          // the first option that failed (New returns it, tokenizer.go:151-155)
};
typedef void (*OptionFn)(Config&, const void*);
struct Option {
    std::string a, b;
    int i = 0;
    int which = 0;
    std::vector<std::string> list;
    std::vector<int> ints;
    VocabularyDataLoader* loader = nullptr;
    void Apply(Config& c) const {
        if (which == 4) c.devices = ints;
        if (which == 5) {  // options.go:52-60
            if (!loader) { c.err.kind = ErrKind::Config; c.err.op = "data_loader"; c.err.detail = "nil"; return; }
            c.dataLoader = loader;
            c.vocabPath.clear();
            c.mergesPath.clear();
        }
        if (which == 1) { c.vocabPath = a; c.mergesPath = b; c.dataLoader = nullptr; }  // (the last data option wins, as in Go)
        if (which == 2) c.device = i;
        if (which == 3) {  // options.go:19-33: each name is <|...|> of at least 5 bytes, and none appears twice
            for (size_t k = 0; k < list.size(); k++) {
                const std::string& t = list[k];
                const bool shaped = t.size() >= 5 && t.compare(0, 2, "<|") == 0 && t.compare(t.size() - 2, 2, "|>") == 0;
                bool dup = false;
                for (size_t j = k + 1; j < list.size() && !dup; j++) dup = list[j] == t;
                if ((!shaped || dup) && !c.err) {
                    c.err.kind = ErrKind::Config;
                    c.err.op = "special_tokens";
                    c.err.detail = std::string(shaped ? "duplicate: " : "validate: ") + t;
                    return;
                }
            }
            c.haveSpecials = true;
            c.specialTokens = list;
        }
    }
};
inline Option WithDataFiles(const std::string& vocabPath, const std::string& mergesPath) {
    Option o; o.which = 1; o.a = vocabPath; o.b = mergesPath; return o;
}
inline Option WithDevice(int device) { Option o; o.which = 2; o.i = device; return o; }
// options.go:52-60: vocabulary and merges from a caller-supplied source.  The loader is called inside New() and not kept.
inline Option WithDataLoader(VocabularyDataLoader* loader) { Option o; o.which = 5; o.loader = loader; return o; }
// One context per entry; EncodeBatch (and everything built on it) cuts a batch into byte-balanced ranges of documents, one per
// GPU.  Single-document calls and Decode use the first device.
inline Option WithDevices(const std::vector<int>& devices) { Option o; o.which = 4; o.ints = devices; return o; }
// options.go:17-37.  Go's nil slice (keep the defaults) is "do not pass the option" here.
inline Option WithSpecialTokens(const std::vector<std::string>& tokens) { Option o; o.which = 3; o.list = tokens; return o; }

// ---- io: the two interfaces the streaming surface needs (Go's io.Reader / io.Writer) -------------------
struct Reader {
    // Reads up to n bytes into p.  Returns the byte count; *eof is set at end of input (possibly together with
    // bytes); a negative return is a read error.
    virtual int Read(uint8_t* p, int n, bool* eof) = 0;
    virtual ~Reader() {}
};
struct Writer {
    virtual bool Write(const uint8_t* p, size_t n) = 0;
    virtual ~Writer() {}
};
struct BytesReader : Reader {  // bytes.NewReader
    const uint8_t* d; size_t n, pos = 0;
    BytesReader(const void* data, size_t len) : d((const uint8_t*)data), n(len) {}
    int Read(uint8_t* p, int k, bool* eof) override {
        if (pos >= n) { *eof = true; return 0; }  // like bytes.Reader: EOF comes on the read after the last byte
        size_t m = n - pos < (size_t)k ? n - pos : (size_t)k;
        memcpy(p, d + pos, m);
        pos += m;
        return (int)m;
    }
};

namespace scanner {

// llama3/scanner/scanner.go:12-15 -- the seam a tokenizer has to fill for the whole streaming surface
struct Tokenizer {
    virtual std::vector<int> Encode(const std::string& text, const EncodeOptions* opts) = 0;
    virtual int GetSpecialTokenID(const std::string& token, Error* err) = 0;
    virtual ~Tokenizer() {}
};

struct Options {  // scanner.go:55-85,93-101
    int bufSize = 4096;
    int maxBuffer = 1024 * 1024;
    EncodeOptions enc{false, false};
};

// The reader / accumulate half of the reference scanner (scanner.go:124-181,281-311,337-478): yields the
// successive chunks the reference would hand to Encode.  Kept separate from the encode call so that chunks of
// many streams can be encoded by one batch call.
class Chunker {
  public:
    Chunker(Reader* r, const Options& o) : r_(r), o_(o) {}
    // Next chunk into *out.  Returns false when the input is exhausted and no chunk is left, or on a reader error;
    // *last is set on the chunk after which the input is known to be at EOF.  A chunk is empty at the empty read
    // at EOF, and (a quirk of the reference kept as is) when maxBuffer is smaller than one UTF-8 sequence: the
    // reference's Scan then returns false without being finished (scanner.go:208-212,271-277).
    bool Next(std::string* out, bool* last, Error* err) {
        out->clear();
        *last = false;
        if (failed_) return false;
        if (done_ && text_.empty()) return false;
        for (;;) {
            bool got_err = false;
            readMoreData(&got_err);
            if ((int)text_.size() >= o_.maxBuffer) { handleMaxBufferReached(); break; }
            if (got_err) {
                failed_ = true;
                if (err) { err->kind = ErrKind::Reader; err->op = "scan"; err->detail = "read error at offset " + std::to_string(text_.size()); }
                return false;
            }
            if (done_) break;
            if (hasTokenizationBoundary()) break;
        }
        out->swap(text_);
        text_.clear();
        *last = done_;
        return true;
    }
    bool Done() const { return done_ && text_.empty(); }

  private:
    void readMoreData(bool* got_err) {  // scanner.go:124-149
        std::vector<uint8_t> buf((size_t)o_.bufSize);
        bool eof = false;
        int n = r_->Read(buf.data(), o_.bufSize, &eof);
        if (n > 0) {
            std::string toWrite;
            if (!pending_.empty()) { toWrite.swap(pending_); pending_.clear(); }
            toWrite.append((const char*)buf.data(), (size_t)n);
            handleBufferLimit(&toWrite);
            text_ += toWrite;
        }
        if (eof) {
            done_ = true;
            if (!pending_.empty()) { text_ += pending_; pending_.clear(); }
            return;
        }
        if (n < 0) *got_err = true;
    }
    void handleBufferLimit(std::string* toWrite) {  // scanner.go:153-166
        if ((int)(text_.size() + toWrite->size()) > o_.maxBuffer) {
            int maxWrite = o_.maxBuffer - (int)text_.size();
            if (maxWrite > 0 && maxWrite < (int)toWrite->size()) {
                int upTo = findUTF8Boundary(*toWrite, maxWrite);
                if (upTo < (int)toWrite->size()) {
                    pending_ = toWrite->substr((size_t)upTo);
                    toWrite->resize((size_t)upTo);
                }
            }
        }
    }
    void handleMaxBufferReached() {  // scanner.go:169-181
        if ((int)text_.size() == o_.maxBuffer && pending_.empty()) {
            if (!text_.empty() && !isValidUTF8Ending(text_)) {
                int splitAt = findLastCompleteUTF8(text_);
                if (splitAt < (int)text_.size()) {
                    pending_ = text_.substr((size_t)splitAt);
                    text_.resize((size_t)splitAt);
                }
            }
        }
    }
    bool hasTokenizationBoundary() const {  // scanner.go:337-367
        if (text_.empty()) return false;
        uint8_t last = (uint8_t)text_.back();
        if (last == ' ' || last == '\n' || last == '\t' || last == '\r') return true;
        if ((last & 0xC0) == 0x80) return false;
        return (int)text_.size() > o_.bufSize / 2;
    }
    static bool isValidUTF8Ending(const std::string& b) {  // scanner.go:370-403
        if (b.empty()) return true;
        uint8_t last = (uint8_t)b.back();
        if (last < 0x80) return true;
        if ((last & 0xC0) == 0x80) return false;
        int expected = 0;
        if ((last & 0xE0) == 0xC0) expected = 2;
        else if ((last & 0xF0) == 0xE0) expected = 3;
        else if ((last & 0xF8) == 0xF0) expected = 4;
        return expected <= 1;
    }
    static int findLastCompleteUTF8(const std::string& b) {  // scanner.go:406-438
        int len = (int)b.size();
        for (int i = len - 1; i >= 0 && i >= len - 4; i--) {
            uint8_t c = (uint8_t)b[(size_t)i];
            if (c < 0x80) return i + 1;
            if ((c & 0xC0) != 0x80) {
                int seq = 0;
                if ((c & 0xE0) == 0xC0) seq = 2;
                else if ((c & 0xF0) == 0xE0) seq = 3;
                else if ((c & 0xF8) == 0xF0) seq = 4;
                return i + seq <= len ? i + seq : i;
            }
        }
        return len;
    }
    static int findUTF8Boundary(const std::string& d, int maxBytes) {  // scanner.go:443-478
        int len = (int)d.size();
        if (maxBytes >= len) return len;
        for (int i = maxBytes; i > 0 && i > maxBytes - 4; i--) {
            if (i >= len) continue;
            uint8_t b = (uint8_t)d[(size_t)i];
            if (b < 0x80 || (b & 0xC0) != 0x80) return i;
        }
        if (maxBytes < len && (((uint8_t)d[(size_t)maxBytes]) & 0xC0) == 0x80) {
            for (int i = maxBytes - 1; i >= 0 && i >= maxBytes - 4; i--)
                if ((((uint8_t)d[(size_t)i]) & 0xC0) != 0x80) return i;
        }
        return maxBytes;
    }

    Reader* r_;
    Options o_;
    std::string text_, pending_;
    bool done_ = false, failed_ = false;
};

// scanner.go:24-29,241-333
class Scanner {
  public:
    Scanner(Tokenizer* t, Reader* r, const Options& o = Options()) : t_(t), ch_(r, o), o_(o) {}
    bool Scan() {
        if (err_) return false;
        if (idx_ < tokens_.size()) { idx_++; return true; }
        if (finished_) return false;
        tokens_.clear();
        idx_ = 0;
        std::string chunk;
        bool last = false;
        bool have = ch_.Next(&chunk, &last, &err_);
        if (err_) return false;
        if (!have) { finished_ = true; return false; }
        if (chunk.empty() && !last) return false;  // nothing to tokenize, not at EOF: scanner.go:208-212,277
        if (chunk.empty()) {  // empty input at EOF: scanner.go:184-205
            finished_ = true;
            Error e, e2;
            if (o_.enc.BOS && !sentBOS_) { int id = t_->GetSpecialTokenID("<|begin_of_text|>", &e); if (!e) { tokens_.push_back(id); sentBOS_ = true; } }
            if (o_.enc.EOS) { int id = t_->GetSpecialTokenID("<|end_of_text|>", &e2); if (!e2) tokens_.push_back(id); }
            if (tokens_.empty()) return false;
            idx_ = 1;
            return true;
        }
        EncodeOptions co;  // scanner.go:214-224: BOS only on the first chunk, EOS handled at the end
        co.BOS = o_.enc.BOS && !sentBOS_;
        co.EOS = false;
        if (co.BOS) sentBOS_ = true;
        tokens_ = t_->Encode(chunk, &co);
        if (last) {
            finished_ = true;
            if (o_.enc.EOS) { Error e; int id = t_->GetSpecialTokenID("<|end_of_text|>", &e); if (!e) tokens_.push_back(id); }
        }
        if (tokens_.empty()) return false;
        idx_ = 1;
        return true;
    }
    int Token() const { return idx_ > 0 && idx_ <= tokens_.size() ? tokens_[idx_ - 1] : 0; }
    std::string Text() const { return std::string(); }  // the reference never assigns lastText (scanner.go:40,323-325)
    Error Err() const { return err_; }

  private:
    Tokenizer* t_;
    Chunker ch_;
    Options o_;
    std::vector<int> tokens_;
    size_t idx_ = 0;
    bool sentBOS_ = false, finished_ = false;
    Error err_;
};

}  // namespace scanner

// llama3/scanner.go:107-132 -- the pair of channels TokenStream returns, as one object: a bounded queue of 100
// tokens filled by a thread that runs the scanner; Next() is `tok, ok := <-tokens`, Err() is `<-errc` after the
// token channel has closed.  The destructor drains and joins, so dropping a stream early is safe.
class TokenChannel {
  public:
    TokenChannel(scanner::Tokenizer* t, Reader* r) : th_([this, t, r] { run(t, r); }) {}
    ~TokenChannel() {
        { std::lock_guard<std::mutex> lk(mu_); abandoned_ = true; }
        not_full_.notify_all();
        if (th_.joinable()) th_.join();
    }
    TokenChannel(const TokenChannel&) = delete;
    TokenChannel& operator=(const TokenChannel&) = delete;
    // Blocks until a token is there (true) or the channel is closed and empty (false).
    bool Next(int* tok) {
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [this] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        *tok = q_.front();
        q_.pop_front();
        lk.unlock();
        not_full_.notify_one();
        return true;
    }
    // The scanner's error, once Next() has returned false (an empty Error when there was none).
    Error Err() {
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [this] { return closed_; });
        return err_;
    }
    static constexpr size_t kCapacity = 100;  // make(chan int, 100)

  private:
    void run(scanner::Tokenizer* t, Reader* r) {
        scanner::Scanner sc(t, r);
        while (sc.Scan()) {
            std::unique_lock<std::mutex> lk(mu_);
            not_full_.wait(lk, [this] { return q_.size() < kCapacity || abandoned_; });  // a full channel blocks the sender
            if (abandoned_) break;
            q_.push_back(sc.Token());
            lk.unlock();
            not_empty_.notify_one();
        }
        { std::lock_guard<std::mutex> lk(mu_); err_ = sc.Err(); closed_ = true; }
        not_empty_.notify_all();
    }
    std::mutex mu_;
    std::condition_variable not_empty_, not_full_;
    std::deque<int> q_;
    bool closed_ = false, abandoned_ = false;
    Error err_;
    std::thread th_;  // last member: starts after everything it uses exists
};

// ---- the tokenizer ------------------------------------------------------------------------------------
class Tokenizer : public scanner::Tokenizer {
  public:
    ~Tokenizer() override {
        if (multi_) l3_multi_destroy(multi_);   // owns every context, ctx_ included
        else if (ctx_) l3_destroy(ctx_);
    }
    Tokenizer(const Tokenizer&) = delete;
    Tokenizer& operator=(const Tokenizer&) = delete;

    // Encode (tokenizer.go:217-265): cannot fail in the reference; a device failure yields an empty result and
    // is reported by LastError().
    std::vector<int> Encode(const std::string& text, const EncodeOptions* opts) override {
        return EncodeBytes((const uint8_t*)text.data(), text.size(), opts);
    }
    std::vector<int> EncodeBytes(const uint8_t* data, size_t n, const EncodeOptions* opts) {
        std::vector<int> out;
        AppendBytes(&out, data, n, opts);
        return out;
    }
    // AppendTokens (tokenizer.go:276-330): Encode(text) appended to dst
    std::vector<int>& AppendTokens(std::vector<int>& dst, const std::string& text, const EncodeOptions* opts) {
        AppendBytes(&dst, (const uint8_t*)text.data(), text.size(), opts);
        return dst;
    }
    // The batched hot path: documents [off[i], off[i+1]) of `text` -> ids + per-document offsets.
    Error EncodeBatch(const uint8_t* text, const std::vector<uint64_t>& off, const EncodeOptions* opts,
                      std::vector<int32_t>* ids, std::vector<uint64_t>* id_off) {
        Error e;
        const uint64_t n_docs = off.size() - 1, n_bytes = off.back();
        ids->resize((size_t)(n_bytes + 2 * n_docs + 64));
        id_off->resize((size_t)n_docs + 1);
        for (;;) {
            uint64_t n = 0;
            int rc = multi_ ? l3_multi_encode_batch(multi_, text, off.data(), n_docs, flagsOf(opts), ids->data(), ids->size(), id_off->data(), &n)
                            : l3_encode_batch(ctx_, text, off.data(), n_docs, flagsOf(opts), ids->data(), ids->size(), id_off->data(), &n);
            if (rc == L3_E_OUTPUT_CAP) { ids->resize((size_t)n); continue; }
            if (rc != L3_OK) {
                e.kind = ErrKind::Device; e.op = "encode"; e.detail = multi_ ? l3_multi_last_error(multi_) : l3_last_error(ctx_);
                last_ = e; ids->clear(); return e;
            }
            ids->resize((size_t)n);
            return e;
        }
    }
    std::string Decode(const std::vector<int>& ids) { return DecodeBytes(ids); }  // tokenizer.go:333-337
    std::string DecodeBytes(const std::vector<int>& ids) {                         // tokenizer.go:339-354
        std::vector<int32_t> v(ids.size());
        for (size_t i = 0; i < ids.size(); i++)
            v[i] = ids[i] < 0 ? -1 : (int32_t)ids[i];  // out of range: skipped
        uint64_t off[2] = {0, v.size()}, ooff[2], n = 0;
        std::string out(v.size() * 8 + 64, '\0');
        for (;;) {
            int rc = l3_decode_batch(ctx_, v.data(), off, 1, (uint8_t*)&out[0], out.size(), ooff, &n);
            if (rc == L3_E_OUTPUT_CAP) { out.resize((size_t)n); continue; }
            if (rc != L3_OK) { last_.kind = ErrKind::Device; last_.op = "decode"; last_.detail = l3_last_error(ctx_); return std::string(); }
            out.resize((size_t)n);
            return out;
        }
    }
    int GetSpecialTokenID(const std::string& token, Error* err) override {  // tokenizer.go:357-368
        int32_t id = 0;
        int rc = l3_special_token_id(ctx_, token.data(), token.size(), &id);
        if (rc != L3_OK) {
            if (err) {
                const bool shaped = token.size() >= 2 && token.compare(0, 2, "<|") == 0 && token.compare(token.size() - 2, 2, "|>") == 0;
                err->kind = shaped ? ErrKind::TokenNotFound : ErrKind::InvalidToken;
                err->op = "get special token";
                err->detail = token;
            }
            return 0;
        }
        return id;
    }
    // OptimisticCount (tokenizer.go:370-417): the token count when anything of the shape <|[a-zA-Z0-9_]+|>
    // (internal/tokens/special.go:15) counts as ONE token, plus BOS and EOS.  The pieces between those matches are
    // pre-tokenized and merged as plain text -- one batch call, a document per piece, L3_NO_SPECIAL.  -1 on a device
    // failure (LastError() says which).
    int OptimisticCount(const std::string& text) {
        std::vector<uint64_t> off{0};
        std::string plain;
        size_t matches = 0, piece_start = 0, i = 0;
        const size_t n = text.size();
        auto word = [](unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; };
        while (i + 1 < n) {
            if (text[i] == '<' && text[i + 1] == '|') {
                size_t j = i + 2;
                while (j < n && word((unsigned char)text[j])) j++;
                if (j > i + 2 && j + 1 < n && text[j] == '|' && text[j + 1] == '>') {   // leftmost match, then on after it
                    if (i > piece_start) { plain.append(text, piece_start, i - piece_start); off.push_back(plain.size()); }
                    matches++;
                    i = j + 2;
                    piece_start = i;
                    continue;
                }
            }
            i++;
        }
        if (n > piece_start) { plain.append(text, piece_start, n - piece_start); off.push_back(plain.size()); }
        uint64_t n_tok = 0;
        if (off.size() > 1) {
            std::vector<int32_t> ids(plain.size() + 64);
            std::vector<uint64_t> id_off(off.size());
            for (;;) {
                int rc = l3_encode_batch(ctx_, (const uint8_t*)plain.data(), off.data(), off.size() - 1, L3_NO_SPECIAL, ids.data(),
                                         ids.size(), id_off.data(), &n_tok);
                if (rc == L3_E_OUTPUT_CAP) { ids.resize((size_t)n_tok); continue; }
                if (rc != L3_OK) { last_.kind = ErrKind::Device; last_.op = "optimistic count"; last_.detail = l3_last_error(ctx_); return -1; }
                break;
            }
        }
        Error no_bos, no_eos;  // counted only when the tokenizer has them (tokenizer.go:378-380,412-414)
        GetSpecialTokenID("<|begin_of_text|>", &no_bos);
        GetSpecialTokenID("<|end_of_text|>", &no_eos);
        return (int)(n_tok + matches + (no_bos ? 0 : 1) + (no_eos ? 0 : 1));
    }
    int VocabSize() const { return l3_vocab_size(ctx_); }  // tokenizer.go:420-422
    // PreTokenize (tokenizer.go:490-506): the pieces of pretokenizer.Tokenize(text) -- which knows no special
    // tokens -- each in the byte-level alphabet of internal/encoding/encoding.go:26-79 (every byte one rune: printable
    // bytes themselves, the other 68 as U+0100...).  As in Go, text passes through []rune first, so every byte
    // that is not part of a valid UTF-8 sequence has become U+FFFD (EF BF BD) before the mapping.
    std::vector<std::string> PreTokenize(const std::string& text) {
        std::vector<std::string> out;
        for (const std::string& piece : PreTokenizeRaw(text)) out.push_back(ByteLevel(piece));
        return out;
    }
    // the same pieces before the byte-level mapping
    std::vector<std::string> PreTokenizeRaw(const std::string& text) {
        std::vector<std::string> out;
        if (text.empty()) return out;
        std::vector<uint8_t> kinds(text.size());
        uint64_t off[2] = {0, text.size()};
        if (l3_pretokenize_batch(ctx_, (const uint8_t*)text.data(), off, 1, L3_NO_SPECIAL, kinds.data()) != L3_OK) {
            last_.kind = ErrKind::Device; last_.op = "pretokenize"; last_.detail = l3_last_error(ctx_);
            return out;
        }
        std::string cur;
        for (size_t i = 0; i < text.size();) {
            if (i && kinds[i]) { out.push_back(cur); cur.clear(); }
            int w = ValidSeqLen((const uint8_t*)text.data() + i, text.size() - i);
            if (w) { cur.append(text, i, (size_t)w); i += (size_t)w; }
            else { cur += "\xEF\xBF\xBD"; i++; }
        }
        out.push_back(cur);
        return out;
    }
    // Length of the valid UTF-8 sequence at p (Go utf8.DecodeRune: shortest form, no surrogates, <= U+10FFFF), 0 if none
    static int ValidSeqLen(const uint8_t* p, size_t n) {
        const uint8_t c = p[0];
        if (c < 0x80) return 1;
        if (c < 0xC2 || c > 0xF4) return 0;
        const int w = c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
        if ((size_t)w > n) return 0;
        uint8_t lo = 0x80, hi = 0xBF;
        if (c == 0xE0) lo = 0xA0; else if (c == 0xED) hi = 0x9F; else if (c == 0xF0) lo = 0x90; else if (c == 0xF4) hi = 0x8F;
        if (p[1] < lo || p[1] > hi) return 0;
        for (int k = 2; k < w; k++) if ((p[k] & 0xC0) != 0x80) return 0;
        return w;
    }
    static std::string ByteLevel(const std::string& raw) {  // encoding.go:26-79, EncodeBytes
        std::string out;
        for (unsigned char b : raw) {
            unsigned cp;
            if ((b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || b >= 0xAE) cp = b;
            else cp = 256u + (b <= 0x20 ? b : b == 0x7F ? 33u : b <= 0xA0 ? 34u + (b - 0x80u) : 67u);  // the 68 others in byte order
            if (cp < 0x80) out += (char)cp;
            else { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        }
        return out;
    }
    // EncodeBPE (the BPE interface, tokenizer.go:80-84,478-488 -> bpe.go:18-71): the tokens of ONE pre-token given in the
    // byte-level alphabet (what PreTokenize returns).  The whole string is looked up first -- the reference's lookup table
    // holds the special tokens too, so EncodeBPE("<|eot_id|>") is that token --, then the alphabet is inverted and the bytes
    // go to the GPU as one pre-token (L3_ONE_PRETOKEN).  Runes outside the alphabet are skipped as bpe.go:84-92 skips them.
    // (One corner differs: when runes WERE skipped and what remains is itself a vocabulary entry, the GPU path returns that
    // entry, the reference the result of its merge loop; no string PreTokenize makes holds such runes.)
    std::vector<int> EncodeBPE(const std::string& pretoken) {
        std::vector<std::vector<int>> r = EncodeBPEBatch(std::vector<std::string>{pretoken});
        return r.empty() ? std::vector<int>() : r[0];
    }
    // the same for many pre-tokens in one call to the GPU
    std::vector<std::vector<int>> EncodeBPEBatch(const std::vector<std::string>& pretokens) {
        std::vector<std::vector<int>> out(pretokens.size());
        std::vector<uint8_t> raw;
        std::vector<uint64_t> off{0};
        std::vector<size_t> which;
        for (size_t i = 0; i < pretokens.size(); i++) {
            int32_t sid = 0;
            if (l3_special_token_id(ctx_, pretokens[i].data(), pretokens[i].size(), &sid) == L3_OK) { out[i].push_back(sid); continue; }
            const std::string b = ByteLevelInverse(pretokens[i]);
            raw.insert(raw.end(), b.begin(), b.end());
            off.push_back(raw.size());
            which.push_back(i);
        }
        if (which.empty()) return out;
        std::vector<int32_t> ids(raw.size() + 64);
        std::vector<uint64_t> id_off(off.size());
        for (;;) {
            uint64_t n = 0;
            int rc = l3_encode_batch(ctx_, raw.data(), off.data(), off.size() - 1, L3_ONE_PRETOKEN, ids.data(), ids.size(), id_off.data(), &n);
            if (rc == L3_E_OUTPUT_CAP) { ids.resize((size_t)n); continue; }
            if (rc != L3_OK) {
                last_.kind = ErrKind::Device; last_.op = "encode bpe"; last_.detail = l3_last_error(ctx_);
                return std::vector<std::vector<int>>(pretokens.size());
            }
            break;
        }
        for (size_t k = 0; k < which.size(); k++) out[which[k]].assign(ids.begin() + (long)id_off[k], ids.begin() + (long)id_off[k + 1]);
        return out;
    }
    // encoding.go:26-79 backwards: the byte of every rune of the byte-level alphabet; other runes (and bytes that are not
    // UTF-8) are skipped
    static std::string ByteLevelInverse(const std::string& s) {
        static const std::vector<uint8_t> others = [] {
            std::vector<uint8_t> v;
            for (unsigned b = 0; b < 256; b++)
                if (!((b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || b >= 0xAE)) v.push_back((uint8_t)b);
            return v;
        }();
        std::string out;
        for (size_t i = 0; i < s.size();) {
            const int w = ValidSeqLen((const uint8_t*)s.data() + i, s.size() - i);
            if (!w) { i++; continue; }
            unsigned cp = (uint8_t)s[i];
            if (w == 2) cp = ((cp & 0x1F) << 6) | ((uint8_t)s[i + 1] & 0x3F);
            else if (w > 2) cp = 0x10000;   // (beyond the alphabet, whatever it is)
            i += (size_t)w;
            if ((cp >= 0x21 && cp <= 0x7E) || (cp >= 0xA1 && cp <= 0xAC) || (cp >= 0xAE && cp <= 0xFF)) out += (char)cp;
            else if (cp >= 256 && cp < 256 + others.size()) out += (char)others[cp - 256];
        }
        return out;
    }
    // NewScanner (llama3/scanner.go:67-75)
    scanner::Scanner NewScanner(Reader* r, const scanner::Options& o = scanner::Options()) { return scanner::Scanner(this, r, o); }
    // Process (llama3/scanner.go:77-102): every token as a little-endian uint32; returns the token count or -1
    long long Process(Reader* r, Writer* w, const scanner::Options& o = scanner::Options()) {
        scanner::Scanner sc(this, r, o);
        long long count = 0;
        while (sc.Scan()) {
            uint32_t v = (uint32_t)sc.Token();
            uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
            if (!w->Write(b, 4)) return -1;
            count++;
        }
        return sc.Err() ? -1 : count;
    }
    // TokenStream (llama3/scanner.go:107-132): tokens of r through a bounded channel, produced concurrently
    std::unique_ptr<TokenChannel> TokenStream(Reader* r) { return std::unique_ptr<TokenChannel>(new TokenChannel(this, r)); }
    Error LastError() const { return last_; }
    l3_ctx* Handle() const { return ctx_; }

  private:
    friend std::unique_ptr<Tokenizer> New(const std::vector<Option>& opts, Error* err);
    Tokenizer() {}
    static uint32_t flagsOf(const EncodeOptions* o) {  // nil => BOS and EOS (tokenizer.go:218-220)
        if (!o) return L3_BOS | L3_EOS;
        return (o->BOS ? L3_BOS : 0u) | (o->EOS ? L3_EOS : 0u);
    }
    void AppendBytes(std::vector<int>* dst, const uint8_t* data, size_t n, const EncodeOptions* opts) {
        std::vector<int32_t> ids;
        std::vector<uint64_t> id_off, off{0, (uint64_t)n};
        if (EncodeBatch(data, off, opts, &ids, &id_off)) return;
        dst->insert(dst->end(), ids.begin(), ids.end());
    }
    l3_ctx* ctx_ = nullptr;
    l3_multi* multi_ = nullptr;
    Error last_;
};

inline bool ReadFile(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    out->clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}

inline std::string Base64(const std::string& in) {
    static const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((in.size() + 2) / 3 * 4);
    for (size_t i = 0; i < in.size(); i += 3) {
        const uint32_t n = (size_t)(in.size() - i) >= 3 ? 3u : (uint32_t)(in.size() - i);
        uint32_t v = (uint32_t)(uint8_t)in[i] << 16;
        if (n > 1) v |= (uint32_t)(uint8_t)in[i + 1] << 8;
        if (n > 2) v |= (uint32_t)(uint8_t)in[i + 2];
        out += A[(v >> 18) & 63]; out += A[(v >> 12) & 63];
        out += n > 1 ? A[(v >> 6) & 63] : '=';
        out += n > 2 ? A[v & 63] : '=';
    }
    return out;
}

// What a VocabularyDataLoader returns, packed into the reference's two file formats: the vocabulary file is
// base64 of the '\n'-separated token strings in id order (decoder.go:16-36); the merges file is base64 of the pairs
// (idA, idB) in priority order, 17 bits per id, most significant bit first (decoder.go:41-126).  Priorities only matter
// through their order (bpe.go:126: rank + position / length), so the map's values need not be 1, 2, 3, ...; two rules of
// EQUAL priority cannot be expressed in the file format and are refused.
inline bool PackLoaderData(VocabularyDataLoader* loader, std::string* vocab_b64, std::string* merges_b64, Error* err) {
    auto fail = [&](ErrKind k, const char* op, const std::string& detail) {
        if (err && !*err) { err->kind = k; err->op = op; err->detail = detail; }
        return false;
    };
    std::vector<std::string> tokens;
    if (!loader->LoadVocabulary(&tokens, err)) return fail(ErrKind::DataNotFound, "load vocabulary", "loader failed");
    if (tokens.size() >= (1u << 17)) return fail(ErrKind::DataNotFound, "load vocabulary", "more than 131,071 tokens");
    std::string lines;
    std::unordered_map<std::string, int> lookup;   // tokenizer.go:201-204: later ids overwrite earlier ones
    for (size_t i = 0; i < tokens.size(); i++) {
        if (tokens[i].empty() || tokens[i].find('\n') != std::string::npos)
            return fail(ErrKind::DataNotFound, "load vocabulary", "empty token or newline in token " + std::to_string(i));
        lines += tokens[i];
        lines += '\n';
        lookup[tokens[i]] = (int)i;
    }
    std::unordered_map<std::string, int> merges;
    if (!loader->LoadMerges(&merges, err)) return fail(ErrKind::DataNotFound, "load merges", "loader failed");
    std::vector<std::pair<int, const std::string*>> order;
    order.reserve(merges.size());
    for (const auto& kv : merges) order.push_back({kv.second, &kv.first});
    std::sort(order.begin(), order.end(), [](const std::pair<int, const std::string*>& a, const std::pair<int, const std::string*>& b) {
        return a.first != b.first ? a.first < b.first : *a.second < *b.second;
    });
    std::string bits;   // packed ids
    uint64_t acc = 0;
    int n_acc = 0;
    auto put17 = [&](uint32_t id) {
        acc = (acc << 17) | id;
        n_acc += 17;
        while (n_acc >= 8) { bits += (char)((acc >> (n_acc - 8)) & 0xFF); n_acc -= 8; }
    };
    for (size_t k = 0; k < order.size(); k++) {
        if (k && order[k].first == order[k - 1].first)
            return fail(ErrKind::DataNotFound, "load merges", "two merge rules share priority " + std::to_string(order[k].first));
        const std::string& key = *order[k].second;
        const size_t sp = key.find(' ');   // byte-level token strings hold no ' ' (the space byte is U+0120)
        if (sp == std::string::npos || key.find(' ', sp + 1) != std::string::npos)
            return fail(ErrKind::DataNotFound, "load merges", "merge identifier is not \"left right\": " + key);
        auto a = lookup.find(key.substr(0, sp)), b = lookup.find(key.substr(sp + 1));
        if (a == lookup.end() || b == lookup.end()) continue;   // names no token: such a rule can never apply
        put17((uint32_t)a->second);
        put17((uint32_t)b->second);
    }
    if (n_acc) bits += (char)((acc << (8 - n_acc)) & 0xFF);   // zero bits pad the last byte
    *vocab_b64 = Base64(lines);
    *merges_b64 = Base64(bits);
    return true;
}

// New (tokenizer.go:142-213).  The Go package embeds the two data files; a C++ build names them with
// WithDataFiles (options.go) or the L3_DATA_DIR environment variable.
inline std::unique_ptr<Tokenizer> New(const std::vector<Option>& opts, Error* err) {
    Config c;
    for (const Option& o : opts) {
        o.Apply(c);
        if (c.err) { if (err) *err = c.err; return nullptr; }  // tokenizer.go:151-155
    }
    std::string vocab, merges;
    if (c.dataLoader) {
        Error le;
        if (!PackLoaderData(c.dataLoader, &vocab, &merges, &le)) { if (err) *err = le; return nullptr; }
    } else {
        if (c.vocabPath.empty()) {
            const char* d = getenv("L3_DATA_DIR");
            std::string dir = d ? d : "tokenizer_amd/data";
            c.vocabPath = dir + "/vocab_base64.txt";
            c.mergesPath = dir + "/merges_binary.txt";
        }
        if (!ReadFile(c.vocabPath, &vocab) || !ReadFile(c.mergesPath, &merges)) {
            if (err) { err->kind = ErrKind::DataNotFound; err->op = "load vocabulary"; err->detail = c.vocabPath; }
            return nullptr;
        }
    }
    std::unique_ptr<Tokenizer> t(new Tokenizer());
    int rc;
    std::vector<const char*> names;
    for (const std::string& n : c.specialTokens) names.push_back(n.c_str());
    static const char* const none[1] = {nullptr};
    const char* const* sp = !c.haveSpecials ? nullptr : names.empty() ? none : names.data();
    if (!c.devices.empty()) {
        rc = l3_multi_create(vocab.data(), vocab.size(), merges.data(), merges.size(), sp, names.size(), c.devices.data(),
                             (int)c.devices.size(), &t->multi_);
        if (rc == L3_OK) t->ctx_ = l3_multi_context(t->multi_, 0);
    } else {
        rc = l3_create_ex(vocab.data(), vocab.size(), merges.data(), merges.size(), sp, names.size(), c.device, &t->ctx_);
    }
    if (rc != L3_OK) {
        if (err) { err->kind = rc == L3_E_DATA ? ErrKind::DataNotFound : ErrKind::Device; err->op = "l3_create"; err->detail = l3_last_error(nullptr); }
        return nullptr;
    }
    return t;
}

// Many streams at once: every stream is cut into the reference scanner's chunks, all chunks of
// all streams are encoded by ONE batch call, BOS goes on a stream's first chunk and EOS after its last, so the
// result per stream equals what Scanner yields for it.
inline Error EncodeStreams(Tokenizer& t, const std::vector<Reader*>& readers, const scanner::Options& o,
                           std::vector<std::vector<int>>* out) {
    Error e;
    std::string text;
    std::vector<uint64_t> off{0};
    std::vector<size_t> first_chunk(readers.size() + 1, 0);
    std::vector<char> at_eof(readers.size(), 0);
    for (size_t s = 0; s < readers.size(); s++) {
        scanner::Chunker ch(readers[s], o);
        std::string chunk;
        bool last = false;
        first_chunk[s] = off.size() - 1;
        at_eof[s] = false;
        while (ch.Next(&chunk, &last, &e)) {
            if (chunk.empty() && !last) break;  // where a `for sc.Scan()` loop over the reference scanner stops
            if (!chunk.empty()) { text += chunk; off.push_back(text.size()); }
            if (last) { at_eof[s] = true; break; }
        }
        if (e) return e;
    }
    first_chunk[readers.size()] = off.size() - 1;
    std::vector<int32_t> ids;
    std::vector<uint64_t> id_off;
    EncodeOptions none{false, false};
    e = t.EncodeBatch((const uint8_t*)text.data(), off, &none, &ids, &id_off);
    if (e) return e;
    out->assign(readers.size(), std::vector<int>());
    Error no_bos, no_eos;  // a custom special-token list may lack either: then none is added (scanner.go:188,195,232)
    const int bos = t.GetSpecialTokenID("<|begin_of_text|>", &no_bos), eos = t.GetSpecialTokenID("<|end_of_text|>", &no_eos);
    for (size_t s = 0; s < readers.size(); s++) {
        std::vector<int>& v = (*out)[s];
        if (o.enc.BOS && !no_bos && (first_chunk[s + 1] > first_chunk[s] || at_eof[s])) v.push_back(bos);
        v.insert(v.end(), ids.begin() + (long)id_off[first_chunk[s]], ids.begin() + (long)id_off[first_chunk[s + 1]]);
        if (o.enc.EOS && !no_eos && at_eof[s]) v.push_back(eos);
    }
    return e;
}

}  // namespace llama3
