// llama3-hip: command line over include/llama3.hpp, with the sub-commands, flags and output formats of the
// reference's `tokenizer llama3 ...` (llama3/cmd/llama3/{encode,decode,info,metrics}.go) -- SURVEY 8 f-3:
//
//   llama3-hip encode [--bos=false] [--eos=false] [-o space|newline|json] [--count] [--count-only] [--metrics] [text...]
//   llama3-hip decode [--skip-special] [id ...]
//   llama3-hip info
//
// No text / ids on the command line: read from stdin.  Differences from the reference, on purpose: stdin is cut into
// the scanner's chunks and ALL chunks are encoded by one batch call (EncodeStreams) instead of one call per chunk;
// --metrics prints the reference's three figures (latency and tps include building the tables, as there) plus the
// same two for the encode call alone.  The data files come from L3_DATA_DIR (default tokenizer_amd/data).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "llama3.hpp"

using namespace llama3;
typedef std::chrono::steady_clock Clock;

// metrics.go:9-19
static std::string format_latency(double ns) {
    char b[64];
    if (ns < 1e3) snprintf(b, sizeof b, "%lldns", (long long)ns);
    else if (ns < 1e6) snprintf(b, sizeof b, "%.2f\xce\xbcs", ns / 1e3);
    else if (ns < 1e9) snprintf(b, sizeof b, "%.2fms", (double)(long long)(ns / 1e3) / 1e3);
    else snprintf(b, sizeof b, "%.2fs", ns / 1e9);
    return b;
}
// metrics.go:21-26
static long long tps(size_t n, double ns) { return ns <= 0 ? 0 : (long long)((double)n / (ns / 1e9)); }

static bool read_stdin(std::string* out) {
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, stdin)) > 0) out->append(buf, k);
    return !ferror(stdin);
}
static bool bool_flag(const char* arg, const char* name, bool* v) {  // --name, --name=true|false
    const size_t l = strlen(name);
    if (strncmp(arg, name, l) != 0) return false;
    if (arg[l] == 0) { *v = true; return true; }
    if (arg[l] != '=') return false;
    *v = !(strcmp(arg + l + 1, "false") == 0 || strcmp(arg + l + 1, "0") == 0);
    return true;
}
static std::unique_ptr<Tokenizer> open_tokenizer() {
    Error e;
    std::unique_ptr<Tokenizer> t = New({}, &e);
    if (!t) fprintf(stderr, "Error: failed to initialize tokenizer: %s\n", e.Message().c_str());
    return t;
}

static int run_encode(int argc, char** argv) {
    bool bos = true, eos = true, count = false, count_only = false, metrics = false, b;
    std::string output = "space", text;
    bool have_text = false;
    for (int i = 0; i < argc; i++) {
        const char* a = argv[i];
        if (bool_flag(a, "--bos", &bos) || bool_flag(a, "--eos", &eos) || bool_flag(a, "--count-only", &count_only) ||
            bool_flag(a, "--count", &count) || bool_flag(a, "--metrics", &metrics)) continue;
        if (bool_flag(a, "--no-bos", &b)) { bos = !b; continue; }
        if (bool_flag(a, "--no-eos", &b)) { eos = !b; continue; }
        if ((strcmp(a, "-o") == 0 || strcmp(a, "--output") == 0) && i + 1 < argc) { output = argv[++i]; continue; }
        if (strncmp(a, "--output=", 9) == 0) { output = a + 9; continue; }
        if (have_text) text += " ";   // strings.Join(args, " ")
        text += a;
        have_text = true;
    }
    if (output != "space" && output != "newline" && output != "json") {
        fprintf(stderr, "Error: unknown output format: %s\n", output.c_str());
        return 1;
    }
    const Clock::time_point t0 = Clock::now();
    std::unique_ptr<Tokenizer> tok = open_tokenizer();
    if (!tok) return 1;
    if (!have_text && !read_stdin(&text)) { fprintf(stderr, "Error: tokenization error: read failed\n"); return 1; }
    const Clock::time_point t1 = Clock::now();
    BytesReader rd(text.data(), text.size());
    scanner::Options so;
    so.enc.BOS = bos;
    so.enc.EOS = eos;
    std::vector<std::vector<int>> res;
    Error e = EncodeStreams(*tok, {&rd}, so, &res);
    if (e) { fprintf(stderr, "Error: tokenization error: %s\n", e.Message().c_str()); return 1; }
    const std::vector<int>& tokens = res[0];
    const Clock::time_point t2 = Clock::now();
    const double ns_all = std::chrono::duration<double, std::nano>(t2 - t0).count();
    const double ns_enc = std::chrono::duration<double, std::nano>(t2 - t1).count();

    if (count_only) {
        if (output == "json") printf("{\"count\":%zu}\n", tokens.size());
        else printf("%zu\n", tokens.size());
        return 0;
    }
    if (output == "json") {  // json.Marshal of a map: keys in alphabetical order
        std::string o = "{";
        if (count) o += "\"count\":" + std::to_string(tokens.size()) + ",";
        if (metrics) {
            o += "\"metrics\":{\"encode_latency\":\"" + format_latency(ns_enc) + "\",\"encode_tps\":" + std::to_string(tps(tokens.size(), ns_enc)) +
                 ",\"input_bytes\":" + std::to_string(text.size()) + ",\"latency\":\"" + format_latency(ns_all) + "\",\"tps\":" +
                 std::to_string(tps(tokens.size(), ns_all)) + "},";
        }
        o += "\"tokens\":[";
        for (size_t i = 0; i < tokens.size(); i++) o += (i ? "," : "") + std::to_string(tokens[i]);
        o += "]}";
        puts(o.c_str());
        return 0;
    }
    if (output == "newline") {
        if (count) printf("count: %zu\n", tokens.size());
        for (int t : tokens) printf("%d\n", t);
    } else {
        if (count) printf("count: %zu\ntokens: ", tokens.size());
        for (size_t i = 0; i < tokens.size(); i++) printf(i ? " %d" : "%d", tokens[i]);
        printf("\n");
    }
    if (metrics) {
        printf("metrics:\n  latency: %s\n  tps: %lld\n  input_bytes: %zu\n", format_latency(ns_all).c_str(), tps(tokens.size(), ns_all), text.size());
        printf("  encode_latency: %s\n  encode_tps: %lld\n", format_latency(ns_enc).c_str(), tps(tokens.size(), ns_enc));
    }
    return 0;
}

static int run_decode(int argc, char** argv) {
    bool skip_special = false;
    std::vector<std::string> words;
    for (int i = 0; i < argc; i++)
        if (!bool_flag(argv[i], "--skip-special", &skip_special)) words.push_back(argv[i]);
    if (words.empty()) {  // bufio.ScanWords over stdin
        std::string in, w;
        if (!read_stdin(&in)) { fprintf(stderr, "Error: failed to read from stdin\n"); return 1; }
        for (char c : in) {
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f') { if (!w.empty()) words.push_back(w); w.clear(); }
            else w += c;
        }
        if (!w.empty()) words.push_back(w);
    }
    std::vector<int> ids;
    for (const std::string& w : words) {
        char* end = nullptr;
        long v = strtol(w.c_str(), &end, 10);
        if (w.empty() || *end) { fprintf(stderr, "Error: invalid token ID \"%s\"\n", w.c_str()); return 1; }
        ids.push_back((int)v);
    }
    if (ids.empty()) { fprintf(stderr, "Error: no token IDs provided\n"); return 1; }
    std::unique_ptr<Tokenizer> tok = open_tokenizer();
    if (!tok) return 1;
    std::string text = tok->Decode(ids);
    if (skip_special) {  // decode.go: every default special-token literal removed from the text
        const int n_special = tok->VocabSize() >= 256 ? 256 : 0;
        for (int k = 0; k < n_special; k++) {
            const std::string lit = tok->Decode({tok->VocabSize() - 256 + k});
            for (size_t p; !lit.empty() && (p = text.find(lit)) != std::string::npos;) text.erase(p, lit.size());
        }
    }
    fwrite(text.data(), 1, text.size(), stdout);
    return 0;
}

static int run_info() {
    std::unique_ptr<Tokenizer> tok = open_tokenizer();
    if (!tok) return 1;
    printf("Llama 3 tokenizer on HIP (gfx950)\n\n");
    printf("  Vocabulary size:   %d tokens (%d regular + 256 special)\n", tok->VocabSize(), tok->VocabSize() - 256);
    static const char* names[] = {"<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>",
                                  "<|eom_id|>", "<|python_tag|>", "<|finetune_right_pad_id|>", "<|reserved_special_token_0|>"};
    for (const char* n : names) {
        Error e;
        const int id = tok->GetSpecialTokenID(n, &e);
        if (!e) printf("  %-30s -> %d\n", n, id);
    }
    printf("\n  Encode / Decode / pre-tokenize run as HIP kernels on batches of documents; the scanner surface cuts\n"
           "  streams into chunks on the host and encodes all chunks of all streams in one batch.\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "encode") == 0) return run_encode(argc - 2, argv + 2);
    if (argc >= 2 && strcmp(argv[1], "decode") == 0) return run_decode(argc - 2, argv + 2);
    if (argc >= 2 && strcmp(argv[1], "info") == 0) return run_info();
    fprintf(stderr, "usage: llama3-hip encode|decode|info [flags] [args]   (see the head of tools/llama3_cli.cpp)\n");
    return 2;
}
