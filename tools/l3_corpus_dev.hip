// l3_corpus_dev.hip -- the C2 / C5 "English" document generator of tools/l3_corpus.c as a HIP kernel, so that a corpus
// larger than any host buffer (64 GiB = 67,108,864 documents of 1 KiB) is made where it is
// encoded: document i is generated from seed + i on the GPU that owns it, byte for byte what l3c_gen_english() writes
// (tests/test_corpus_device.py compares them).  Measurement / test support only: nothing in the product links it.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr uint64_t kSeedEnglish = 0x4C6C616D61330001ull;
constexpr int kDocsPerBlock = 64;
constexpr int kMaxDocBytes = 1024;

__device__ __forceinline__ uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t pick(const uint64_t* cdf, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (u <= cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// One lane per document, written into the workgroup's LDS image (one row per document); the rows then leave as
// coalesced 16-byte stores.
__global__ __launch_bounds__(kDocsPerBlock) void english_docs_kernel(uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes,
                                                                    const uint8_t* words, const uint32_t* word_off,
                                                                    const uint64_t* cdf, uint32_t n_words, uint8_t* out) {
    __shared__ __align__(16) uint8_t img[kDocsPerBlock][kMaxDocBytes];
    const uint64_t d = (uint64_t)blockIdx.x * kDocsPerBlock + threadIdx.x;
    if (d < n_docs) {
        uint8_t* row = img[threadIdx.x];
        uint64_t s = kSeedEnglish + first_doc + d;
        uint32_t pos = 0;
        while (pos < doc_bytes) {
            uint8_t buf[48];
            uint32_t n = 0;
            if (sm64(s) % 100 < 1) {
                const uint32_t nd = 1 + (uint32_t)(sm64(s) % 6);
                for (uint32_t i = 0; i < nd; i++) buf[n++] = (uint8_t)('0' + sm64(s) % 10);
            } else {
                const uint32_t w = pick(cdf, n_words, sm64(s));
                uint32_t l = word_off[w + 1] - word_off[w];
                if (l > 40) l = 40;
                for (uint32_t i = 0; i < l; i++) buf[i] = words[word_off[w] + i];
                n = l;
                if (sm64(s) % 10 == 0 && buf[0] >= 'a' && buf[0] <= 'z') buf[0] = (uint8_t)(buf[0] - 32);
            }
            const uint32_t r = (uint32_t)(sm64(s) % 100);
            if (r < 86) { buf[n++] = ' '; }
            else if (r < 91) { buf[n++] = ','; buf[n++] = ' '; }
            else if (r < 96) { buf[n++] = '.'; buf[n++] = ' '; }
            else if (r < 98) { buf[n++] = '\n'; }
            else if (r < 99) { buf[n++] = '\''; buf[n++] = 's'; buf[n++] = ' '; }
            else { buf[n++] = 'n'; buf[n++] = '\''; buf[n++] = 't'; buf[n++] = ' '; }
            if (n > doc_bytes - pos) n = doc_bytes - pos;
            for (uint32_t i = 0; i < n; i++) row[pos + i] = buf[i];
            pos += n;
        }
    }
    __syncthreads();
    const uint64_t d0 = (uint64_t)blockIdx.x * kDocsPerBlock;
    const uint64_t nd = n_docs - d0 < (uint64_t)kDocsPerBlock ? n_docs - d0 : (uint64_t)kDocsPerBlock;
    uint8_t* dst = out + d0 * doc_bytes;
    if (doc_bytes % 16 == 0 && (((uintptr_t)dst) & 15u) == 0) {
        const uint32_t per = doc_bytes / 16;
        for (uint64_t i = threadIdx.x; i < nd * per; i += kDocsPerBlock) {
            const uint32_t doc = (uint32_t)(i / per), k = (uint32_t)(i % per);
            reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(&img[doc][k * 16]);
        }
    } else {
        for (uint64_t i = threadIdx.x; i < nd * doc_bytes; i += kDocsPerBlock) dst[i] = img[i / doc_bytes][i % doc_bytes];
    }
}

}  // namespace

// All pointers are device pointers; `stream` is a hipStream_t (NULL: the default stream).  Returns 0, or a HIP error code.
// doc_bytes <= 1024.
extern "C" int l3c_gen_english_device(uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes, const uint8_t* d_words,
                                      const uint32_t* d_word_off, const uint64_t* d_cdf, uint32_t n_words, uint8_t* d_out,
                                      void* stream) {
    if (doc_bytes == 0 || doc_bytes > (uint32_t)kMaxDocBytes || n_words == 0) return (int)hipErrorInvalidValue;
    if (n_docs == 0) return 0;
    const uint64_t blocks = (n_docs + kDocsPerBlock - 1) / kDocsPerBlock;
    if (blocks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(english_docs_kernel, dim3((uint32_t)blocks), dim3(kDocsPerBlock), 0, (hipStream_t)stream, first_doc, n_docs,
                       doc_bytes, d_words, d_word_off, d_cdf, n_words, d_out);
    return (int)hipGetLastError();
}
