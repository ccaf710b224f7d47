// l3_decode.hip -- decode_sum / decode_scan / decode_write_kernel: ids -> bytes in blocks of 4,096 ids (tokenizer.go:339-354).
#include "l3_kernels.h"

#include "l3_bpe.h"
#include "l3_rules.h"
#include "l3_dev.h"

namespace l3 {

namespace {

constexpr int kScanThreads = 1024;
// ---- decode (tokenizer.go:339-354): ids -> bytes, out-of-range ids skipped ------------------------------
// Sequence borders play no part in the bytes: the output of the batch is the concatenation of the bytes of all ids in
// order, and sequence s starts at the byte position of id id_off[s].  So the work is cut into blocks of kDecBlock ids
// whatever the sequences are (one 4 MiB document and 65,536 short ones cost the same):
//   decode_sum_kernel    bytes of every block of ids
//   decode_scan_kernel   exclusive scan of the block sums (one workgroup; 3,400 blocks for 14 M ids), total
//   decode_write_kernel  per block: positions by a workgroup scan, bytes staged in LDS and stored as aligned dwords,
//                        out_off[s] for the sequences that start inside the block
constexpr int kDecThreads = 256, kDecPer = 4, kDecRound = kDecThreads * kDecPer, kDecRounds = 4, kDecBlock = kDecRound * kDecRounds;
constexpr int kDecStage = 16384;   // bytes of one round staged in LDS (1,024 ids: 4.5 KB typical, 128 KB at the very most)

__device__ __forceinline__ uint32_t dec_len(const uint32_t* tok_off, int n_tokens, int32_t id, uint32_t* src) {
    if (id < 0 || id >= n_tokens) { *src = 0; return 0; }
    const uint32_t a = tok_off[id];
    *src = a;
    return tok_off[id + 1] - a;
}

__global__ __launch_bounds__(kDecThreads) void decode_sum_kernel(const uint32_t* tok_off, int n_tokens, const int32_t* ids,
                                                                 uint64_t n_ids, uint64_t* block_base) {
    __shared__ uint32_t tmp[16];
    const uint64_t lo = (uint64_t)blockIdx.x * kDecBlock;
    uint32_t mine = 0, src;
#pragma unroll 4
    for (int k = 0; k < kDecBlock / kDecThreads; k++) {
        const uint64_t i = lo + (uint64_t)k * kDecThreads + threadIdx.x;
        if (i < n_ids) mine += dec_len(tok_off, n_tokens, ids[i], &src);
    }
    uint32_t tot;
    (void)block_escan<kDecThreads / 64>(mine, tmp, &tot);
    if (threadIdx.x == 0) block_base[blockIdx.x + 1] = tot;   // decode_scan_kernel turns these into positions
}

__global__ __launch_bounds__(kScanThreads) void decode_scan_kernel(uint64_t* out_off, uint64_t n_seqs, unsigned long long* total) {
    __shared__ unsigned long long wsum[kScanThreads / 64];
    __shared__ unsigned long long carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_s = 0; out_off[0] = 0; }
    __syncthreads();
    for (uint64_t base = 0; base < n_seqs; base += kScanThreads) {
        uint64_t s = base + tid;
        unsigned long long v = s < n_seqs ? out_off[s + 1] : 0ull, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(inc, d, 64);
            if (lane_id() >= d) inc += o;
        }
        if (lane_id() == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned long long pre = carry_s, tot = 0;
        for (int i = 0; i < kScanThreads / 64; i++) {
            if (i < (tid >> 6)) pre += wsum[i];
            tot += wsum[i];
        }
        if (s < n_seqs) out_off[s + 1] = pre + inc;
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
}

__global__ __launch_bounds__(kDecThreads) void decode_write_kernel(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens,
                                                                   const int32_t* ids, uint64_t n_ids, const uint64_t* id_off,
                                                                   uint64_t n_seqs, const uint64_t* block_base, uint8_t* out,
                                                                   uint64_t out_cap, uint64_t* out_off,
                                                                   const unsigned long long* total) {
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t pos[kDecRound];          // byte position of every id of the round, relative to the round
    __shared__ __align__(16) uint8_t stage[kDecStage];
    __shared__ unsigned long long s_first;
    const int tid = threadIdx.x;
    const bool store = *total <= out_cap;        // too small a buffer: offsets only (the caller learns the size and returns)
    const uint64_t lo = (uint64_t)blockIdx.x * kDecBlock;
    const uint64_t hi = lo + kDecBlock < n_ids ? lo + kDecBlock : n_ids;
    if (tid == 0) {   // first sequence that starts at or after this block's first id
        uint64_t a = 0, b = n_seqs + 1;
        while (a < b) { const uint64_t m = (a + b) / 2; if (id_off[m] < lo) a = m + 1; else b = m; }
        s_first = a;
    }
    __syncthreads();
    uint64_t s_cur = s_first;                    // the same in every thread from here on
    uint64_t run = block_base[blockIdx.x];       // byte position of the round's first id
    for (uint64_t rlo = lo; rlo < hi; rlo += kDecRound) {
        const uint64_t rhi = rlo + kDecRound < hi ? rlo + kDecRound : hi;
        const uint64_t i0 = rlo + (uint64_t)tid * kDecPer;
        uint32_t len[kDecPer], src[kDecPer], tsum = 0;
#pragma unroll
        for (int k = 0; k < kDecPer; k++) {
            len[k] = (i0 + k < rhi) ? dec_len(tok_off, n_tokens, ids[i0 + k], &src[k]) : (src[k] = 0, 0u);
            tsum += len[k];
        }
        uint32_t tot;
        uint32_t p = block_escan<kDecThreads / 64>(tsum, tmp, &tot);
        const uint32_t p0 = p;
#pragma unroll
        for (int k = 0; k < kDecPer; k++) { pos[tid * kDecPer + k] = p; p += len[k]; }
        const bool staged = tot <= (uint32_t)kDecStage;
        if (store) {
            uint32_t q = p0;
            if (staged) {
#pragma unroll
                for (int k = 0; k < kDecPer; k++) {
                    for (uint32_t j = 0; j < len[k]; j++) stage[q + j] = tok_bytes[src[k] + j];
                    q += len[k];
                }
            } else {   // a round of very long vocabulary entries: straight to memory
                uint8_t* dst = out + run + q;
#pragma unroll
                for (int k = 0; k < kDecPer; k++) {
                    for (uint32_t j = 0; j < len[k]; j++) dst[j] = tok_bytes[src[k] + j];
                    dst += len[k];
                }
            }
        }
        __syncthreads();
        // sequences that start inside the round (ascending offsets: the hits are a prefix of what is left)
        for (;;) {
            const uint64_t s = s_cur + (uint64_t)tid;
            uint64_t v = ~0ull;
            if (tid < 64 && s <= n_seqs) v = id_off[s];
            const bool hit = v < rhi;
            if (hit) out_off[s] = run + pos[v - rlo];
            const int cnt = __syncthreads_count(hit);
            s_cur += (uint64_t)cnt;
            if (cnt < 64) break;
        }
        if (store && staged && tot) {
            // aligned dwords of the output: dword j covers stage bytes [4 j - a, 4 j - a + 4), a = misalignment of the start
            uint8_t* dst0 = out + run;
            const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(dst0) & 3u);
            const uint32_t n_dw = (a + tot + 3u) >> 2;
            uint32_t* dw = reinterpret_cast<uint32_t*>(dst0 - a);
            for (uint32_t j = tid; j < n_dw; j += kDecThreads) {
                const int32_t b0 = (int32_t)(4u * j) - (int32_t)a;
                if (b0 >= 0 && (uint32_t)b0 + 4u <= tot) {
                    dw[j] = (uint32_t)stage[b0] | ((uint32_t)stage[b0 + 1] << 8) | ((uint32_t)stage[b0 + 2] << 16) |
                            ((uint32_t)stage[b0 + 3] << 24);
                } else {
                    for (int k = 0; k < 4; k++) {
                        const int32_t bk = b0 + k;
                        if (bk >= 0 && (uint32_t)bk < tot) dst0[bk] = stage[bk];
                    }
                }
            }
        }
        run += tot;
        __syncthreads();
    }
    // sequences that start at the very end (empty ones behind the last id, and the closing offset)
    if (blockIdx.x == gridDim.x - 1)
        for (uint64_t s = s_cur + (uint64_t)tid; s <= n_seqs; s += kDecThreads) out_off[s] = run;
}

}  // namespace

uint64_t decode_blocks(uint64_t n_ids) { return n_ids ? (n_ids + kDecBlock - 1) / kDecBlock : 1; }

hipError_t launch_decode(const uint32_t* tok_off, const uint8_t* tok_bytes, int n_tokens, const int32_t* ids, uint64_t n_ids,
                         const uint64_t* id_off, uint64_t n_seqs, uint64_t* block_base, uint8_t* out, uint64_t out_cap,
                         uint64_t* out_off, unsigned long long* total, hipStream_t stream) {
    const uint64_t n_blocks = decode_blocks(n_ids);   // block_base: n_blocks + 1 entries
    if (n_blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_sum_kernel, dim3((uint32_t)n_blocks), dim3(kDecThreads), 0, stream, tok_off, n_tokens, ids, n_ids,
                       block_base);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, block_base, n_blocks, total);
    L3_LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_write_kernel, dim3((uint32_t)n_blocks), dim3(kDecThreads), 0, stream, tok_off, tok_bytes, n_tokens,
                       ids, n_ids, id_off, n_seqs, block_base, out, out_cap, out_off, total);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3
