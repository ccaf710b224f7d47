// l3_dev.h -- wave / workgroup primitives shared by the kernel translation units (l3_tile.hip, l3_merge.hip, l3_slow.hip,
// l3_emit.hip, l3_decode.hip).  Device code only; everything sits in an unnamed namespace, every unit gets its own copy.
#pragma once
#include "l3_kernels.h"

namespace l3 {

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- workgroup primitives (256 threads = 4 waves) --------------------------------------------
// Data movement between the lanes of a wave by DPP (one VALU operation, no trip through the LDS crossbar as __shfl makes):
// CTRL 0x111 / 0x112 / 0x114 / 0x118 = row_shr:1/2/4/8 inside rows of 16 lanes, 0x142 / 0x143 = row_bcast:15 / :31 (the last
// lane of a row to the rows behind it: ROWS 0xA / 0xC), 0x138 / 0x130 = wave_shr:1 / wave_shl:1 (previous / next lane of the
// wave).  A lane whose source does not exist gets `idle`.  Checked on the device by tools/probes/dpp_shift.hip.  All 64 lanes
// must be active.
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ uint32_t dpp_move(uint32_t v, uint32_t idle = 0u) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)idle, (int)v, CTRL, ROWS, 0xF, false);
}
// inclusive sum scan over the wave: six v_add_u32_dpp
__device__ __forceinline__ uint32_t wave_iscan(uint32_t v) {
    v += dpp_move<0x111>(v);
    v += dpp_move<0x112>(v);
    v += dpp_move<0x114>(v);
    v += dpp_move<0x118>(v);
    v += dpp_move<0x142, 0xA>(v);
    v += dpp_move<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ unsigned long long wave_iscan64(unsigned long long v) {
#define L3_STEP(CTRL, ROWS) v += ((unsigned long long)dpp_move<CTRL, ROWS>((uint32_t)(v >> 32)) << 32) | dpp_move<CTRL, ROWS>((uint32_t)v)
    L3_STEP(0x111, 0xF); L3_STEP(0x112, 0xF); L3_STEP(0x114, 0xF); L3_STEP(0x118, 0xF); L3_STEP(0x142, 0xA); L3_STEP(0x143, 0xC);
#undef L3_STEP
    return v;
}
// exclusive sum scan over the workgroup; tmp: >= 4 words of LDS.  Contains two barriers.
template <int NW = kTileThreads / 64>
__device__ __forceinline__ uint32_t block_escan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    uint32_t inc = wave_iscan(v);
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t x = tmp[i];
        if (i < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
template <int NW = kTileThreads / 64>
__device__ __forceinline__ unsigned long long block_min64(unsigned long long v, unsigned long long* tmp) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    if (lane_id() == 0) tmp[wave_id()] = v;
    __syncthreads();
    unsigned long long r = tmp[0];
#pragma unroll
    for (int i = 1; i < NW; i++) r = tmp[i] < r ? tmp[i] : r;
    __syncthreads();
    return r;
}
// sum over the workgroup; tmp: >= NW words of LDS.  Two barriers.
template <int NW = kTileThreads / 64>
__device__ __forceinline__ unsigned long long block_sum64(unsigned long long v, unsigned long long* tmp) {
    v = wave_iscan64(v);
    __syncthreads();
    if (lane_id() == 63) tmp[wave_id()] = v;
    __syncthreads();
    unsigned long long r = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) r += tmp[i];
    return r;
}
// segmented counter: `cnt` = length of the run at the end of the span, `open` = the span holds no break
struct Seg { uint32_t cnt, open; };
__device__ __forceinline__ Seg seg_comb(Seg a, Seg b) {
    Seg r;
    r.cnt = b.open ? a.cnt + b.cnt : b.cnt;
    r.open = a.open & b.open;
    return r;
}
// byte source over global memory restricted to one segment [lo, hi)
struct GlobalSrc {
    const uint8_t* t;
    int64_t lo_, hi_;
    __device__ uint32_t b(int64_t q) const { return t[q]; }
    __device__ bool ds(int64_t q) const { return q == lo_ || q == hi_; }
    __device__ int64_t lo() const { return lo_; }
    __device__ int64_t hi() const { return hi_; }
};

__device__ __forceinline__ int find_next_bit64(const u64* bits, int from, int limit) {
    if (from >= limit) return -1;
    int w = from >> 6;
    u64 x = bits[w] & (~0ull << (from & 63));
    const int wl = (limit - 1) >> 6;
    for (;;) {
        if (x) {
            int q = (w << 6) + __builtin_ctzll(x);
            return q < limit ? q : -1;
        }
        if (++w > wl) return -1;
        x = bits[w];
    }
}
__device__ __forceinline__ bool any_bit_in(const u64* bits, int from, int to) {  // [from, to), from < to
    int w0 = from >> 6, w1 = (to - 1) >> 6;
    u64 m0 = ~0ull << (from & 63), m1 = ~0ull >> (63 - ((to - 1) & 63));
    if (w0 == w1) return (bits[w0] & m0 & m1) != 0;
    if (bits[w0] & m0) return true;
    for (int w = w0 + 1; w < w1; w++)
        if (bits[w]) return true;
    return (bits[w1] & m1) != 0;
}

__device__ __forceinline__ uint32_t up32(uint32_t v) { return dpp_move<0x138>(v); }  // value of the previous lane (lane 0: 0)
__device__ __forceinline__ uint32_t dn32(uint32_t v) { return dpp_move<0x130>(v); }  // value of the next lane (lane 63: 0)
// exclusive segmented scan over the wave; (0, open) is the identity of seg_comb
__device__ __forceinline__ Seg wave_seg_escan(Seg v) {
    Seg inc = v;
#define L3_STEP(CTRL, ROWS) { Seg t; t.cnt = dpp_move<CTRL, ROWS>(inc.cnt, 0u); t.open = dpp_move<CTRL, ROWS>(inc.open, 1u); inc = seg_comb(t, inc); }
    L3_STEP(0x111, 0xF) L3_STEP(0x112, 0xF) L3_STEP(0x114, 0xF) L3_STEP(0x118, 0xF) L3_STEP(0x142, 0xA) L3_STEP(0x143, 0xC)
#undef L3_STEP
    Seg ex;
    ex.cnt = dpp_move<0x138>(inc.cnt, 0u);
    ex.open = dpp_move<0x138>(inc.open, 1u);
    return ex;
}

}  // namespace

}  // namespace l3

#define L3_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e_ = hipGetLastError();        \
        if (e_ != hipSuccess) return e_;          \
    } while (0)
