// l3_giant.hip -- the round-based exact merge (see merge_rounds in l3_kernels.hip) run by the WHOLE GPU.
//
// A piece of a long pre-token that no cut divides and that has kGiantMin symbols or more -- one character repeated a
// million times, a DNA string, glued words without end -- takes one workgroup tens of milliseconds: a workgroup cannot
// keep enough memory requests in flight.  slow_kernel lists such pieces (Workspace::giants) instead; the host then runs
// the kernels below over ALL listed pieces as ONE flat sequence of symbols, and afterwards the tail of the pipeline.
//
// The algorithm is merge_rounds' (bpe.go:49-59, priority_queue.go:20-25: always the pair with the smallest
// (rank, position); a round applies every member of an independent set S whose key is below theta): each phase of a
// round is a kernel, what merge_rounds passes through workgroup barriers and reductions goes through GiantState.
// Pieces are independent sequences: the first symbol of each carries kSegBit, a pair is never probed across it, and
// greedy merging of the concatenation is the greedy merging of each piece (keys order by flat position, which orders
// the positions inside a piece).  Every symbol carries the flat position of its first byte (`o`), so at the end a
// token goes to the image slot of its first symbol: the image convention of slow_kernel.
#include "l3_kernels.h"

#include "l3_bpe.h"

namespace l3 {

namespace {

constexpr int32_t kIdMask = 0x1FFFF;
constexpr int32_t kSelBit = 1 << 30;
constexpr int32_t kDirtyBit = 1 << 29;
constexpr int32_t kSegBit = 1 << 28;     // first symbol of a piece
constexpr int32_t kKeepMask = kIdMask | kSegBit;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// the flat arrays: symbols, pair priorities, original positions, two of each (a round compacts from one into the other)
struct Flat {
    int32_t* s[2];
    uint32_t* p[2];
    uint32_t* o[2];
};
__host__ __device__ inline Flat flat_of(void* base, uint64_t n_syms) {
    Flat F;
    uint32_t* b = reinterpret_cast<uint32_t*>(base);
    F.s[0] = reinterpret_cast<int32_t*>(b); F.s[1] = reinterpret_cast<int32_t*>(b + n_syms);
    F.p[0] = b + 2 * n_syms; F.p[1] = b + 3 * n_syms;
    F.o[0] = b + 4 * n_syms; F.o[1] = b + 5 * n_syms;
    return F;
}

__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ bool finished(const GiantState& st) { return st.done || st.n < 2 || (uint32_t)(st.best >> 32) == kNoPrio; }

// every wave of the grid owns one contiguous range of the n symbols, a multiple of 64 long.  The grid is the same in every
// launch of one merge (giant_grid(), at most kGiantGrid workgroups: the per-wave arrays of GiantState hold kGiantWaves entries).
__device__ __forceinline__ uint32_t grid_waves() { return gridDim.x * (uint32_t)(kGiantThreads / 64); }
__device__ __forceinline__ void wave_range(uint32_t n, uint32_t* gw, uint32_t* lo, uint32_t* hi) {
    const uint32_t g = blockIdx.x * (uint32_t)(kGiantThreads / 64) + (threadIdx.x >> 6);
    const uint32_t n_waves = grid_waves();
    const uint32_t lw = (((n + n_waves - 1) / n_waves) + 63u) & ~63u;
    const unsigned long long a = (unsigned long long)g * lw;
    *gw = g;
    *lo = a < n ? (uint32_t)a : n;
    *hi = a + lw < n ? (uint32_t)(a + lw) : n;
}

// ---- set-up: the end of every piece listed with an open end (the first cut in a later window of its pre-token, else
//      the pre-token's end), the pieces' places in the flat sequence (in list order), the state
__global__ __launch_bounds__(1024) void giant_order_kernel(Workspace W, uint32_t n_giants) {
    __shared__ unsigned long long wsum[16];
    GiantState& st = *W.gstate;
    const uint32_t t = threadIdx.x;
    unsigned long long n = 0;
    if (t < n_giants) {
        Giant& G = W.giants[t];
        if (G.n == 0) {
            const SlowEntry& e = W.slow[G.entry];
            const uint32_t L = e.pad;
            const uint32_t* first = W.win_first + (e.win0 & ~kEntrySequential);
            const uint32_t n_win = (L + (uint32_t)kWin - 1) / (uint32_t)kWin;
            uint32_t end = L;
            for (uint32_t w = G.k + 1; w < n_win; w++) {
                const uint32_t f = first[w];
                if (f != 0xFFFFFFFFu) { end = f; break; }
            }
            G.n = end - (uint32_t)G.q;
        }
        n = G.n;
        st.order[t] = t;
    }
    unsigned long long inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane_id() >= d) inc += o;
    }
    if (lane_id() == 63) wsum[t >> 6] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (uint32_t i = 0; i < 16; i++) { if (i < (t >> 6)) pre += wsum[i]; tot += wsum[i]; }
    if (t < n_giants) W.giants[t].flat = pre + inc - n;
    if (t == 0) {
        st.n = tot >= (1ull << 31) ? 0xFFFFFFFFu : (uint32_t)tot;
        st.n_next = 0; st.round = 0; st.done = 0;
        st.best = ~0ull; st.best_next = ~0ull; st.theta = ~0ull;
        st.n_giants = n_giants;
    }
}

// ---- symbols, pair priorities, positions of every piece; -1 into its image slots; the smallest key
__global__ __launch_bounds__(kGiantThreads) void giant_init_kernel(Tables T, Workspace W, uint32_t n_giants, Flat F) {
    unsigned long long best = ~0ull;
    for (uint32_t g = 0; g < n_giants; g++) {
        const Giant G = W.giants[g];
        for (uint32_t i = blockIdx.x * kGiantThreads + threadIdx.x; i < G.n; i += gridDim.x * kGiantThreads) {
            const uint32_t b0 = G.sp[G.q + i];
            const uint32_t f = (uint32_t)G.flat + i;
            const uint32_t pr = i + 1 < G.n ? T.bytepair[(b0 << 8) | (uint32_t)G.sp[G.q + i + 1]] : kNoPrio;
            F.s[0][f] = (int32_t)T.byte_tok[b0] | (i == 0 ? kSegBit : 0);
            F.p[0][f] = pr;
            F.o[0][f] = f;
            G.img[G.q + i] = -1;
            const unsigned long long key = ((unsigned long long)pr << 32) | f;
            best = key < best ? key : best;
        }
    }
    best = wave_min64(best);
    if (lane_id() == 0 && best != ~0ull) atomicMin(&W.gstate->best, best);
}

// ---- a round, phase 1 (only when the smallest rank r joins two EQUAL symbols: then its occurrences can overlap,
//      x x x ..., and are taken alternately from the left): per wave, the run of r at the end of its range
__global__ __launch_bounds__(kGiantThreads) void giant_seg_kernel(Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n = st.n, r = (uint32_t)(st.best >> 32), ipos = (uint32_t)st.best, cur = st.round & 1u;
    const int32_t* sa = F.s[cur];
    const uint32_t* pa = F.p[cur];
    if ((sa[ipos] & kIdMask) != (sa[ipos + 1] & kIdMask)) return;
    uint32_t gw, lo, hi;
    wave_range(n, &gw, &lo, &hi);
    const uint32_t lane = (uint32_t)lane_id();
    GiantSeg sg; sg.cnt = 0; sg.open = 1;
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t vn = hi - base < 64u ? hi - base : 64u;
        const unsigned long long full = vn == 64u ? ~0ull : (1ull << vn) - 1ull;
        const unsigned long long m = __ballot(i < hi && pa[i] == r);
        if (m == full) {
            sg.cnt += vn;
        } else {
            const unsigned long long z = ~m & full;
            sg.cnt = vn - 1u - (63u - (uint32_t)__builtin_clzll(z));
            sg.open = 0;
        }
    }
    if (lane == 0) st.wave_seg[gw] = sg;
}

// ---- phase 2: the set S (kSelBit): every strict local minimum of the key, and of the runs of r every other one
__global__ __launch_bounds__(kGiantThreads) void giant_select_kernel(Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n = st.n, r = (uint32_t)(st.best >> 32), ipos = (uint32_t)st.best, cur = st.round & 1u;
    int32_t* sa = F.s[cur];
    const uint32_t* pa = F.p[cur];
    const bool same = (sa[ipos] & kIdMask) == (sa[ipos + 1] & kIdMask);   // (sa[ipos .. ipos + 1] keep their ids in this kernel)
    uint32_t gw, lo, hi;
    wave_range(n, &gw, &lo, &hi);
    const uint32_t lane = (uint32_t)lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st.theta = ~0ull; st.best_next = ~0ull; }
    uint32_t carry = 0;   // same: length of the run of r that ends right before the current 64 symbols
    if (same && lo < hi) {
        long long w = (long long)gw - 1;
        while (w >= 0) {
            const long long idx = w - (long long)lane;
            GiantSeg sg; sg.cnt = 0; sg.open = 0;
            if (idx >= 0) sg = st.wave_seg[idx];
            const unsigned long long closed = __ballot(!sg.open);
            const uint32_t upto = closed ? (uint32_t)__builtin_ctzll(closed) : 63u;   // lanes 0 .. upto belong to the run
            uint32_t c = lane <= upto ? sg.cnt : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            carry += c;
            if (closed) break;
            w -= 64;
        }
    }
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane;
        const bool in = i < hi;
        const uint32_t p = in ? pa[i] : kNoPrio;
        const bool isr = in && p == r;
        bool sel = isr;
        if (same) {
            const unsigned long long m = __ballot(isr);
            const unsigned long long z = ~m & lt;
            const uint32_t idx = z ? lane - (63u - (uint32_t)__builtin_clzll(z)) - 1u : lane + carry;
            sel = isr && !(idx & 1u);
            const unsigned long long zz = ~m;
            carry = zz ? (uint32_t)__builtin_clzll(zz) : carry + 64u;
        }
        if (in) {
            if (!isr && p != kNoPrio) {  // strict local minimum of (rank, position); the last pair of a piece is kNoPrio
                const uint32_t pl = i >= 1 ? pa[i - 1] : kNoPrio, pn = pa[i + 1];
                sel = pl > p && pn >= p;
            }
            sa[i] = (sa[i] & kKeepMask) | (sel ? kSelBit : 0);
        }
    }
}

// ---- phase 3: theta
__global__ __launch_bounds__(kGiantThreads) void giant_theta_kernel(Tables T, Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n = st.n, cur = st.round & 1u;
    const int32_t* sa = F.s[cur];
    const uint32_t* pa = F.p[cur];
    unsigned long long th = ~0ull;
    for (uint32_t i = blockIdx.x * kGiantThreads + threadIdx.x; i < n; i += gridDim.x * kGiantThreads) {
        const uint32_t im1 = i >= 1u ? i - 1u : 0u, im2 = i >= 2u ? i - 2u : 0u;
        const uint32_t ip1 = i + 1u < n ? i + 1u : n - 1u, ip2 = i + 2u < n ? i + 2u : n - 1u;
        const uint32_t p = pa[i], pm1 = pa[im1], pm2 = pa[im2], pp1 = pa[ip1], pp2 = pa[ip2];
        const int32_t s0 = sa[i], sm1 = sa[im1], sm2 = sa[im2], sp1 = sa[ip1], sp2 = sa[ip2];
        const bool valid = p != kNoPrio;   // then i + 1 < n and i + 1 lies in the same piece
        const bool sel = valid && (s0 & kSelBit);
        const unsigned long long key = ((unsigned long long)p << 32) | i;
        const bool covered = (i >= 1u && (sm1 & kSelBit) && pm1 <= p) || ((sp1 & kSelBit) && pp1 < p);
        if (valid && !sel && !covered) th = key < th ? key : th;
        const uint32_t m = prio_merged_id(T, p);
        const bool hasl = sel && i >= 1u && !(s0 & kSegBit);              // a left neighbour in the same piece
        const bool hasr = sel && i + 2u < n && !(sp2 & kSegBit);          // a symbol behind the pair in the same piece
        uint32_t lsym = (uint32_t)(sm1 & kIdMask);
        if (i >= 2u && (sm2 & kSelBit) && pm2 <= p) lsym = prio_merged_id(T, pm2);
        uint32_t rsym = (uint32_t)(sp2 & kIdMask);
        if ((sp2 & kSelBit) && pp2 < p) rsym = prio_merged_id(T, pp2);
        const uint32_t a1 = hasl ? lsym : 0u, b1 = hasl ? m : 0u;   // idle lanes probe (0, 0)
        const uint32_t a2 = hasr ? m : 0u, b2 = hasr ? rsym : 0u;
        uint32_t r1, r2;
        pair_prio2(T, a1, b1, a2, b2, &r1, &r2);
        if (!hasl) r1 = kNoPrio;
        if (!hasr) r2 = kNoPrio;
        const uint32_t v = r2 < r1 ? r2 : r1;
        unsigned long long beta = v == kNoPrio ? ~0ull : ((unsigned long long)v << 32);
        if (beta < key + 1) beta = key + 1;
        if (sel) th = beta < th ? beta : th;
    }
    th = wave_min64(th);
    if (lane_id() == 0 && th != ~0ull) atomicMin(&st.theta, th);
}

__device__ __forceinline__ bool is_dead(const int32_t* sa, const uint32_t* pa, uint32_t i, unsigned long long th) {
    return i >= 1 && (sa[i - 1] & kSelBit) && (((unsigned long long)pa[i - 1] << 32) | (i - 1)) < th;
}

// ---- phase 4: symbols that survive the round, per wave
__global__ __launch_bounds__(kGiantThreads) void giant_count_kernel(Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n = st.n, cur = st.round & 1u;
    const unsigned long long th = st.theta;
    const int32_t* sa = F.s[cur];
    const uint32_t* pa = F.p[cur];
    uint32_t gw, lo, hi;
    wave_range(n, &gw, &lo, &hi);
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t cnt = 0;
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane;
        const bool in = i < hi;
        cnt += (uint32_t)__popcll(__ballot(in && !is_dead(sa, pa, i, th)));
    }
    if (lane == 0) st.wave_live[gw] = cnt;
}

// ---- phase 5: apply the members of S below theta, compacting into the other arrays
__global__ __launch_bounds__(kGiantThreads) void giant_apply_kernel(Tables T, Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n = st.n, cur = st.round & 1u;
    const unsigned long long th = st.theta;
    const int32_t* sa = F.s[cur];
    const uint32_t* pa = F.p[cur];
    const uint32_t* oa = F.o[cur];
    int32_t* sb = F.s[cur ^ 1u];
    uint32_t* pb = F.p[cur ^ 1u];
    uint32_t* ob = F.o[cur ^ 1u];
    uint32_t gw, lo, hi;
    wave_range(n, &gw, &lo, &hi);
    const uint32_t lane = (uint32_t)lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t j0 = 0;   // survivors in the ranges of the waves before this one
    for (uint32_t w = lane; w < gw; w += 64) j0 += st.wave_live[w];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) j0 += __shfl_xor(j0, d, 64);
    if (gw == grid_waves() - 1u && lane == 0) st.n_next = j0 + st.wave_live[gw];
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane;
        const bool in = i < hi;
        const bool live = in && !is_dead(sa, pa, i, th);
        const unsigned long long lm = __ballot(live);
        if (live) {
            const uint32_t j = j0 + (uint32_t)__popcll(lm & lt);
            const uint32_t p = pa[i];
            const int32_t x = sa[i];
            const bool applied = (x & kSelBit) && (((unsigned long long)p << 32) | i) < th;
            sb[j] = applied ? (int32_t)(prio_merged_id(T, p) | (uint32_t)kDirtyBit | (uint32_t)(x & kSegBit)) : (x & kKeepMask);
            pb[j] = p;
            ob[j] = oa[i];
        }
        j0 += (uint32_t)__popcll(lm);
    }
}

// ---- phase 6: the pairs that touch a merged symbol are probed again; the smallest key of the next round
__global__ __launch_bounds__(kGiantThreads) void giant_reprobe_kernel(Tables T, Workspace W, Flat F) {
    GiantState& st = *W.gstate;
    if (finished(st)) return;
    const uint32_t n2 = st.n_next, nxt = (st.round & 1u) ^ 1u;
    const int32_t* sb = F.s[nxt];
    uint32_t* pb = F.p[nxt];
    unsigned long long best = ~0ull;
    for (uint32_t q = blockIdx.x * kGiantThreads + threadIdx.x; q < n2; q += gridDim.x * kGiantThreads) {
        const bool has_next = q + 1u < n2;
        const int32_t x = sb[q];
        const int32_t y = has_next ? sb[q + 1u] : kSegBit;
        uint32_t p = pb[q];
        if ((x | (has_next ? y : 0)) & kDirtyBit) p = (has_next && !(y & kSegBit)) ? pair_prio(T, (uint32_t)(x & kIdMask), (uint32_t)(y & kIdMask)) : kNoPrio;
        pb[q] = p;
        const unsigned long long val = ((unsigned long long)p << 32) | q;
        best = val < best ? val : best;
    }
    best = wave_min64(best);
    if (lane_id() == 0 && best != ~0ull) atomicMin(&st.best_next, best);
}

// ---- phase 7: the next round's state
__global__ void giant_next_kernel(Workspace W) {
    GiantState& st = *W.gstate;
    if (st.done) return;
    if (st.n < 2 || (uint32_t)(st.best >> 32) == kNoPrio) { st.done = 1; return; }
    st.n = st.n_next;
    st.best = st.best_next;
    st.round++;
    if (st.n < 2 || (uint32_t)(st.best >> 32) == kNoPrio) st.done = 1;
}

// ---- the tokens to the images: each to the slot of its first symbol; window counters, token counts
__global__ __launch_bounds__(kGiantThreads) void giant_store_kernel(Workspace W, Flat F) {
    const GiantState& st = *W.gstate;
    const uint32_t n = st.n, cur = st.round & 1u, ng = st.n_giants;
    const int32_t* sa = F.s[cur];
    const uint32_t* oa = F.o[cur];
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t total_threads = gridDim.x * kGiantThreads;
    for (uint32_t j0 = blockIdx.x * kGiantThreads + (threadIdx.x & ~63u); j0 < n; j0 += total_threads) {   // whole waves
        const uint32_t j = j0 + lane;
        const bool in = j < n;
        unsigned long long key = ~0ull;   // piece << 32 | window of the slot
        uint32_t g = 0;
        if (in) {
            const uint32_t f = oa[j];
            uint32_t a = 0, b = ng - 1;   // last piece (in flat order) that starts at or before f
            while (a < b) {
                const uint32_t mid = (a + b + 1) >> 1;
                if ((uint32_t)W.giants[st.order[mid]].flat <= f) a = mid; else b = mid - 1;
            }
            g = st.order[a];
            const Giant& G = W.giants[g];
            const uint64_t slot = G.q + (uint64_t)(f - (uint32_t)G.flat);
            G.img[slot] = sa[j] & kIdMask;
            key = ((unsigned long long)g << 32) | (unsigned long long)(slot / (uint64_t)kWin);
        }
        // one atomic per run of lanes with the same piece and window (neighbours nearly always share both)
        const unsigned long long prev = __shfl_up(key, 1, 64);
        const bool head = in && (lane == 0 || prev != key);
        const unsigned long long heads = __ballot(head), ins = __ballot(in);
        if (head) {
            const unsigned long long later = heads & ~((2ull << lane) - 1ull);
            const uint32_t end = later ? (uint32_t)__builtin_ctzll(later) : (uint32_t)__popcll(ins);   // lanes in use are 0 .. popc - 1
            const uint32_t run = end - lane;
            const Giant& G = W.giants[g];
            atomicAdd(&G.live[(uint32_t)key], run);
            atomicAdd(&W.slow[G.entry].count, run);
            atomicAdd(&W.tile_extra[W.slow[G.entry].tile], run);
        }
    }
}

}  // namespace

#define L3_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e_ = hipGetLastError();        \
        if (e_ != hipSuccess) return e_;          \
    } while (0)

hipError_t launch_giants_order(const Workspace& W, uint32_t n_giants, hipStream_t stream) {
    if (n_giants == 0 || n_giants > kMaxGiants) return hipErrorInvalidValue;
    hipLaunchKernelGGL(giant_order_kernel, dim3(1), dim3(1024), 0, stream, W, n_giants);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

// Workgroups of every launch of one grid-wide merge: about 4,096 symbols each, a power of two from 128 to kGiantGrid.  The cost
// of a launch grows with its grid (dispatch, and every wave adds up the survivor counts of the waves before it), the time of
// a round falls with it while there are symbols to spread: measured on 1 M / 3 M / 7.7 M symbols the best grids were 256 /
// 512 / 1,024 (1 MiB of one letter: 0.41 ms of host-driven passes against 0.61 at 1,024 and 0.47 at 128).
static uint32_t giant_grid(uint64_t n_syms) {
    uint32_t g = 128;
    while (g * 2u <= (uint32_t)kGiantGrid && (uint64_t)(g * 2u) * 4096u <= n_syms) g *= 2u;
    return g;
}

hipError_t launch_giants_init(const Tables& T, const Workspace& W, uint32_t n_giants, uint64_t n_syms, void* flat, hipStream_t stream) {
    if (n_giants == 0 || n_giants > kMaxGiants || n_syms == 0 || n_syms >= (1ull << 31)) return hipErrorInvalidValue;
    const Flat F = flat_of(flat, n_syms);
    hipLaunchKernelGGL(giant_init_kernel, dim3(giant_grid(n_syms)), dim3(kGiantThreads), 0, stream, T, W, n_giants, F);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_giants_rounds(const Tables& T, const Workspace& W, uint64_t n_syms, void* flat, int rounds, hipStream_t stream) {
    const Flat F = flat_of(flat, n_syms);
    const uint32_t grid = giant_grid(n_syms);
    for (int r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(giant_seg_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, W, F);
        hipLaunchKernelGGL(giant_select_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, W, F);
        hipLaunchKernelGGL(giant_theta_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, T, W, F);
        hipLaunchKernelGGL(giant_count_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, W, F);
        hipLaunchKernelGGL(giant_apply_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, T, W, F);
        hipLaunchKernelGGL(giant_reprobe_kernel, dim3(grid), dim3(kGiantThreads), 0, stream, T, W, F);
        hipLaunchKernelGGL(giant_next_kernel, dim3(1), dim3(1), 0, stream, W);
        L3_LAUNCH_CHECK();
    }
    return hipSuccess;
}

hipError_t launch_giants_store(const Workspace& W, uint64_t n_syms, void* flat, hipStream_t stream) {
    const Flat F = flat_of(flat, n_syms);
    hipLaunchKernelGGL(giant_store_kernel, dim3(giant_grid(n_syms)), dim3(kGiantThreads), 0, stream, W, F);
    L3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace l3
