// lds_unaligned.hip -- does gfx950 execute the unaligned LDS loads that hipcc emits for align-1 accesses (ds_read_b32 / b64 /
// b96 at any byte address), and what do they cost next to aligned loads + v_alignbit?  Stand-alone probe:
//   hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct __attribute__((packed)) U3 { uint32_t a, b, c; };
struct __attribute__((packed)) U1 { uint32_t a; };
constexpr int kBytes = 4096;
template <int MODE, bool DEP = true>
__global__ __launch_bounds__(256) void k(const uint8_t* in, const uint16_t* pos, uint32_t* out, int iters) {
    __shared__ __align__(16) uint8_t buf[kBytes + 32];
    for (int i = threadIdx.x; i < (kBytes + 32) / 4; i += 256) reinterpret_cast<uint32_t*>(buf)[i] = reinterpret_cast<const uint32_t*>(in)[i];
    __syncthreads();
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    uint32_t s = pos[blockIdx.x * 256 + threadIdx.x];
    for (int it = 0; it < iters; it++) {
        uint32_t a, b, c;
        if (MODE == 0) {   // unaligned 12-byte load
            U3 v; __builtin_memcpy(&v, buf + s, 12); a = v.a; b = v.b; c = v.c;
        } else if (MODE == 1) {  // aligned words + funnel shifts
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(buf) + (s >> 2);
            const uint32_t sh = (s & 3) * 8;
            const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2], x3 = wp[3];
            a = __funnelshift_r(x0, x1, sh); b = __funnelshift_r(x1, x2, sh); c = __funnelshift_r(x2, x3, sh);
        } else {  // unaligned 4-byte load of a 16-bit-aligned pair
            U1 v; __builtin_memcpy(&v, buf + (s & ~1u), 4); a = v.a; b = 0; c = 0;
        }
        acc0 ^= a; acc1 += b; acc2 ^= c;
        s = DEP ? (s + (a & 7u) + 13u) & (kBytes - 1) : (s + 13u) & (kBytes - 1);
    }
    out[(blockIdx.x * 256 + threadIdx.x) * 3 + 0] = acc0;
    out[(blockIdx.x * 256 + threadIdx.x) * 3 + 1] = acc1;
    out[(blockIdx.x * 256 + threadIdx.x) * 3 + 2] = acc2;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int blocks = 2048, iters = 256;
    std::vector<uint8_t> h(kBytes + 32);
    uint32_t r = 12345;
    for (auto& x : h) { r = r * 1664525u + 1013904223u; x = (uint8_t)(r >> 24); }
    std::vector<uint16_t> pos(blocks * 256);
    for (auto& p : pos) { r = r * 1664525u + 1013904223u; p = (uint16_t)((r >> 12) & (kBytes - 1)); }
    uint8_t* d_in; uint16_t* d_pos; uint32_t* d_out[3];
    CK(hipMalloc(&d_in, h.size())); CK(hipMalloc(&d_pos, pos.size() * 2));
    for (int m = 0; m < 3; m++) CK(hipMalloc(&d_out[m], blocks * 256 * 3 * 4));
    CK(hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pos, pos.data(), pos.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[3];
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[0], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[1], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[2], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
    }
    std::vector<uint32_t> o0(blocks * 256 * 3), o1(blocks * 256 * 3), o2(blocks * 256 * 3);
    CK(hipMemcpy(o0.data(), d_out[0], o0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out[1], o1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), d_out[2], o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < o0.size(); i++) bad += o0[i] != o1[i];
    // mode 2 against the host
    size_t bad2 = 0;
    for (int t = 0; t < blocks * 256; t++) {
        uint32_t s = pos[t], acc = 0;
        for (int it = 0; it < iters; it++) {
            uint32_t q = s & ~1u, a = h[q] | (h[q + 1] << 8) | (h[q + 2] << 16) | ((uint32_t)h[q + 3] << 24);
            acc ^= a;
            s = (s + (a & 7u) + 13u) & (kBytes - 1);
        }
        bad2 += acc != o2[t * 3];
    }
    printf("unaligned b96 vs aligned+alignbit: %zu mismatching words of %zu; unaligned b32 vs host: %zu mismatches\n", bad, o0.size(), bad2);
    {   // throughput: addresses do not depend on the loaded data
        float t[3];
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<0, false>), dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[0], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[0], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<1, false>), dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[1], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[1], e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<2, false>), dim3(blocks), dim3(256), 0, 0, d_in, d_pos, d_out[2], iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[2], e0, e1));
        printf("independent addresses, ms: unaligned b96 %.3f, aligned + alignbit %.3f, unaligned b32 %.3f\n", t[0], t[1], t[2]);
    }
    printf("ms: unaligned b96 %.3f, aligned 4 x b32 + 3 alignbit %.3f, unaligned b32 %.3f  (%d blocks x 256 threads x %d dependent loads)\n", ms[0], ms[1], ms[2], blocks, iters);
    return bad || bad2 ? 2 : 0;
}
