// dpp_shift.hip -- semantics of the DPP controls the kernels rely on, checked on the device: wave_shr:1 / wave_shl:1 (value of
// the previous / next lane, 0 at the ends with bound_ctrl) and the six-step DPP inclusive scan.
//   hipcc -O3 --offload-arch=gfx950 -o dpp_shift dpp_shift.hip && ./dpp_shift
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t wave_iscan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__global__ void k(const uint32_t* in, uint32_t* out) {
    const uint32_t v = in[threadIdx.x];
    out[threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);        // wave_shr:1
    out[256 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);  // wave_shl:1
    out[512 + threadIdx.x] = wave_iscan_dpp(v & 0xFFFFu);
}
int main() {
    uint32_t h[256], o[768], *d_in, *d_out;
    uint32_t r = 99;
    for (int i = 0; i < 256; i++) { r = r * 1664525u + 1013904223u; h[i] = r | 1u; }
    if (hipMalloc(&d_in, sizeof h) != hipSuccess || hipMalloc(&d_out, sizeof o) != hipSuccess) return 1;
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_in, d_out);
    if (hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int bad = 0;
    for (int i = 0; i < 256; i++) {
        const int l = i & 63;
        const uint32_t up = l ? h[i - 1] : 0u, dn = l < 63 ? h[i + 1] : 0u;
        uint32_t sc = 0;
        for (int j = i - l; j <= i; j++) sc += h[j] & 0xFFFFu;
        bad += (o[i] != up) + (o[256 + i] != dn) + (o[512 + i] != sc);
    }
    printf("dpp wave_shr:1 / wave_shl:1 / six-step scan: %d mismatches\n", bad);
    return bad ? 2 : 0;
}
