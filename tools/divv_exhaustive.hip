// tools/divv_exhaustive.hip — run ON THE GPU BOX: division by a VARIABLE divisor without the scaling / fix-up of the IEEE expansion:
//     V1:  r0 = v_rcp_f32(b);  e = fma(-b, r0, 1);  r = fma(e, r0, r0);  q0 = a r;  t = fma(-q0, b, a);  q = fma(t, r, q0)     (6)
//     V0:  r = v_rcp_f32(b);                                             q0 = a r;  t = fma(-q0, b, a);  q = fma(t, r, q0)     (4)
// against RN(a / b) for EVERY pair of binary32 significands (b in [1, 2), a in [1, 4): 2^47 quotients; scale-free away from
// overflow / underflow).  Prints, per variant, how many divisors have a failing dividend, and the first few.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__global__ void __launch_bounds__(256) k_divv(uint32_t b0, unsigned* __restrict__ nfail, uint32_t* __restrict__ fb, uint32_t* __restrict__ fa, unsigned cap) {
    const uint32_t bbits = 0x3f800000u + b0 + blockIdx.x;
    const float b = __uint_as_float(bbits);
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    uint32_t first0 = 0, first1 = 0;
    for (uint32_t m = threadIdx.x; m < (1u << 24); m += 256) {
        const float a = __uint_as_float(0x3f800000u + m);
        const float want = (float)((double)a / (double)b);
        float q0 = a * r1, t = __builtin_fmaf(-q0, b, a);
        const float q1 = __builtin_fmaf(t, r1, q0);
        q0 = a * r0; t = __builtin_fmaf(-q0, b, a);
        const float qz = __builtin_fmaf(t, r0, q0);
        if (__float_as_uint(q1) != __float_as_uint(want) && first1 == 0) first1 = 0x3f800000u + m;
        if (__float_as_uint(qz) != __float_as_uint(want) && first0 == 0) first0 = 0x3f800000u + m;
    }
    if (first1) { const unsigned i = atomicAdd(&nfail[1], 1u); if (i < cap) { fb[cap + i] = bbits; fa[cap + i] = first1; } }
    if (first0) { const unsigned i = atomicAdd(&nfail[0], 1u); if (i < cap) { fb[i] = bbits; fa[i] = first0; } }
}

int main() {
    const unsigned cap = 1u << 12;
    unsigned* nfail; uint32_t *fb, *fa;
    hipMalloc(&nfail, 8); hipMalloc(&fb, 2 * cap * 4); hipMalloc(&fa, 2 * cap * 4);
    hipMemset(nfail, 0, 8);
    const uint32_t per = 1u << 16;
    for (uint32_t b0 = 0; b0 < (1u << 23); b0 += per) {
        hipLaunchKernelGGL(k_divv, dim3(per), dim3(256), 0, 0, b0, nfail, fb, fa, cap);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed at b0 = %u\n", b0); return 1; }
    }
    unsigned n[2]; hipMemcpy(n, nfail, 8, hipMemcpyDeviceToHost);
    std::vector<uint32_t> hb(2 * cap), ha(2 * cap);
    hipMemcpy(hb.data(), fb, hb.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), fa, ha.size() * 4, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
        printf("V%d (%s): thread-level reports of a failing dividend: %u (0 = every one of the 2^47 quotients is the IEEE quotient)\n", v,
               v ? "rcp + one Newton step" : "rcp as it comes", n[v]);
        for (unsigned i = 0; i < n[v] && i < 8; ++i) { float b, a; memcpy(&b, &hb[v * cap + i], 4); memcpy(&a, &ha[v * cap + i], 4); printf("    b = %a  a = %a\n", b, a); }
    }
    return 0;
}
