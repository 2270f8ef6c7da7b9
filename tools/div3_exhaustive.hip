// tools/div3_exhaustive.hip — run ON THE GPU BOX: is the three-instruction binary32 division by a known divisor
//     r = RN(1 / d) (once);   q0 = RN(a * r);   e = fma(-q0, d, a);   q = fma(e, r, q0)
// equal to the IEEE quotient RN(a / d) for EVERY pair of binary32 significands?  Division is scale-free away from overflow and
// underflow, so d in [1, 2) (2^23 values) against a in [1, 4) (2^24 values: both relative positions of the significands) covers all
// pairs: 2^47 quotients.  Prints every divisor that has a failing dividend (with the first one).
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div3_exhaustive.hip -o /tmp/div3 && /tmp/div3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__global__ void __launch_bounds__(256) k_div3(uint32_t d0, unsigned* __restrict__ nfail, uint32_t* __restrict__ fail_d, uint32_t* __restrict__ fail_a,
                                              unsigned cap) {
    // one workgroup per divisor; its 256 threads stride over the 2^24 dividends
    const uint32_t dbits = 0x3f800000u + d0 + blockIdx.x;
    const float d = __uint_as_float(dbits);
    const float r = 1.0f / d;                                  // correctly rounded (build flag)
    uint32_t first = 0;
    for (uint32_t m = threadIdx.x; m < (1u << 24); m += 256) {
        const float a = __uint_as_float(0x3f800000u + m);
        const float q0 = a * r;
        const float e = __builtin_fmaf(-q0, d, a);
#ifdef DIV3_CONTROL          // negative control: without the correction step the checker must report failures
        const float q = q0; (void)e;
#else
        const float q = __builtin_fmaf(e, r, q0);
#endif
        const float want = (float)((double)a / (double)d);     // binary64 quotient of two binary32 numbers rounds correctly to binary32
        if (__float_as_uint(q) != __float_as_uint(want) && first == 0) first = 0x3f800000u + m;
    }
    if (first) {
        const unsigned i = atomicAdd(nfail, 1u);
        if (i < cap) { fail_d[i] = dbits; fail_a[i] = first; }
    }
}

int main() {
    const unsigned cap = 1u << 20;
    unsigned* nfail; uint32_t *fd, *fa;
    hipMalloc(&nfail, 4); hipMalloc(&fd, cap * 4); hipMalloc(&fa, cap * 4);
    hipMemset(nfail, 0, 4);
    const uint32_t per = 1u << 16;                             // divisors per launch
#ifdef DIV3_CONTROL
    const uint32_t total = per;
#else
    const uint32_t total = 1u << 23;
#endif
    for (uint32_t d0 = 0; d0 < total; d0 += per) {
        hipLaunchKernelGGL(k_div3, dim3(per), dim3(256), 0, 0, d0, nfail, fd, fa, cap);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed at d0 = %u\n", d0); return 1; }
        if ((d0 / per) % 16 == 15) { unsigned n; hipMemcpy(&n, nfail, 4, hipMemcpyDeviceToHost); printf("# %u of 2^23 divisors done, %u failing so far\n", d0 + per, n); fflush(stdout); }
    }
    unsigned n; hipMemcpy(&n, nfail, 4, hipMemcpyDeviceToHost);
    std::vector<uint32_t> hd(n < cap ? n : cap), ha(hd.size());
    hipMemcpy(hd.data(), fd, hd.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), fa, ha.size() * 4, hipMemcpyDeviceToHost);
    printf("divisors with a failing dividend: %u of 8388608\n", n);
    for (size_t i = 0; i < hd.size() && i < 200; ++i) { float d, a; memcpy(&d, &hd[i], 4); memcpy(&a, &ha[i], 4); printf("  d = %a (0x%08x)  first failing a = %a\n", d, hd[i], a); }
    return 0;
}
