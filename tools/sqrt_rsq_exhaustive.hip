// tools/sqrt_rsq_exhaustive.hip — run ON THE GPU BOX: is
//     rs = v_rsq_f32(x);  y0 = x * rs;  h = .5 * rs;  r = fma(-y0, y0, x);  y = fma(r, h, y0)
// the IEEE square root for every binary32 x?  (sqrt is scale-free in steps of 4: two binades of significands would do, but all
// 2^31 positive values take a second.)  Prints the number of differing arguments per exponent range and the first few.
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/sqrt_rsq_exhaustive.hip -o build/sqrt_rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void __launch_bounds__(256) k(unsigned long long* __restrict__ bad_by_exp, uint32_t* __restrict__ first, unsigned* nfirst) {
    const uint32_t bits = blockIdx.x * 256u + threadIdx.x + (blockIdx.y << 24);
    if (bits >= 0x7f800000u) return;
    const float x = __uint_as_float(bits);
    const float rs = __builtin_amdgcn_rsqf(x);
    const float y0 = x * rs;
    const float h = .5f * rs;
    const float r = __builtin_fmaf(-y0, y0, x);
    const float y = __builtin_fmaf(r, h, y0);
    const float want = (float)__builtin_sqrt((double)x);       // binary64 sqrt of a binary32 number rounds correctly to binary32
    if (__float_as_uint(y) != __float_as_uint(want)) {
        atomicAdd(&bad_by_exp[bits >> 23], 1ull);
        const unsigned i = atomicAdd(nfirst, 1u);
        if (i < 64) first[i] = bits;
    }
}
int main() {
    unsigned long long* bad; uint32_t* first; unsigned* nf;
    hipMalloc(&bad, 256 * 8); hipMalloc(&first, 64 * 4); hipMalloc(&nf, 4);
    hipMemset(bad, 0, 256 * 8); hipMemset(nf, 0, 4);
    hipLaunchKernelGGL(k, dim3(1u << 16, 128), dim3(256), 0, 0, bad, first, nf);
    hipDeviceSynchronize();
    unsigned long long h[256]; uint32_t f[64]; unsigned n;
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost); hipMemcpy(&n, nf, 4, hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (int e = 0; e < 255; ++e) if (h[e]) { printf("biased exponent %3d (2^%d): %llu differing\n", e, e - 127, h[e]); tot += h[e]; }
    printf("total differing: %llu of 2139095040 positive finite arguments\n", tot);
    for (unsigned i = 0; i < n && i < 16; ++i) { float x; memcpy(&x, &f[i], 4); printf("  x = %a\n", x); }
    return 0;
}
