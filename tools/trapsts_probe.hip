// tools/trapsts_probe.hip — run ON THE GPU BOX: do the sticky IEEE exception bits of TRAPSTS accumulate without traps enabled?
// (a per-wave witness "no invalid operation / division by zero happened" would let kernels use forms that are exact except in such cases)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, unsigned* sts) {
    unsigned s0, s1, s2, mode;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(s0));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
    const float x = in[threadIdx.x];
    float a = x * 1.5f + 2.0f;                       // ordinary (inexact at most)
    asm volatile("s_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(s1) : "v"(a));
    float b = (x - x) / (x - x);                     // 0 / 0: invalid
    float c = 1.0f / (x - x);                        // division by zero
    asm volatile("s_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(s2) : "v"(b), "v"(c));
    out[threadIdx.x] = a + b + c;
    if (threadIdx.x == 0) { sts[0] = s0; sts[1] = s1; sts[2] = s2; sts[3] = mode; }
}
int main() {
    float *in, *out; unsigned* sts;
    hipMalloc(&in, 256); hipMalloc(&out, 256); hipMalloc(&sts, 16);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i;
    hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, sts);
    unsigned r[4]; hipMemcpy(r, sts, 16, hipMemcpyDeviceToHost);
    printf("TRAPSTS at start 0x%08x, after ordinary arithmetic 0x%08x, after 0/0 and 1/0 0x%08x; MODE 0x%08x (EXCP_EN = bits 12-20)\n", r[0], r[1], r[2], r[3]);
    printf("EXCP bits (0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 underflow, 5 inexact): start %03x ordinary %03x after %03x\n", r[0] & 0x1ff, r[1] & 0x1ff, r[2] & 0x1ff);
    return 0;
}
