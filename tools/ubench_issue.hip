// tools/ubench_issue.hip — VALU issue-rate microbenchmark for gfx950 (MI355X), cycles from the shader clock.
//
// Question it answers: what is the sustained issue cost of the instructions k_clouds is made of, in SHADER
// CYCLES (s_memtime; the wall clock is reported beside it, so DVFS is visible), as a function of waves per SIMD and of the
// operand pattern?  round 1's tools/ubench_valu.hip divided wall time by a nominal 2.4 GHz and used three distinct VGPR
// sources per FMA; it read 2.87 "cycles" for v_fma_f32 where MI355X_MICROARCH.md says 2.
//
// Method: every wave runs ITERS iterations of a block of UNROLL x CHAINS inline-asm instructions on CHAINS independent
// registers (dependent distance = CHAINS instructions), reads s_memtime before and after, and stores the difference.
// With W waves resident per SIMD, cycles per wave-instruction per SIMD = mean wave cycles / (W x instructions per wave).
// One workgroup = 256 threads = one wave per SIMD; W = workgroups per CU = grid / 256 CUs (all resident: few registers).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_issue tools/ubench_issue.hip ; run: build/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int CHAINS = 8, UNROLL = 16, ITERS = 400;

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
__device__ __forceinline__ unsigned long long realtime() {     // constant 100 MHz
    unsigned long long t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

struct Stamp { unsigned long long cyc, ref; };

#define BENCH_F32(name, INSN)                                                                        \
__global__ void __launch_bounds__(256) k_##name(Stamp* out, float* sink, float b, float c) {         \
    float a[CHAINS];                                                                                 \
    for (int i = 0; i < CHAINS; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;                        \
    float vb = b + (float)threadIdx.x * 1e-9f, vc = c;                                               \
    asm volatile("" : "+v"(vb), "+v"(vc));                                                           \
    const unsigned long long r0 = realtime(), t0 = memtime();                                        \
    for (int it = 0; it < ITERS; ++it) {                                                             \
        _Pragma("unroll") for (int j = 0; j < UNROLL; ++j) {                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { INSN; }                             \
        }                                                                                            \
    }                                                                                                \
    const unsigned long long t1 = memtime(), r1 = realtime();                                        \
    float s = 0; for (int i = 0; i < CHAINS; ++i) s += a[i];                                         \
    sink[blockIdx.x * 256 + threadIdx.x] = s;                                                        \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};\
}

// operand patterns of v_fma_f32
BENCH_F32(fma_vvv3, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(vb), "v"(vc)))       // 3 distinct VGPR sources
BENCH_F32(fma_self, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])))                          // one VGPR source
BENCH_F32(fma_vss, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(b)))                  // VGPR + one SGPR (twice)
BENCH_F32(fma_vlit, asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(a[i])))                        // inline constants
BENCH_F32(mul_vv, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(mul_vs, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b)))
BENCH_F32(mul_self, asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i])))
BENCH_F32(add_vv, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(add_self, asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i])))
BENCH_F32(sub_lit, asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[i])))
BENCH_F32(floor_, asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(cvt_i32, asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(and_lit, asm volatile("v_and_b32 %0, 63, %0" : "+v"(a[i])))
BENCH_F32(lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(cmp_vv, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(vb) : "vcc"))
BENCH_F32(cmp_sdst, asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(vb) : "s20", "s21"))
BENCH_F32(cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(mov, asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(vb)))
BENCH_F32(exp_, asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(rcp_, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(sqrt_, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(readlane, asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20"))
BENCH_F32(readfirst, asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20"))
BENCH_F32(writelane, asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a[i]) : "s"(b)))
BENCH_F32(salu_add, asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc"))
BENCH_F32(salu_and64, asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc"))
BENCH_F32(mul_lit32, asm volatile("v_mul_f32 %0, 0x4028f5c3, %0" : "+v"(a[i])))                      // 32-bit literal (2.64f)
BENCH_F32(sub_lit32, asm volatile("v_sub_f32 %0, 0x40400000, %0" : "+v"(a[i])))                      // 3.0f is not an inline constant
BENCH_F32(sub_vs, asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b)))
BENCH_F32(add_vs, asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b)))
BENCH_F32(mov_lit32, asm volatile("v_mov_b32 %0, 0x4028f5c3" : "=v"(a[i])))
BENCH_F32(mov_s, asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(b)))
BENCH_F32(cmp_cnd, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(vb) : "vcc"))
BENCH_F32(cnd_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(cnd_exec, asm volatile("v_cndmask_b32_e64 %0, %0, %1, exec" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(lshlrev, asm volatile("v_lshlrev_b32 %0, 5, %0" : "+v"(a[i])))
BENCH_F32(add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(fract_, asm volatile("v_fract_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(rndne_, asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(trunc_, asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i])))
BENCH_F32(cvt_f_i, asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i])))
BENCH_F32(max_vv, asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(min_lit, asm volatile("v_min_f32 %0, 1.0, %0" : "+v"(a[i])))
BENCH_F32(med3, asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[i])))
// BENCH_F32(fma_lit32, asm volatile("v_fma_f32 %0, %0, %1, 0x40400000" : "+v"(a[i]) : "v"(vb)))
BENCH_F32(fmac, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(vb), "v"(vc)))
BENCH_F32(fmaak, asm volatile("v_fmaak_f32 %0, %0, %1, 0x40400000" : "+v"(a[i]) : "v"(vb)))
// BENCH_F32(pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&vb)))
BENCH_F32(mov_dpp, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(vb)))
BENCH_F32(swizzle, asm volatile("ds_swizzle_b32 %0, %0 offset:0x041f\n s_waitcnt lgkmcnt(0)" : "+v"(a[i])))
BENCH_F32(bperm, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(vb)))
// mixed stream of the kernel's inner blend: mul, mul, add on independent chains with a scalar operand in one of them
BENCH_F32(mix3, asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %2, %0\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(vb), "s"(c)))
// one VALU + one SALU per pair (does scalar work share the issue slot?)
BENCH_F32(valu_salu, asm volatile("v_mul_f32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(vb) : "s20", "scc"))

#define BENCH_F64(name, INSN)                                                                        \
__global__ void __launch_bounds__(256) k_##name(Stamp* out, float* sink, float bf, float cf) {       \
    double a[CHAINS];                                                                                \
    for (int i = 0; i < CHAINS; ++i) a[i] = (double)(threadIdx.x + i) * 1e-3;                        \
    double vb = bf + (double)threadIdx.x * 1e-12, vc = cf;                                           \
    float fa[CHAINS];                                                                                \
    for (int i = 0; i < CHAINS; ++i) fa[i] = (float)(threadIdx.x + i) * 1e-3f;                       \
    asm volatile("" : "+v"(vb), "+v"(vc));                                                           \
    const unsigned long long r0 = realtime(), t0 = memtime();                                        \
    for (int it = 0; it < ITERS; ++it) {                                                             \
        _Pragma("unroll") for (int j = 0; j < UNROLL; ++j) {                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { INSN; }                             \
        }                                                                                            \
    }                                                                                                \
    const unsigned long long t1 = memtime(), r1 = realtime();                                        \
    double s = 0; for (int i = 0; i < CHAINS; ++i) s += a[i] + fa[i];                                \
    sink[blockIdx.x * 256 + threadIdx.x] = (float)s;                                                 \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};\
}
BENCH_F64(fma64_vvv, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(vb), "v"(vc)))
BENCH_F64(fma64_vss, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "s"(vb)))
BENCH_F64(fma64_self, asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i])))
BENCH_F64(mul64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F64(add64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(vb)))
BENCH_F64(cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(fa[i])))
BENCH_F64(cvt_f32_f64, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[i]) : "v"(a[i])))
BENCH_F64(rndne64, asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i])))
BENCH_F64(cvt_i32_f64, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(fa[i]) : "v"(a[i])))

typedef void (*kern_t)(Stamp*, float*, float, float);
struct Case { const char* name; kern_t k; int insn_per_stmt; };

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device: %s  CUs=%d  nominal clock=%d MHz  CHAINS=%d UNROLL=%d ITERS=%d\n", prop.name, cus, prop.clockRate / 1000, CHAINS, UNROLL, ITERS);
    printf("# cyc = shader cycles (s_memtime) per wave-instruction per SIMD; GHz = s_memtime / s_memrealtime (100 MHz) over the loop\n");
    const int max_w = 8;
    Stamp* out; float* sink;
    CK(hipMalloc(&out, (size_t)cus * max_w * 4 * sizeof(Stamp)));
    CK(hipMalloc(&sink, (size_t)cus * max_w * 256 * sizeof(float)));
    std::vector<Stamp> h((size_t)cus * max_w * 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Case> cases = {
#define C(n, k) {#n, k_##n, k}
        C(fma_vvv3, 1), C(fma_self, 1), C(fma_vss, 1), C(fma_vlit, 1), C(mul_vv, 1), C(mul_vs, 1), C(mul_self, 1), C(add_vv, 1),
        C(add_self, 1), C(sub_lit, 1), C(floor_, 1), C(cvt_i32, 1), C(and_lit, 1), C(lshl_add, 1), C(cmp_vv, 1), C(cmp_sdst, 1),
        C(cndmask, 1), C(mov, 1), C(exp_, 1), C(rcp_, 1), C(sqrt_, 1), C(readlane, 1), C(readfirst, 1), C(writelane, 1),
        C(salu_add, 1), C(salu_and64, 1), C(mix3, 3), C(valu_salu, 2),
        C(mul_lit32, 1), C(sub_lit32, 1), C(sub_vs, 1), C(add_vs, 1), C(mov_lit32, 1), C(mov_s, 1), C(cmp_cnd, 2), C(cnd_e64, 1),
        C(cnd_exec, 1), C(lshlrev, 1), C(add_u32, 1), C(mad_u24, 1), C(fract_, 1), C(rndne_, 1), C(trunc_, 1), C(cvt_f_i, 1),
        C(max_vv, 1), C(min_lit, 1), C(med3, 1), C(fmac, 1), C(fmaak, 1), C(mov_dpp, 1), C(swizzle, 1), C(bperm, 1),
        C(fma64_vvv, 1), C(fma64_vss, 1), C(fma64_self, 1), C(mul64, 1), C(add64, 1), C(cvt_f64_f32, 1), C(cvt_f32_f64, 1),
        C(rndne64, 1), C(cvt_i32_f64, 1),
    };
    const int ws[] = {1, 2, 4, 5, 8};
    printf("%-14s", "instruction");
    for (int w : ws) printf("  W=%d cyc  wall  GHz ", w);
    printf("\n");
    for (auto& c : cases) {
        printf("%-14s", c.name);
        for (int w : ws) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, sink, 1.0001f, 0.5f);   // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, sink, 1.0001f, 0.5f);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float wall_ms; CK(hipEventElapsedTime(&wall_ms, e0, e1));
            CK(hipMemcpy(h.data(), out, (size_t)blocks * 4 * sizeof(Stamp), hipMemcpyDeviceToHost));
            double cyc = 0, ref = 0;
            for (int i = 0; i < blocks * 4; ++i) { cyc += (double)h[i].cyc; ref += (double)h[i].ref; }
            cyc /= blocks * 4; ref /= blocks * 4;
            const double insns = (double)ITERS * UNROLL * CHAINS * c.insn_per_stmt;
            const double ghz = cyc / ref * 0.1;
            // wall view: the whole launch (ramp and drain included): SIMD-cycles per wave-instruction at the measured clock
            const double wall_cyc = wall_ms * 1e-3 * ghz * 1e9 * (cus * 4.0) / ((double)blocks * 4 * insns);
            printf("  %6.2f %6.2f %4.2f ", cyc / (w * insns), wall_cyc, ghz);
        }
        printf("\n");
    }
    return 0;
}
