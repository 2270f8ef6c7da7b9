// tools/ubench_valu.hip — instruction-issue microbenchmark for gfx950 (MI355X).
//
// Measures the sustained wave-instruction rate of the VALU/LDS instructions the sbx
// kernels are built from (fp32 scalar and packed, fp64, conversions, transcendentals,
// LDS reads), so that design choices (packed math? fp64 angle addition? LDS trig tables?)
// rest on measured issue costs instead of datasheet guesses.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip
// Run  : build/ubench_valu            (prints one line per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 8;   // independent chains per thread
constexpr int INNER = 16;   // asm statements per chain per loop iteration

// one asm op on a float chain
#define OP_F32(name, insn)                                                              \
__global__ void __launch_bounds__(256) k_##name(float* out, int iters, float b, float c) { \
    float a[UNROLL];                                                                    \
    for (int i = 0; i < UNROLL; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;           \
    for (int it = 0; it < iters; ++it) {                                                \
        _Pragma("unroll") for (int j = 0; j < INNER; ++j) {                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                          \
                asm volatile(insn : "+v"(a[i]) : "v"(b), "v"(c));                       \
        }                                                                               \
    }                                                                                   \
    float s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i];                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                     \
}

OP_F32(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP_F32(mul_f32, "v_mul_f32 %0, %0, %1")
OP_F32(add_f32, "v_add_f32 %0, %0, %1")
OP_F32(floor_f32, "v_floor_f32 %0, %0")
OP_F32(rndne_f32, "v_rndne_f32 %0, %0")
OP_F32(exp_f32, "v_exp_f32 %0, %0")
OP_F32(rcp_f32, "v_rcp_f32 %0, %0")
OP_F32(sqrt_f32, "v_sqrt_f32 %0, %0")
OP_F32(sin_f32, "v_sin_f32 %0, %0")
OP_F32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP_F32(cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
OP_F32(xor_b32, "v_xor_b32 %0, %0, %1")
OP_F32(lshl_b32, "v_lshlrev_b32 %0, 1, %0")
OP_F32(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OP_F32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")

typedef float float2v __attribute__((ext_vector_type(2)));
#define OP_PK(name, insn)                                                               \
__global__ void __launch_bounds__(256) k_##name(float* out, int iters, float b, float c) { \
    float2v a[UNROLL]; float2v bb = {b, b}, cc = {c, c};                                \
    for (int i = 0; i < UNROLL; ++i) a[i] = float2v{(float)(threadIdx.x + i) * 1e-3f, 1.f}; \
    for (int it = 0; it < iters; ++it) {                                                \
        _Pragma("unroll") for (int j = 0; j < INNER; ++j) {                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                          \
                asm volatile(insn : "+v"(a[i]) : "v"(bb), "v"(cc));                     \
        }                                                                               \
    }                                                                                   \
    float s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i].x + a[i].y;                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                     \
}
OP_PK(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP_PK(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OP_PK(pk_add_f32, "v_pk_add_f32 %0, %0, %1")

#define OP_F64(name, insn)                                                              \
__global__ void __launch_bounds__(256) k_##name(float* out, int iters, float bf, float cf) { \
    double a[UNROLL]; double b = bf, c = cf;                                            \
    for (int i = 0; i < UNROLL; ++i) a[i] = (double)(threadIdx.x + i) * 1e-3;           \
    for (int it = 0; it < iters; ++it) {                                                \
        _Pragma("unroll") for (int j = 0; j < INNER; ++j) {                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                          \
                asm volatile(insn : "+v"(a[i]) : "v"(b), "v"(c));                       \
        }                                                                               \
    }                                                                                   \
    double s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i];                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                              \
}
OP_F64(fma_f64, "v_fma_f64 %0, %0, %1, %2")
OP_F64(mul_f64, "v_mul_f64 %0, %0, %1")
OP_F64(add_f64, "v_add_f64 %0, %0, %1")

// conversions: f32 -> f64 -> f32 round trip (2 instructions per statement)
__global__ void __launch_bounds__(256) k_cvt_roundtrip(float* out, int iters, float b, float c) {
    float a[UNROLL];
    for (int i = 0; i < UNROLL; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INNER; ++j) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                double d;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a[i]));
                asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d));
            }
        }
    }
    float s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// IEEE fp32 division as the compiler emits it (v_div_scale/v_rcp/fma.../v_div_fixup)
__global__ void __launch_bounds__(256) k_div_ieee(float* out, int iters, float b, float c) {
    float a[UNROLL];
    for (int i = 0; i < UNROLL; ++i) a[i] = (float)(threadIdx.x + i + 1) * 1e3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INNER; ++j) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) { a[i] = a[i] / b; asm volatile("" : "+v"(a[i])); }
        }
    }
    float s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_sqrt_ieee(float* out, int iters, float b, float c) {
    float a[UNROLL];
    for (int i = 0; i < UNROLL; ++i) a[i] = (float)(threadIdx.x + i + 1) * 1e3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INNER; ++j) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) { a[i] = __builtin_sqrtf(a[i]) + b; asm volatile("" : "+v"(a[i])); }
        }
    }
    float s = 0; for (int i = 0; i < UNROLL; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS reads at pseudo-random (per-lane) 16-byte-aligned addresses out of a 32 KiB table
template <int BYTES>
__global__ void __launch_bounds__(256) k_lds_read(float* out, int iters, float b, float c) {
    __shared__ __attribute__((aligned(16))) float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = (float)i;
    __syncthreads();
    unsigned idx[UNROLL];
    for (int i = 0; i < UNROLL; ++i) idx[i] = (threadIdx.x * 2654435761u + i * 40503u) >> 8;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INNER; ++j) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                unsigned o = (idx[i] & 2047u) * 4u;   // float index, 16-byte aligned
                if (BYTES == 16) { float4 v = *(const float4*)&tab[o]; acc += v.x + v.w; }
                else if (BYTES == 8) { float2 v = *(const float2*)&tab[o]; acc += v.x + v.y; }
                else { acc += tab[o]; }
                idx[i] = idx[i] * 1664525u + 1013904223u + (unsigned)j;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, int, float, float);

struct Case { const char* name; kern_t k; double insn_per_stmt; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("# device: %s  CUs=%d  clock=%d MHz  (UNROLL=%d INNER=%d)\n", prop.name, cus, prop.clockRate / 1000, UNROLL, INNER);
    const int blocks = cus * 8, threads = 256;   // 8 blocks x 4 waves = 32 waves per CU
    float* out; CK(hipMalloc(&out, (size_t)blocks * threads * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Case> cases = {
        {"v_fma_f32", k_fma_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_add_f32", k_add_f32, 1},
        {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_mul_f32", k_pk_mul_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1},
        {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
        {"v_cvt_f64_f32+v_cvt_f32_f64", k_cvt_roundtrip, 2},
        {"v_floor_f32", k_floor_f32, 1}, {"v_rndne_f32", k_rndne_f32, 1}, {"v_cvt_i32_f32", k_cvt_i32_f32, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_f32", k_cmp_f32, 1}, {"v_xor_b32", k_xor_b32, 1},
        {"v_lshlrev_b32", k_lshl_b32, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
        {"v_exp_f32", k_exp_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1}, {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_sin_f32", k_sin_f32, 1},
        {"a/b (IEEE f32 div seq)", k_div_ieee, 1}, {"sqrtf (IEEE seq)+add", k_sqrt_ieee, 1},
        {"ds_read_b32 random", k_lds_read<4>, 1}, {"ds_read_b64 random", k_lds_read<8>, 1}, {"ds_read_b128 random", k_lds_read<16>, 1},
    };
    printf("%-30s %12s %14s %16s\n", "instruction", "ms", "Gwave-instr/s", "cycles/instr/SIMD@2.4GHz");
    for (auto& c : cases) {
        int iters = 2000;
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0001f, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double stmts = (double)blocks * (threads / 64) * (double)iters * INNER * UNROLL;   // wave-level statements
        double rate = stmts * c.insn_per_stmt / (ms * 1e-3);
        double cyc = (double)cus * 4 * 2.4e9 / rate;
        printf("%-30s %12.3f %14.1f %16.2f\n", c.name, ms, rate / 1e9, cyc);
    }
    return 0;
}
