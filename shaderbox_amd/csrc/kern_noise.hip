// shaderbox_amd/csrc/kern_noise.hip — the noise library as standalone kernels.
//
// noise_worley.h is named by the north star but sits on no default app path (only the texture-baked
// USE_NOISE_TEX variant of APP_CLOUDS, app_func.h and util/ddsvolgen use it), so
// it is delivered as library functions: hash_w (/root/reference/src/noise_worley.h:5-17), noise_w
// (:20-51), the 4-octave tiled fBm DECL_FBM_FUNC_TILE(fbm_worley_tile, 4, (1. - (noise_w(p, L).r + .25)))
// (src/fbm.h:8, util/ddsvolgen/src/ddsvolgen.cpp:52), and the 128^3 volume ddsvolgen bakes (:101-117).
// hash_w multiplies sin by 43758.5453123, so only the shared correctly rounded sin gives parity.
#include "sbx_device.h"
#include "sbx_noise.h"
#include "sbx_witness.h"

namespace sbx {

__device__ __forceinline__ v3 hash_w(v3 x) {                               // noise_worley.h:5-17
    const v3 xx = V3(dot(x, V3(127.1f, 311.7f, 74.7f)), dot(x, V3(269.5f, 183.3f, 246.1f)),
                     dot(x, V3(113.5f, 271.9f, 124.6f)));
    return V3(fract_(sin_(xx.x) * 43758.5453123f), fract_(sin_(xx.y) * 43758.5453123f),
              fract_(sin_(xx.z) * 43758.5453123f));
}

// closest, second closest, |cell id| over the 27 neighbour cells, domain repeating every `rep`  :20-51
__device__ __forceinline__ v3 noise_w(v3 pos, float rep) {
    const v3 x = pos * rep;
    const v3 p = V3(floor_(x.x), floor_(x.y), floor_(x.z));
    const v3 f = V3(x.x - p.x, x.y - p.y, x.z - p.z);
    float id = 0.0f, r0 = 100.0f, r1 = 100.0f;
    for (int k = -1; k <= 1; k++)
        for (int j = -1; j <= 1; j++)
            for (int i = -1; i <= 1; i++) {
                const v3 b = V3((float)i, (float)j, (float)k);
                const v3 pb = p + b;
                const v3 r = b - f + hash_w(V3(mod_(pb.x, rep), mod_(pb.y, rep), mod_(pb.z, rep)));
                const float d = dot(r, r);
                if (d < r0) {
                    id = dot(p + b, V3(1.0f, 57.0f, 113.0f));
                    r1 = r0;
                    r0 = d;
                } else if (d < r1) {
                    r1 = d;
                }
            }
    return V3(sqrt_(r0), sqrt_(r1), abs_(id));
}

__device__ __forceinline__ float fbm_worley_tile(v3 pos, float lacunarity, float init_gain, float gain) {
    float H = init_gain, L = lacunarity, t = 0.f;
    for (int i = 0; i < 4; ++i) {                     // p is never scaled: the octave scale is the repeat L
        t += (1.f - (noise_w(pos, L).x + .25f)) * H;
        L *= lacunarity;
        H *= gain;
    }
    return t;
}

__global__ void __launch_bounds__(256) k_noise_eval(int fn, const float* __restrict__ xyz, float p0, float p1, float p2,
                                                     float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v3 p = V3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    v3 r = V3(0, 0, 0);
    switch (fn) {
    case 0: r.x = noise_iq(p); break;
    case 1: r = hash_w(p); break;
    case 2: r = noise_w(p, p0); break;
    case 3: r.x = fbm_worley_tile(p, p0, p1, p2); break;
    // test hooks of sbx_witness.h: normalize in its IEEE form, in the recorded-domain fast form, and the record itself
    case 4: r = normalize(p); break;
    case 5: { Wit<true> w; r = w.normalize(p); break; }
    case 6: { Wit<true> w; (void)w.normalize(p); r.x = w.bad ? 1.f : 0.f; break; }
    }
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

// one thread per voxel, x fastest: float4 stores coalesce along x
__global__ void __launch_bounds__(256) k_worley_volume(int size, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)size * size * size;
    if (i >= total) return;
    const int x = (int)(i % size), y = (int)((i / size) % size), z = (int)(i / ((size_t)size * size));
    const v3 pos = (V3((float)x, (float)y, (float)z) + .5f) / (float)size;          // ddsvolgen.cpp:107
    out[i] = make_float4(fbm_worley_tile(pos, 2.f, 1.f, .5f), 0.f, 0.f, 0.f);       // :52-61,108-111
}

int launch_noise_eval(int fn, const float* xyz, const float* par, float* out, size_t n, hipStream_t s) {
    if (fn < 0 || fn > 6) return -1;
    hipLaunchKernelGGL(k_noise_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, xyz, par[0], par[1], par[2], out, n);
    return 0;
}
void launch_worley_volume(int size, float* out, hipStream_t s) {
    const size_t total = (size_t)size * size * size;
    hipLaunchKernelGGL(k_worley_volume, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, size,
                       reinterpret_cast<float4*>(out));
}

}  // namespace sbx
