// shaderbox_amd/csrc/kern_atmosphere.hip — APP_ATMOSPHERE: Rayleigh/Mie single scattering.
//
// Follows /root/reference/src/app_atmosphere.h (FROM_SPACE defined, :162): render :183-228 (sky
// dome branch :190-209), get_incident_light :78-160 (16 view samples), get_sun_light :50-76
// (8 light samples), isect_sphere :15-26, phase functions src/volumetric.h:13-33 (hg_g = .76).
// sun_dir (a _mutable global rotated by setup_scene, :177-181) is a frame constant built on the
// host from (0,1,0), i.e. it restarts from its initialiser for every pixel (GLSL semantics).
// Magnitudes are ~6.4e6 in binary32, so the evaluation order below is part of the result.
#include <cmath>
#ifndef ATM_NO_FIN
#define ATM_NO_FIN 0
#endif
#include "sbx_atmosphere.h"

namespace sbx {

#ifndef ATM_MIN_WAVES
#define ATM_MIN_WAVES 6     // (the max-ILP strategy of this source, build.py, takes what registers it is given: 147 for the plain kernel without a bound)
#endif
template <bool FIN, int PREC = 0>
__global__ void __launch_bounds__(64 * ATM_TX, ATM_MIN_WAVES) k_atmosphere(FrameAtmosphere F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
    constexpr bool T64 = FIN && ATM_EXP_REG && ATM_EXP64 && !ATM_EXP4K;
    __shared__ double etab[32];
    __shared__ double etab64[T64 ? 64 : 1];               // exp_reg64_'s table (builds without exp_reg4k_)
    if (threadIdx.x < 32) etab[threadIdx.x] = kExp2Tab[threadIdx.x];
    if (T64 && threadIdx.x < 64) etab64[threadIdx.x] = kExp2Tab64[threadIdx.x];
    if (ATM_TX > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const Pixel px = pixel_of_thread<8, ATM_TX>(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);

    // sky dome mapping :195-207
    const float z2 = pc.x * pc.x + pc.y * pc.y;
#ifndef ATM_FREE_EXIT
#define ATM_FREE_EXIT 1
#endif
    // Outside the dome's circle z2 = 2 the argument of acos is below -1 (z2 > 2 is a multiple of ulp(2), so 1 - z2 is exact and
    // < -1): acos_ returns its NaN, the direction is NaN in all three components, isect_sphere's `d2 < radius2` (:25) is false and
    // get_incident_light returns (0, 0, 0) (:85-88) — 28 % of a 16:9 frame.  A wave whose pixels are ALL out there (the test is
    // wave-uniform) skips the atan2 / acos / two sin / two cos of the mapping, ~600 instructions, and encodes that black.
    if (ATM_FREE_EXIT && __builtin_amdgcn_ballot_w64(!(z2 > 2.0f)) == 0ull) {
        tile_cost_store(M, tl_t0);
        store_rgba(M, out, px.idx, to_srgb(V3(0.f, 0.f, 0.f)));
        return;
    }
    const float phi = atan2_(pc.y, pc.x);
    const float theta = acos_(1.0f - z2);
    const v3 rd = V3(sin_(theta) * cos_(phi), cos_(theta), sin_(theta) * sin_(phi));
    const v3 ro = V3(0, ATM_EARTH_R + 1.f, 0);

    const v3 col = atm_incident_light<FIN, PREC>(ro, rd, F.sun_dir, etab, etab64);
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(col));
}

// exp_reg4k_ as a standalone function (sbx_math_eval "exp_reg4k"): the same table, read from global memory, the same instruction
// sequence as inside k_atmosphere — for the exhaustive comparison with exp_
__global__ void __launch_bounds__(256) k_exp4k_eval(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = exp_reg4k_(a[i], kExp2Tab4096);
}
void launch_exp4k_eval(const float* a, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_exp4k_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, out, n);
}

dim3 atmosphere_grid(const RowMap& M) { return grid_for<8, ATM_TX>(M); }

void launch_atmosphere(const FrameAtmosphere& F, const RowMap& M, float* out, hipStream_t s, int precision) {
    // FIN: camera and sun direction are finite numbers (they are for every finite u_res / u_time)
    const float chk[] = {F.cam.res_x, F.cam.res_y, F.cam.aspect_x, F.cam.fov, F.sun_dir.x, F.sun_dir.y, F.sun_dir.z,
                         F.cam.fwd.x, F.cam.fwd.y, F.cam.fwd.z, F.cam.up.x, F.cam.up.y, F.cam.up.z, F.cam.right.x, F.cam.right.y, F.cam.right.z};
    bool fin = std::isfinite(F.cam.rres_x) && std::isfinite(F.cam.rres_y);
    for (float v : chk) fin = fin && std::isfinite(v);
#if ATM_NO_FIN
    fin = false;
#endif
    if (fin && precision == 1) hipLaunchKernelGGL((k_atmosphere<true, 1>), (grid_for<8, ATM_TX>(M)), dim3(64 * ATM_TX), 0, s, F, M, out);   // the tolerance tier
    else if (fin) hipLaunchKernelGGL(k_atmosphere<true>, (grid_for<8, ATM_TX>(M)), dim3(64 * ATM_TX), 0, s, F, M, out);
    else hipLaunchKernelGGL(k_atmosphere<false>, (grid_for<8, ATM_TX>(M)), dim3(64 * ATM_TX), 0, s, F, M, out);
}

}  // namespace sbx
