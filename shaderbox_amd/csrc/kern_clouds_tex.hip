// shaderbox_amd/csrc/kern_clouds_tex.hip — APP_CLOUDS compiled with USE_NOISE_TEX.
//
// Follows /root/reference/src/app_clouds.h with USE_NOISE_TEX defined (:9): density_func :62-86 takes its
// shape from u_tex_noise (t1) and erodes it with u_tex_noise_2 (t2) through remap() (src/util.h:127-138):
//     shape = u_tex_noise.SampleLevel(u_sampler0, pos, 0).r                          :69-70
//     w     = u_tex_noise_2.SampleLevel(u_sampler0, pos, 0).r                        :75-78
//     ww    = mix(w, 1 - w, height) ;  shape = remap(shape, ww * .7, 1, 0, 1)        :79-80
// with `height` = i / steps of the march that samples (:108, :183) — the one place the reference uses it.
// The volumes are the RGBA32F size^3 textures util/ddsvolgen bakes (ddsvolgen.cpp:101-117) and hlsltoy binds
// with a MIN_MAG_MIP_LINEAR / WRAP sampler (util/hlsltoy/src/hlsltoy.cpp:227-249, 437).
//
// The reference leaves the texture filter to the GPU; the sbx spec fixes it ("SampleLevel"):
//   per axis  u = c * size - .5 ;  i = floor(u) ;  f = u - i ;  i0 = wrap(i), i1 = wrap(i0 + 1)
//   texel    = mix(mix(mix(t000, t100, fx), mix(t010, t110, fx), fy),
//                  mix(mix(t001, t101, fx), mix(t011, t111, fx), fy), fz)   with mix(a, b, t) = a (1 - t) + b t
//   wrap(i)  = i - size * floor(i / size), evaluated in binary32 and folded once into [0, size); NaN -> 0
// i.e. texel centres at (i + .5) / size, full binary32 weights (a D3D11 sampler quantises them to 8 bits:
// against real texture hardware parity is unpinned, against the oracle it is bit-exact).
//
// MI355X shape of the problem: this is the one kernel of the library that READS memory — 16 texel fetches per
// density sample, ~157 samples per pixel.  The library keeps its own R32F copy of the .r channel the shader
// uses (8.4 MB per 128^3 volume instead of 33.5 MB: both volumes sit in the 256 MB Infinity Cache and mostly
// in the 4 MB L2 of each XCD), neighbouring pixels of a wave tile sample neighbouring texels, and the straight
// per-lane kernel below needs few registers, so 8 waves per SIMD hide the L2 latency.  HBM traffic stays the
// framebuffer (16 B/pixel) plus one pass over the volumes; roofline and FETCH_SIZE in DESIGN.md §4.8.
#include <cmath>
#include "sbx_device.h"

#ifndef TEX_ZL
#define TEX_ZL 1           // 0: never use the z-only light march (A/B timing)
#endif
#ifndef TEX_MIN_WAVES
#define TEX_MIN_WAVES 5
#endif

namespace sbx {

struct NoiseTex { const float* r; int size; float fsize; double rsize; int lg; };   // rsize = recip64(fsize): fl / size as an exact multiply
                                                                                  // (sbx_math.h div_by); lg = log2(size) when size is a power of two

__device__ __forceinline__ void tex_axis(float c, const NoiseTex& T, int& i0, int& i1, float& f) {
    const float u = c * T.fsize - .5f;
    const float fl = floor_(u);
    f = u - fl;
    float m = fl - T.fsize * floor_(div_by(fl, T.rsize));      // GLSL mod(fl, size) = fl - size * floor(fl / size); the quotient
                                                               // through the binary64 reciprocal is the IEEE quotient, bit for bit
    if (m < 0.f) m += T.fsize;                                 // a rounded quotient can land one period off
    if (m >= T.fsize) m -= T.fsize;
    const int i = (m >= 0.f && m < T.fsize) ? (int)m : 0;      // NaN / infinite coordinates sample texel 0
    i0 = i;
    i1 = (i + 1 == T.size) ? 0 : i + 1;
}

// wave-uniform "any lane" (the opaque form of sbx_hashcache.h: the branch must not be folded back into a per-lane one)
__device__ __forceinline__ bool tex_wave_any(bool x) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(x);
    unsigned any = (unsigned)m | (unsigned)(m >> 32);
    asm volatile("" : "+s"(any));
    return any != 0;
}

// POW2 (both volume sizes are powers of two, decided on the host): for |floor(u)| < 2^31 the spec's wrap
// fl - size * floor(fl / size) — every operation exact for a power-of-two size — IS the two's-complement (int)fl & (size - 1),
// so the float division, two floors, the fold-in selects and the range test of tex_axis collapse into cvt + and.  Coordinates
// beyond that (|c| >= 2^30 / size, or NaN) send the whole wave through the general form.
template <bool POW2>
__device__ __forceinline__ float tex3d_r(const NoiseTex& T, v3 p) {     // SampleLevel(linear, wrap, lod 0).r
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    const float lim = 1073741824.0f / T.fsize;
    if (POW2 && !tex_wave_any(!(abs_(p.x) < lim) || !(abs_(p.y) < lim) || !(abs_(p.z) < lim))) {
        const int mask = T.size - 1;
        const float ux = p.x * T.fsize - .5f, uy = p.y * T.fsize - .5f, uz = p.z * T.fsize - .5f;
        const float lx = floor_(ux), ly = floor_(uy), lz = floor_(uz);
        fx = ux - lx; fy = uy - ly; fz = uz - lz;
        x0 = (int)lx & mask; y0 = (int)ly & mask; z0 = (int)lz & mask;
        x1 = (x0 + 1) & mask; y1 = (y0 + 1) & mask; z1 = (z0 + 1) & mask;
    } else {
        tex_axis(p.x, T, x0, x1, fx);
        tex_axis(p.y, T, y0, y1, fy);
        tex_axis(p.z, T, z0, z1, fz);
    }
    // 32-bit texel indices from one base pointer (a volume has at most 2^30 texels).  Power-of-two sizes: shifts and ORs (the
    // three fields do not overlap) instead of v_mul_lo_u32, a quarter-rate instruction, six times per sample.
    unsigned r0, r1, a0, a1;
    if (POW2) {
        const unsigned k = (unsigned)T.lg;
        r0 = (unsigned)z0 << (2 * k); r1 = (unsigned)z1 << (2 * k);
        a0 = (unsigned)y0 << k; a1 = (unsigned)y1 << k;
    } else {
        const unsigned s1 = (unsigned)T.size;
        r0 = (unsigned)z0 * s1 * s1; r1 = (unsigned)z1 * s1 * s1;
        a0 = (unsigned)y0 * s1; a1 = (unsigned)y1 * s1;
    }
    const float t000 = T.r[r0 + a0 + (unsigned)x0], t100 = T.r[r0 + a0 + (unsigned)x1];
    const float t010 = T.r[r0 + a1 + (unsigned)x0], t110 = T.r[r0 + a1 + (unsigned)x1];
    const float t001 = T.r[r1 + a0 + (unsigned)x0], t101 = T.r[r1 + a0 + (unsigned)x1];
    const float t011 = T.r[r1 + a1 + (unsigned)x0], t111 = T.r[r1 + a1 + (unsigned)x1];
    const float a = mix_(t000, t100, fx), b = mix_(t010, t110, fx);
    const float c = mix_(t001, t101, fx), d = mix_(t011, t111, fx);
    return mix_(mix_(a, b, fy), mix_(c, d, fy), fz);
}

// ---- the z-only light march (ZL kernels: power-of-two volumes, light step without x and y component) --------------------
// illuminate_volume's samples (:106-113) are lp = pos + (j + 1) * L dt.  With L dt = (0, 0, lz) — the reference's default sun
// (0, 0, -1), src/uniform_buffer.h:42 — lp.x = pos.x + 0 and lp.y = pos.y + 0 are the main sample's own coordinates (a -0 turned
// +0 changes no result bit: the next operation is `* fsize - .5`), so for BOTH volumes the x and y texel indices and weights of
// all light samples of a step are the main sample's, and as long as a light sample stays in the main sample's z cell
// (one light step is dt * .001 * size = .16 texels of the 128^3 volume) so are its two bilinear plane values
//     p0 = mix(mix(t000, t100, fx), mix(t010, t110, fx), fy),   p1 = the same in plane z1:
// the sample is mix(p0, p1, fz) with only fz new — the same operations on the same values as tex3d_r, 2 instead of ~110
// instructions and no loads.  A lane that leaves its cell fetches the two planes of its new cell (per-lane branch).
struct TexCell { unsigned x0, x1, a0, a1; float fx, fy, fl, p0, p1; };   // a0/a1: y * size; fl = floor(uz) of the planes held

__device__ __forceinline__ void tex_planes(const NoiseTex& T, TexCell& c, float fl) {      // power-of-two sizes, |fl| < 2^31
    const int mask = T.size - 1;
    const unsigned k = (unsigned)T.lg;
    const int z0 = (int)fl & mask, z1 = (z0 + 1) & mask;
    const unsigned r0 = (unsigned)z0 << (2 * k), r1 = (unsigned)z1 << (2 * k);
    const float t000 = T.r[r0 + c.a0 + c.x0], t100 = T.r[r0 + c.a0 + c.x1];
    const float t010 = T.r[r0 + c.a1 + c.x0], t110 = T.r[r0 + c.a1 + c.x1];
    const float t001 = T.r[r1 + c.a0 + c.x0], t101 = T.r[r1 + c.a0 + c.x1];
    const float t011 = T.r[r1 + c.a1 + c.x0], t111 = T.r[r1 + c.a1 + c.x1];
    c.p0 = mix_(mix_(t000, t100, c.fx), mix_(t010, t110, c.fx), c.fy);
    c.p1 = mix_(mix_(t001, t101, c.fx), mix_(t011, t111, c.fx), c.fy);
    c.fl = fl;
}
// tex3d_r<true>'s fast form, keeping what the light march reuses.  The caller has made the range test.
__device__ __forceinline__ float tex3d_seed(const NoiseTex& T, v3 p, TexCell& c) {
    const int mask = T.size - 1;
    const unsigned k = (unsigned)T.lg;
    const float ux = p.x * T.fsize - .5f, uy = p.y * T.fsize - .5f, uz = p.z * T.fsize - .5f;
    const float lx = floor_(ux), ly = floor_(uy), lz = floor_(uz);
    c.fx = ux - lx; c.fy = uy - ly;
    const float fz = uz - lz;
    const int x0 = (int)lx & mask, y0 = (int)ly & mask;
    c.x0 = (unsigned)x0; c.x1 = (unsigned)((x0 + 1) & mask);
    c.a0 = (unsigned)y0 << k; c.a1 = (unsigned)((y0 + 1) & mask) << k;
    tex_planes(T, c, lz);
    return mix_(c.p0, c.p1, fz);
}
// a light sample: z only
__device__ __forceinline__ float tex3d_z(const NoiseTex& T, float pz, TexCell& c) {
    const float uz = pz * T.fsize - .5f;
    const float lz = floor_(uz);
    if (lz != c.fl) tex_planes(T, c, lz);
    return mix_(c.p0, c.p1, uz - lz);
}

__device__ __forceinline__ float remap_(float v, float omin, float omax, float nmin, float nmax) {   // util.h:127-138
    return nmin + (((v - omin) / (omax - omin)) * (nmax - nmin));
}

template <bool POW2>
__device__ __forceinline__ float tex_density(const FrameClouds& F, const NoiseTex& T1, const NoiseTex& T2, v3 pos_in, float height) {
    const v3 pos = pos_in * .001f;                             // cld_noise_factor :20,66
    float shape = tex3d_r<POW2>(T1, pos);                      // :69-70
    const float w = tex3d_r<POW2>(T2, pos);                    // :75-78
    const float ww = mix_(w, 1.f - w, height);                 // :79
    shape = remap_(shape, ww * .7f, 1.f, 0.f, 1.f);            // :80
    return shape * smoothstep_rd(F.cov, F.cov_rd, shape);      // :83-84 (division by the frame constant through its exact reciprocal)
}

__device__ __forceinline__ float hg_phase_tex(float mu, float g) {   // volumetric.h:27-33, note (4 + PI)
    return (1.f - g * g) / ((4.f + 3.14159265359f) * pow_(1.f + g * g - 2.f * g * mu, 1.5f));
}

template <bool POW2, bool ZL>      // ZL (decided on the host): POW2 and the light step has no x and no y component
__global__ void __launch_bounds__(WG_THREADS, TEX_MIN_WAVES) k_clouds_tex(FrameClouds F, RowMap M, float* __restrict__ out,
                                                            NoiseTex T1, NoiseTex T2, double rsteps, double rlsteps) {
    // rsteps = recip64(float(steps)), rlsteps = recip64(float(lsteps)): `i / steps` and `j / lsteps` as exact multiplies
    // (sbx_math.h div_by: the IEEE quotient, bit for bit; ~13 instead of ~42 issue cycles, 230 times per marching pixel)
    const Pixel px = pixel_of_thread<8>(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, (float)px.x + .5f, (float)px.y + .5f);
    const v3 dir = primary_dir(F.cam, pc);

    float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);        // render_sky_color :36-46
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);
    sky = abs3(sky);

    const float cutoff = dot(dir, V3(0, 1, 0));
    v3 col = sky;
    if (!(cutoff < 0.05f)) {                                   // :212
        const v3 projection = dir / dir.y;                     // render_clouds :153-202
        v3 origin = F.cam.eye + projection * 150.f;
        origin = origin + F.wind_off;
        const float phase = hg_phase_tex(clamp_(dot(F.sun_dir, dir), 0.f, 1.f), .2f);
        const v3 lstep = F.sun_dir * F.dt;
        float transmittance = 1.f, radiance = 0.f, alpha = 0.f, t = 0.f;
        for (int i = 0; i < F.steps; ++i) {
            const float height = div_by((float)i, rsteps);      // :183  i / steps
            const v3 pos = origin + t * projection;
            t += F.dt;
            // ZL: the main sample through the seeding form of the filter when every coordinate of the wave — and the far end of its
            // light march — is inside the fast form's range (wave-uniform; otherwise this step runs the general code below)
            TexCell c1, c2;
            bool fast = false;
            float density;
            if (ZL) {
                const v3 q = pos * .001f;
                const float zend = (pos.z + (float)(F.lsteps + 1) * lstep.z) * .001f;
                const float lim = fmin_(1073741824.0f / T1.fsize, 1073741824.0f / T2.fsize) * .5f;
                fast = !tex_wave_any(!(abs_(q.x) < lim) || !(abs_(q.y) < lim) || !(abs_(q.z) < lim) || !(abs_(zend) < lim));
                if (fast) {
                    float shape = tex3d_seed(T1, q, c1);                            // tex_density with the cells kept
                    const float w = tex3d_seed(T2, q, c2);
                    const float ww = mix_(w, 1.f - w, height);
                    shape = remap_(shape, ww * .7f, 1.f, 0.f, 1.f);
                    density = shape * smoothstep_rd(F.cov, F.cov_rd, shape);
                } else {
                    density = tex_density<POW2>(F, T1, T2, pos, height);
                }
            } else {
                density = tex_density<POW2>(F, T1, T2, pos, height);
            }
            if (!(density < .005f)) {                          // integrate_volume :132
                const float T_i = exp_(-density * F.sigma * F.dt);
                transmittance *= T_i;
                v3 lp = pos + lstep;                           // illuminate_volume :91-123
                float ltrans = 1.f;
                if (ZL && fast) {
                    for (int j = 0; j < F.lsteps; ++j) {
                        const float lh = div_by((float)j, rlsteps);                 // :108  j / lsteps
                        const float qz = lp.z * .001f;
                        float shape = tex3d_z(T1, qz, c1);
                        const float w = tex3d_z(T2, qz, c2);
                        const float ww = mix_(w, 1.f - w, lh);
                        shape = remap_(shape, ww * .7f, 1.f, 0.f, 1.f);
                        const float d = shape * smoothstep_rd(F.cov, F.cov_rd, shape);
                        ltrans *= exp_(-d * F.sigma * F.dt);
                        lp.z = lp.z + lstep.z;
                    }
                } else
                for (int j = 0; j < F.lsteps; ++j) {
                    const float lh = div_by((float)j, rlsteps);                     // :108  j / lsteps
                    const float d = tex_density<POW2>(F, T1, T2, lp, lh);
                    ltrans *= exp_(-d * F.sigma * F.dt);
                    lp = lp + lstep;
                }
                radiance += (density * F.sigma) * (ltrans * F.sun_power * phase) * transmittance * F.dt;
                alpha += (1.f - T_i) * (1.f - alpha);
            }
            if (alpha > .999f) break;
        }
        const float a = alpha * smoothstep_(.0f, .2f, cutoff);
        col = abs3(mix3(sky, V3s(radiance), a));               // :215-217
    }
    store_rgba(M, out, px.idx, to_srgb(col));
}

// .r of an RGBA32F volume -> the library's R32F copy (one float4 read, one float written per voxel)
__global__ void __launch_bounds__(256) k_extract_r(const float4* __restrict__ rgba, float* __restrict__ r, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) r[i] = rgba[i].x;
}

void launch_extract_r(const float* rgba, float* r, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_extract_r, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(rgba), r, n);
}

// the filter on its own, straight from an RGBA32F volume (parity tests of the spec): same operations as tex3d_r
__global__ void __launch_bounds__(256) k_tex3d_eval(int size, const float* __restrict__ rgba, const float* __restrict__ xyz,
                                                    float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const NoiseTex T{nullptr, size, (float)size, recip64((float)size), 0};      // (tex_axis does not use lg)
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    tex_axis(xyz[3 * i], T, x0, x1, fx);
    tex_axis(xyz[3 * i + 1], T, y0, y1, fy);
    tex_axis(xyz[3 * i + 2], T, z0, z1, fz);
    auto at = [&](int x, int y, int z) { return rgba[(((size_t)z * size + y) * size + x) * 4]; };
    const float a = mix_(at(x0, y0, z0), at(x1, y0, z0), fx), b = mix_(at(x0, y1, z0), at(x1, y1, z0), fx);
    const float c = mix_(at(x0, y0, z1), at(x1, y0, z1), fx), d = mix_(at(x0, y1, z1), at(x1, y1, z1), fx);
    out[i] = mix_(mix_(a, b, fy), mix_(c, d, fy), fz);
}
void launch_tex3d_eval(int size, const float* rgba, const float* xyz, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_tex3d_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, size, rgba, xyz, out, n);
}

void launch_clouds_tex(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s, const float* shape_r, int shape_size,
                       const float* detail_r, int detail_size) {
    auto lg2 = [](int n) { int k = 0; while ((1 << k) < n) ++k; return k; };
    const NoiseTex T1{shape_r, shape_size, (float)shape_size, recip64((float)shape_size), lg2(shape_size)};
    const NoiseTex T2{detail_r, detail_size, (float)detail_size, recip64((float)detail_size), lg2(detail_size)};
    const bool pow2 = (shape_size & (shape_size - 1)) == 0 && (detail_size & (detail_size - 1)) == 0;
    const double rs = recip64((float)F.steps), rl = recip64((float)F.lsteps);     // loops with 0 steps never use them
    const v3 lstep = F.sun_dir * F.dt;                           // the kernel's own expression
    const bool zl = pow2 && lstep.x == 0.f && lstep.y == 0.f && std::isfinite(lstep.z) && TEX_ZL;
    if (zl) hipLaunchKernelGGL((k_clouds_tex<true, true>), grid_for<8>(M), dim3(WG_THREADS), 0, s, F, M, out, T1, T2, rs, rl);
    else if (pow2) hipLaunchKernelGGL((k_clouds_tex<true, false>), grid_for<8>(M), dim3(WG_THREADS), 0, s, F, M, out, T1, T2, rs, rl);
    else hipLaunchKernelGGL((k_clouds_tex<false, false>), grid_for<8>(M), dim3(WG_THREADS), 0, s, F, M, out, T1, T2, rs, rl);
}

}  // namespace sbx
