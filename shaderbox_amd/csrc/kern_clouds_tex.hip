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
// framebuffer (16 B/pixel) plus one pass over the volumes; roofline and FETCH_SIZE in DESIGN.md §5.6.
#include <cmath>
#include <algorithm>
#include "sbx_device.h"
#include "sbx_exp4k_table.h"

#ifndef TEX_ZL
#define TEX_ZL 1           // 0: never use the z-only light march (A/B timing)
#endif
#ifndef TEX_MIN_WAVES
#define TEX_MIN_WAVES 5
#endif
#ifndef TEX_TW
#define TEX_TW 32          // wave tile TEX_TW x 64/TEX_TW pixels
#endif
#ifndef TEX_SKIP
#define TEX_SKIP 1         // REG frames: no exp of a zero density
#endif
#ifndef TEX_YTAB
#define TEX_YTAB 1         // the y terms of the main march (and of the z-only light march) from a per-workgroup table in LDS
#endif
#ifndef TEX_EXP_SMALL
#define TEX_EXP_SMALL 1    // exp_small_ where every active lane's argument lies in its domain [-0.205, 0] (tested per call)
#endif
#ifndef TEX_DIVN
#define TEX_DIVN 1        // 0: remap's division always through the IEEE expansion (A/B timing)
#endif
#ifndef TEX_XB
#define TEX_XB 1          // 0: never exp_reg4k_ (A/B timing)
#endif
#ifndef TEX_LH
#define TEX_LH 1          // 0: `j / lsteps` computed per light sample (A/B timing)
#endif
#ifndef TEX_TX
#define TEX_TX 4           // waves per workgroup
#endif
#ifndef TEX_YROWS_N
#define TEX_YROWS_N 256
#endif
constexpr int TEX_YROWS = TEX_YROWS_N;   // march steps the LDS table covers (8 KB per workgroup); more steps: the per-lane form

namespace sbx {

struct NoiseTex { const float* r; int size; float fsize; double rsize; int lg; };   // rsize = recip64(fsize): fl / size as an exact multiply
                                                                                  // (sbx_math.h div_by); lg = log2(size) when size is a power of two

// One texel of a power-of-two volume by its 32-bit BYTE offset (size <= 512: at most 2^29 bytes): the address is the uniform base
// plus a zero-extended 32-bit register, which is exactly what global_load_dword's scalar-base form takes — written as T.r[index]
// the compiler builds a 64-bit address per texel (v_lshl_add_u64, 16 per density sample).
__device__ __forceinline__ float texel_b(const NoiseTex& T, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(T.r) + byte_off);
}

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
    if (POW2) {                                                // byte offsets (texel_b): the POW2 kernels run for sizes <= 512 only
        const unsigned k = (unsigned)T.lg;
        const unsigned b0 = (unsigned)x0 << 2, b1 = (unsigned)x1 << 2;
        r0 = (unsigned)z0 << (2 * k + 2); r1 = (unsigned)z1 << (2 * k + 2);
        a0 = (unsigned)y0 << (k + 2); a1 = (unsigned)y1 << (k + 2);
        const float t000 = texel_b(T, r0 + a0 + b0), t100 = texel_b(T, r0 + a0 + b1);
        const float t010 = texel_b(T, r0 + a1 + b0), t110 = texel_b(T, r0 + a1 + b1);
        const float t001 = texel_b(T, r1 + a0 + b0), t101 = texel_b(T, r1 + a0 + b1);
        const float t011 = texel_b(T, r1 + a1 + b0), t111 = texel_b(T, r1 + a1 + b1);
        const float a = mix_(t000, t100, fx), b = mix_(t010, t110, fx);
        const float c = mix_(t001, t101, fx), d = mix_(t011, t111, fx);
        return mix_(mix_(a, b, fy), mix_(c, d, fy), fz);
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
struct TexCell { unsigned x0, x1, a0, a1; float fx, fy, fl, p0, p1; };   // x0/x1, a0/a1: BYTE offsets x * 4, y * size * 4; fl = floor(uz) of the planes held

// (fs, mask: the volume's size as a float and size - 1, handed in by the caller in VGPRs — an SGPR source halves the issue rate of
//  a VALU instruction on gfx950, profiles/r02_ubench_issue.txt, and these are used ~10 times per sample)
__device__ __forceinline__ void tex_planes(const NoiseTex& T, TexCell& c, float fl, int mask) {      // power-of-two sizes, |fl| < 2^31
    const unsigned k = (unsigned)T.lg;
    const int z0 = (int)fl & mask, z1 = (z0 + 1) & mask;
    const unsigned r0 = (unsigned)z0 << (2 * k + 2), r1 = (unsigned)z1 << (2 * k + 2);
    const float t000 = texel_b(T, r0 + c.a0 + c.x0), t100 = texel_b(T, r0 + c.a0 + c.x1);
    const float t010 = texel_b(T, r0 + c.a1 + c.x0), t110 = texel_b(T, r0 + c.a1 + c.x1);
    const float t001 = texel_b(T, r1 + c.a0 + c.x0), t101 = texel_b(T, r1 + c.a0 + c.x1);
    const float t011 = texel_b(T, r1 + c.a1 + c.x0), t111 = texel_b(T, r1 + c.a1 + c.x1);
    c.p0 = mix_(mix_(t000, t100, c.fx), mix_(t010, t110, c.fx), c.fy);
    c.p1 = mix_(mix_(t001, t101, c.fx), mix_(t011, t111, c.fx), c.fy);
    c.fl = fl;
}
// tex3d_r<true>'s fast form, keeping what the light march reuses.  The caller has made the range test.
__device__ __forceinline__ float tex3d_seed(const NoiseTex& T, v3 p, TexCell& c, float fs, int mask) {
    const unsigned k = (unsigned)T.lg;
    const float ux = p.x * fs - .5f, uy = p.y * fs - .5f, uz = p.z * fs - .5f;
    const float lx = floor_(ux), ly = floor_(uy), lz = floor_(uz);
    c.fx = ux - lx; c.fy = uy - ly;
    const float fz = uz - lz;
    const int x0 = (int)lx & mask, y0 = (int)ly & mask;
    c.x0 = (unsigned)x0 << 2; c.x1 = (unsigned)((x0 + 1) & mask) << 2;
    c.a0 = (unsigned)y0 << (k + 2); c.a1 = (unsigned)((y0 + 1) & mask) << (k + 2);
    tex_planes(T, c, lz, mask);
    return mix_(c.p0, c.p1, fz);
}
// the same with the y terms of the step given (frame constants: the march's y is the same for every pixel, see k_clouds_tex)
__device__ __forceinline__ float tex3d_seed_y(const NoiseTex& T, float px, float pz, float fy, unsigned a0, unsigned a1, TexCell& c,
                                              float fs, int mask) {
    const float ux = px * fs - .5f, uz = pz * fs - .5f;
    const float lx = floor_(ux), lz = floor_(uz);
    c.fx = ux - lx; c.fy = fy;
    const float fz = uz - lz;
    const int x0 = (int)lx & mask;
    c.x0 = (unsigned)x0 << 2; c.x1 = (unsigned)((x0 + 1) & mask) << 2;
    c.a0 = a0; c.a1 = a1;
    tex_planes(T, c, lz, mask);
    return mix_(c.p0, c.p1, fz);
}
// a light sample: z only
__device__ __forceinline__ float tex3d_z(const NoiseTex& T, float pz, TexCell& c, float fs, int mask) {
    const float uz = pz * fs - .5f;
    const float lz = floor_(uz);
    if (lz != c.fl) tex_planes(T, c, lz, mask);
    return mix_(c.p0, c.p1, uz - lz);
}

// The kernel's exp: exp_small_ (10 instructions) when all active lanes' arguments lie in [-0.205, 0] — most samples: the argument
// is -density * sigma * dt with densities well below the bound the host can prove — else exp_reg4k_ (xb) or exp_ itself.  All three
// are equal to exp_ on their domains, so the choice changes no bit.  A NaN fails the test.
__device__ __forceinline__ float tex_exp(float x, int xb) {
    if (TEX_EXP_SMALL && __builtin_amdgcn_ballot_w64(!(x >= EXP_SMALL_MIN && x <= 0.f)) == 0ull) return exp_small_(x);
    return xb ? exp_reg4k_(x, kExp2Tab4096) : exp_(x);
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

// REG (decided on the host per launch, clouds_tex_regular): the coverage edge, sigma and dt are finite.  Then, and only then,
//      a zero density needs no exp: -(+-0) * sigma * dt = -+0 and exp(-+0) = 1 exactly, so `ltrans *= 1` is skipped;
//      (Deciding a zero density from the SHAPE sample alone — with a detail volume known to lie in [0, 1], a shape value below
//      the coverage edge cannot be lifted over it by the remap — was built and measured: per lane +8 %, per wave +5 % at
//      3840x2160; the detail sample is cheap next to the branch.  Not in.)
// YT (decided on the host: power-of-two volumes, steps <= TEX_YROWS, the march's y range inside the fast filter's domain):
//      render_clouds marches along dir / dir.y, whose y component is exactly 1 (:165), from origin.y = (eye.y + 150) + wind.y, so
//      the height of main step i, `i / steps`, and per volume floor / fract of uy and the two row offsets are the same for
//      every pixel: each workgroup computes them once into LDS (thread i the row of step i, the operations a lane would do) and
//      the march reads the row with two broadcast loads.  The z-only light march keeps x and y of its main sample: same row.
// xb (decided on the host, clouds_tex_exp_bound): REG, and the texel values of both volumes (scanned when they were bound) bound
//      every density so that |density * sigma * dt| <= 80: the exps are exp_reg4k_ of sbx_math.h (no range guard, 4096-entry table,
//      degree 3: 15 instructions against exp_'s 21), equal to exp_ on that whole range.
// dv (clouds_tex_bounds): xb, and the texel ranges keep remap's divisor 1 - .7 ww within [1e-3, 2^40] and its dividend below 2^40, with
//      the coverage edge in [2^-20, 2^20]: the division is divn_ of sbx_math.h (six instructions, equal to the IEEE quotient for all
//      significand pairs).  A dividend so small that its quotient is below 2^-89 — both forms then differ from the IEEE quotient at
//      most in such a value — is far under the coverage edge: the density is quotient * smoothstep(..) = quotient * 0, a zero of the
//      quotient's sign either way (and remap's `0 +` turns a zero quotient into +0 in both).
struct TexArgs { NoiseTex T1, T2; double rsteps, rlsteps; int xb, dv; };
#define TEX_EXP(x) tex_exp((x), A.xb)
// remap(v, o, 1, 0, 1) = 0 + ((v - o) / (1 - o)) * (1 - 0)     util.h:127-138 (the multiply by 1 and the sum with 0 kept: -0 -> +0)
#define TEX_REMAP(v, o) ((TEX_DIVN && A.dv) ? 0.f + (divn_((v) - (o), 1.f - (o)) * (1.f - 0.f)) : remap_((v), (o), 1.f, 0.f, 1.f))
constexpr int TEX_LH_N = 64;     // light steps whose `j / lsteps` comes from the workgroup's LDS table (more: computed per sample)
template <bool POW2, bool ZL, bool REG, bool YT>      // ZL (decided on the host): POW2 and the light step has no x and no y component
__global__ void __launch_bounds__(64 * TEX_TX, TEX_MIN_WAVES) k_clouds_tex(FrameClouds F, RowMap M, float* __restrict__ out, TexArgs A) {
    // rsteps = recip64(float(steps)), rlsteps = recip64(float(lsteps)): `i / steps` and `j / lsteps` as exact multiplies
    // (sbx_math.h div_by: the IEEE quotient, bit for bit; ~13 instead of ~42 issue cycles, 230 times per marching pixel)
    const NoiseTex& T1 = A.T1;
    const NoiseTex& T2 = A.T2;
    const double rsteps = A.rsteps, rlsteps = A.rlsteps;
    // illuminate_volume's height `j / lsteps` (:108) is the same for every pixel: once per workgroup into LDS, one broadcast read per
    // light sample instead of cvt + binary64 multiply + cvt
    __shared__ float lh_tab[TEX_LH_N];
    if (threadIdx.x < TEX_LH_N) lh_tab[threadIdx.x] = div_by((float)(int)threadIdx.x, rlsteps);
    __shared__ float4 yrow_f[YT ? TEX_YROWS : 1];          // {height, fy of volume 1, fy of volume 2, -}
    __shared__ uint4 yrow_i[YT ? TEX_YROWS : 1];           // {a0, a1 of volume 1, a0, a1 of volume 2}: byte offsets y0 * size * 4, y1 * size * 4
    if (YT) {
        for (int i = threadIdx.x; i < F.steps; i += 64 * TEX_TX) {
            float t = 0.f;
            for (int j = 0; j < i; ++j) t += F.dt;                              // t after i steps of `t += dt` (:186)
            const float origin_y = (F.cam.eye.y + 1.0f * 150.f) + F.wind_off.y;  // :166-167 with projection.y = 1
            const float qy = (origin_y + t * 1.0f) * .001f;                      // :185, :66
            const float u1 = qy * T1.fsize - .5f, u2 = qy * T2.fsize - .5f;      // tex3d_seed's y terms, both volumes
            const float l1 = floor_(u1), l2 = floor_(u2);
            const int y1 = (int)l1 & (T1.size - 1), y2 = (int)l2 & (T2.size - 1);
            yrow_f[i] = make_float4(div_by((float)i, rsteps), u1 - l1, u2 - l2, 0.f);
            yrow_i[i] = make_uint4((unsigned)y1 << (T1.lg + 2), (unsigned)((y1 + 1) & (T1.size - 1)) << (T1.lg + 2),
                                   (unsigned)y2 << (T2.lg + 2), (unsigned)((y2 + 1) & (T2.size - 1)) << (T2.lg + 2));
        }
    }
    if (TEX_TX > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const bool lh_lds = TEX_LH && F.lsteps <= TEX_LH_N;
    const Pixel px = pixel_of_thread<TEX_TW, TEX_TX>(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
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
        float fs1 = T1.fsize, fs2 = T2.fsize;
        int m1 = T1.size - 1, m2 = T2.size - 1;
        float vcov = F.cov, vsig = F.sigma, vdt = F.dt;
        asm volatile("" : "+v"(fs1), "+v"(fs2), "+v"(m1), "+v"(m2), "+v"(vcov), "+v"(vsig), "+v"(vdt));   // VGPR-resident (full-rate operands)
        const float lim = fmin_(1073741824.0f / T1.fsize, 1073741824.0f / T2.fsize) * .5f;
        // YT: the fast filter's range test once per pixel instead of once per step — |origin| + reach bounds every sample of the
        // march (either sign of dt) and of its light march in z; y is the table's (checked on the host)
        bool fast_all = false;
        if (YT) {
            const float tend = (float)F.steps * abs_(F.dt) * 1.001f;
            const float zreach = (float)(F.lsteps + 1) * abs_(lstep.z);
            const float ex = abs_(origin.x) + tend * abs_(projection.x) * 1.001f;           // |origin + t proj| for any t in [-tend, tend]
            const float ez = abs_(origin.z) + tend * abs_(projection.z) * 1.001f + zreach;
            fast_all = !tex_wave_any(!(ex * .001f < lim * .5f) || !(ez * .001f < lim * .5f));
        }
        float transmittance = 1.f, radiance = 0.f, alpha = 0.f, t = 0.f;
        for (int i = 0; i < F.steps; ++i) {
            const v3 pos = origin + t * projection;
            t += F.dt;
            // ZL: the main sample through the seeding form of the filter when every coordinate of the wave — and the far end of its
            // light march — is inside the fast form's range (wave-uniform; otherwise this step runs the general code below)
            TexCell c1, c2;
            bool fast = false;
            float density;
            float height;
            if (YT && fast_all) {
                const float4 rf = yrow_f[i];                                        // all lanes one address: a broadcast
                const uint4 ri = yrow_i[i];
                height = rf.x;
                fast = true;
                const float qx = pos.x * .001f, qz = pos.z * .001f;
                float shape = tex3d_seed_y(T1, qx, qz, rf.y, ri.x, ri.y, c1, fs1, m1);
                const float w = tex3d_seed_y(T2, qx, qz, rf.z, ri.z, ri.w, c2, fs2, m2);
                const float ww = mix_(w, 1.f - w, height);
                shape = TEX_REMAP(shape, ww * .7f);
                density = REG ? x_smoothstep_rd_med3(vcov, F.cov_rd, shape) : shape * smoothstep_rd(F.cov, F.cov_rd, shape);
            } else {
                height = div_by((float)i, rsteps);                                  // :183  i / steps
                if (ZL) {
                    const v3 q = pos * .001f;
                    const float zend = (pos.z + (float)(F.lsteps + 1) * lstep.z) * .001f;
                    fast = !tex_wave_any(!(abs_(q.x) < lim) || !(abs_(q.y) < lim) || !(abs_(q.z) < lim) || !(abs_(zend) < lim));
                    if (fast) {
                        float shape = tex3d_seed(T1, q, c1, fs1, m1);                            // tex_density with the cells kept
                        const float w = tex3d_seed(T2, q, c2, fs2, m2);
                        const float ww = mix_(w, 1.f - w, height);
                        shape = TEX_REMAP(shape, ww * .7f);
                        density = shape * smoothstep_rd(F.cov, F.cov_rd, shape);
                    } else {
                        density = tex_density<POW2>(F, T1, T2, pos, height);
                    }
                } else {
                    density = tex_density<POW2>(F, T1, T2, pos, height);
                }
            }
            if (!(density < .005f)) {                          // integrate_volume :132
                const float T_i = TEX_EXP(-density * F.sigma * F.dt);
                transmittance *= T_i;
                v3 lp = pos + lstep;                           // illuminate_volume :91-123
                float ltrans = 1.f;
                if (ZL && fast) {
                    for (int j = 0; j < F.lsteps; ++j) {
                        const float qz = lp.z * .001f;
                        float shape = tex3d_z(T1, qz, c1, fs1, m1);
                        lp.z = lp.z + lstep.z;
                        const float lh = lh_lds ? lh_tab[j] : div_by((float)j, rlsteps);   // :108  j / lsteps
                        const float w = tex3d_z(T2, qz, c2, fs2, m2);
                        const float ww = mix_(w, 1.f - w, lh);
                        shape = TEX_REMAP(shape, ww * .7f);
                        const float d = REG ? x_smoothstep_rd_med3(vcov, F.cov_rd, shape) : shape * smoothstep_rd(F.cov, F.cov_rd, shape);
                        if (REG && TEX_SKIP && d == 0.f) continue;                  // exp(-+0) = 1
                        ltrans *= TEX_EXP(-d * vsig * vdt);
                    }
                } else
                for (int j = 0; j < F.lsteps; ++j) {
                    const float lh = lh_lds ? lh_tab[j] : div_by((float)j, rlsteps);   // :108  j / lsteps
                    const float d = tex_density<POW2>(F, T1, T2, lp, lh);
                    lp = lp + lstep;
                    if (REG && TEX_SKIP && d == 0.f) continue;                      // exp(-+0) = 1
                    ltrans *= TEX_EXP(-d * F.sigma * F.dt);
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

// min / max of the R32F copy (NaN texels set the flag): the bounds the host derives the exp form from (clouds_tex_exp_bound).
// res[0] = min, res[1] = max as order-preserving unsigned keys, res[2] = 1 if a NaN was seen; initialised by the launcher.
__device__ __forceinline__ unsigned f_key(float x) { const unsigned u = f2u(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__global__ void __launch_bounds__(256) k_minmax_r(const float* __restrict__ r, size_t n, unsigned* __restrict__ res) {
    unsigned lo = 0xffffffffu, hi = 0u, nan = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = r[i];
        if (v != v) { nan = 1u; continue; }
        const unsigned k = f_key(v);
        lo = k < lo ? k : lo; hi = k > hi ? k : hi;
    }
    atomicMin(&res[0], lo); atomicMax(&res[1], hi);
    if (nan) atomicOr(&res[2], 1u);
}
void launch_minmax_r(const float* r, size_t n, unsigned* res, hipStream_t s) {      // res: 3 device words, set to {~0, 0, 0} here
    (void)hipMemsetAsync(res, 0xff, sizeof(unsigned), s);
    (void)hipMemsetAsync(res + 1, 0, 2 * sizeof(unsigned), s);
    hipLaunchKernelGGL(k_minmax_r, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, r, n, res);
}
float minmax_key_to_float(unsigned k) { return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

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

// "regular frame" for the REG shortcuts of k_clouds_tex: everything they rely on is checked here, per launch
static bool clouds_tex_regular(const FrameClouds& F) {
    return std::isfinite(F.cov) && std::isfinite(F.cov_rd) && F.cov_rd > 0.0 && std::isfinite(F.sigma) && std::isfinite(F.dt);
}
// |density| <= B for every sample, from the texel ranges [lo1, hi1] (shape) and [lo2, hi2] (detail): a trilinear blend is a convex
// combination (binary32 weights in [0, 1]) and stays within the texels' range up to rounding; ww = mix(w, 1 - w, h) with h in [0, 1]
// lies between min(lo2, 1 - hi2) and max(hi2, 1 - lo2); remap = (shape - .7 ww) / (1 - .7 ww); density = remap * smoothstep in [0, 1].
// Returns < 0 when there is no bound (unknown or non-finite ranges, or 1 - .7 ww can come near 0).
static double clouds_tex_density_bound(const float* b, bool* dv_ok) {      // b = {lo1, hi1, lo2, hi2}
    if (dv_ok) *dv_ok = false;
    for (int i = 0; i < 4; ++i) if (!std::isfinite(b[i])) return -1.0;
    const double wlo = std::min((double)b[2], 1.0 - b[3]), whi = std::max((double)b[3], 1.0 - b[2]);
    const double olo = std::min(.7 * wlo, .7 * whi) - 1e-5, ohi = std::max(.7 * wlo, .7 * whi) + 1e-5;
    const double bmin = 1.0 - ohi;
    if (!(bmin >= 1e-3)) return -1.0;
    const double amax = std::max(std::fabs(b[0] - ohi), std::fabs(b[1] - olo)) + 1e-5;
    if (dv_ok) *dv_ok = amax <= 0x1p40 && (1.0 - olo) <= 0x1p40;       // divn_'s exponent ranges (bmin >= 1e-3 already)
    return amax / bmin * 1.001;
}
void launch_clouds_tex(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s, const float* shape_r, int shape_size,
                       const float* detail_r, int detail_size, const float* bounds) {
    auto lg2 = [](int n) { int k = 0; while ((1 << k) < n) ++k; return k; };
    const NoiseTex T1{shape_r, shape_size, (float)shape_size, recip64((float)shape_size), lg2(shape_size)};
    const NoiseTex T2{detail_r, detail_size, (float)detail_size, recip64((float)detail_size), lg2(detail_size)};
    const bool pow2 = (shape_size & (shape_size - 1)) == 0 && (detail_size & (detail_size - 1)) == 0 &&
                      shape_size <= 512 && detail_size <= 512;   // 32-bit byte offsets (texel_b): 2^29 bytes at most
    const double rs = recip64((float)F.steps), rl = recip64((float)F.lsteps);     // loops with 0 steps never use them
    const v3 lstep = F.sun_dir * F.dt;                           // the kernel's own expression
    const bool zl = pow2 && lstep.x == 0.f && lstep.y == 0.f && std::isfinite(lstep.z) && TEX_ZL;
    const bool reg = clouds_tex_regular(F);
    // the LDS y table: the march's y range (the kernel's own expressions for its two ends) inside the fast filter's domain
    bool yt = false;
    if (TEX_YTAB && pow2 && F.steps > 0 && F.steps <= TEX_YROWS && std::isfinite(F.dt)) {
        const float origin_y = (F.cam.eye.y + 1.0f * 150.f) + F.wind_off.y;
        const float far = std::fabs(origin_y) + (float)F.steps * std::fabs(F.dt) * 1.001f;
        const float lim = std::fmin(1073741824.0f / T1.fsize, 1073741824.0f / T2.fsize) * .25f;
        yt = far * .001f < lim;                                  // NaN compares false
    }
    // exp_reg4k_'s domain: |density * sigma * dt| <= 80 (bounds == nullptr: the volumes were bound without a scan)
    bool dv_ok = false;
    const double dens_max = bounds ? clouds_tex_density_bound(bounds, &dv_ok) : -1.0;
    const int xb = (TEX_XB && reg && dens_max >= 0.0 && dens_max * std::fabs((double)F.sigma) * std::fabs((double)F.dt) * 1.001 <= 80.0) ? 1 : 0;
    const int dv = (TEX_DIVN && xb && dv_ok && std::fabs(F.cov) >= 0x1p-20f && std::fabs(F.cov) <= 0x1p20f && F.cov > 0.f) ? 1 : 0;
    const TexArgs A{T1, T2, rs, rl, xb, dv};
    const dim3 grid = grid_for<TEX_TW, TEX_TX>(M), block(64 * TEX_TX);
#define SBX_TEX_LAUNCH(P, Z) do {                                                                                          \
        if (reg && yt) hipLaunchKernelGGL((k_clouds_tex<P, Z, true, true>), grid, block, 0, s, F, M, out, A);               \
        else if (reg) hipLaunchKernelGGL((k_clouds_tex<P, Z, true, false>), grid, block, 0, s, F, M, out, A);              \
        else if (yt) hipLaunchKernelGGL((k_clouds_tex<P, Z, false, true>), grid, block, 0, s, F, M, out, A);               \
        else hipLaunchKernelGGL((k_clouds_tex<P, Z, false, false>), grid, block, 0, s, F, M, out, A);                      \
    } while (0)
    if (zl) SBX_TEX_LAUNCH(true, true);
    else if (pow2) SBX_TEX_LAUNCH(true, false);
    else hipLaunchKernelGGL((k_clouds_tex<false, false, false, false>), grid, block, 0, s, F, M, out, A);
#undef SBX_TEX_LAUNCH
}

}  // namespace sbx
