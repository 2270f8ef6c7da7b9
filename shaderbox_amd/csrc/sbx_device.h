// shaderbox_amd/csrc/sbx_device.h — thread->pixel mapping and framebuffer store shared by the kernels.
//
// One thread per pixel.  A workgroup is 256 threads = 4 wave64; each wave owns a small pixel tile
// (8x8 by default, lane = ly*TW + lx) so that the rays of a wave stay coherent (same lattice cells,
// same march exits); the workgroup covers 4 tiles side by side.  Every lane stores one float4 (16 B):
// a wave writes full 128..512-byte row segments, fully coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include "sbx_frame.h"

namespace sbx {

// Tile shape of a wave (TW x 64/TW pixels) is chosen per kernel from measurements on MI355X
// (profiles/r01_tile_shapes.txt): wide tiles suit APP_CLOUDS (a pixel row shares dir.y, hence march
// length and lit pattern), 16x4 suits APP_EGG, 8x8 the others.  Tile heights divide the 8-row blocks of the
// multi-GPU split.
#ifndef SBX_WG_WAVES
#define SBX_WG_WAVES 4
#endif
constexpr int WG_TILES_X = SBX_WG_WAVES, WG_THREADS = 64 * SBX_WG_WAVES;

#ifndef SBX_PIXEL_POINTS
#define SBX_PIXEL_POINTS 1     // A/B switches (tools/ab_build.py): the point-list and span forms of pixel_of compiled in
#endif
#ifndef SBX_PIXEL_SPANS
#define SBX_PIXEL_SPANS 1
#endif
struct Pixel { int x, y; size_t idx; bool valid; float fx, fy; };   // (fx, fy) = fragCoord: the pixel centre, or a listed point

// TOP_FIRST: workgroups are dispatched in increasing blockIdx (x fastest, then y); with TOP_FIRST the first ones take
// the TOP rows of the launch.  For APP_CLOUDS the bottom rows never march (src/app_clouds.h:212), so the launch then ends
// on its cheapest waves and the end-of-launch drain is not a few 100-step marches on a mostly empty chip.
// the tile a workgroup renders: its own index, or — with a dispatch-order table (RowMap.order) — the tile the table names (wave-uniform)
__device__ __forceinline__ void ordered_tile(const RowMap& M, int& bx, int& by) {
    if (M.order) {
        const unsigned t = M.order[by * (int)gridDim.x + bx];
        bx = (int)(t & 0xffffu);
        by = (int)(t >> 16);
    }
}
// a wave's duration into the cost table of the next frame's dispatch order (lane 0 of the workgroup's first wave; t0: s_memrealtime
// at the kernel's start)
// (the same for a kernel with an order of its own — APP_EGG's hot-first —: (bx, by) = the tile it hands to pixel_of.  Until this
// existed k_egg's costs were filed under the WORKGROUP's index, not the tile's: the table built from them was noise, and "the
// order loses 15 % on APP_EGG" was a measurement of that)
__device__ __forceinline__ void tile_cost_store_at(const RowMap& M, unsigned long long t0, int bx, int by) {
    if (M.cost && threadIdx.x == 0) {
        ordered_tile(M, bx, by);
        M.cost[by * (int)gridDim.x + bx] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
    }
}
__device__ __forceinline__ void tile_cost_store(const RowMap& M, unsigned long long t0) {
    if (M.cost && threadIdx.x == 0) {
        int bx = (int)blockIdx.x, by = (int)blockIdx.y;
        ordered_tile(M, bx, by);
        M.cost[by * (int)gridDim.x + bx] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
    }
}
template <int TW = 8, int TX = WG_TILES_X, bool TOP_FIRST = false>
__device__ __forceinline__ Pixel pixel_of(const RowMap& M, int tid, int bx, int by_in, int grid_y) {
    constexpr int TH = 64 / TW;
    const int lane = tid & 63, wave = tid >> 6;
    ordered_tile(M, bx, by_in);
    Pixel p;
    p.x = bx * (TW * TX) + wave * TW + (lane % TW);
    // (dispatching the block rows from the middle of the launch outwards — heaviest tiles first for centred scenes — LOSES:
    // EGG 1080p 0.54 -> 0.77 ms, VINYL 4K 3.73 -> 4.07: heavy waves that run together share their SIMDs' issue slots.
    // Re-measured after the hit block left the trace loop: one EGG 1080p launch 0.269 -> 0.241 ms, but 0.160 -> 0.168 ms per
    // frame with two in flight and 4K 0.632 -> 0.648 ms: a shorter tail, lower throughput.  Not in.)
    // (so does visiting the rows with a stride: APP_CLOUDS 4K 2.86 ms in row order, 2.87 / 2.91 / 3.41 with strides 7 / 37 / 269)
    const int by = TOP_FIRST ? (grid_y - 1 - by_in) : by_in;
    const int r = by * TH + (lane / TW);
    p.valid = (p.x < M.width) && (r < M.nrows);
    if (SBX_PIXEL_POINTS && M.frag) {                        // wave-uniform (a kernel argument): point list, see RowMap
        p.idx = (size_t)r * M.width + p.x;
        p.valid = p.valid && p.idx < (size_t)M.npoints;
        p.y = r;
        p.fx = p.valid ? M.frag[2 * p.idx] : .5f;
        p.fy = p.valid ? M.frag[2 * p.idx + 1] : .5f;
        return p;
    }
    p.y = row_to_y(M, r);
    if (SBX_PIXEL_SPANS && M.span_mode) {                    // wave-uniform: the span forms of the multi-GPU split, see RowMap
        const int yc = p.y < M.height ? p.y : M.height - 1;  // (rows past the frame are invalid anyway; keep the read in range)
        const int g = yc / M.block_rows;
        const int4 T = M.span[g];
        if (M.span_mode == 1) {
            p.x += T.x;
            p.valid = p.valid && p.x < T.y;
            p.idx = (size_t)T.z + (size_t)(yc - g * M.block_rows) * (size_t)(T.y - T.x) + (size_t)(p.x - T.x);
        } else if (M.span_mode == 3) {                       // a peer of the store exchange: its spans, at their place in the owner's frame
            p.valid = p.valid && p.x >= T.x && p.x < T.y;
            p.idx = (size_t)p.y * M.width + p.x;
        } else {
            p.valid = p.valid && (T.w == 0 || p.x < T.x || p.x >= T.y);
            p.idx = (size_t)p.y * M.width + p.x;
        }
        p.fx = (float)p.x + .5f;
        p.fy = (float)p.y + .5f;
        return p;
    }
    p.idx = (size_t)(M.in_place ? p.y : r) * M.width + p.x;
    p.fx = (float)p.x + .5f;                                 // fragCoord of pixel (x, y) is its centre
    p.fy = (float)p.y + .5f;
    return p;
}
template <int TW = 8, int TX = WG_TILES_X, bool TOP_FIRST = false>
__device__ __forceinline__ Pixel pixel_of_thread(const RowMap& M) {
    return pixel_of<TW, TX, TOP_FIRST>(M, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
template <int TW = 8, int TX = WG_TILES_X>
inline dim3 grid_for(const RowMap& M) {
    constexpr int TH = 64 / TW, W = TW * TX;
    return dim3((M.width + W - 1) / W, (M.nrows + TH - 1) / TH);
}
// float -> UNORM8 by the Direct3D rule (the write into hlsltoy's R8G8B8A8_UNORM back buffer, util/hlsltoy/src/hlsltoy.cpp:79,192):
// NaN -> 0, clamp to [0, 1], scale by 255, add .5, truncate
__device__ __forceinline__ unsigned unorm8_(float v) {
    if (!(v > 0.f)) return 0u;                       // NaN, -x, -0, +0
    if (v > 1.f) v = 1.f;
    return (unsigned)(v * 255.f + .5f);
}
__device__ __forceinline__ void store_rgba(const RowMap& M, float* out, size_t idx, v3 c) {
    if (M.rgb == 1) {                                                          // wave-uniform (a kernel argument)
        float* o = out + idx * 3;                                              // 64 lanes: 768 contiguous bytes
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
    } else if (M.rgb == 2) {                                                   // SBX_FORMAT_RGBA8: one 4-byte store per pixel, alpha 255
        reinterpret_cast<unsigned*>(out)[idx] = unorm8_(c.x) | (unorm8_(c.y) << 8) | (unorm8_(c.z) << 16) | 0xff000000u;
    } else if (M.rgb == 3) {                                                   // float4 pixels whose alpha is already in the frame: one
        float* o = out + idx * 4;                                              // three-dword store (the store exchange, sbx_shared_*:
        o[0] = c.x; o[1] = c.y; o[2] = c.z;                                    // 12 instead of 16 bytes per pixel over xGMI)
    } else {
        reinterpret_cast<float4*>(out)[idx] = make_float4(c.x, c.y, c.z, 1.0f);   // main.h:52
    }
}

// launchers (one per app), defined next to their kernels
void launch_clouds(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s, int variant, void* ytab, int ytab_rows,
                   bool build_table);
constexpr int CLOUDS_YTAB_ROWS = 4096;      // march steps covered by the per-frame y table (192 KB per table; beyond it the
                                            // table-less kernels run: 1100 steps at 4K took 48 ms without a table)
constexpr int CLOUDS_YTAB_BYTES = CLOUDS_YTAB_ROWS * 48;
constexpr int CLOUDS_YTAB_BIG_MAX = 1 << 20;   // longest march served by the context's one on-demand table (48 MB); beyond it: the table-less kernels
constexpr int CLOUDS_YTAB_RING = 8;         // eager tables: one per REBUILD (key change), round robin; reuse of a slot waits
                                            // for the launches that may still read it (sbx_capi.hip render_clouds)
constexpr int CLOUDS_YTAB_CAPTURE = 8;      // tables used only by launches recorded into a stream capture
void launch_clouds_tex(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s, const float* shape_r, int shape_size,
                       const float* detail_r, int detail_size, const float* bounds);     // bounds: {lo1, hi1, lo2, hi2} of the texels, or NULL
void launch_minmax_r(const float* r, size_t n, unsigned* res, hipStream_t s);
float minmax_key_to_float(unsigned k);
void launch_extract_r(const float* rgba, float* r, size_t n, hipStream_t s);
void launch_tex3d_eval(int size, const float* rgba, const float* xyz, float* out, size_t n, hipStream_t s);
void launch_egg(const FrameEgg& F, const RowMap& M, float* out, hipStream_t s, int variant, void* side);   // side: egg_side_create() or nullptr
void* egg_side_create();
void egg_side_destroy(void* side);
hipError_t bind_fault_egg(unsigned* word);
void launch_raytracer(const FrameRaytracer& F, const RowMap& M, float* out, hipStream_t s, int variant);
void launch_atmosphere(const FrameAtmosphere& F, const RowMap& M, float* out, hipStream_t s, int precision = 0);
void launch_sdf_ao(const FrameSdfAo& F, const RowMap& M, float* out, hipStream_t s, int variant);
void launch_vinyl(const FrameVinyl& F, const RowMap& M, float* out, hipStream_t s, int variant);
void launch_clouds_best(const FrameCloudsBest& F, const RowMap& M, float* out, hipStream_t s);
void launch_clouds_ue4(const FrameCloudsUe4& F, const RowMap& M, float* out, hipStream_t s);
void launch_planet(const FramePlanet& F, const RowMap& M, float* out, hipStream_t s, int variant);
void launch_assemble(int width, int height, int block_rows, int nranks, int root_rounds, int rounds, int rows_max,
                     const float* gathered, float* frame, hipStream_t s, bool rgba8);
void launch_assemble_peers(int width, int height, int block_rows, int nranks, int root_rounds, int rounds, int rows_max,
                           int channels, const float* peers, float* frame, hipStream_t s);
hipError_t bind_fault_clouds(unsigned* word);
hipError_t bind_fault_clouds_ue4(unsigned* word);
hipError_t bind_fault_planet(unsigned* word);
void launch_raise_fault(unsigned code, hipStream_t s);
void launch_assemble_spans(int width, int height, int block_rows, const int4* span, const float* peers, size_t stride_pixels,
                           float* frame, hipStream_t s, bool rgba8);
void launch_pack_unorm8(int width, int rows, int flip, const float* in, unsigned char* out, hipStream_t s);
void launch_order_build(const unsigned* cost, unsigned* cls, unsigned* ghist, unsigned* order, int gx, int gy, hipStream_t s);
size_t order_build_scratch_words();   // tiles by cost, longest first (kern_util.hip)
// the grid each launcher uses for a map (the tile shapes are private to the kernel files): what a dispatch-order table is built for
dim3 clouds_grid(const RowMap& M);
dim3 egg_grid(const RowMap& M);
dim3 raytracer_grid(const RowMap& M);
dim3 atmosphere_grid(const RowMap& M);
dim3 planet_grid(const RowMap& M);
dim3 sdf_ao_grid(const RowMap& M);
dim3 vinyl_grid(const RowMap& M);
dim3 clouds_best_grid(const RowMap& M);
dim3 clouds_ue4_grid(const RowMap& M);
int launch_noise_eval(int fn, const float* xyz, const float* par, float* out, size_t n, hipStream_t s);
void launch_worley_volume(int size, float* out, hipStream_t s);
void launch_exp4k_eval(const float* a, float* out, size_t n, hipStream_t s);
int launch_math_eval(int fn, const float* a, const float* b, float* out, size_t n, hipStream_t s);
void launch_cl_exp_eval(const float* a, float* out, size_t n, hipStream_t s, int form);   // exp_reg_ / exp_reg128_ forms (kern_clouds.hip)

}  // namespace sbx
