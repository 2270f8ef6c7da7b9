// shaderbox_amd/csrc/sbx_device.h — thread->pixel mapping and framebuffer store shared by the kernels.
//
// One thread per pixel.  A workgroup is 256 threads = 4 wave64; each wave owns an 8x8 pixel tile
// (lane = ly*8 + lx) so that the rays of a wave stay coherent (same lattice cells, same march
// exits) and so that a tile of 8 rows lines up with the 8-row cyclic blocks of the multi-GPU
// split; the workgroup covers 32x8 pixels.  Every lane stores one float4 (16 B): a wave writes
// eight 128-byte row segments, fully coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include "sbx_frame.h"

namespace sbx {

constexpr int TILE_W = 8, TILE_H = 8, WG_TILES_X = 4;
constexpr int WG_W = TILE_W * WG_TILES_X, WG_H = TILE_H, WG_THREADS = 256;

struct Pixel { int x, y; size_t idx; bool valid; };

__device__ __forceinline__ Pixel pixel_of_thread(const RowMap& M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Pixel p;
    p.x = blockIdx.x * WG_W + wave * TILE_W + (lane & 7);
    const int r = blockIdx.y * WG_H + (lane >> 3);
    p.valid = (p.x < M.width) && (r < M.nrows);
    p.y = row_to_y(M, r);
    p.idx = (size_t)r * M.width + p.x;
    return p;
}
inline dim3 grid_for(const RowMap& M) {
    return dim3((M.width + WG_W - 1) / WG_W, (M.nrows + WG_H - 1) / WG_H);
}
__device__ __forceinline__ void store_rgba(float* out, size_t idx, v3 c) {
    reinterpret_cast<float4*>(out)[idx] = make_float4(c.x, c.y, c.z, 1.0f);   // main.h:52
}

// launchers (one per app), defined next to their kernels
void launch_clouds(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s, int variant, void* ytab, int ytab_rows);
constexpr int CLOUDS_YTAB_ROWS = 1024;      // march steps covered by the per-frame y table
constexpr int CLOUDS_YTAB_BYTES = CLOUDS_YTAB_ROWS * 48;
constexpr int CLOUDS_YTAB_RING = 8;         // tables in flight (one per launch, round robin)
void launch_egg(const FrameEgg& F, const RowMap& M, float* out, hipStream_t s);
void launch_raytracer(const FrameRaytracer& F, const RowMap& M, float* out, hipStream_t s);
void launch_atmosphere(const FrameAtmosphere& F, const RowMap& M, float* out, hipStream_t s);
void launch_sdf_ao(const FrameSdfAo& F, const RowMap& M, float* out, hipStream_t s);
void launch_vinyl(const FrameVinyl& F, const RowMap& M, float* out, hipStream_t s);
void launch_planet(const FramePlanet& F, const RowMap& M, float* out, hipStream_t s);
void launch_assemble(int width, int height, int block_rows, int nranks, int rows_max,
                     const float* gathered, float* frame, hipStream_t s);
int launch_noise_eval(int fn, const float* xyz, const float* par, float* out, size_t n, hipStream_t s);
void launch_worley_volume(int size, float* out, hipStream_t s);
int launch_math_eval(int fn, const float* a, const float* b, float* out, size_t n, hipStream_t s);

}  // namespace sbx
