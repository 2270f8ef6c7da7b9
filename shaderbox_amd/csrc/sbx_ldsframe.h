// shaderbox_amd/csrc/sbx_ldsframe.h — a kernel's frame block (its first argument) in LDS, read where it is used.
//
// gfx950, measured (profiles/r02_ubench_issue.txt): an fp32 VALU instruction with an SGPR source issues at HALF rate, and the
// SGPR file holds ~100 values — a frame block of rotations and primitive frames does not fit, the compiler parks the rest in
// VGPR lanes and fetches it back with v_readlane (half-rate as well; 15 % of k_vinyl's instructions were that).  So the block
// is copied once per workgroup from the kernarg segment to LDS and every member is read AT ITS USE with a volatile LDS load
// (all lanes one address: a broadcast): the constants arrive in VGPRs — full-rate operands — the loads ride the LDS pipe beside
// the VALU, and because they are volatile nothing is hoisted out of the march loops or kept alive across them.
// Same values, same operations: bit-identical by construction (k_vinyl 2.31 -> 1.40 ms).  It pays where the block overflows the
// SGPR file; APP_EGG's ~100 floats fit, and there the per-use LDS latency LOSES (1080p 0.271 -> 0.296 ms, 4K 0.64 -> 0.67).
#pragma once
#include <hip/hip_runtime.h>
#include "sbx_frame.h"

namespace sbx {

// copy the kernel's FIRST argument (at offset 0 of the kernarg segment) into the workgroup's LDS block; includes the barrier
template <class Frame, int THREADS>
__device__ __forceinline__ void lds_frame_fill(Frame& Fs) {
    static_assert(sizeof(Frame) % 4 == 0, "frame blocks are made of 32-bit words");
    const float* ka = (const float*)__builtin_amdgcn_kernarg_segment_ptr();
    float* dst = reinterpret_cast<float*>(&Fs);
    for (int i = (int)threadIdx.x; i < (int)(sizeof(Frame) / 4); i += THREADS) dst[i] = ka[i];
    __syncthreads();
}

// (the reference passed in is into a __shared__ block: the cast names the LDS address space, else the volatile read is a flat load)
#define SBX_LDS_PTR(T, r) ((const volatile __attribute__((address_space(3))) T*)(&(r)))
__device__ __forceinline__ float lds_ld(const float& r) { return *SBX_LDS_PTR(float, r); }
__device__ __forceinline__ int lds_ld(const int& r) { return *SBX_LDS_PTR(int, r); }
__device__ __forceinline__ double lds_ld(const double& r) { return *SBX_LDS_PTR(double, r); }
__device__ __forceinline__ v2 lds_ld(const v2& r) { return V2(lds_ld(r.x), lds_ld(r.y)); }
__device__ __forceinline__ v3 lds_ld(const v3& r) { return V3(lds_ld(r.x), lds_ld(r.y), lds_ld(r.z)); }
__device__ __forceinline__ m3 lds_ld(const m3& r) { return m3{lds_ld(r.c0), lds_ld(r.c1), lds_ld(r.c2)}; }
__device__ __forceinline__ BezierFrame lds_ld(const BezierFrame& r) {
    return BezierFrame{lds_ld(r.b), lds_ld(r.u), lds_ld(r.v), lds_ld(r.w), lds_ld(r.a2), lds_ld(r.c2), lds_ld(r.bc), lds_ld(r.br)};
}
__device__ __forceinline__ CylFrame lds_ld(const CylFrame& r) { return CylFrame{lds_ld(r.dir), lds_ld(r.len1), lds_ld(r.len0)}; }

}  // namespace sbx
