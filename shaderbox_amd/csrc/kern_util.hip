// shaderbox_amd/csrc/kern_util.hip — frame assembly after the multi-GPU gather, and the
// elementwise math-spec evaluator used by the parity tests.
#include "sbx_device.h"
#include "sbx_noise.h"

namespace sbx {

// `gathered` = nranks slabs (rank-major) of rows_max rows; slab r holds rank r's cyclic row-blocks
// densely in increasing y (sbx_render_rank).  One thread moves one float4 pixel; consecutive
// threads move consecutive pixels of a row, so both the read and the write are coalesced 16-B
// accesses.  Each XCD streams whole rows; there is no reuse to tile for.
// The split is RowMap's (sbx_frame.h): cycles of `rounds` rounds, rank 0 left out of the rounds >= root_rounds.
// (PX: float4 pixels, or `unsigned` = one R8G8B8A8_UNORM word per pixel, sbx_set_output_format)
template <class PX>
__global__ void __launch_bounds__(256) k_assemble(int width, int height, int block_rows, int nranks, int root_rounds,
                                                   int rounds, int rows_max, const PX* __restrict__ gathered,
                                                   PX* __restrict__ frame) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)width * height;
    if (i >= total) return;
    const int y = (int)(i / width), x = (int)(i - (size_t)y * width);
    const int blk = y / block_rows, in_blk = y - blk * block_rows;
    const int V = split_cycle_blocks(nranks, root_rounds, rounds);
    const int cycle = blk / V, v = blk - cycle * V;
    int rank, round;
    if (v < root_rounds * nranks) { round = v / nranks; rank = v - round * nranks; }
    else { const int w = v - root_rounds * nranks; const int q = w / (nranks - 1); round = root_rounds + q; rank = 1 + (w - q * (nranks - 1)); }
    const int cnt = rank == 0 ? root_rounds : rounds;
    const int local_blk = cycle * cnt + round;
    const size_t src = ((size_t)rank * rows_max + (size_t)local_blk * block_rows + in_blk) * width + x;
    frame[i] = gathered[src];
}

void launch_assemble(int width, int height, int block_rows, int nranks, int root_rounds, int rounds, int rows_max,
                     const float* gathered, float* frame, hipStream_t s, bool rgba8) {
    const size_t total = (size_t)width * height;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (rgba8)
        hipLaunchKernelGGL(k_assemble<unsigned>, grid, block, 0, s, width, height, block_rows, nranks, root_rounds, rounds, rows_max,
                           reinterpret_cast<const unsigned*>(gathered), reinterpret_cast<unsigned*>(frame));
    else
        hipLaunchKernelGGL(k_assemble<float4>, grid, block, 0, s, width, height, block_rows, nranks, root_rounds, rounds, rows_max,
                           reinterpret_cast<const float4*>(gathered), reinterpret_cast<float4*>(frame));
}

// The root of the direct exchange renders its own row-blocks IN PLACE (sbx_render_split_in_place), so only the peers' rows
// move: `peers` = the slabs of ranks 1 .. nranks-1 (rank-major, rows_max rows each), C floats per pixel — 3 when the slabs
// crossed xGMI without their alpha (RowMap.rgb), which is the constant 1 of main.h:52 and is written here.  Rows of rank 0
// are left alone.  One thread per frame pixel; reads and writes of a wave are contiguous (64 x 12 or 16 B in, 64 x 16 B out).
template <int C>      // C = 1: R8G8B8A8_UNORM words on both sides (sbx_set_output_format)
__global__ void __launch_bounds__(256) k_assemble_peers(int width, int height, int block_rows, int nranks, int root_rounds,
                                                         int rounds, int rows_max, const float* __restrict__ peers,
                                                         float4* __restrict__ frame) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)width * height;
    if (i >= total) return;
    const int y = (int)(i / width), x = (int)(i - (size_t)y * width);
    const int blk = y / block_rows, in_blk = y - blk * block_rows;
    const int V = split_cycle_blocks(nranks, root_rounds, rounds);
    const int cycle = blk / V, v = blk - cycle * V;
    int rank, round;
    if (v < root_rounds * nranks) { round = v / nranks; rank = v - round * nranks; }
    else { const int w = v - root_rounds * nranks; const int q = w / (nranks - 1); round = root_rounds + q; rank = 1 + (w - q * (nranks - 1)); }
    if (rank == 0) return;                                   // rendered where it belongs
    const int local_blk = cycle * rounds + round;
    const size_t src = ((size_t)(rank - 1) * rows_max + (size_t)local_blk * block_rows + in_blk) * width + x;
    if (C == 1) {
        reinterpret_cast<unsigned*>(frame)[i] = reinterpret_cast<const unsigned*>(peers)[src];
    } else if (C == 4) {
        frame[i] = reinterpret_cast<const float4*>(peers)[src];
    } else {
        const float* p = peers + src * 3;
        frame[i] = make_float4(p[0], p[1], p[2], 1.0f);
    }
}
void launch_assemble_peers(int width, int height, int block_rows, int nranks, int root_rounds, int rounds, int rows_max,
                           int channels, const float* peers, float* frame, hipStream_t s) {
    const size_t total = (size_t)width * height;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (channels == 1)
        hipLaunchKernelGGL(k_assemble_peers<1>, grid, block, 0, s, width, height, block_rows, nranks, root_rounds, rounds, rows_max,
                           peers, reinterpret_cast<float4*>(frame));
    else if (channels == 3)
        hipLaunchKernelGGL(k_assemble_peers<3>, grid, block, 0, s, width, height, block_rows, nranks, root_rounds, rounds, rows_max,
                           peers, reinterpret_cast<float4*>(frame));
    else
        hipLaunchKernelGGL(k_assemble_peers<4>, grid, block, 0, s, width, height, block_rows, nranks, root_rounds, rounds, rows_max,
                           peers, reinterpret_cast<float4*>(frame));
}

// The span exchange (RowMap.span): a peer's slab holds, block after block, only the span [x0, x1) of each of its row-blocks,
// 3 floats per pixel; slab r - 1 starts at peers + (r - 1) * stride_pixels * 3.  One thread per frame pixel; pixels of rank 0's
// blocks and pixels outside their block's span were rendered in place by the owner (sbx_render_span_root) and are left alone.
template <bool RGBA8>      // RGBA8: one R8G8B8A8_UNORM word per pixel on both sides (sbx_set_output_format)
__global__ void __launch_bounds__(256) k_assemble_spans(int width, int height, int block_rows, const int4* __restrict__ span,
                                                         const float* __restrict__ peers, size_t stride_pixels,
                                                         float4* __restrict__ frame) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)width * height;
    if (i >= total) return;
    const int y = (int)(i / width), x = (int)(i - (size_t)y * width);
    const int g = y / block_rows;
    const int4 T = span[g];
    if (T.w == 0 || x < T.x || x >= T.y) return;
    const size_t src = (size_t)(T.w - 1) * stride_pixels + (size_t)T.z + (size_t)(y - g * block_rows) * (size_t)(T.y - T.x) + (size_t)(x - T.x);
    if (RGBA8) {
        reinterpret_cast<unsigned*>(frame)[i] = reinterpret_cast<const unsigned*>(peers)[src];
        return;
    }
    const float* p = peers + src * 3;
    frame[i] = make_float4(p[0], p[1], p[2], 1.0f);                // alpha: the constant of main.h:52
}
void launch_assemble_spans(int width, int height, int block_rows, const int4* span, const float* peers, size_t stride_pixels,
                           float* frame, hipStream_t s, bool rgba8) {
    const size_t total = (size_t)width * height;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (rgba8)
        hipLaunchKernelGGL(k_assemble_spans<true>, grid, block, 0, s, width, height, block_rows, span, peers, stride_pixels,
                           reinterpret_cast<float4*>(frame));
    else
        hipLaunchKernelGGL(k_assemble_spans<false>, grid, block, 0, s, width, height, block_rows, span, peers, stride_pixels,
                           reinterpret_cast<float4*>(frame));
}

// float RGBA -> R8G8B8A8_UNORM, the back-buffer write of hlsltoy (util/hlsltoy/src/hlsltoy.cpp:79,192), by the
// Direct3D float -> UNORM rule: NaN -> 0, clamp to [0, 1], scale by 255, add .5, truncate.  One pixel per thread:
// a 16-byte load and a 4-byte store, both coalesced.  flip != 0 writes the top row first (D3D / PPM order).
__global__ void __launch_bounds__(256) k_pack_unorm8(int width, int rows, int flip, const float4* __restrict__ in,
                                                      unsigned* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)width * rows;
    if (i >= total) return;
    const int y = (int)(i / width), x = (int)(i - (size_t)y * width);
    const float4 c = in[i];
    auto q = [](float v) -> unsigned { return unorm8_(v); };     // sbx_device.h: the rule store_rgba's RGBA8 mode applies too
    const size_t o = (size_t)(flip ? rows - 1 - y : y) * width + x;
    out[o] = q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
}
void launch_pack_unorm8(int width, int rows, int flip, const float* in, unsigned char* out, hipStream_t s) {
    const size_t total = (size_t)width * rows;
    hipLaunchKernelGGL(k_pack_unorm8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, width, rows, flip,
                       reinterpret_cast<const float4*>(in), reinterpret_cast<unsigned*>(out));
}

__global__ void __launch_bounds__(256) k_math_eval(int fn, const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    const float y = b ? b[i] : 0.f;
    float r;
    switch (fn) {
    case 0: r = sin_(x); break;
    case 1: r = cos_(x); break;
    case 2: r = tan_(x); break;
    case 3: r = exp_(x); break;
    case 4: r = pow_(x, y); break;
    case 5: r = acos_(x); break;
    case 6: r = atan2_(x, y); break;
    case 7: r = hash1(x); break;
    case 8: r = x / y; break;                       // IEEE division
    case 9: r = div_by(x, recip64(y)); break;       // the same through the binary64 reciprocal (sbx_math.h)
    case 11: r = pow_h_(x, y); break;               // the former series pow (comparison with the table form)
    case 10: r = exp_h13_(x); break;                // the former 13-term exp (equivalence test against the table form)
    case 12: r = sqrt_n_(x); break;                 // v_sqrt_f32 + fix-up (exact outside (0, 2^-96))
    case 13: r = sqrt_ieee_(x); break;              // the compiler's IEEE expansion
    case 21: r = sin_b40_(x); break;                // sin with the degree-15 polynomial (equal to sin_ for |x| < 2^44.6)
    case 22: r = div3_(x, y, 1.0f / y); break;      // division by a known divisor in three binary32 instructions
    case 23: r = sqrt_rs_(x); break;                // v_rsq_f32 + one corrected step (exact for finite x >= 2^-102)
    case 24: r = divn_(x, y); break;                // v_rcp_f32 + a Newton step + div3_'s three instructions
    case 25: r = srgb_pow_(x); break;
    case 26: r = pow_spec_(x, y); break;            // pow as stated (pow_ is its lean device form)
               // pow_(x, 1 / 2.2f) in its short form (equal on all 2^32 arguments)
    default: r = 0.f;
    }
    out[i] = r;
}

int launch_math_eval(int fn, const float* a, const float* b, float* out, size_t n, hipStream_t s) {
    if (fn < 0 || (fn > 13 && fn != 21 && fn != 22 && fn != 23 && fn != 24 && fn != 25 && fn != 26)) return -1;
    hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, a, b, out, n);
    return 0;
}

// DISPATCH ORDER (RowMap.order): the launch's gx x gy tiles sorted by the cost the previous frames measured for them, longest first
// — a counting sort over 1024 cost classes (cost >> 6: 0.64 us each, the last one open), so the table is a permutation of the tiles
// whatever the cost words hold (uninitialised memory included): the order is a hint, a missing or a doubled tile would be a wrong frame.
// Three small kernels behind each other on the launch's stream, ORDER_G workgroups over contiguous chunks of the tiles:
//   k_order_count: the class of every tile, written to `cls` (launches of other streams may be rewriting `cost` meanwhile, and a tile
//                  counted in one class and placed in another would run a class's cursor into its neighbour's range), and the
//                  chunk's histogram (LDS) to ghist[chunk][class]
//   k_order_scan : ghist[chunk][class] := the first place of that chunk's tiles of that class (classes in order, chunks within a class)
//   k_order_place: every tile takes the next place of its (chunk, class) cursor
// (Round 6's first form was ONE workgroup doing all of it: 150 us for the 129 600 tiles of a 4K frame, bound by the one CU's VALU —
// ~100 instructions per tile and pass — not by memory; ~25 us this way.  A MILD order was tried too — row order, the trivial tiles
// last, only the long tiles of the launch's final stretch moved to the front, to keep the neighbours row order gives a wave: no better
// with frames in flight and it loses the strips' gain; removed.  profiles/r06_tile_order.txt.)
constexpr int ORDER_G = 64, ORDER_T = 256, ORDER_CLASSES = 1024;
__device__ __forceinline__ void order_chunk(int n, int& lo, int& hi) {
    const int chunk = (n + ORDER_G - 1) / ORDER_G;
    lo = (int)blockIdx.x * chunk;
    hi = lo + chunk < n ? lo + chunk : n;
}
__global__ void __launch_bounds__(ORDER_T) k_order_count(const unsigned* __restrict__ cost, unsigned* __restrict__ cls,
                                                         unsigned* __restrict__ ghist, int n) {
    __shared__ unsigned h[ORDER_CLASSES];
    const int tid = (int)threadIdx.x;
    for (int c = tid; c < ORDER_CLASSES; c += ORDER_T) h[c] = 0u;
    __syncthreads();
    int lo, hi;
    order_chunk(n, lo, hi);
    for (int i = lo + tid; i < hi; i += ORDER_T) {
        const unsigned k = __hip_atomic_load(&cost[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 6;
        const unsigned c = (unsigned)(ORDER_CLASSES - 1) - (k < (unsigned)(ORDER_CLASSES - 1) ? k : (unsigned)(ORDER_CLASSES - 1));   // class 0 = the longest
        cls[i] = c;
        atomicAdd(&h[c], 1u);
    }
    __syncthreads();
    for (int c = tid; c < ORDER_CLASSES; c += ORDER_T) ghist[(int)blockIdx.x * ORDER_CLASSES + c] = h[c];
}
__global__ void __launch_bounds__(ORDER_CLASSES) k_order_scan(unsigned* __restrict__ ghist) {
    __shared__ unsigned part[2 * (ORDER_CLASSES / 64)];
    const int c = (int)threadIdx.x;                                          // one thread per class
    unsigned tot = 0u;
    for (int g = 0; g < ORDER_G; ++g) { const unsigned v = ghist[g * ORDER_CLASSES + c]; ghist[g * ORDER_CLASSES + c] = tot; tot += v; }
    unsigned incl = tot;                                                     // inclusive scan of the classes' totals: wave, then workgroup
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o); if ((c & 63) >= o) incl += v; }
    if ((c & 63) == 63) part[c >> 6] = incl;
    __syncthreads();
    if (c == 0) { unsigned acc = 0u; for (int w = 0; w < ORDER_CLASSES / 64; ++w) { part[ORDER_CLASSES / 64 + w] = acc; acc += part[w]; } }
    __syncthreads();
    const unsigned base = incl - tot + part[ORDER_CLASSES / 64 + (c >> 6)];
    for (int g = 0; g < ORDER_G; ++g) ghist[g * ORDER_CLASSES + c] += base;
}
__global__ void __launch_bounds__(ORDER_T) k_order_place(const unsigned* __restrict__ cls, const unsigned* __restrict__ ghist,
                                                         unsigned* __restrict__ order, int n, int gx) {
    __shared__ unsigned cur[ORDER_CLASSES];
    const int tid = (int)threadIdx.x;
    for (int c = tid; c < ORDER_CLASSES; c += ORDER_T) cur[c] = ghist[(int)blockIdx.x * ORDER_CLASSES + c];
    __syncthreads();
    int lo, hi;
    order_chunk(n, lo, hi);
    for (int i = lo + tid; i < hi; i += ORDER_T) {
        const unsigned pos = atomicAdd(&cur[cls[i]], 1u);
        order[pos] = (unsigned)(i % gx) | ((unsigned)(i / gx) << 16);
    }
}
void launch_order_build(const unsigned* cost, unsigned* cls, unsigned* ghist, unsigned* order, int gx, int gy, hipStream_t s) {
    const int n = gx * gy;
    hipLaunchKernelGGL(k_order_count, dim3(ORDER_G), dim3(ORDER_T), 0, s, cost, cls, ghist, n);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_CLASSES), 0, s, ghist);
    hipLaunchKernelGGL(k_order_place, dim3(ORDER_G), dim3(ORDER_T), 0, s, cls, ghist, order, n, gx);
}
size_t order_build_scratch_words() { return (size_t)ORDER_G * ORDER_CLASSES; }

}  // namespace sbx
