// shaderbox_amd/csrc/sbx_hashcache.h — per-wave lattice-hash cache in LDS, shared by the kernels whose time
// goes into noise_iq (/root/reference/src/noise_iq.h:5-29): APP_CLOUDS and APP_PLANET.
//
// hash(n) = fract(sin(n)*753.5453123) depends only on the integer lattice index n = p.x + 157 p.y + 113 p.z,
// and the 64 rays of an 8x8 pixel tile sample almost the same place, so the 8 corner hashes of a lattice
// cell are computed once per wave and kept in a small direct-mapped table in LDS: HC_TABLES tables (one
// per octave / noise stream) of 64 slots {tag = bits of n, 8 hashes}, slot = int(n) & 63, 9.3 KB per wave.
// Lookups are per lane (lanes of a tile mostly read the same slot: LDS broadcast); misses are served by the
// whole wave together (leader election with ballot/readlane, then lane (r, c) hashes corner c of pending
// cell r).  Every hash is hash1() of the same binary32 argument as in a per-lane evaluation, so results
// are bit-identical by construction.  DESIGN.md §5.1 has the measurements and the compiler hazards.
#pragma once
#include "sbx_device.h"
#include "sbx_noise.h"

namespace sbx {

// "does any lane of the wave have x?" as an OPAQUE wave-uniform value.  Every branch and loop that
// encloses cross-lane steps (ballot/readlane leader election, the 64-lane hash pass) must stay
// wave-uniform with all lanes enabled.  Written as `if (__ballot(x))` / `while (__ballot(x))` the
// optimizer is free to fold the test back to the per-lane condition x and to unswitch or re-nest the
// region on per-lane terms; the region then runs with lanes masked off and the cooperative steps
// silently lose their workers (observed: endless miss loops).  Passing the mask through an empty asm
// with an SGPR constraint keeps the value uniform and hides its origin.
// ballot of a comparison is the comparison's own SGPR result; ballots of combined predicates cost a
// v_cndmask + v_cmp round trip, so hot tests OR/AND the 64-bit masks of the individual comparisons instead
__device__ __forceinline__ unsigned long long wave_mask(bool x) { return __builtin_amdgcn_ballot_w64(x); }
__device__ __forceinline__ bool wave_any_mask(unsigned long long m) {
    unsigned any = (unsigned)m | (unsigned)(m >> 32);
    asm volatile("" : "+s"(any));
    return any != 0;
}
__device__ __forceinline__ bool wave_any(bool x) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(x);    // the i1 mask itself, no 0/1 round trip through a VGPR
    unsigned any = (unsigned)m | (unsigned)(m >> 32);
    asm volatile("" : "+s"(any));
    return any != 0;
}

#ifndef SBX_HC_SLOTS
#define SBX_HC_SLOTS 64
#endif
#ifndef SBX_HC_MAGIC_SLOT
#define SBX_HC_MAGIC_SLOT 1
#endif
constexpr int HC_SLOTS = SBX_HC_SLOTS;
struct alignas(16) WaveCache {
    float h[4][HC_SLOTS][8];          // corner order: +0,+1,+157,+158,+113,+114,+270,+271
    unsigned tag[4][HC_SLOTS];        // bits of n; 0x7fc00001 (a NaN) = empty
    unsigned ins_tag[8], ins_slot[8]; // cells being inserted in the current pass
    unsigned spec;                    // hc_insert<., SPEC>: 0, or 8 | sx | sy << 1 | sz << 2 — the signs (1 = negative) of the direction the
                                      // wave's sample points travel in (see hc_insert)
#ifdef SBX_CL_STATS
    float stat[8];                    // census build only (tools/clouds_census.py): slow calls, passes, cells, re-lookups,
                                      // main samples past the first / second stage, Lipschitz-skipped steps
#endif
};

// miss path, one octave: the lanes in `need` lack their cell.  Leaders (one per distinct slot) are
// elected with ballot/readlane, up to 8 cells per pass; lane (r, c) evaluates corner c of pending cell r.
// SPEC (the marches of APP_PLANET): a pass costs the wave the same instructions whether one cell is pending or eight — lane (r, c)
// hashes corner c of cell r, and with ONE pending cell 56 of the 64 lanes idle through the binary64 sin.  When the wave has said
// which way its sample points travel (WaveCache.spec: a march along a ray enters, next, one of the cells of the 2 x 2 x 2 block
// AHEAD of the current one), the seven idle rows hash the seven other cells of that block — those not in the table yet — in the same
// pass.  What a table holds never changes what a lookup returns (every entry is hash1 of its own lattice index, whoever put it
// there), only how often the next lookups miss; an evicted cell is hashed again if it is wanted again.  The eight cells of a block
// fall into eight different slots for HC_SLOTS >= 32 (offsets +-1, +-157, +-113 and their sums are distinct mod 32).
template <bool B40 = false, bool SPEC = false>
__device__ __forceinline__ void hc_insert(WaveCache& S, int k, unsigned nbits, int slot, bool need, int lane) {
    const int corner = lane & 7;
    const float off = (corner & 1 ? 1.0f : 0.0f) + (corner & 2 ? 157.0f : 0.0f) + (corner & 4 ? 113.0f : 0.0f);
    unsigned long long m = __builtin_amdgcn_ballot_w64(need);
    if (SPEC && HC_SLOTS >= 32) {
        const unsigned sp = (unsigned)__builtin_amdgcn_readfirstlane((int)S.spec);
        if (sp && m) {
            const int leader = __ffsll((long long)m) - 1;
            const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nbits, leader);
            const int s0 = __builtin_amdgcn_readlane(slot, leader);
            if ((m & ~__builtin_amdgcn_ballot_w64(slot == s0)) == 0) {       // one pending cell (wave-uniform): rows 1..7 are free
                const int r = lane >> 3;
                const float dx = (sp & 1u) ? -1.0f : 1.0f, dy = (sp & 2u) ? -157.0f : 157.0f, dz = (sp & 4u) ? -113.0f : 113.0f;
                const float nr = u2f(n0) + ((r & 1 ? dx : 0.0f) + (r & 2 ? dy : 0.0f) + (r & 4 ? dz : 0.0f));
                const int sr = (int)nr & (HC_SLOTS - 1);
                const bool present = r != 0 && S.tag[k][sr] == f2u(nr);
                __builtin_amdgcn_wave_barrier();
                if (!present) {
                    S.h[k][sr][corner] = hash1_b<B40>(nr + off);
                    if (corner == 0) S.tag[k][sr] = f2u(nr);
                }
                __builtin_amdgcn_wave_barrier();
                return;
            }
        }
    }
    // (Filling the free rows also when SEVERAL cells are pending — lanes on both sides of a cell face — was built and measured:
    //  k_planet 7680x4320 5.77 -> 5.91 ms; the bookkeeping in this loop costs more than the few passes it saves.  profiles/r05_log.md)
    while (m) {
        int cnt = 0;
        while (m && cnt < 8) {
            const int leader = __ffsll((long long)m) - 1;
            const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nbits, leader);
            const int s0 = __builtin_amdgcn_readlane(slot, leader);
            if (lane == cnt) { S.ins_tag[cnt] = n0; S.ins_slot[cnt] = (unsigned)s0; }
            m &= ~__builtin_amdgcn_ballot_w64(slot == s0);          // one cell per slot per call; losers are served in the next round
            ++cnt;
        }
        __builtin_amdgcn_wave_barrier();
#ifdef SBX_CL_STATS
        if (lane == 0) { S.stat[1] += 1.f; S.stat[2] += (float)cnt; }
#endif
        if (lane < cnt * 8) {
            const int r = lane >> 3;
            const unsigned n0 = S.ins_tag[r];
            const int s0 = (int)S.ins_slot[r];
            S.h[k][s0][corner] = hash1_b<B40>(u2f(n0) + off);
            if (corner == 0) S.tag[k][s0] = n0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// Slow path of one octave, out of line on purpose: called only when some active lane misses.  Inserts
// the missing cells round by round (one cell per slot per round) and hands every active lane the 8
// hashes of its cell BY VALUE: a lane latches its hashes as soon as its cell is present, so a later
// round that reuses the slot cannot take them away.  Kept in its own function, with a single read site
// and an opaque wave-uniform loop test (wave_any), because inlined into the callers the optimizer merged
// the latch with the caller's pre-loop read and turned the loop into a per-lane (divergent) one — the
// hash pass then ran with its worker lanes masked off and the loop never finished.
struct H8 { float4 lo, hi; };
// The miss loop of hc_slow needs at most 64 rounds (one cell per slot per round, 64 lanes); its bound exists because a compiler
// that re-nests the loop on per-lane terms (the hazard described above) would otherwise hang the GPU.  Reaching the bound means
// some lane's hashes were never delivered: the pixels of that launch are WRONG.  That must not pass silently: the wave sets the
// device's sticky fault word — one word of pinned host memory per device, bound to this translation unit's pointer by
// sbx_create — and every later call on a context of that device fails with SBX_ERR_FAULT until sbx_clear_fault.
static __device__ unsigned* g_hc_fault = nullptr;
__device__ __noinline__ void hc_fault(unsigned code) {
    unsigned* p = g_hc_fault;
    if (p) __hip_atomic_fetch_or(p, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
inline hipError_t hc_bind_fault_word(unsigned* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_hc_fault), &p, sizeof(p)); }
#ifndef SBX_HC_SLOW_INLINE
#define SBX_HC_SLOW_INLINE __forceinline__
#endif
// B40: the caller has shown |n| <= 2^40 for every lattice index it can produce (hash1_b)
template <bool B40 = false, bool SPEC = false>
__device__ SBX_HC_SLOW_INLINE H8 hc_slow(WaveCache& S, int k, unsigned nbits, int slot, bool active, int lane) {
    H8 r;
    r.lo = make_float4(0.f, 0.f, 0.f, 0.f);
    r.hi = r.lo;
    bool need = active;
#ifdef SBX_CL_STATS
    if (lane == 0) S.stat[0] += 1.f;
#endif
    int round = 0;
    for (; round < 4096; ++round) {                       // bounded on principle; needs <= 64 rounds
        if (need && S.tag[k][slot] == nbits) {
            r.lo = *reinterpret_cast<const float4*>(&S.h[k][slot][0]);
            r.hi = *reinterpret_cast<const float4*>(&S.h[k][slot][4]);
            need = false;
        }
        if (!wave_any(need)) break;
        hc_insert<B40, SPEC>(S, k, nbits, slot, need, lane);
    }
#ifndef SBX_HC_FAULT
#define SBX_HC_FAULT 1
#endif
    if (SBX_HC_FAULT && round >= 4096) hc_fault(1u);                      // (wave-uniform) never observed; see g_hc_fault
    return r;
}

// trilinear blend of the 8 corner hashes (noise_iq.h:20-23)
__device__ __forceinline__ float hc_blend(float4 lo, float4 hi, float fx, float fy, float fz) {
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    const float a = lo.x * gx + lo.y * fx;
    const float b = lo.z * gx + lo.w * fx;
    const float c = hi.x * gx + hi.y * fx;
    const float d = hi.z * gx + hi.w * fx;
    const float ab = a * gy + b * fy;
    const float cd = c * gy + d * fy;
    return ab * gz + cd * fz;
}


// initialise the calling wave's table (all tags empty)
__device__ __forceinline__ void hc_init(WaveCache& S, int lane) {
    for (int i = lane; i < 4 * HC_SLOTS; i += 64) (&S.tag[0][0])[i] = 0x7fc00001u;
    if (lane == 0) S.spec = 0u;
    __builtin_amdgcn_wave_barrier();
}

// Tell hc_insert<., SPEC> which way the wave's sample points travel: `d` = the direction in lattice space (any positive scale) of
// the first lane in `on`; on == none or d == 0 switches the speculation off.
__device__ __forceinline__ void hc_set_direction(WaveCache& S, v3 d, bool on, int lane) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
    unsigned bits = 8u | (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    const int first = m ? __ffsll((long long)m) - 1 : 0;
    bits = (unsigned)__builtin_amdgcn_readlane((int)bits, first);
    if (lane == 0) S.spec = m ? bits : 0u;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void hc_no_direction(WaveCache& S, int lane) {
    if (lane == 0) S.spec = 0u;
    __builtin_amdgcn_wave_barrier();
}

// N independent noise_iq evaluations at p[0..N-1], evaluation i using table tab[i]: all lattice terms and
// tag reads first, one wave-uniform all-hit test, then reads + blends.  `active` = lanes whose result is used.
// INVARIANT: the tab[i] are pairwise DISTINCT (N <= 4 tables).  All tags are sampled up front; if two evaluations shared a
// table, the insert that serves evaluation i could evict a cell evaluation j > i had seen present, and j would then read
// another cell's hashes without looking at the tag again.  The callers pass (base + i) & 3 with N <= 4, or {0, 1, 2, 3};
// the static_assert below holds N to the number of tables, the distinctness is the callers' contract.
// XI ("exact integers"): the caller guarantees lattice coordinates so small that n = px + 157 py + 113 pz is an integer below
// 2^24 at every step (APP_PLANET: |p| <= 145 in the finest octave).  Both products are then exact, the reference's
// RN(RN(px + RN(157 py)) + RN(113 pz)) has no rounding at all, and two fmas return the same integer.
// SPEC: hc_insert's forward-block speculation (the caller sets WaveCache.spec around its marches).
template <int N, bool XI = false, bool SPEC = false>
__device__ __forceinline__ void coop_noise_n(WaveCache& S, const v3 (&p)[N], const int (&tab)[N], bool active, int lane,
                                             float (&out)[N]) {
    static_assert(N >= 1 && N <= 4, "one table per evaluation: at most four per batch");
    float fx[N], fy[N], fz[N];
    unsigned nbits[N];
    int slot[N];
    bool ne[N];
    bool miss = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float px = floor_(p[i].x), py = floor_(p[i].y), pz = floor_(p[i].z);
        const float ax = p[i].x - px, ay = p[i].y - py, az = p[i].z - pz;
        fx[i] = ax * ax * tm2_(ax);
        fy[i] = ay * ay * tm2_(ay);
        fz[i] = az * az * tm2_(az);
        const float n = XI ? __builtin_fmaf(113.0f, pz, __builtin_fmaf(py, 157.0f, px)) : px + py * 157.0f + 113.0f * pz;
        nbits[i] = f2u(n);
        // XI: n is an integer below 2^22 in magnitude, so the low bits of RN(n + 1.5 * 2^23) ARE n's low bits (two's complement, as
        // (int)n & mask): one full-rate add instead of the half-rate conversion
        slot[i] = (XI && SBX_HC_MAGIC_SLOT) ? (int)(f2u(n + 12582912.0f) & (unsigned)(HC_SLOTS - 1)) : ((int)n & (HC_SLOTS - 1));
        ne[i] = (S.tag[tab[i]][slot[i]] != nbits[i]);
        miss |= ne[i];
    }
    if (!wave_any(active && miss)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float4 lo = *reinterpret_cast<const float4*>(&S.h[tab[i]][slot[i]][0]);
            const float4 hi = *reinterpret_cast<const float4*>(&S.h[tab[i]][slot[i]][4]);
            out[i] = hc_blend(lo, hi, fx[i], fy[i], fz[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            H8 h;
            if (wave_any(active && ne[i])) {
                h = hc_slow<XI, SPEC>(S, tab[i], nbits[i], slot[i], active, lane);       // XI: indices below 2^24
            } else {
                h.lo = *reinterpret_cast<const float4*>(&S.h[tab[i]][slot[i]][0]);
                h.hi = *reinterpret_cast<const float4*>(&S.h[tab[i]][slot[i]][4]);
            }
            out[i] = hc_blend(h.lo, h.hi, fx[i], fy[i], fz[i]);
        }
    }
}

}  // namespace sbx
