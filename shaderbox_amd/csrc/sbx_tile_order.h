// sbx_tile_order.h — the host side of the DISPATCH ORDER (RowMap.order / .cost): which launches get a table, when one is built, adopted
// and retired.  Included by sbx_capi.hip only; the sort itself is kern_util.hip (launch_order_build).  DESIGN.md 5.1,
// profiles/r06_tile_order.txt.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include "sbx_device.h"
#include "sbx_frame.h"

namespace sbx {

// DISPATCH ORDER of a full-frame launch (RowMap.order / .cost): tiles sorted by the duration the previous frames measured for them,
// longest first, so that a launch ends on its SHORTEST waves instead of on whichever rows come last — one k_clouds launch of the
// 3840x2160 frame kept the chip full for 2.04 ms and then drained for 0.29 ms (tools/clouds_timeline.py).  A ring of tables
// (a launch in flight may still read the one before), rebuilt from the cost table after the first frames of a key, 16 and 32 launches
// later and every TILE_ORDER_REFRESH launches after that.  A context keeps up to TILE_ORDER_KEYS shapes per app (least recently used
// replaced): the ranks of an emulated multi-GPU frame driven through one context each keep their table, as separate processes would.
constexpr int TILE_ORDER_RING = 4, TILE_ORDER_REFRESH = 64, TILE_ORDER_KEYS = 8;
// What a MOVING scene means for an app's table (measured on animated frames, profiles/r06_tile_order.txt section 11):
//   TILE_SCENE_FREE    the costs do not follow the scene (APP_VINYL: -7 % standing or moving): the plain refresh schedule
//   TILE_SCENE_REFRESH they drift with it (APP_CLOUDS: a table 8-64 frames old keeps 0.6-3 % of the 6 % a fresh one gives): while the
//                      scene moves the table is rebuilt behind EVERY launch (23 us of a 2.2-3.4 ms frame: -3.6 ... -4.5 %)
//   TILE_SCENE_KEYED   they jump with it (APP_EGG: the silhouette's 16 x 4-pixel tiles are others a frame later; with tables even one
//                      frame old an animated launch is 8-27 % SLOWER than in the kernel's own hot-first order): the scene is part of
//                      the table's key — a scene that stands still gets its table, a moving one never does
enum { TILE_SCENE_FREE = 0, TILE_SCENE_REFRESH = 1, TILE_SCENE_KEYED = 2 };
struct TileOrder {
    unsigned* mem = nullptr;               // cost | classes | TILE_ORDER_RING tables, `cap` words each | the sort's histograms
    size_t cap = 0;
    int key[14] = {-1};                     // app, width, nrows, y0, grid x, grid y, the split the rows belong to, and (TILE_SCENE_KEYED) the scene
    unsigned long long scene = 0;          // the scene (hash of the frame's uniforms) of the last launch; `moving`: it differed from the one before
    bool moving = false;
    int cur = -1, pending = -1, age = 0, built = 0;   // pending: a table whose build is queued, current once `ready` has passed
    unsigned long long stamp = 0;          // last use (least recently used entry of an app is the one a new shape takes)
    hipStream_t pending_stream = nullptr;  // the stream the pending table's build is queued on: ITS later launches may read the table at once (stream order)
    hipEvent_t ready{}, seen{};            // the pending table is built / behind the shape's first launch (the first table waits for its costs, on the host)
    bool have_ready = false, seen_recorded = false;
    std::vector<std::pair<hipStream_t, hipEvent_t>> users[TILE_ORDER_RING];     // streams that launched readers of the CURRENT table
    std::vector<hipEvent_t> retired[TILE_ORDER_RING];   // recorded behind the last readers of a table that is current no more: the slot is free once all have passed
};
// a context's tables: by app id (enum sbx_app), a few launch shapes each
struct TileOrderSet {
    TileOrder tab[16][TILE_ORDER_KEYS];
    unsigned long long clock = 0;
    hipStream_t last_stream = nullptr;     // the stream of the last launch that could take an order, and how many in a row came on it
    int same_stream = 0;
    std::vector<hipEvent_t>* pool = nullptr;   // the context's pool of timing-less events (taken from, returned to)
};

// The dispatch order of a launch (TileOrder above).  tile_order_begin: the table and the cost words of this launch go into M (or
// nothing: a point list, a sub-range of a slab, a stream being captured, SBX_TILE_ORDER=0); tile_order_end: after
// the launch is enqueued — remembers the stream as a reader, rebuilds the table when one is due.
// A table that stops being current: an event behind the last launch of every stream that read it.  Nothing waits for these on the
// device — tile_order_end rewrites a slot only once they have all passed.
static void tile_order_retire(TileOrderSet& S, TileOrder& T, int slot) {
    if (slot < 0) return;
    for (auto& u : T.users[slot]) {
        if (hipEventRecord(u.second, u.first) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); (void)hipGetLastError(); (*S.pool).push_back(u.second); continue; }   // (a stream that is gone)
        T.retired[slot].push_back(u.second);
    }
    T.users[slot].clear();
}
static bool tile_order_slot_free(TileOrderSet& S, TileOrder& T, int slot) {
    auto& rs = T.retired[slot];
    while (!rs.empty()) {
        if (hipEventQuery(rs.back()) != hipSuccess) { (void)hipGetLastError(); return false; }
        (*S.pool).push_back(rs.back());
        rs.pop_back();
    }
    return true;
}
static TileOrder* tile_order_begin(TileOrderSet& S, int app, RowMap& M, dim3 grid, hipStream_t s, bool capturing, unsigned long long scene, int scene_policy) {
    static const int mode = [] { const char* v = getenv("SBX_TILE_ORDER"); return v ? atoi(v) : 1; }();   // 0 off; 2: costs and tables but no order (debugging)
    if (mode == 0 || capturing || app < 0 || app >= 16 || M.frag || M.r0 != 0 || grid.x == 0 || grid.x > 0xffffu || grid.y > 0xffffu) return nullptr;
    const size_t n = (size_t)grid.x * grid.y;
    if (n < 4096) return nullptr;                                 // (small launches: nothing to order)
    const bool keyed = scene_policy == TILE_SCENE_KEYED;
    const int key[14] = {app, M.width, M.nrows, M.y0, (int)grid.x, (int)grid.y, M.nranks, M.rank, M.block_rows, M.root_rounds, M.rounds,
                         M.span_mode * 4 + M.in_place, keyed ? (int)(unsigned)scene : 0, keyed ? (int)(unsigned)(scene >> 32) : 0};
    TileOrder* hit = nullptr;
    TileOrder* lru = nullptr;                                     // the least recently used entry that has no build still queued
    for (auto& E : S.tab[app]) {
        if (std::memcmp(key, E.key, sizeof(key)) == 0) { hit = &E; break; }
        if (E.pending >= 0 && hipEventQuery(E.ready) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (!lru || E.stamp < lru->stamp) lru = &E;
    }
    if (!hit && !lru) return nullptr;                             // (every entry waits for a build: this launch goes without — the host never blocks here)
    TileOrder& T = hit ? *hit : *lru;
    T.stamp = ++S.clock;
    if (!hit) {
        tile_order_retire(S, T, T.pending);                       // (a build of the old shape, if any, has finished: see above; its early readers)
        T.pending = -1;
        if (n > T.cap) {
            if (T.mem) {
                (void)hipDeviceSynchronize(); (void)hipFree(T.mem); T.mem = nullptr; T.cap = 0;
                for (auto& us : T.users) { for (auto& u : us) (*S.pool).push_back(u.second); us.clear(); }
                for (auto& rs : T.retired) { for (auto& e : rs) (*S.pool).push_back(e); rs.clear(); }
            }
            if (hipMalloc((void**)&T.mem, (n * (2 + TILE_ORDER_RING) + order_build_scratch_words()) * 4) != hipSuccess) { (void)hipGetLastError(); T.key[0] = -1; return nullptr; }
            T.cap = n;
            (void)hipMemsetAsync(T.mem, 0, n * 4, s);            // (tiles no wave reports for — all of it outside the frame — sort last, not anywhere)
        }
        std::memcpy(T.key, key, sizeof(key));
        tile_order_retire(S, T, T.cur);                         // (the old shape's table stays readable for its launches in flight)
        T.cur = -1; T.age = 0; T.built = 0; T.seen_recorded = false;
    }
    if (!T.have_ready) {
        if (hipEventCreateWithFlags(&T.ready, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipEventCreateWithFlags(&T.seen, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(T.ready); return nullptr; }
        T.have_ready = true;
    }
    M.cost = T.mem;
    T.moving = scene_policy == TILE_SCENE_REFRESH && hit && scene != T.scene;
    T.scene = scene;
    if (T.pending >= 0 && hipEventQuery(T.ready) == hipSuccess) {              // the new table is complete: current from this launch on
        tile_order_retire(S, T, T.cur);
        T.cur = T.pending; T.pending = -1; ++T.built;
    }
    (void)hipGetLastError();                                                  // (hipErrorNotReady is not an error)
    // ONE AT A TIME only.  A host that keeps frames in flight (launches alternating over streams) already fills the end of one launch
    // with the start of the next; there the sorted order buys nothing (4K CLOUDS 2.203 -> 2.217 ms per frame with three in flight) and
    // costs 4-12 % on an eighth-frame strip, while one launch at a time gains 7 % (full frame) to 18 % (strip).  The sign of frames in
    // flight: this launch comes on another stream than the last one.  The costs are collected either way.
    if (s == S.last_stream) ++S.same_stream; else { S.last_stream = s; S.same_stream = 0; }
    // the table of this launch: the current one (complete before this call: no stream waits for it) — or, on the stream its build is
    // queued on, the PENDING one: stream order puts the build ahead of this launch, so a host that never looks at its frames still
    // renders frame k + 1 by frame k's costs
    const int use = (T.pending >= 0 && T.pending_stream == s) ? T.pending : T.cur;
    if (use >= 0 && S.same_stream >= 3 && mode == 1) {
        bool found = false;
        for (auto& u : T.users[use]) if (u.first == s) { found = true; break; }
        if (!found) {
            hipEvent_t ev{};
            if (!(*S.pool).empty()) { ev = (*S.pool).back(); (*S.pool).pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return &T; }
            T.users[use].emplace_back(s, ev);
        }
        M.order = T.mem + T.cap * (size_t)(2 + use);
    }
    return &T;
}
static void tile_order_end(TileOrderSet& S, TileOrder* Tp, hipStream_t s) {
    TileOrder& T = *Tp;
    ++T.age;
    // the first table of a shape after TWO launches of it (a host that renders a shape once never pays for a table it would not use)
    // and once they have FINISHED (their costs are what the table is made of; an event recorded behind the second and queried here),
    // later ones 16 and 32 launches on, then every TILE_ORDER_REFRESH launches
    if (T.cur < 0 && !T.seen_recorded && T.age >= 2) {          // (at the SECOND launch of a shape: a shape — or, TILE_SCENE_KEYED, a scene
        if (hipEventRecord(T.seen, s) != hipSuccess) { (void)hipGetLastError(); return; }   // — that is rendered once costs no packet)
        T.seen_recorded = true;
    }
    // (16, 32, then every 64 launches: the costs drift with the scene, slowly, and the first table is made of the first launch's)
    const int refresh = T.cur < 0 ? 2 : T.moving ? 1 : T.built <= 1 ? 16 : T.built == 2 ? 32 : TILE_ORDER_REFRESH;
    if (T.pending >= 0 || T.age < refresh) return;
    // not while the host keeps frames in flight (this launch came on another stream than the last): the table would not be used, and
    // the sort, queued behind a launch of ONE stream, has to find room beside the other streams' launches (as one 1 024-thread
    // workgroup it waited up to 8 ms for a whole CU to drain: rocprofv3, three 4K frames in flight).  The costs keep being collected;
    // the build is due again at the first launch that follows another one on its stream.
    if (S.same_stream < 1) return;
    if (T.cur < 0 && (!T.seen_recorded || hipEventQuery(T.seen) != hipSuccess)) { (void)hipGetLastError(); return; }
    // The build goes IN LINE, on the stream of the launch that is due, and nothing on the device waits for it except that stream's
    // own next launch (~25 us once per 64 launches): the other render streams take the table once its event has passed (queried on
    // the host, tile_order_begin), the cost words may hold any mixture of frames (a table is a permutation whatever they hold), and
    // the slot it writes is one whose last readers have finished (their events are queried here; if no slot is free yet the build
    // is tried again at the next launch).  What round 6 tried first, and what it cost (profiles/r06_tile_order.txt sections 5-7):
    // other streams WAITING for the new table (~170 us per refresh, each); a side stream that waits for events of the render
    // streams (HIP multiplexes streams over a few hardware queues: the wait parks in a queue it shares with a render stream and
    // holds back the launches behind it, 8-10 % of eighth-frame strips in flight); a fifth normal-priority stream at all (two of a
    // host's three render streams then share a queue, 0.278 -> 0.338 ms per strip); a high-priority side stream (free in one
    // process, but eight processes sharing one GPU ran 45 % slower: the priority queues preempt each other's processes).
    int next = -1;
    for (int k = 1; k < TILE_ORDER_RING && next < 0; ++k) {
        const int c = ((T.cur < 0 ? 0 : T.cur) + k) % TILE_ORDER_RING;
        if (c != T.cur && tile_order_slot_free(S, T, c)) next = c;
    }
    if (next < 0) return;
    launch_order_build(T.mem, T.mem + T.cap, T.mem + T.cap * (size_t)(2 + TILE_ORDER_RING), T.mem + T.cap * (size_t)(2 + next), T.key[4], T.key[5], s);
    if (hipEventRecord(T.ready, s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); (void)hipGetLastError(); return; }
    T.pending = next; T.pending_stream = s; T.age = 0;
}

// the shape of `app` used last: its pending table adopted (the host waits for it), for sbx_debug_tile_order
static TileOrder& tile_order_latest(TileOrderSet& S, int app) {
    TileOrder* mru = &S.tab[app][0];
    for (auto& E : S.tab[app]) if (E.stamp > mru->stamp) mru = &E;
    TileOrder& T = *mru;
    if (T.pending >= 0 && hipEventSynchronize(T.ready) == hipSuccess) { tile_order_retire(S, T, T.cur); T.cur = T.pending; T.pending = -1; ++T.built; }
    return T;
}
static void tile_order_destroy(TileOrderSet& S) {
    for (auto& per_app : S.tab) for (auto& T : per_app) {
        if (T.mem) (void)hipFree(T.mem);
        if (T.have_ready) { (void)hipEventDestroy(T.ready); (void)hipEventDestroy(T.seen); }
        for (auto& us : T.users) for (auto& u : us) (void)hipEventDestroy(u.second);
        for (auto& rs : T.retired) for (auto& e : rs) (void)hipEventDestroy(e);
        T = TileOrder();
    }
}

}  // namespace sbx
