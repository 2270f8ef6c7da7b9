// shaderbox_amd/csrc/sbx_multi.hip — multi-GPU frames inside the library (include/sbx.h, "sbx_multi_*").
//
// SURVEY.md §8b "Ownership" / §8e "Collective": the library owns the communicator; one PROCESS drives all the GPUs of the
// node (single-process ncclCommInitAll, /opt/rocm/include/rccl/rccl.h:236), so a plain C or C++ host — the role of the
// reference's frame-granular hosts, util/hlsltoy/src/hlsltoy.cpp:494-516 — can shard a frame without MPI, torchrun or a
// second process.  The frame shards as the same cyclic row-blocks as the per-process path (sbx_split_*):
//
//   rank 0 (the owner of the frame) renders its blocks IN PLACE into the caller's frame (sbx_render_split_in_place);
//   rank i > 0 renders its blocks densely into a slab on its own GPU (3 floats per pixel: alpha is the constant 1 of
//       src/main.h:52) and sends the WHOLE slab with ONE ncclSend (rccl.h:700); the root posts ONE ncclRecv per peer (:722)
//       into a staging area, all in one group — N-1 point-to-point transfers over N-1 distinct xGMI links — and one small
//       kernel (k_assemble_peers) scatters the rows and writes the alpha.  (SBX_MULTI_EXCHANGE_BLOCKS, the round-2 form:
//       one send/recv pair per 8-row block straight into the final rows, no staging and no scatter, but ~34 x (N-1)
//       point-to-point operations in one group per 4K frame — kept behind sbx_multi_set_exchange for comparison on a node.)
// Every entry point leaves RANK 0's device current (hipSetDevice), whatever devices it visited.
//
// Two frames may be in flight (double-buffered slabs and two stream sets per rank, alternating per call) so that the
// drain of one frame's kernels overlaps the next frame, exactly as bench.py pipelines the per-process path.
//
// Ranks may share a device (devices[] with repeats).  That is how the N-rank schedule is exercised on fewer GPUs than
// ranks (tests on a 1-GPU box): transfers then are device-to-device copies on the sending rank's stream, and RCCL is not
// loaded.  RCCL itself is dlopen'ed on first use, so a single-GPU host of libsbx never needs it.
#include "../../include/sbx.h"
#include "../../include/sbx_test.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// the slice of the RCCL API this file uses (types as in rccl.h)
typedef void* nccl_comm_t;
typedef int nccl_result_t;
constexpr int kNcclFloat = 7;            // ncclFloat32 (rccl.h ncclDataType_t)
struct RcclApi {
    void* handle = nullptr;
    nccl_result_t (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
    nccl_result_t (*GroupStart)() = nullptr;
    nccl_result_t (*GroupEnd)() = nullptr;
    nccl_result_t (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    nccl_result_t (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(nccl_result_t) = nullptr;
    bool loaded = false;
    std::string load_error;
    bool load(std::string& err) {
        if (loaded) return true;
        if (handle) { (void)dlclose(handle); handle = nullptr; }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            const char* de = dlerror();
            err = load_error = std::string("cannot load librccl: ") + (de ? de : "?");
            return false;
        }
        std::string missing;
        auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) { missing += missing.empty() ? "" : ", "; missing += n; } return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!missing.empty()) {
            // a half-resolved table must never be reachable: the next load() starts from scratch
            CommInitAll = nullptr; CommDestroy = nullptr; GroupStart = nullptr; GroupEnd = nullptr; Send = nullptr; Recv = nullptr;
            GetErrorString = nullptr;
            (void)dlclose(handle);
            handle = nullptr;
            err = load_error = "librccl lacks " + missing;
            return false;
        }
        loaded = true;
        return true;
    }
};
RcclApi g_rccl;

constexpr int kInFlight = 2;

struct Rank {
    int device = 0;
    sbx_ctx* ctx = nullptr;
    hipStream_t render[kInFlight] = {nullptr, nullptr};   // this rank's launches (and its sends / copies)
    hipStream_t recv[kInFlight] = {nullptr, nullptr};     // rank 0 only: RCCL receives, beside its own rendering
    hipEvent_t done[kInFlight] = {};                      // the rank's part of the frame is where it belongs
    hipEvent_t recv_done[kInFlight] = {};
    float* slab[kInFlight] = {nullptr, nullptr};
    size_t slab_floats = 0;
    float* stage[kInFlight] = {nullptr, nullptr};         // rank 0 only: where the peers' slabs land (slab exchange)
    size_t stage_floats = 0;
    nccl_comm_t comm = nullptr;
};

std::string g_create_error;

}  // namespace

struct sbx_multi {
    std::vector<Rank> ranks;
    bool use_rccl = false;
    bool peer_stores_ok = true;                           // every peer's device can address rank 0's memory (or is rank 0's device)
    int block_rows = 8, root_rounds = 1, rounds = 1;
    int exchange = SBX_MULTI_EXCHANGE_SLABS;
    int out_format = SBX_FORMAT_RGBA32F;                  // SBX_FORMAT_RGBA8: every pixel anywhere is ONE 32-bit word (counted as one "float" below)
    unsigned calls = 0;
    hipEvent_t start[kInFlight] = {};
    bool recv_recorded[kInFlight] = {false, false};       // root.recv_done[k] has been recorded at least once
    // the span exchange's slab sizes: sbx_span_table is host code that probes the app's early-exit tests tile by tile (14 ms for an
    // 8K ATMOSPHERE frame at 8 ranks against ~0.5 ms of GPU time per frame), and its result depends on (app, u_res,
    // u_mouse, split) only — computed once per such key, not once per frame
    std::vector<uint32_t> span_key;
    std::vector<int64_t> span_pix;
    std::string err;
};

// every device of the world idle: before buffers that launches of EITHER in-flight frame may still touch are freed (hipFree
// waits for the current device only; a peer's copy into the root's landing area runs on the peer's stream)
static void sync_all_ranks(sbx_multi* m) {
    for (Rank& r : m->ranks) { (void)hipSetDevice(r.device); (void)hipDeviceSynchronize(); }
}

static int mfail(sbx_multi* m, int code, const std::string& what, hipError_t e = hipSuccess) {
    if (m) {
        m->err = what;
        if (e != hipSuccess) { m->err += ": "; m->err += hipGetErrorString(e); }
    }
    return code;
}

extern "C" {

void sbx_multi_destroy(sbx_multi* m) {
    if (!m) return;
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.device);
        (void)hipDeviceSynchronize();
        if (r.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(r.comm);
        for (int k = 0; k < kInFlight; ++k) {
            if (r.slab[k]) (void)hipFree(r.slab[k]);
            if (r.stage[k]) (void)hipFree(r.stage[k]);
            if (r.render[k]) (void)hipStreamDestroy(r.render[k]);
            if (r.recv[k]) (void)hipStreamDestroy(r.recv[k]);
            if (r.done[k]) (void)hipEventDestroy(r.done[k]);
            if (r.recv_done[k]) (void)hipEventDestroy(r.recv_done[k]);
        }
        if (r.ctx) sbx_destroy(r.ctx);
    }
    if (!m->ranks.empty()) {
        (void)hipSetDevice(m->ranks[0].device);
        for (int k = 0; k < kInFlight; ++k) if (m->start[k]) (void)hipEventDestroy(m->start[k]);
    }
    delete m;
}

const char* sbx_multi_create_error(void) { return g_create_error.c_str(); }

int sbx_multi_create(int nranks, const int* devices, sbx_multi** out) {
    if (!out) return SBX_ERR_ARG;
    *out = nullptr;
    g_create_error.clear();
    if (nranks < 1 || nranks > 64 || !devices) { g_create_error = "bad rank count or NULL device list"; return SBX_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device"; return SBX_ERR_NO_DEVICE; }
    bool distinct = true;
    for (int i = 0; i < nranks; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) { g_create_error = "device id out of range"; return SBX_ERR_ARG; }
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) distinct = false;
    }
    sbx_multi* m = new sbx_multi();
    m->ranks.resize(nranks);
    for (int i = 0; i < nranks; ++i) {
        Rank& r = m->ranks[i];
        r.device = devices[i];
        int rc = sbx_create(r.device, &r.ctx);
        if (rc != SBX_OK) { g_create_error = "sbx_create failed on a rank's device"; sbx_multi_destroy(m); return rc; }
        hipError_t e = hipSetDevice(r.device);
        for (int k = 0; k < kInFlight && e == hipSuccess; ++k) {
            if ((e = hipStreamCreateWithFlags(&r.render[k], hipStreamNonBlocking)) != hipSuccess) break;
            if ((e = hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming)) != hipSuccess) break;
            if (i == 0) {
                if ((e = hipStreamCreateWithFlags(&r.recv[k], hipStreamNonBlocking)) != hipSuccess) break;
                if ((e = hipEventCreateWithFlags(&r.recv_done[k], hipEventDisableTiming)) != hipSuccess) break;
                if ((e = hipEventCreateWithFlags(&m->start[k], hipEventDisableTiming)) != hipSuccess) break;
            }
        }
        if (e != hipSuccess) { g_create_error = std::string("stream / event creation: ") + hipGetErrorString(e); sbx_multi_destroy(m); return SBX_ERR_HIP; }
    }
    // peers write into the root's frame (copy mode) / RCCL sets its own peer mappings up: enable peer access where possible
    for (int i = 1; i < nranks; ++i) {
        if (m->ranks[i].device == m->ranks[0].device) continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, m->ranks[i].device, m->ranks[0].device);
        if (can) {
            (void)hipSetDevice(m->ranks[i].device);
            hipError_t e = hipDeviceEnablePeerAccess(m->ranks[0].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); m->peer_stores_ok = false; }
        } else {
            m->peer_stores_ok = false;
        }
    }
    const char* no_rccl = getenv("SBX_MULTI_NO_RCCL");      // diagnostic: peer copies instead of RCCL
    if (distinct && nranks > 1 && !(no_rccl && no_rccl[0] == '1')) {
        std::string err;
        if (!g_rccl.load(err)) { g_create_error = err; fprintf(stderr, "libsbx: %s\n", err.c_str()); sbx_multi_destroy(m); return SBX_ERR_UNSUPPORTED; }
        std::vector<nccl_comm_t> comms(nranks, nullptr);
        const nccl_result_t rc = g_rccl.CommInitAll(comms.data(), nranks, devices);
        if (rc != 0) {
            g_create_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc);
            sbx_multi_destroy(m);
            return SBX_ERR_HIP;
        }
        for (int i = 0; i < nranks; ++i) m->ranks[i].comm = comms[i];
        m->use_rccl = true;
    }
    (void)hipSetDevice(m->ranks[0].device);                  // the caller continues on the frame owner's device
    *out = m;
    return SBX_OK;
}

// Self-test of the RCCL slice this file uses, runnable on ONE GPU: load librccl, ncclCommInitAll over {device}, one grouped
// ncclSend / ncclRecv pair from the rank to itself on two streams (the shape of sbx_multi_render's exchange), compare.
// Returns SBX_OK, or a negative status with the failing step in *step (1 load, 2 init, 3 alloc, 4 group, 5 data).
int sbx_multi_rccl_selftest(int device, int* step) {
    int dummy = 0;
    int& st = step ? *step : dummy;
    std::string err;
    st = 1;
    if (!g_rccl.load(err)) return SBX_ERR_UNSUPPORTED;
    st = 2;
    if (hipSetDevice(device) != hipSuccess) return SBX_ERR_HIP;
    nccl_comm_t comm = nullptr;
    if (g_rccl.CommInitAll(&comm, 1, &device) != 0) return SBX_ERR_HIP;
    st = 3;
    const size_t n = 3840 * 8 * 4;                              // one 8-row block of a 4K frame
    float *a = nullptr, *b = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    int rc = SBX_OK;
    if (hipMalloc((void**)&a, n * 4) != hipSuccess || hipMalloc((void**)&b, n * 4) != hipSuccess ||
        hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) rc = SBX_ERR_HIP;
    std::vector<float> h(n), g(n, 0.f);
    if (rc == SBX_OK) {
        for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 977) * .25f;
        if (hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(b, 0, n * 4) != hipSuccess) rc = SBX_ERR_HIP;
    }
    if (rc == SBX_OK) {
        st = 4;
        nccl_result_t nr = g_rccl.GroupStart();
        if (nr == 0) nr = g_rccl.Send(a, n, kNcclFloat, 0, comm, s1);
        if (nr == 0) nr = g_rccl.Recv(b, n, kNcclFloat, 0, comm, s2);
        const nccl_result_t ne = g_rccl.GroupEnd();
        if (nr != 0 || ne != 0) rc = SBX_ERR_HIP;
        if (hipStreamSynchronize(s1) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess) rc = SBX_ERR_HIP;
    }
    if (rc == SBX_OK) {
        st = 5;
        if (hipMemcpy(g.data(), b, n * 4, hipMemcpyDeviceToHost) != hipSuccess || std::memcmp(g.data(), h.data(), n * 4) != 0) rc = SBX_ERR_HIP;
    }
    if (s1) (void)hipStreamDestroy(s1);
    if (s2) (void)hipStreamDestroy(s2);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)g_rccl.CommDestroy(comm);
    if (rc == SBX_OK) st = 0;
    return rc;
}

int sbx_multi_ranks(const sbx_multi* m) { return m ? (int)m->ranks.size() : SBX_ERR_ARG; }
int sbx_multi_uses_rccl(const sbx_multi* m) { return m ? (m->use_rccl ? 1 : 0) : SBX_ERR_ARG; }
const char* sbx_multi_last_error(sbx_multi* m) { return m ? m->err.c_str() : "no multi-GPU context"; }

int sbx_multi_set_split(sbx_multi* m, int block_rows, int root_rounds, int rounds) {
    if (!m) return SBX_ERR_ARG;
    const int n = (int)m->ranks.size();
    if (block_rows <= 0 || rounds < 1 || root_rounds < 0 || root_rounds > rounds || (n == 1 && root_rounds != rounds))
        return mfail(m, SBX_ERR_ARG, "bad split");
    m->block_rows = block_rows; m->root_rounds = root_rounds; m->rounds = rounds;
    return SBX_OK;
}

int sbx_multi_set_exchange(sbx_multi* m, int mode) {
    if (!m) return SBX_ERR_ARG;
    if (mode != SBX_MULTI_EXCHANGE_SLABS && mode != SBX_MULTI_EXCHANGE_BLOCKS && mode != SBX_MULTI_EXCHANGE_SPANS &&
        mode != SBX_MULTI_EXCHANGE_PEER_STORES)
        return mfail(m, SBX_ERR_ARG, "unknown exchange mode");
    if (mode == SBX_MULTI_EXCHANGE_PEER_STORES && !m->peer_stores_ok)
        return mfail(m, SBX_ERR_UNSUPPORTED, "peer stores need every rank's device to have peer access to rank 0's");
    if (mode == SBX_MULTI_EXCHANGE_PEER_STORES && m->use_rccl) {
        // distinct devices: this form has never run on more than one GPU (no such box was available to any round), and within one
        // process it orders a slot's reuse through the caller's stream only.  It stays available for the first multi-GPU run, but a
        // host has to ask for it twice; the multi-process form with explicit flags is sbx_shared_* (include/sbx.h).
        const char* en = std::getenv("SBX_ENABLE_PEER_STORES");
        if (!en || en[0] != '1')
            return mfail(m, SBX_ERR_UNSUPPORTED, "peer stores across distinct devices are unvalidated: set SBX_ENABLE_PEER_STORES=1 to use them "
                                                  "(or use the store exchange of sbx_shared_*)");
    }
    m->exchange = mode;
    return SBX_OK;
}

int sbx_multi_set_output_format(sbx_multi* m, int format) {
    if (!m) return SBX_ERR_ARG;
    if (format != SBX_FORMAT_RGBA32F && format != SBX_FORMAT_RGBA8) return mfail(m, SBX_ERR_ARG, "unknown output format");
    for (Rank& r : m->ranks) {
        const int rc = sbx_set_output_format(r.ctx, format);
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
    }
    m->out_format = format;
    return SBX_OK;
}
int sbx_multi_set_variant(sbx_multi* m, int variant) {
    if (!m) return SBX_ERR_ARG;
    for (Rank& r : m->ranks) {
        const int rc = sbx_set_variant(r.ctx, variant);
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
    }
    return SBX_OK;
}

int sbx_multi_set_noise_volumes(sbx_multi* m, int shape_size, const float* shape_rgba, int detail_size, const float* detail_rgba) {
    if (!m) return SBX_ERR_ARG;
    if (!shape_rgba || !detail_rgba || shape_size <= 0 || detail_size <= 0) return mfail(m, SBX_ERR_ARG, "bad noise volume arguments");
    Rank& root = m->ranks[0];
    struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{root.device};   // whatever path returns
    const size_t b1 = (size_t)shape_size * shape_size * shape_size * 16, b2 = (size_t)detail_size * detail_size * detail_size * 16;
    for (Rank& r : m->ranks) {
        hipError_t e = hipSetDevice(r.device);
        if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        int rc;
        if (r.device == root.device) {
            rc = sbx_set_noise_volumes(r.ctx, shape_size, shape_rgba, detail_size, detail_rgba, nullptr);
            (void)hipDeviceSynchronize();
        } else {                               // the volumes live on rank 0's device: stage a copy on this rank's device
            float *t1 = nullptr, *t2 = nullptr;
            if ((e = hipMalloc((void**)&t1, b1)) != hipSuccess || (e = hipMalloc((void**)&t2, b2)) != hipSuccess) {
                if (t1) (void)hipFree(t1);
                return mfail(m, SBX_ERR_HIP, "hipMalloc", e);
            }
            e = hipMemcpyPeer(t1, r.device, shape_rgba, root.device, b1);
            if (e == hipSuccess) e = hipMemcpyPeer(t2, r.device, detail_rgba, root.device, b2);
            rc = e == hipSuccess ? sbx_set_noise_volumes(r.ctx, shape_size, t1, detail_size, t2, nullptr) : SBX_ERR_HIP;
            (void)hipDeviceSynchronize();
            (void)hipFree(t1); (void)hipFree(t2);
            if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipMemcpyPeer", e);
        }
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
    }
    return SBX_OK;
}

// global block index of local block `lb` of `rank` (the arithmetic of sbx_frame.h row_to_y / shard.py)
static int global_block(int lb, int rank, int nranks, int root_rounds, int rounds) {
    const int cnt = rank == 0 ? root_rounds : rounds;
    const int cycle = lb / cnt, round = lb - cycle * cnt;
    const int v = round < root_rounds ? round * nranks + rank
                                      : root_rounds * nranks + (round - root_rounds) * (nranks - 1) + (rank - 1);
    return cycle * (root_rounds * nranks + (rounds - root_rounds) * (nranks - 1)) + v;
}

// The span exchange (include/sbx.h "span exchange") inside the library: peers render and send only the spans of their blocks,
// packed; rank 0 renders its blocks and everything outside the spans in one launch over the frame and scatters the packed slabs.
static int render_spans(sbx_multi* m, int app, const sbx_uniforms* uni, const void* aux, float* frame, hipStream_t user, int k, int W, int H) {
    const int n = (int)m->ranks.size();
    const int br = m->block_rows, m0 = m->root_rounds, mr = m->rounds;
    Rank& root = m->ranks[0];
    hipError_t e;
    const size_t epp = m->out_format == SBX_FORMAT_RGBA8 ? 1 : 3;      // 32-bit words per pixel of a span slab
    std::vector<uint32_t> key(10, 0u);
    key[0] = (uint32_t)app; key[1] = (uint32_t)br; key[2] = (uint32_t)n; key[3] = (uint32_t)m0; key[4] = (uint32_t)mr;
    std::memcpy(&key[5], uni->u_res, 8);
    std::memcpy(&key[7], uni->u_mouse, 8);
    if (key != m->span_key || (int)m->span_pix.size() != n) {
        std::vector<int64_t> fresh(n, 0);
        const int nb = sbx_span_table(app, uni, aux, br, n, m0, mr, nullptr, fresh.data(), nullptr);
        if (nb < 0) return mfail(m, nb, "bad span table arguments (app, u_res or split)");
        m->span_pix = fresh;
        m->span_key = key;
    }
    const std::vector<int64_t>& pix = m->span_pix;
    int64_t stride = 0;
    for (int i = 1; i < n; ++i) stride = pix[i] > stride ? pix[i] : stride;
    stride = (stride + 63) / 64 * 64;
    const size_t need_stage = (size_t)(n - 1) * (size_t)stride * epp;
    if (need_stage > root.stage_floats) {
        sync_all_ranks(m);
        (void)hipSetDevice(root.device);
        for (int q = 0; q < kInFlight; ++q) {
            if (root.stage[q]) (void)hipFree(root.stage[q]);
            root.stage[q] = nullptr;
            if ((e = hipMalloc((void**)&root.stage[q], (need_stage ? need_stage : 1) * sizeof(float))) != hipSuccess) { root.stage_floats = 0; return mfail(m, SBX_ERR_HIP, "hipMalloc stage", e); }
        }
        root.stage_floats = need_stage;
    }
    for (int i = 0; i < n; ++i) {
        Rank& r = m->ranks[i];
        if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        if ((e = hipStreamWaitEvent(r.render[k], m->start[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        if (m->recv_recorded[k] && (e = hipStreamWaitEvent(r.render[k], root.recv_done[k], 0)) != hipSuccess)
            return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        int rc;
        if (i == 0) {
            rc = sbx_render_span_root(r.ctx, app, uni, aux, br, n, m0, mr, frame, r.render[k]);
        } else {
            const size_t need = (size_t)(pix[i] > 0 ? pix[i] : 1) * epp;
            if (need > r.slab_floats) {
                sync_all_ranks(m);
                (void)hipSetDevice(r.device);
                for (int q = 0; q < kInFlight; ++q) {
                    if (r.slab[q]) (void)hipFree(r.slab[q]);
                    r.slab[q] = nullptr;
                    if ((e = hipMalloc((void**)&r.slab[q], need * sizeof(float))) != hipSuccess) { r.slab_floats = 0; return mfail(m, SBX_ERR_HIP, "hipMalloc slab", e); }
                }
                r.slab_floats = need;
            }
            rc = sbx_render_span_peer(r.ctx, app, uni, aux, br, i, n, m0, mr, 0, 0x7fffffff, r.slab[k], r.render[k]);
        }
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
    }
    if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
    if ((e = hipStreamWaitEvent(root.recv[k], m->start[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    if (m->use_rccl) {
        std::string where = "ncclGroupStart";
        nccl_result_t nr = g_rccl.GroupStart();
        auto check = [&](nccl_result_t rc, const char* call, int rank) {
            if (nr == 0 && rc != 0) { nr = rc; where = std::string(call) + " (rank " + std::to_string(rank) + ")"; }
            return nr == 0;
        };
        for (int i = 1; i < n && nr == 0; ++i) {
            if (pix[i] <= 0) continue;
            Rank& r = m->ranks[i];
            const size_t floats = (size_t)pix[i] * epp;
            if (check(g_rccl.Send(r.slab[k], floats, kNcclFloat, 0, r.comm, r.render[k]), "ncclSend", i))
                check(g_rccl.Recv(root.stage[k] + (size_t)(i - 1) * (size_t)stride * epp, floats, kNcclFloat, i, root.comm, root.recv[k]), "ncclRecv", i);
        }
        const nccl_result_t ne = g_rccl.GroupEnd();
        if (nr != 0) return mfail(m, SBX_ERR_HIP, "RCCL " + where + ": " + g_rccl.GetErrorString(nr));
        if (ne != 0) return mfail(m, SBX_ERR_HIP, std::string("RCCL ncclGroupEnd: ") + g_rccl.GetErrorString(ne));
    } else {
        for (int i = 1; i < n; ++i) {
            Rank& r = m->ranks[i];
            if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
            if (pix[i] > 0) {
                float* dst = root.stage[k] + (size_t)(i - 1) * (size_t)stride * epp;
                const size_t bytes = (size_t)pix[i] * epp * sizeof(float);
                if (r.device == root.device) e = hipMemcpyAsync(dst, r.slab[k], bytes, hipMemcpyDeviceToDevice, r.render[k]);
                else e = hipMemcpyPeerAsync(dst, root.device, r.slab[k], r.device, bytes, r.render[k]);
                if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "span slab copy", e);
            }
            if ((e = hipEventRecord(r.done[k], r.render[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
        }
        if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        for (int i = 1; i < n; ++i)
            if ((e = hipStreamWaitEvent(root.recv[k], m->ranks[i].done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    }
    // the scatter follows the receives only: it and rank 0's own launch over the frame write disjoint pixels (that launch leaves
    // the peers' spans alone), so they may run side by side
    {
        const int rc = sbx_assemble_spans(root.ctx, app, uni, aux, br, n, m0, mr, root.stage[k], stride, frame, root.recv[k]);
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(root.ctx));
    }
    if ((e = hipEventRecord(root.recv_done[k], root.recv[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
    m->recv_recorded[k] = true;
    for (int i = 0; i < n; ++i) {
        Rank& r = m->ranks[i];
        if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        if ((e = hipEventRecord(r.done[k], r.render[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
    }
    if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
    for (int i = 0; i < n; ++i)
        if ((e = hipStreamWaitEvent(user, m->ranks[i].done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    if ((e = hipStreamWaitEvent(user, root.recv_done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    return SBX_OK;
}

int sbx_multi_render(sbx_multi* m, int app, const sbx_uniforms* uni, const void* aux, float* frame, void* stream) {
    if (!m) return SBX_ERR_ARG;
    if (!uni || !frame) return mfail(m, SBX_ERR_ARG, "NULL uniforms or frame");
    const int W = (int)uni->u_res[0], H = (int)uni->u_res[1];
    if (W <= 0 || H <= 0 || (float)W != uni->u_res[0] || (float)H != uni->u_res[1]) return mfail(m, SBX_ERR_ARG, "bad u_res");
    const int n = (int)m->ranks.size();
    const int br = m->block_rows, m0 = m->root_rounds, mr = m->rounds;
    const int k = (int)(m->calls++ % kInFlight);
    hipStream_t user = (hipStream_t)stream;
    Rank& root = m->ranks[0];
    hipError_t e = hipSetDevice(root.device);
    if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
    // everything this frame does starts after what the caller has already enqueued on `stream` (e.g. the last reader of `frame`)
    if ((e = hipEventRecord(m->start[k], user)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
    struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{root.device};   // whatever path returns
    if (m->exchange == SBX_MULTI_EXCHANGE_SPANS && n > 1) return render_spans(m, app, uni, aux, frame, user, k, W, H);
    if (m->exchange == SBX_MULTI_EXCHANGE_PEER_STORES) {
        // No slab, no landing area, no scatter, no RCCL: every rank renders its row-blocks IN PLACE into rank 0's frame — a peer's
        // stores are 16-byte float4 writes, 1 KB per wave, that travel over its xGMI link as the kernel produces them (peer access
        // to rank 0's memory, enabled at creation).  The exchange step is fused into the render kernels' stores; rank 0 does
        // nothing for the others.  16 instead of 12 bytes per pixel cross the link.
        for (int i = 0; i < n; ++i) {
            Rank& r = m->ranks[i];
            if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
            if ((e = hipStreamWaitEvent(r.render[k], m->start[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
            const int rc = sbx_render_split_in_place(r.ctx, app, uni, aux, br, i, n, m0, mr, frame, r.render[k]);
            if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
            if ((e = hipEventRecord(r.done[k], r.render[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
        }
        if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        for (int i = 0; i < n; ++i)
            if ((e = hipStreamWaitEvent(user, m->ranks[i].done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        return SBX_OK;
    }
    const bool slabs = m->exchange != SBX_MULTI_EXCHANGE_BLOCKS;
    const bool rgba8 = m->out_format == SBX_FORMAT_RGBA8;
    const int ch = rgba8 ? 1 : (slabs ? 3 : 4);                      // 32-bit words per pixel of a peer's slab
    const size_t row_floats = (size_t)W * (rgba8 ? 1 : 4), slab_row = (size_t)W * ch;
    const int rows_max = sbx_split_rows_max(H, br, n, m0, mr);
    if (rows_max < 0) return mfail(m, SBX_ERR_ARG, "bad split");
    if (n > 1 && slabs) {                                           // the root's landing area: (n - 1) slabs of rows_max rows
        const size_t need = (size_t)(n - 1) * rows_max * slab_row;
        if (need > root.stage_floats) {
            sync_all_ranks(m);                                      // the other in-flight frame may still be landing in stage[1 - k]
            (void)hipSetDevice(root.device);
            for (int q = 0; q < kInFlight; ++q) {
                if (root.stage[q]) (void)hipFree(root.stage[q]);
                root.stage[q] = nullptr;
                if ((e = hipMalloc((void**)&root.stage[q], need * sizeof(float))) != hipSuccess) { root.stage_floats = 0; return mfail(m, SBX_ERR_HIP, "hipMalloc stage", e); }
            }
            root.stage_floats = need;
        }
    }
    // ---- every rank renders its share -------------------------------------------------------------------
    for (int i = 0; i < n; ++i) {
        Rank& r = m->ranks[i];
        if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        if ((e = hipStreamWaitEvent(r.render[k], m->start[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        // slot k's landing area (and, in copy mode, the rows the peers write into the frame) is free again only when the root has
        // scattered the frame that used it last: order that explicitly instead of through the caller's stream, which may differ
        // from frame to frame
        if (m->recv_recorded[k] && (e = hipStreamWaitEvent(r.render[k], root.recv_done[k], 0)) != hipSuccess)
            return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        int rc;
        if (i == 0) {
            rc = sbx_render_split_in_place(r.ctx, app, uni, aux, br, 0, n, m0, mr, frame, r.render[k]);
        } else {
            const size_t need = (size_t)rows_max * row_floats;       // sized for either exchange
            if (need > r.slab_floats) {
                sync_all_ranks(m);                                    // a transfer out of slab[1 - k] may still be running (RCCL: on the root's side too)
                (void)hipSetDevice(r.device);
                for (int q = 0; q < kInFlight; ++q) {
                    if (r.slab[q]) (void)hipFree(r.slab[q]);
                    r.slab[q] = nullptr;
                    if ((e = hipMalloc((void**)&r.slab[q], need * sizeof(float))) != hipSuccess) { r.slab_floats = 0; return mfail(m, SBX_ERR_HIP, "hipMalloc slab", e); }
                }
                r.slab_floats = need;
            }
            rc = slabs ? sbx_render_split_rgb(r.ctx, app, uni, aux, br, i, n, m0, mr, 0, 0x7fffffff, r.slab[k], r.render[k])
                       : sbx_render_split(r.ctx, app, uni, aux, br, i, n, m0, mr, 0, 0x7fffffff, r.slab[k], r.render[k]);
        }
        if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(r.ctx));
    }
    // ---- the one exchange step ----------------------------------------------------------------------------
    if (n > 1) {
        if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        if ((e = hipStreamWaitEvent(root.recv[k], m->start[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        if (m->use_rccl) {
            // every call inside the group is checked; the first failure is remembered with its call and rank, the group is
            // closed all the same (an open group would swallow every later RCCL call of the process), and the failure is what
            // the caller gets, with ncclGetErrorString's text
            std::string where = "ncclGroupStart";
            nccl_result_t nr = g_rccl.GroupStart();
            auto check = [&](nccl_result_t rc, const char* call, int rank) {
                if (nr == 0 && rc != 0) { nr = rc; where = std::string(call) + " (rank " + std::to_string(rank) + ")"; }
                return nr == 0;
            };
            for (int i = 1; i < n && nr == 0; ++i) {
                Rank& r = m->ranks[i];
                const int rows = sbx_split_rank_rows(H, br, i, n, m0, mr);
                if (slabs) {                                        // one send / one receive per peer: the whole slab
                    if (rows <= 0) continue;
                    const size_t floats = (size_t)rows * slab_row;
                    if (check(g_rccl.Send(r.slab[k], floats, kNcclFloat, 0, r.comm, r.render[k]), "ncclSend", i))
                        check(g_rccl.Recv(root.stage[k] + (size_t)(i - 1) * rows_max * slab_row, floats, kNcclFloat, i, root.comm, root.recv[k]), "ncclRecv", i);
                    continue;
                }
                for (int lr = 0, lb = 0; lr < rows && nr == 0; lr += br, ++lb) {       // one pair per row-block, into the final rows
                    const int y = global_block(lb, i, n, m0, mr) * br;
                    const int cnt = (y + br <= H) ? br : (H - y);
                    const size_t floats = (size_t)cnt * row_floats;
                    if (check(g_rccl.Send(r.slab[k] + (size_t)lr * row_floats, floats, kNcclFloat, 0, r.comm, r.render[k]), "ncclSend", i))
                        check(g_rccl.Recv(frame + (size_t)y * row_floats, floats, kNcclFloat, i, root.comm, root.recv[k]), "ncclRecv", i);
                }
            }
            const nccl_result_t ne = g_rccl.GroupEnd();
            if (nr != 0) return mfail(m, SBX_ERR_HIP, "RCCL " + where + ": " + g_rccl.GetErrorString(nr));
            if (ne != 0) return mfail(m, SBX_ERR_HIP, std::string("RCCL ncclGroupEnd: ") + g_rccl.GetErrorString(ne));
        } else {
            for (int i = 1; i < n; ++i) {
                Rank& r = m->ranks[i];
                if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
                const int rows = sbx_split_rank_rows(H, br, i, n, m0, mr);
                if (slabs) {
                    if (rows > 0) {
                        float* dst = root.stage[k] + (size_t)(i - 1) * rows_max * slab_row;
                        const size_t bytes = (size_t)rows * slab_row * sizeof(float);
                        if (r.device == root.device) e = hipMemcpyAsync(dst, r.slab[k], bytes, hipMemcpyDeviceToDevice, r.render[k]);
                        else e = hipMemcpyPeerAsync(dst, root.device, r.slab[k], r.device, bytes, r.render[k]);
                        if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "slab copy", e);
                    }
                } else {
                    for (int lr = 0, lb = 0; lr < rows; lr += br, ++lb) {
                        const int y = global_block(lb, i, n, m0, mr) * br;
                        const int cnt = (y + br <= H) ? br : (H - y);
                        const size_t bytes = (size_t)cnt * row_floats * sizeof(float);
                        if (r.device == root.device)
                            e = hipMemcpyAsync(frame + (size_t)y * row_floats, r.slab[k] + (size_t)lr * row_floats, bytes, hipMemcpyDeviceToDevice, r.render[k]);
                        else
                            e = hipMemcpyPeerAsync(frame + (size_t)y * row_floats, root.device, r.slab[k] + (size_t)lr * row_floats, r.device, bytes, r.render[k]);
                        if (e != hipSuccess) return mfail(m, SBX_ERR_HIP, "row-block copy", e);
                    }
                }
                // the copies of this peer are complete at this point of its stream: the root's scatter waits for it
                if ((e = hipEventRecord(r.done[k], r.render[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
            }
            if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
            if (slabs)
                for (int i = 1; i < n; ++i)
                    if ((e = hipStreamWaitEvent(root.recv[k], m->ranks[i].done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
        }
        if (slabs) {
            // the peers' rows to their places, alpha = 1 (rank 0's rows, rendered in place beside this, are not touched)
            const int rc = sbx_assemble_peers(root.ctx, W, H, br, n, m0, mr, 3, root.stage[k], frame, root.recv[k]);
            if (rc != SBX_OK) return mfail(m, rc, sbx_last_error(root.ctx));
        }
        if ((e = hipEventRecord(root.recv_done[k], root.recv[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
        m->recv_recorded[k] = true;
    }
    // ---- the caller's stream continues when every part of the frame is in place ---------------------------------
    for (int i = 0; i < n; ++i) {
        Rank& r = m->ranks[i];
        if ((e = hipSetDevice(r.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
        if ((e = hipEventRecord(r.done[k], r.render[k])) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipEventRecord", e);
    }
    if ((e = hipSetDevice(root.device)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipSetDevice", e);
    for (int i = 0; i < n; ++i)
        if ((e = hipStreamWaitEvent(user, m->ranks[i].done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    if (n > 1)
        if ((e = hipStreamWaitEvent(user, root.recv_done[k], 0)) != hipSuccess) return mfail(m, SBX_ERR_HIP, "hipStreamWaitEvent", e);
    return SBX_OK;
}

}  // extern "C"
