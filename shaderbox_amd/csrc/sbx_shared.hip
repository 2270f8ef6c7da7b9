// shaderbox_amd/csrc/sbx_shared.hip — the store exchange (include/sbx.h sbx_shared_*): a frame that the ranks of a multi-GPU
// split render INTO, in place, each from its own device.
//
// The reference has no multi-device path.  The split's one exchange step — the peers' row-blocks reaching the
// frame's owner — is here the peers' own pixel stores: a peer maps the owner's frame (hipIpcOpenMemHandle across processes; the
// pointer itself inside one process) and its render kernel writes its rows where they belong, over xGMI.  The owner receives
// nothing, lands nothing, scatters nothing: it is an ordinary rank.
//
// Ordering.  Pixel visibility rides on KERNEL BOUNDARIES (the HIP peer-access model): a peer's render kernel ends before its signal
// kernel starts, and between the two the stream records an event made with hipEventReleaseToSystem — the boundary of two kernels of
// one stream by itself promises an AGENT-scope release only, and the signal kernel's own __threadfence_system runs in one wave
// (no run on two devices has been possible yet, so the release is asked for explicitly rather than assumed); the owner's wait kernel ends before the
// stream's next kernel starts (system-scope acquire).  The flags only carry the order, and they are polled INSIDE a running kernel,
// so they live in their own page of FINE-GRAINED device memory (hipDeviceMallocFinegrained: uncached for system-scope atomics on
// every agent) next to the coarse-grained frame:
//     word 0             release: the owner's frame counter — frame q of this object may be overwritten once release >= q
//     word 16 * r        done[r]: peer r's frame counter — its rows of frame q are in place once done[r] >= q
// Every side counts its own frames (begin increments), so no sequence number crosses the API.
//
// Where the flag kernels run: ALL FOUR on the caller's stream.  Moving the two sets (which never spin) to a side stream behind an
// event, so that the render stream carries [wait][render] instead of [wait][render][set], was measured and is SLOWER — an event
// record plus a cross-stream wait per frame cost more than a one-thread kernel in line (eighth-frames of an 8-rank CLOUDS split:
// 0.304 -> 0.355 ms per frame, profiles/r05_log.md).
#include "../../include/sbx.h"
#include "../../include/sbx_test.h"
#include "sbx_device.h"
#include <cstring>
#include <unistd.h>

namespace sbx {

constexpr int FLAG_STRIDE = 16;                  // words between two flags: one 64-byte line each
constexpr size_t FLAG_BYTES = 4096;

// `abort` (a word of the object's own device, or nullptr): the frame number whose wait gave up on this side.  A side that never saw
// the frame released must not report its rows as in place: the owner's wait then gives up too, and both sides hold the
// fault word instead of one of them holding a frame that was overwritten under it.
__global__ void k_flag_set(unsigned* flag, unsigned seq, const unsigned* abort) {
    if (abort && *abort == seq) return;
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave: lane i < count polls flags[(first + i) * FLAG_STRIDE] until it has reached `seq` (counters wrap: signed distance).
// s_sleep between polls keeps the wave off its SIMD's issue slots; the bound turns a lost peer into a fault word instead of a hung
// device (wall_clock64 = the constant 100 MHz counter).
__global__ void __launch_bounds__(64) k_flag_wait(const unsigned* flags, int first, int count, unsigned seq, unsigned* fault,
                                                   long long timeout_ticks, unsigned* abort) {
    const int i = (int)threadIdx.x;
    const long long t0 = wall_clock64();
    bool ok = i >= count;
    for (;;) {
        if (!ok) {
            const unsigned v = __hip_atomic_load(flags + (size_t)(first + i) * FLAG_STRIDE, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = (int)(v - seq) >= 0;
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > timeout_ticks) {
            if (i == 0 && fault) __hip_atomic_store(fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (i == 0 && abort) *abort = seq;
            break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    __threadfence_system();
}

// float4(0, 0, 0, 1) everywhere: the alpha that three-dword pixel stores (RowMap.rgb == 3) never touch
__global__ void __launch_bounds__(256) k_fill_alpha(float4* frame, size_t pixels) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < pixels) frame[i] = make_float4(0.f, 0.f, 0.f, 1.0f);
}

// The landing model of sbx_test.h: `workgroups` resident workgroups copy `bytes` at the pace that makes the whole take
// `ticks` of the 100 MHz clock — chunk c of a workgroup is not started before its share of the time has passed.
__global__ void __launch_bounds__(256) k_model_landing(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                        long long ticks) {
    const long long t0 = wall_clock64();
    const size_t per_wg = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per_wg, hi = lo + per_wg < n16 ? lo + per_wg : n16;
    constexpr size_t CHUNK = 256 * 8;                               // 32 KB per workgroup and step
    const size_t nchunks = hi > lo ? (hi - lo + CHUNK - 1) / CHUNK : 0;
    for (size_t c = 0; c < nchunks; ++c) {
        const long long due = t0 + (long long)((double)ticks * (double)c / (double)nchunks);
        while (wall_clock64() < due) __builtin_amdgcn_s_sleep(8);
        for (size_t k = 0; k < 8; ++k) {
            const size_t i = lo + c * CHUNK + k * 256 + threadIdx.x;
            if (i < hi) dst[i] = src[i];
        }
    }
    while (wall_clock64() < t0 + ticks) __builtin_amdgcn_s_sleep(8);
}

unsigned* fault_word_device(int device);         // sbx_capi.hip: the device's sticky fault word, as a device pointer
int ctx_device(const sbx_ctx* ctx);
int ctx_fail(sbx_ctx* ctx, int code, const char* what, hipError_t e);

}  // namespace sbx

using namespace sbx;

struct HandleBlob {                              // what sbx_shared_handle.opaque holds
    hipIpcMemHandle_t frame, flags;              // 2 x 64 bytes
    uint64_t pid;
    uint64_t frame_bytes;
    uint64_t frame_ptr, flags_ptr;               // valid in process `pid` only (ranks of one process share the pointers)
    int32_t nranks, device;
    uint32_t magic;
    uint32_t ipc_ok;                             // the two IPC handles are valid (hipIpcGetMemHandle worked in the exporting process)
};
static_assert(sizeof(HandleBlob) <= sizeof(sbx_shared_handle), "handle too small");
static const uint32_t kMagic = 0x53425853u;      // "SXBS"

struct sbx_shared {
    sbx_ctx* ctx = nullptr;
    int device = 0;
    bool owner = false, mapped = false;          // mapped: the pointers came from hipIpcOpenMemHandle
    float* frame = nullptr;
    unsigned* flags = nullptr;
    size_t frame_bytes = 0;
    int nranks = 1;
    unsigned* abort = nullptr;                   // device word: the frame whose wait gave up here (k_flag_wait / k_flag_set)
    hipEvent_t release = nullptr;                // hipEventReleaseToSystem: recorded between a peer's render and its signal
    unsigned seq[64] = {0};                      // per rank driven through this object (one process may drive several)
    long long timeout_ticks = 10000ll * 100000ll;  // 10 s at 100 MHz
    HandleBlob blob{};
};

static hipError_t shared_extras(sbx_shared* s) {
    hipError_t e = hipMalloc((void**)&s->abort, 64);
    if (e == hipSuccess) e = hipMemset(s->abort, 0, 64);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->release, hipEventDisableTiming | hipEventReleaseToSystem);
    return e;
}
static void shared_extras_free(sbx_shared* s) {
    if (s->abort) (void)hipFree(s->abort);
    if (s->release) (void)hipEventDestroy(s->release);
}

extern "C" {

int sbx_shared_create(sbx_ctx* ctx, size_t frame_bytes, int nranks, sbx_shared** out) {
    if (!ctx || !out) return SBX_ERR_ARG;
    *out = nullptr;
    if (frame_bytes == 0 || (frame_bytes & 3u) || nranks < 1 || nranks > 64) return ctx_fail(ctx, SBX_ERR_ARG, "sbx_shared_create: bad size or rank count", hipSuccess);
    hipError_t e = hipSetDevice(ctx_device(ctx));
    if (e != hipSuccess) return ctx_fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    sbx_shared* s = new sbx_shared();
    s->ctx = ctx; s->device = ctx_device(ctx); s->owner = true; s->frame_bytes = frame_bytes; s->nranks = nranks;
    if ((e = hipMalloc((void**)&s->frame, frame_bytes)) != hipSuccess ||
        (e = hipExtMallocWithFlags((void**)&s->flags, FLAG_BYTES, hipDeviceMallocFinegrained)) != hipSuccess ||
        (e = hipMemset(s->flags, 0, FLAG_BYTES)) != hipSuccess || (e = shared_extras(s)) != hipSuccess) {
        if (s->frame) (void)hipFree(s->frame);
        if (s->flags) (void)hipFree(s->flags);
        shared_extras_free(s);
        delete s;
        return ctx_fail(ctx, SBX_ERR_HIP, "sbx_shared_create: allocation", e);
    }
    if ((frame_bytes & 15u) == 0) {
        const size_t px = frame_bytes / 16;
        hipLaunchKernelGGL(k_fill_alpha, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, nullptr, reinterpret_cast<float4*>(s->frame), px);
    } else {
        (void)hipMemsetAsync(s->frame, 0, frame_bytes, nullptr);
    }
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) {
        (void)hipFree(s->frame); (void)hipFree(s->flags); shared_extras_free(s); delete s;
        return ctx_fail(ctx, SBX_ERR_HIP, "sbx_shared_create: first touch", e);
    }
    HandleBlob& b = s->blob;
    std::memset(&b, 0, sizeof(b));
    b.pid = (uint64_t)getpid(); b.frame_bytes = frame_bytes; b.nranks = nranks; b.device = s->device; b.magic = kMagic;
    b.frame_ptr = (uint64_t)(uintptr_t)s->frame; b.flags_ptr = (uint64_t)(uintptr_t)s->flags;
    // the IPC handles are taken lazily (sbx_shared_export): ranks of one process never need them
    *out = s;
    return SBX_OK;
}

int sbx_shared_export(sbx_shared* s, sbx_shared_handle* handle) {
    if (!s || !handle) return SBX_ERR_ARG;
    if (!s->owner) return ctx_fail(s->ctx, SBX_ERR_ARG, "sbx_shared_export: only the owner exports", hipSuccess);
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return ctx_fail(s->ctx, SBX_ERR_HIP, "hipSetDevice", e);
    // Ranks of THIS process open the blob by its pointers and never look at the IPC handles: a host without HIP IPC still gets a blob
    // that works inside the process (ipc_ok = 0 makes sbx_shared_open in another process say why it cannot).
    s->blob.ipc_ok = 1;
    if ((e = hipIpcGetMemHandle(&s->blob.frame, s->frame)) != hipSuccess || (e = hipIpcGetMemHandle(&s->blob.flags, s->flags)) != hipSuccess) {
        (void)hipGetLastError();
        s->blob.ipc_ok = 0;
    }
    std::memset(handle, 0, sizeof(*handle));
    std::memcpy(handle->opaque, &s->blob, sizeof(s->blob));
    return SBX_OK;
}

int sbx_shared_open(sbx_ctx* ctx, const sbx_shared_handle* handle, sbx_shared** out) {
    if (!ctx || !handle || !out) return SBX_ERR_ARG;
    *out = nullptr;
    HandleBlob b;
    std::memcpy(&b, handle->opaque, sizeof(b));
    if (b.magic != kMagic || b.nranks < 1 || b.nranks > 64 || b.frame_bytes == 0) return ctx_fail(ctx, SBX_ERR_ARG, "sbx_shared_open: not a handle of sbx_shared_export", hipSuccess);
    hipError_t e = hipSetDevice(ctx_device(ctx));
    if (e != hipSuccess) return ctx_fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    sbx_shared* s = new sbx_shared();
    s->ctx = ctx; s->device = ctx_device(ctx); s->frame_bytes = (size_t)b.frame_bytes; s->nranks = b.nranks; s->blob = b;
    if (b.pid == (uint64_t)getpid()) {
        // another rank of the owner's process: the same pointers (a rank on another device reaches them through peer access)
        s->frame = (float*)(uintptr_t)b.frame_ptr;
        s->flags = (unsigned*)(uintptr_t)b.flags_ptr;
        if (b.device != s->device) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, s->device, b.device);
            if (!can) { delete s; return ctx_fail(ctx, SBX_ERR_UNSUPPORTED, "sbx_shared_open: no peer access to the owner's device", hipSuccess); }
            e = hipDeviceEnablePeerAccess(b.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { delete s; return ctx_fail(ctx, SBX_ERR_HIP, "hipDeviceEnablePeerAccess", e); }
            (void)hipGetLastError();
        }
    } else {
        void *pf = nullptr, *pg = nullptr;
        if (!b.ipc_ok) { delete s; return ctx_fail(ctx, SBX_ERR_UNSUPPORTED, "sbx_shared_open: the exporting process could not take HIP IPC handles (is HSA_ENABLE_IPC_MODE_LEGACY=0 set there?)", hipSuccess); }
        if ((e = hipIpcOpenMemHandle(&pf, b.frame, hipIpcMemLazyEnablePeerAccess)) != hipSuccess ||
            (e = hipIpcOpenMemHandle(&pg, b.flags, hipIpcMemLazyEnablePeerAccess)) != hipSuccess) {
            if (pf) (void)hipIpcCloseMemHandle(pf);
            delete s;
            return ctx_fail(ctx, SBX_ERR_HIP, "hipIpcOpenMemHandle (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", e);
        }
        s->frame = (float*)pf; s->flags = (unsigned*)pg; s->mapped = true;
    }
    if ((e = shared_extras(s)) != hipSuccess) {
        if (s->mapped) { (void)hipIpcCloseMemHandle(s->frame); (void)hipIpcCloseMemHandle(s->flags); }
        shared_extras_free(s);
        delete s;
        return ctx_fail(ctx, SBX_ERR_HIP, "sbx_shared_open: allocation", e);
    }
    *out = s;
    return SBX_OK;
}

void sbx_shared_close(sbx_shared* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->owner) {
        (void)hipDeviceSynchronize();
        (void)hipFree(s->frame);
        (void)hipFree(s->flags);
    } else if (s->mapped) {
        (void)hipDeviceSynchronize();
        (void)hipIpcCloseMemHandle(s->frame);
        (void)hipIpcCloseMemHandle(s->flags);
    }
    shared_extras_free(s);
    delete s;
}

float* sbx_shared_frame(sbx_shared* s) { return s ? s->frame : nullptr; }
size_t sbx_shared_bytes(sbx_shared* s) { return s ? s->frame_bytes : 0; }

int sbx_shared_set_timeout_ms(sbx_shared* s, int ms) {
    if (!s || ms <= 0) return SBX_ERR_ARG;
    s->timeout_ticks = (long long)ms * 100000ll;
    return SBX_OK;
}

static int flag_launch_done(sbx_shared* s, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctx_fail(s->ctx, SBX_ERR_HIP, what, e);
    return SBX_OK;
}

int sbx_shared_frame_begin(sbx_shared* s, int rank, void* stream) {
    if (!s) return SBX_ERR_ARG;
    if (rank < 0 || rank >= s->nranks || (rank == 0) != s->owner) {
        // (inside one process the owner's object may be shared: rank 0 must use the owner's, peers an opened one)
        return ctx_fail(s->ctx, SBX_ERR_ARG, "sbx_shared_frame_begin: rank 0 drives the created object, the peers an opened one", hipSuccess);
    }
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return ctx_fail(s->ctx, SBX_ERR_HIP, "hipSetDevice", e);
    unsigned* fault = fault_word_device(s->device);
    if (!fault) return ctx_fail(s->ctx, SBX_ERR_HIP, "sbx_shared_frame_begin: the device's fault word is not bound (a wait that gives up could not say so)", hipSuccess);
    const unsigned q = s->seq[rank] + 1u;
    if (rank == 0)
        hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, (hipStream_t)stream, s->flags, q, (const unsigned*)nullptr);
    else
        hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)s->flags, 0, 1, q, fault, s->timeout_ticks,
                           s->abort);
    const int rc = flag_launch_done(s, "sbx_shared_frame_begin launch");
    if (rc == SBX_OK) s->seq[rank] = q;                      // (a launch that failed has not started frame q: the counters stay in step)
    return rc;
}

int sbx_shared_frame_end(sbx_shared* s, int rank, void* stream) {
    if (!s) return SBX_ERR_ARG;
    if (rank < 0 || rank >= s->nranks || (rank == 0) != s->owner)
        return ctx_fail(s->ctx, SBX_ERR_ARG, "sbx_shared_frame_end: rank 0 drives the created object, the peers an opened one", hipSuccess);
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return ctx_fail(s->ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const unsigned q = s->seq[rank];
    if (q == 0) return ctx_fail(s->ctx, SBX_ERR_ARG, "sbx_shared_frame_end before sbx_shared_frame_begin", hipSuccess);
    if (rank == 0) {
        if (s->nranks > 1)
            hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)s->flags, 1, s->nranks - 1, q,
                               fault_word_device(s->device), s->timeout_ticks, s->abort);
    } else {
        (void)hipEventRecord(s->release, (hipStream_t)stream);            // system-scope release of the render kernel's stores
        hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, (hipStream_t)stream, s->flags + (size_t)rank * FLAG_STRIDE, q, (const unsigned*)s->abort);
    }
    return flag_launch_done(s, "sbx_shared_frame_end launch");
}

int sbx_model_landing(sbx_ctx* ctx, const void* src, void* dst, size_t bytes, int workgroups, float duration_us, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!src || !dst || (bytes & 15u) || (((uintptr_t)src | (uintptr_t)dst) & 15u) || workgroups < 1 || workgroups > 4096 || !(duration_us >= 0.f))
        return ctx_fail(ctx, SBX_ERR_ARG, "sbx_model_landing: bad arguments", hipSuccess);
    hipError_t e = hipSetDevice(ctx_device(ctx));
    if (e != hipSuccess) return ctx_fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    hipLaunchKernelGGL(k_model_landing, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       bytes / 16, (long long)(duration_us * 100.f));
    e = hipGetLastError();
    if (e != hipSuccess) return ctx_fail(ctx, SBX_ERR_HIP, "sbx_model_landing launch", e);
    return SBX_OK;
}

}  // extern "C"
