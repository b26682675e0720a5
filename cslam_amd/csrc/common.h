// Internal helpers shared by the .hip translation units of libcslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "../../include/cslam_hip.h"
#include "../../include/cslam_hip_experimental.h"

#define CSLAM_API extern "C" __attribute__((visibility("default")))

void cslam_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            cslam_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                   \
            return CSLAM_E_HIP;                                                    \
        }                                                                          \
    } while (0)

#define ARG_CHECK(cond, msg)                  \
    do {                                      \
        if (!(cond)) {                        \
            cslam_set_error("invalid argument: %s", msg); \
            return CSLAM_E_INVALID;           \
        }                                     \
    } while (0)

// ---- device selection ------------------------------------------------------------------
// Every entry point runs on the device that owns its data and leaves the caller's current device as it
// found it (a process may hold banks on one GPU and run its extractor on another).
struct DeviceGuard {
    int prev;
    bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(true) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (dev >= 0 && dev != cur) {
            ok = (hipSetDevice(dev) == hipSuccess);
            if (ok) prev = cur;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

int cslam_visible_devices();   // cached hipGetDeviceCount (bank.hip)

// device owning a device pointer; -1 = unknown (null / host pointer) or only one device visible (nothing to choose)
static inline int device_of_ptr(const void *p) {
    if (!p || cslam_visible_devices() <= 1) return -1;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}

#define BANK_DEVICE(b)                                              \
    DeviceGuard _dev_guard((b)->device);                            \
    if (!_dev_guard.ok) { cslam_set_error("hipSetDevice(%d) failed", (b)->device); return CSLAM_E_HIP; }
#define PTR_DEVICE(p)                                               \
    DeviceGuard _dev_guard(device_of_ptr(p));                       \
    if (!_dev_guard.ok) { cslam_set_error("hipSetDevice failed for the device owning %s", #p); return CSLAM_E_HIP; }

// ---- once-per-device set-up (thread-safe) ---------------------------------------------
// hipFuncSetAttribute is per device, and entry points may be called from several host threads: a plain
// `static bool done` is a data race and a per-process latch.  Bit d of the mask = device d configured; two threads
// racing through the same set-up both perform it (idempotent) before either sets the bit.
//     static DeviceOnce once;  int dev;
//     if (once.todo(&dev)) { HIP_TRY(hipFuncSetAttribute(...)); once.done(dev); }
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool todo(int *dev) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        *dev = d;
        return d < 0 || d >= 64 || !((mask.load(std::memory_order_acquire) >> d) & 1ull);
    }
    void done(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};
int cslam_cu_count();          // compute units of the CURRENT device, cached per device (bank.hip); 0 on error

// Scratch of an entry point that needs device memory between its own kernels (split-K partial sums, operand pairs): one
// grow-only buffer per (device, stream).  Launches on ONE stream run in order, so they can share a buffer; two streams -- two
// extraction lanes, two host threads -- must not.  A superseded buffer stays allocated: a pointer captured in a hipGraph (the
// online path replays one) has to remain valid when a later, larger batch needs more room.  `floor_bytes` = smallest allocation.
// Returns nullptr with *status = CSLAM_E_NOMEM when the allocation fails, CSLAM_E_HIP when the stream is being captured into a graph
// and has no buffer of that size yet (the message says which); SCRATCH_GET turns that into the entry point's return value.
struct StreamScratch {
    struct Entry { int dev; void *stream; char *ptr; size_t bytes; };
    std::mutex mu;
    std::vector<Entry> entries;
    char *get(int dev, void *stream, size_t need, size_t floor_bytes, int *status) {
        *status = CSLAM_OK;
        std::lock_guard<std::mutex> lock(mu);
        Entry *e = nullptr;
        for (Entry &x : entries)
            if (x.dev == dev && x.stream == stream) e = &x;
        if (e && e->bytes >= need) return e->ptr;
        size_t want = e && need < 2 * e->bytes ? 2 * e->bytes : need;
        if (want < floor_bytes) want = floor_bytes;
        // a capturing stream cannot allocate (and the attempt would invalidate the capture): the caller has to have run the entry
        // point once, at this size, on the stream it captures on -- cslam_amd.vpr.heads.OnlineGraph warms up on its capture stream
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
            cslam_set_error("scratch of %zu bytes needed on a stream that is being captured: run the call once on that stream first", need);
            *status = CSLAM_E_HIP;
            return nullptr;
        }
        char *fresh = nullptr;
        if (hipMalloc((void **)&fresh, want) != hipSuccess) {
            (void)hipGetLastError();
            cslam_set_error("out of device memory: %zu bytes of scratch", want);
            *status = CSLAM_E_NOMEM;
            return nullptr;
        }
        if (e) { e->ptr = fresh; e->bytes = want; }                // the previous buffer is left alive on purpose (see above)
        else entries.push_back(Entry{dev, stream, fresh, want});
        return fresh;
    }
};

#define SCRATCH_GET(var, type, scratch, dev, stream, need, floor_bytes)                         \
    type var;                                                                                  \
    { int _st; var = (type)(scratch).get((dev), (stream), (need), (floor_bytes), &_st); if (!var) return _st; }

static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div64(int64_t x, int64_t m) { return (x + m - 1) / m; }

// ---- ranking order shared by every kernel -------------------------------------------
// rank key of a float64 similarity: NaN ranks first (reference: argsort()[::-1] puts NaN
// first, cslam/nns_matching.py:60), an empty slot is -inf with index -1.
__device__ __forceinline__ double rank_key(double sim) { return (sim != sim) ? INFINITY : sim; }
// "a ranks before b": larger key, ties -> larger row index
__device__ __forceinline__ bool ranks_before(double ka, int ia, double kb, int ib) {
    return ka > kb || (ka == kb && ia > ib);
}

// reference similarity from the three dots (scipy correlation(centered=False) + clip)
__device__ __forceinline__ double sim_from_dots(double uv, double uu, double vv) {
    double dist = 1.0 - uv / sqrt(uu * vv);
    dist = dist < 0.0 ? 0.0 : dist;   // NaN compares false -> propagates
    dist = dist > 2.0 ? 2.0 : dist;
    return 1.0 - dist;
}

// butterfly all-reduce of a double over the 64 lanes; every lane ends with identical bits
__device__ __forceinline__ double wave_allreduce_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// A sorted list of up to 64 (key, idx) entries held one-per-lane across a wave
// (lane 0 = best).  Empty entries: key = -inf, idx = -1.
struct WaveList {
    double key;
    int idx;
    __device__ __forceinline__ void init() { key = -INFINITY; idx = -1; }
    // wave-uniform candidate (ck, ci); returns nothing, list stays sorted; entry 63 falls out
    __device__ __forceinline__ void insert(double ck, int ci, int lane) {
        double pk = __shfl_up(key, 1, 64);
        int pi = __shfl_up(idx, 1, 64);
        bool before_me = ranks_before(ck, ci, key, idx);
        bool before_prev = (lane > 0) && ranks_before(ck, ci, pk, pi);
        if (before_prev) { key = pk; idx = pi; }
        else if (before_me) { key = ck; idx = ci; }
    }
    __device__ __forceinline__ double key_at(int pos) const { return __shfl(key, pos, 64); }
    __device__ __forceinline__ int idx_at(int pos) const { return __shfl(idx, pos, 64); }
};
