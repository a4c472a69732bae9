// grhip_internal.h -- shared plumbing of libgrhip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/grhip.h"

namespace grhip {

// ---- thread-local error detail -------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define GRHIP_HIP(call)                                                                     \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return ::grhip::fail(e__ == hipErrorOutOfMemory ? GRHIP_ENOMEM : GRHIP_ERUNTIME, \
                                 "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                                 __FILE__, __LINE__);                                       \
    } while (0)

// ---- device buffer that only ever grows -----------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return GRHIP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        GRHIP_HIP(hipMalloc(&p, want));
        cap = want;
        return GRHIP_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Staging buffer of the host-buffer entry points.  Scheduler-sized calls (<= MAPPED_MAX bytes) get host memory that is
// pinned AND mapped into the device's address space: the caller's items are copied into it by the CPU, the kernels read
// and write it in place over the host link, and no copy engine is involved (a 32 KB call spent more time submitting its
// two DMA transfers than moving them: 45 -> 26 us per 1024-output call of the fused block, 61 -> 34 at 8192;
// tools/bench_small_calls.py).  Once a call needs more than MAPPED_MAX, the buffer becomes device memory for good.
struct StageBuf : DevBuf {
#ifndef GRHIP_MAPPED_MAX
#define GRHIP_MAPPED_MAX (2 * 1024 * 1024)
#endif
    static constexpr size_t MAPPED_MAX = GRHIP_MAPPED_MAX;
    void *host = nullptr;       // host address of a mapped buffer (p is its device address); null: device memory
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return GRHIP_OK;
        if (bytes <= MAPPED_MAX && (cap == 0 || host)) {
            size_t want = 64 * 1024;
            while (want < bytes) want *= 2;
            release();
            GRHIP_HIP(hipHostMalloc(&host, want, hipHostMallocMapped));
            if (hipHostGetDevicePointer(&p, host, 0) != hipSuccess) { (void)hipHostFree(host); host = nullptr; p = nullptr; }
            else { cap = want; return GRHIP_OK; }
        }
        release();
        return DevBuf::reserve(bytes);
    }
    void release()
    {
        if (host) { (void)hipHostFree(host); host = nullptr; p = nullptr; cap = 0; }
        else DevBuf::release();
    }
    // host address of a device address inside a mapped buffer, or null
    void *host_of(const void *dev) const
    {
        if (!host || (const char *)dev < (const char *)p || (const char *)dev >= (const char *)p + cap) return nullptr;
        return (char *)host + ((const char *)dev - (const char *)p);
    }
};

// the two counters of the tiled kernel's tile queue: zeroed once, re-armed by every launch
struct SchedBuf {
    DevBuf b;
    unsigned *get()
    {
        if (!b.p) {
            if (b.reserve(2 * sizeof(unsigned)) != GRHIP_OK) return nullptr;
            if (hipMemset(b.p, 0, 2 * sizeof(unsigned)) != hipSuccess) { b.release(); return nullptr; }
        }
        return b.as<unsigned>();
    }
    void release() { b.release(); }
};

// ---- base of every block handle -------------------------------------------
#define GRHIP_H2D(h, dst, src, bytes, st) do { int rc__ = (h)->h2d((dst), (src), (bytes), (st)); if (rc__) return rc__; } while (0)
#define GRHIP_D2H(h, dst, src, bytes, st) do { int rc__ = (h)->d2h((dst), (src), (bytes), (st)); if (rc__) return rc__; } while (0)

struct HandleBase {
    int device = 0;
    hipStream_t own_stream = nullptr;
    std::mutex setter_mutex;   // setters vs. the work thread
    StageBuf stage_in, stage_out;
    // Pinned staging of the host-buffer entry points (SURVEY 8b "Ownership": the handle owns it, the caller owns its I/O
    // buffers).  Scheduler-sized transfers (<= PIN_MAX) go caller buffer -> pinned slot -> DMA and back: no page pinning or
    // runtime staging per call; two slots per direction, so a call split in chunks copies one while the other is in flight.
    // Larger transfers are left to the runtime: measured (tools/bench_small_calls.py) its own staging wins from a few hundred KB.
    static constexpr size_t PIN_SLOT = 32 * 1024, PIN_MAX = 64 * 1024;
    struct PinRing {
        void *buf[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        int next = 0;
    } pin_up, pin_down;
    int pin_init(PinRing &r);
    void pin_release(PinRing &r);
    int h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
    int d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st);     // complete on return (<= PIN_MAX) or queued

    int init_device(int dev);
    void destroy_base();
    hipStream_t pick(void *stream) const { return stream ? (hipStream_t)stream : own_stream; }
    int bind() const;          // hipSetDevice for the calling thread
};

int default_mode();
inline bool mode_valid(int m) { return m == GRHIP_MODE_FAST || m == GRHIP_MODE_GENERIC || m == GRHIP_MODE_FAST_VALU || m == GRHIP_MODE_FAST_REFTAPS; }
inline bool mode_fast(int m) { return m != GRHIP_MODE_GENERIC; }          // FAST, FAST_VALU or FAST_REFTAPS
inline bool mode_matrix(int m) { return m == GRHIP_MODE_FAST || m == GRHIP_MODE_FAST_REFTAPS; }   // matrix cores allowed

// shared read-only device tables (per device, created on first use)
struct DeviceTables {
    float *atan_tab = nullptr;       // 257 floats   (gr_fast_atan2f table)
    float *mmse_rev = nullptr;       // [8][129] floats, tap-major, reversed taps
};
int get_device_tables(int device, const DeviceTables **out);

}  // namespace grhip
