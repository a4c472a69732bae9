// capi_common.hip -- error reporting, device tables, memory helpers of the C ABI.
#include <atomic>
#include <cstdlib>
#include <map>

#include "grhip_internal.h"
#include <cstring>
#include "tables.inc"

namespace grhip {

static thread_local char t_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<int> g_mode{-1};

int default_mode()
{
    int m = g_mode.load();
    if (m >= 0) return m;
    const char *e = getenv("GRHIP_MODE");
    m = (e && (!strcmp(e, "generic") || !strcmp(e, "1"))) ? GRHIP_MODE_GENERIC
        : (e && (!strcmp(e, "fast_valu") || !strcmp(e, "2"))) ? GRHIP_MODE_FAST_VALU
        : (e && (!strcmp(e, "fast_reftaps") || !strcmp(e, "3"))) ? GRHIP_MODE_FAST_REFTAPS : GRHIP_MODE_FAST;
    g_mode.store(m);
    return m;
}

int HandleBase::init_device(int dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GRHIP_ENODEV, "no HIP device available (%s): libgrhip has no CPU fallback",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (dev < 0 || dev >= n) return fail(GRHIP_EINVAL, "device %d out of range [0,%d)", dev, n);
    device = dev;
    GRHIP_HIP(hipSetDevice(dev));
    GRHIP_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    return GRHIP_OK;
}

int HandleBase::bind() const
{
    GRHIP_HIP(hipSetDevice(device));
    return GRHIP_OK;
}

int HandleBase::pin_init(PinRing &r)
{
    if (r.buf[0]) return GRHIP_OK;
    for (int k = 0; k < 2; ++k) {
        GRHIP_HIP(hipHostMalloc(&r.buf[k], PIN_SLOT, hipHostMallocDefault));
        GRHIP_HIP(hipEventCreateWithFlags(&r.ev[k], hipEventDisableTiming));
        r.busy[k] = false;
    }
    r.next = 0;
    return GRHIP_OK;
}

void HandleBase::pin_release(PinRing &r)
{
    for (int k = 0; k < 2; ++k) {
        if (r.ev[k]) { if (r.busy[k]) (void)hipEventSynchronize(r.ev[k]); (void)hipEventDestroy(r.ev[k]); }
        if (r.buf[k]) (void)hipHostFree(r.buf[k]);
        r.buf[k] = nullptr; r.ev[k] = nullptr; r.busy[k] = false;
    }
}

int HandleBase::h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (!bytes) return GRHIP_OK;
    // mapped staging: the CPU writes the buffer the kernels are about to read (nothing of an earlier call is in flight:
    // every host-buffer entry ends with a synchronisation)
    for (StageBuf *b : {&stage_in, &stage_out})
        if (void *hp = b->host_of(dst_dev)) { memcpy(hp, src_host, bytes); return GRHIP_OK; }
    if (bytes > PIN_MAX) {
        GRHIP_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return GRHIP_OK;
    }
    int rc = pin_init(pin_up);
    if (rc) return rc;
    PinRing &r = pin_up;
    for (size_t off = 0; off < bytes; off += PIN_SLOT) {
        const size_t n = bytes - off < PIN_SLOT ? bytes - off : PIN_SLOT;
        const int k = r.next;
        if (r.busy[k]) { GRHIP_HIP(hipEventSynchronize(r.ev[k])); r.busy[k] = false; }    // its last DMA has read it
        memcpy(r.buf[k], (const char *)src_host + off, n);
        GRHIP_HIP(hipMemcpyAsync((char *)dst_dev + off, r.buf[k], n, hipMemcpyHostToDevice, st));
        GRHIP_HIP(hipEventRecord(r.ev[k], st));
        r.busy[k] = true;
        r.next = k ^ 1;
    }
    return GRHIP_OK;
}

int HandleBase::d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
    if (!bytes) return GRHIP_OK;
    for (StageBuf *b : {&stage_in, &stage_out})
        if (void *hp = b->host_of(src_dev)) {      // mapped staging: the kernels wrote host memory; wait for them, copy out
            GRHIP_HIP(hipStreamSynchronize(st));
            memcpy(dst_host, hp, bytes);
            return GRHIP_OK;
        }
    if (bytes > PIN_MAX) {
        GRHIP_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        return GRHIP_OK;
    }
    int rc = pin_init(pin_down);
    if (rc) return rc;
    PinRing &r = pin_down;
    size_t p_off = 0, p_n = 0;
    int p_k = -1;
    for (size_t off = 0; off < bytes; off += PIN_SLOT) {
        const size_t n = bytes - off < PIN_SLOT ? bytes - off : PIN_SLOT;
        const int k = r.next;       // free: its previous content was copied out before this slot came round again
        GRHIP_HIP(hipMemcpyAsync(r.buf[k], (const char *)src_dev + off, n, hipMemcpyDeviceToHost, st));
        GRHIP_HIP(hipEventRecord(r.ev[k], st));
        if (p_k >= 0) {
            GRHIP_HIP(hipEventSynchronize(r.ev[p_k]));
            memcpy((char *)dst_host + p_off, r.buf[p_k], p_n);
        }
        p_off = off; p_n = n; p_k = k;
        r.next = k ^ 1;
    }
    GRHIP_HIP(hipEventSynchronize(r.ev[p_k]));
    memcpy((char *)dst_host + p_off, r.buf[p_k], p_n);
    return GRHIP_OK;
}

void HandleBase::destroy_base()
{
    (void)hipSetDevice(device);
    pin_release(pin_up);
    pin_release(pin_down);
    stage_in.release();
    stage_out.release();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    own_stream = nullptr;
}

static std::mutex g_tab_mutex;
static std::map<int, DeviceTables> g_tabs;

int get_device_tables(int device, const DeviceTables **out)
{
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    auto it = g_tabs.find(device);
    if (it == g_tabs.end()) {
        DeviceTables t;
        GRHIP_HIP(hipSetDevice(device));
        GRHIP_HIP(hipMalloc((void **)&t.atan_tab, sizeof(grhip_atan_bits)));
        GRHIP_HIP(hipMemcpy(t.atan_tab, grhip_atan_bits, sizeof(grhip_atan_bits), hipMemcpyHostToDevice));
        GRHIP_HIP(hipMalloc((void **)&t.mmse_rev, sizeof(grhip_mmse_rev_bits)));
        GRHIP_HIP(hipMemcpy(t.mmse_rev, grhip_mmse_rev_bits, sizeof(grhip_mmse_rev_bits),
                            hipMemcpyHostToDevice));
        it = g_tabs.emplace(device, t).first;
    }
    *out = &it->second;
    return GRHIP_OK;
}

}  // namespace grhip

using namespace grhip;

extern "C" {

const char *grhip_strerror(int status)
{
    switch (status) {
    case GRHIP_OK: return "ok";
    case GRHIP_EINVAL: return "invalid argument";
    case GRHIP_ERANGE: return "out of range";
    case GRHIP_ERUNTIME: return "HIP runtime error";
    case GRHIP_ENOMEM: return "out of memory";
    case GRHIP_ENODEV: return "no usable HIP device";
    default: return status > 0 ? "ok (item count)" : "unknown error";
    }
}

const char *grhip_last_error(void) { return t_err; }

const char *grhip_version(void) { return "grhip 0.1 (gfx950)"; }

int grhip_device_count(int *count)
{
    if (!count) return fail(GRHIP_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(GRHIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return GRHIP_OK;
}

int grhip_device_synchronize(int device)
{
    GRHIP_HIP(hipSetDevice(device));
    GRHIP_HIP(hipDeviceSynchronize());
    return GRHIP_OK;
}

int grhip_set_default_mode(int mode)
{
    if (!mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode %d", mode);
    g_mode.store(mode);
    return GRHIP_OK;
}

int grhip_get_default_mode(void) { return default_mode(); }

int grhip_malloc(void **d_ptr, size_t bytes, int device)
{
    if (!d_ptr) return fail(GRHIP_EINVAL, "d_ptr is NULL");
    GRHIP_HIP(hipSetDevice(device));
    GRHIP_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return GRHIP_OK;
}

int grhip_free(void *d_ptr)
{
    if (d_ptr) GRHIP_HIP(hipFree(d_ptr));
    return GRHIP_OK;
}

int grhip_memcpy_h2d(void *d_dst, const void *src, size_t bytes)
{
    if (bytes) GRHIP_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return GRHIP_OK;
}

int grhip_memcpy_d2h(void *dst, const void *d_src, size_t bytes)
{
    if (bytes) GRHIP_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return GRHIP_OK;
}

int grhip_stream_synchronize(void *stream)
{
    GRHIP_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GRHIP_OK;
}

}  // extern "C"
