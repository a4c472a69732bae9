// capi_digital.hip -- C ABI for digital_clock_recovery_mm_ff,
// digital_binary_slicer_fb, digital_correlate_access_code_bb.
#include <cmath>

#include "digital_kernels.h"
#include "grhip_internal.h"

using namespace grhip;

namespace grhip {

// set_omega (gr-digital/include/digital_clock_recovery_mm_ff.h:70-75), mixed
// float/double arithmetic kept as written there.
void mm_set_omega(MMState &s, float omega)
{
    s.omega = omega;
    s.min_omega = omega * (1.0 - s.omega_relative_limit);
    s.max_omega = omega * (1.0 + s.omega_relative_limit);
    s.omega_mid = 0.5 * (s.min_omega + s.max_omega);
}

int mm_init_state(MMState &s, float omega, float gain_omega, float mu, float gain_mu, float rel)
{
    if (omega < 1) return fail(GRHIP_ERANGE, "clock rate must be > 0");               // .cc:58-59
    if (gain_mu < 0 || gain_omega < 0) return fail(GRHIP_ERANGE, "Gains must be non-negative");   // .cc:60-61
    memset(&s, 0, sizeof(s));
    s.mu = mu; s.gain_omega = gain_omega; s.gain_mu = gain_mu;
    s.last_sample = 0; s.omega_relative_limit = rel;
    mm_set_omega(s, omega);
    return GRHIP_OK;
}

// set_access_code (gr-digital/lib/digital_correlate_access_code_bb.cc:64-85)
int corr_set_code(CorrParams &p, unsigned long long &flag_bit, const char *code, size_t len)
{
    if (len > 64) return fail(GRHIP_ERANGE, "access_code is > 64 bits");
    if (len == 0) { p.mask = 0; flag_bit = 0; }
    else {
        p.mask = ((~0ULL) >> (64 - len)) << (64 - len);
        flag_bit = 1ULL << (64 - len);
    }
    p.access_code = 0;
    for (unsigned i = 0; i < 64; i++) {
        p.access_code <<= 1;
        if (i < len) p.access_code |= code[i] & 1;
    }
    p.len = (unsigned)len;
    return GRHIP_OK;
}

}  // namespace grhip

struct grhip_clock_recovery_mm_ff : HandleBase {
    const DeviceTables *tabs = nullptr;
    DevBuf d_state, d_counts;
    int read_state(MMState &s)
    {
        GRHIP_HIP(hipMemcpy(&s, d_state.p, sizeof(s), hipMemcpyDeviceToHost));
        return GRHIP_OK;
    }
    int write_state(const MMState &s)
    {
        GRHIP_HIP(hipMemcpy(d_state.p, &s, sizeof(s), hipMemcpyHostToDevice));
        return GRHIP_OK;
    }
};

struct grhip_binary_slicer_fb : HandleBase {};

struct grhip_pager_slicer_fb : HandleBase {
    float alpha = 0, beta = 1;
    DevBuf d_avg;
};

struct grhip_unpack_k_bits_bb : HandleBase {
    unsigned k = 1;
};

struct grhip_stream_adapter : HandleBase {
    bool split = true;
    size_t item_size = 1, nstreams = 1;
};

struct grhip_correlate_access_code_bb : HandleBase {
    CorrParams p;
    unsigned long long flag_bit = 0;
    DevBuf d_state;
};

extern "C" {

// ---- clock_recovery_mm_ff ---------------------------------------------------
int grhip_clock_recovery_mm_ff_create(grhip_clock_recovery_mm_ff **h, float omega, float gain_omega, float mu,
                                      float gain_mu, float omega_relative_limit, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    MMState s;
    int rc = mm_init_state(s, omega, gain_omega, mu, gain_mu, omega_relative_limit);
    if (rc) return rc;
    auto *m = new (std::nothrow) grhip_clock_recovery_mm_ff();
    if (!m) return fail(GRHIP_ENOMEM, "alloc");
    rc = m->init_device(device);
    if (!rc) rc = get_device_tables(device, &m->tabs);
    if (!rc) rc = m->d_state.reserve(sizeof(MMState));
    if (!rc) rc = m->d_counts.reserve(2 * sizeof(int));
    if (!rc) rc = m->write_state(s);
    if (rc) { m->d_state.release(); m->d_counts.release(); m->destroy_base(); delete m; return rc; }
    *h = m;
    return GRHIP_OK;
}

void grhip_clock_recovery_mm_ff_destroy(grhip_clock_recovery_mm_ff *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_state.release(); h->d_counts.release();
    h->destroy_base();
    delete h;
}

int grhip_clock_recovery_mm_ff_forecast(const grhip_clock_recovery_mm_ff *h, int noutput_items)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    MMState s;
    int rc = const_cast<grhip_clock_recovery_mm_ff *>(h)->bind();
    if (!rc) rc = const_cast<grhip_clock_recovery_mm_ff *>(h)->read_state(s);
    if (rc) return rc;
    return (int)ceil((noutput_items * s.omega) + 8);     // .cc:80-87, d_interp->ntaps() == 8
}

int grhip_clock_recovery_mm_ff_general_work_device(grhip_clock_recovery_mm_ff *h, int noutput_items,
                                                   int ninput_items, const float *d_in, float *d_out,
                                                   int *d_counts, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || ninput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    int rc = h->bind();
    if (rc) return rc;
    return launch_mm(h->d_state.as<MMState>(), 1, noutput_items, ninput_items, d_in, 0, d_out, 0,
                     d_counts ? d_counts : h->d_counts.as<int>(), h->tabs->mmse_rev, h->pick(stream));
}

int grhip_clock_recovery_mm_ff_general_work(grhip_clock_recovery_mm_ff *h, int noutput_items, int ninput_items,
                                            const float *in, float *out, int *consumed)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || ninput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    int rc = h->bind();
    if (rc) return rc;
    if ((rc = h->stage_in.reserve((size_t)(ninput_items > 0 ? ninput_items : 1) * 4))) return rc;
    if ((rc = h->stage_out.reserve((size_t)(noutput_items > 0 ? noutput_items : 1) * 4))) return rc;
    hipStream_t st = h->own_stream;
    if (ninput_items) GRHIP_H2D(h, h->stage_in.p, in, (size_t)ninput_items * 4, st);
    rc = launch_mm(h->d_state.as<MMState>(), 1, noutput_items, ninput_items, h->stage_in.as<float>(), 0,
                   h->stage_out.as<float>(), 0, h->d_counts.as<int>(), h->tabs->mmse_rev, st);
    if (rc) return rc;
    int counts[2] = {0, 0};
    GRHIP_HIP(hipMemcpyAsync(counts, h->d_counts.p, sizeof(counts), hipMemcpyDeviceToHost, st));
    GRHIP_HIP(hipStreamSynchronize(st));
    if (counts[0] > 0)
        GRHIP_HIP(hipMemcpy(out, h->stage_out.p, (size_t)counts[0] * 4, hipMemcpyDeviceToHost));
    if (consumed) *consumed = counts[1];
    return counts[0];
}

#define MM_GETTER(name, field)                                              \
    float grhip_clock_recovery_mm_ff_##name(grhip_clock_recovery_mm_ff *h) \
    {                                                                       \
        MMState s;                                                          \
        if (!h || h->bind() || h->read_state(s)) return NAN;                \
        return s.field;                                                     \
    }
MM_GETTER(mu, mu)
MM_GETTER(omega, omega)
MM_GETTER(gain_mu, gain_mu)
MM_GETTER(gain_omega, gain_omega)

#define MM_SETTER(name, stmt)                                                           \
    int grhip_clock_recovery_mm_ff_set_##name(grhip_clock_recovery_mm_ff *h, float v)  \
    {                                                                                   \
        if (!h) return fail(GRHIP_EINVAL, "null handle");                               \
        std::lock_guard<std::mutex> lk(h->setter_mutex);                                \
        MMState s;                                                                      \
        int rc = h->bind();                                                             \
        if (!rc) rc = h->read_state(s);                                                 \
        if (rc) return rc;                                                              \
        stmt;                                                                           \
        return h->write_state(s);                                                       \
    }
MM_SETTER(gain_mu, s.gain_mu = v)
MM_SETTER(gain_omega, s.gain_omega = v)
MM_SETTER(mu, s.mu = v)
MM_SETTER(omega, mm_set_omega(s, v))

// ---- binary_slicer_fb ----------------------------------------------------------
int grhip_binary_slicer_fb_create(grhip_binary_slicer_fb **h, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    auto *b = new (std::nothrow) grhip_binary_slicer_fb();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    int rc = b->init_device(device);
    if (rc) { b->destroy_base(); delete b; return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_binary_slicer_fb_destroy(grhip_binary_slicer_fb *h)
{
    if (!h) return;
    h->destroy_base();
    delete h;
}

int grhip_binary_slicer_fb_work_device(grhip_binary_slicer_fb *h, int noutput_items, const float *d_in,
                                       unsigned char *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = launch_binary_slicer(d_in, d_out, noutput_items, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_binary_slicer_fb_work(grhip_binary_slicer_fb *h, int noutput_items, const float *in,
                                unsigned char *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t n = (size_t)noutput_items;
    if ((rc = h->stage_in.reserve(n * 4))) return rc;
    if ((rc = h->stage_out.reserve(n))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n * 4, st);
    if ((rc = launch_binary_slicer(h->stage_in.as<float>(), h->stage_out.as<unsigned char>(), (long long)n, st)))
        return rc;
    GRHIP_D2H(h, out, h->stage_out.p, n, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

// ---- pager_slicer_fb -------------------------------------------------------------
int grhip_pager_slicer_fb_create(grhip_pager_slicer_fb **h, float alpha, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    auto *b = new (std::nothrow) grhip_pager_slicer_fb();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->alpha = alpha;
    b->beta = (float)(1.0 - (double)alpha);          // pager_slicer_fb.cc:40
    int rc = b->init_device(device);
    if (!rc) rc = b->d_avg.reserve(sizeof(float));
    if (!rc) { hipError_t e = hipMemset(b->d_avg.p, 0, sizeof(float)); if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "memset"); }
    if (rc) { b->d_avg.release(); b->destroy_base(); delete b; return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_pager_slicer_fb_destroy(grhip_pager_slicer_fb *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_avg.release();
    h->destroy_base();
    delete h;
}

int grhip_pager_slicer_fb_work_device(grhip_pager_slicer_fb *h, int noutput_items, const float *d_in,
                                      unsigned char *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = launch_pager_slicer(h->d_avg.as<float>(), 1, h->alpha, h->beta, d_in, 0, d_out, 0, noutput_items,
                             h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_pager_slicer_fb_work(grhip_pager_slicer_fb *h, int noutput_items, const float *in, unsigned char *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t n = (size_t)noutput_items;
    if ((rc = h->stage_in.reserve(n * 4))) return rc;
    if ((rc = h->stage_out.reserve(n))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n * 4, st);
    if ((rc = launch_pager_slicer(h->d_avg.as<float>(), 1, h->alpha, h->beta, h->stage_in.as<float>(), 0,
                                  h->stage_out.as<unsigned char>(), 0, (long long)n, st)))
        return rc;
    GRHIP_D2H(h, out, h->stage_out.p, n, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

int grhip_pager_slicer_fb_dc_offset(grhip_pager_slicer_fb *h, float *dc_offset)
{
    if (!h || !dc_offset) return fail(GRHIP_EINVAL, "null argument");
    int rc = h->bind();
    if (rc) return rc;
    GRHIP_HIP(hipDeviceSynchronize());
    GRHIP_HIP(hipMemcpy(dc_offset, h->d_avg.p, sizeof(float), hipMemcpyDeviceToHost));
    return GRHIP_OK;
}

// ---- unpack_k_bits_bb --------------------------------------------------------------
int grhip_unpack_k_bits_bb_create(grhip_unpack_k_bits_bb **h, unsigned k, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (k == 0) return fail(GRHIP_ERANGE, "interpolation must be > 0");          // .cc:45-46
    if (k > 32) return fail(GRHIP_EINVAL, "k > 32: the reference shifts an unsigned int");
    auto *b = new (std::nothrow) grhip_unpack_k_bits_bb();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->k = k;
    int rc = b->init_device(device);
    if (rc) { b->destroy_base(); delete b; return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_unpack_k_bits_bb_destroy(grhip_unpack_k_bits_bb *h)
{
    if (!h) return;
    h->destroy_base();
    delete h;
}

int grhip_unpack_k_bits_bb_work_device(grhip_unpack_k_bits_bb *h, int noutput_items, const unsigned char *d_in,
                                       unsigned char *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || (unsigned)noutput_items % h->k) return fail(GRHIP_EINVAL, "noutput_items must be a multiple of k");
    int rc = h->bind();
    if (rc) return rc;
    rc = launch_unpack_k_bits(h->k, d_in, d_out, noutput_items, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_unpack_k_bits_bb_work(grhip_unpack_k_bits_bb *h, int noutput_items, const unsigned char *in,
                                unsigned char *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || (unsigned)noutput_items % h->k) return fail(GRHIP_EINVAL, "noutput_items must be a multiple of k");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t n = (size_t)noutput_items, ni = n / h->k;
    if ((rc = h->stage_in.reserve(ni))) return rc;
    if ((rc = h->stage_out.reserve(n))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, ni, st);
    if ((rc = launch_unpack_k_bits(h->k, h->stage_in.as<unsigned char>(), h->stage_out.as<unsigned char>(), (long long)n, st)))
        return rc;
    GRHIP_D2H(h, out, h->stage_out.p, n, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

// ---- stream_to_streams / streams_to_stream ------------------------------------------
int grhip_stream_adapter_create(grhip_stream_adapter **h, int split, size_t item_size, size_t nstreams, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (item_size == 0 || nstreams == 0 || nstreams > 65536) return fail(GRHIP_EINVAL, "bad item_size / nstreams");
    auto *b = new (std::nothrow) grhip_stream_adapter();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->split = split != 0; b->item_size = item_size; b->nstreams = nstreams;
    int rc = b->init_device(device);
    if (rc) { b->destroy_base(); delete b; return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_stream_adapter_destroy(grhip_stream_adapter *h)
{
    if (!h) return;
    h->destroy_base();
    delete h;
}

int grhip_stream_adapter_work_device(grhip_stream_adapter *h, int n, void *d_single, void *d_streams,
                                     size_t stream_stride_items, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (n < 0) return fail(GRHIP_EINVAL, "negative item count");
    if (stream_stride_items < (size_t)n) return fail(GRHIP_EINVAL, "stream stride shorter than the streams");
    int rc = h->bind();
    if (rc) return rc;
    rc = launch_streams(h->split, d_single, d_streams, (long long)stream_stride_items, (int)h->nstreams, h->item_size, n,
                        h->pick(stream));
    return rc ? rc : n;
}

int grhip_stream_adapter_work(grhip_stream_adapter *h, int n, void *single, void *const *streams)
{
    if (!h || !streams) return fail(GRHIP_EINVAL, "null argument");
    if (n < 0) return fail(GRHIP_EINVAL, "negative item count");
    if (n == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    const size_t per = (size_t)n * h->item_size, tot = per * h->nstreams;
    if ((rc = h->stage_in.reserve(tot))) return rc;
    if ((rc = h->stage_out.reserve(tot))) return rc;
    hipStream_t st = h->own_stream;
    // stage_in holds the single stream, stage_out the nstreams streams back to back
    if (h->split) {
        GRHIP_H2D(h, h->stage_in.p, single, tot, st);
    } else {
        for (size_t j = 0; j < h->nstreams; ++j)
            GRHIP_H2D(h, (char *)h->stage_out.p + j * per, streams[j], per, st);
    }
    if ((rc = launch_streams(h->split, h->stage_in.p, h->stage_out.p, n, (int)h->nstreams, h->item_size, n, st))) return rc;
    if (h->split) {
        for (size_t j = 0; j < h->nstreams; ++j)
            GRHIP_D2H(h, streams[j], (char *)h->stage_out.p + j * per, per, st);
    } else {
        GRHIP_D2H(h, single, h->stage_in.p, tot, st);
    }
    GRHIP_HIP(hipStreamSynchronize(st));
    return n;
}

// ---- correlate_access_code_bb ----------------------------------------------------
int grhip_correlate_access_code_bb_create(grhip_correlate_access_code_bb **h, const char *access_code,
                                          size_t len, int threshold, int device)
{
    if (!h || (len && !access_code)) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    CorrParams p;
    memset(&p, 0, sizeof(p));
    unsigned long long fb = 0;
    int rc = corr_set_code(p, fb, access_code, len);
    if (rc) return rc;
    p.threshold = (unsigned)threshold;
    auto *c = new (std::nothrow) grhip_correlate_access_code_bb();
    if (!c) return fail(GRHIP_ENOMEM, "alloc");
    c->p = p; c->flag_bit = fb;
    rc = c->init_device(device);
    if (!rc) rc = c->d_state.reserve(sizeof(CorrState));
    if (!rc) { hipError_t e = hipMemset(c->d_state.p, 0, sizeof(CorrState)); if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "memset"); }
    if (rc) { c->d_state.release(); c->destroy_base(); delete c; return rc; }
    *h = c;
    return GRHIP_OK;
}

void grhip_correlate_access_code_bb_destroy(grhip_correlate_access_code_bb *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_state.release();
    h->destroy_base();
    delete h;
}

int grhip_correlate_access_code_bb_set_access_code(grhip_correlate_access_code_bb *h, const char *access_code,
                                                   size_t len)
{
    if (!h || (len && !access_code)) return fail(GRHIP_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    CorrParams p = h->p;
    unsigned long long fb = 0;
    int rc = corr_set_code(p, fb, access_code, len);
    if (rc) return rc;               // reference: returns false, keeps the old code
    h->p = p; h->flag_bit = fb;
    return GRHIP_OK;
}

int grhip_correlate_access_code_bb_work_device(grhip_correlate_access_code_bb *h, int noutput_items,
                                               const unsigned char *d_in, unsigned char *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    CorrParams p;
    { std::lock_guard<std::mutex> lk(h->setter_mutex); p = h->p; }
    rc = launch_correlate(p, h->d_state.as<CorrState>(), 1, d_in, nullptr, 0, d_out, 0, noutput_items, nullptr,
                          0, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_correlate_access_code_bb_work(grhip_correlate_access_code_bb *h, int noutput_items,
                                        const unsigned char *in, unsigned char *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t n = (size_t)noutput_items;
    if ((rc = h->stage_in.reserve(n + 8))) return rc;
    if ((rc = h->stage_out.reserve(n + 8))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n, st);
    rc = grhip_correlate_access_code_bb_work_device(h, noutput_items, h->stage_in.as<unsigned char>(),
                                                    h->stage_out.as<unsigned char>(), st);
    if (rc < 0) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, n, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

}  // extern "C"
