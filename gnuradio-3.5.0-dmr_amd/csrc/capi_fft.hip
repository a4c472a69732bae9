// capi_fft.hip -- C ABI for gr_fft_vcc and gr_pfb_channelizer_ccf.
#include <cmath>

#include "digital_kernels.h"
#include "fft_kernels.h"
#include "grhip_internal.h"

using namespace grhip;

struct grhip_fft_vcc : HandleBase {
    int N = 0, forward = 1, shift = 0;
    std::vector<float> window;     // empty or N
    DevBuf d_window;
    FftPlan plan;              // any fft_size > 0 (fft_any.hip)
    bool has_window = false;
    int upload_window()
    {
        has_window = !window.empty();
        if (!has_window) return GRHIP_OK;
        int rc = d_window.reserve(window.size() * 4);
        if (rc) return rc;
        GRHIP_HIP(hipMemcpy(d_window.p, window.data(), window.size() * 4, hipMemcpyHostToDevice));
        return GRHIP_OK;
    }
};

struct grhip_pfb_channelizer_ccf : HandleBase {
    unsigned M = 0;
    float oversample_rate = 1.f;
    int rate_ratio = 1, output_multiple = 1;
    unsigned taps_per_filter = 0;
    bool updated = false;
    std::vector<float> ftaps;      // [M][tpf] reversed
    std::vector<int> idxlut;
    DevBuf d_ftaps, d_idxlut, d_dft;
    DevBuf d_hier_in, d_hier_vec;     // hier entry, shapes without a fused kernel: de-interleaved streams, output vectors

    // set_taps (filter/gr_pfb_channelizer_ccf.cc:104-139)
    int set_taps(const float *taps, size_t ntaps)
    {
        taps_per_filter = (unsigned)ceil((double)ntaps / (double)M);
        size_t tot = (size_t)M * taps_per_filter;
        std::vector<float> tmp(tot, 0.f);
        for (size_t i = 0; i < ntaps; ++i) tmp[i] = taps[i];
        ftaps.assign(tot ? tot : 1, 0.f);
        for (unsigned i = 0; i < M; i++)
            for (unsigned j = 0; j < taps_per_filter; j++)
                ftaps[(size_t)i * taps_per_filter + (taps_per_filter - 1 - j)] = tmp[i + (size_t)j * M];
        int rc = d_ftaps.reserve(ftaps.size() * 4);
        if (rc) return rc;
        GRHIP_HIP(hipMemcpy(d_ftaps.p, ftaps.data(), ftaps.size() * 4, hipMemcpyHostToDevice));
        updated = true;
        return GRHIP_OK;
    }
};

extern "C" {

int grhip_fft_vcc_create(grhip_fft_vcc **h, int fft_size, int forward, const float *window, size_t window_len,
                         int shift, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (fft_size <= 0) return fail(GRHIP_ERANGE, "gri_fftw: invalid fft_size");      // gri_fft.cc:104-105
    if (!FftPlan::size_ok(fft_size))
        return fail(GRHIP_EINVAL, "fft_size %d: more than 2^26 points (2^25 when not a power of two)", fft_size);
    if (window_len && !window) return fail(GRHIP_EINVAL, "window is NULL");
    auto *f = new (std::nothrow) grhip_fft_vcc();
    if (!f) return fail(GRHIP_ENOMEM, "alloc");
    f->N = fft_size; f->forward = forward ? 1 : 0; f->shift = shift ? 1 : 0;
    // set_window accepts only size 0 or fft_size (gr_fft_vcc.cc:55-64); the ctor ignores the result
    if (window_len == (size_t)fft_size) f->window.assign(window, window + window_len);
    int rc = f->init_device(device);
    if (!rc) rc = f->plan.build(fft_size, f->forward);
    if (!rc) rc = f->upload_window();
    if (rc) { f->d_window.release(); f->plan.release(); f->destroy_base(); delete f; return rc; }
    *h = f;
    return GRHIP_OK;
}

void grhip_fft_vcc_destroy(grhip_fft_vcc *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_window.release(); h->plan.release();
    h->destroy_base();
    delete h;
}

int grhip_fft_vcc_set_window(grhip_fft_vcc *h, const float *window, size_t window_len)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (!(window_len == 0 || window_len == (size_t)h->N)) return 0;     // false
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    int rc = h->bind();
    if (rc) return rc;
    if (window_len) h->window.assign(window, window + window_len); else h->window.clear();
    rc = h->upload_window();
    return rc ? rc : 1;
}

int grhip_fft_vcc_work_device(grhip_fft_vcc *h, int noutput_items, const void *d_in, void *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = h->plan.exec(h->shift, h->has_window ? h->d_window.as<float>() : nullptr, (const float2 *)d_in, (float2 *)d_out,
                      noutput_items, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_fft_vcc_work(grhip_fft_vcc *h, int noutput_items, const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t bytes = (size_t)noutput_items * h->N * 8;
    if ((rc = h->stage_in.reserve(bytes))) return rc;
    if ((rc = h->stage_out.reserve(bytes))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, bytes, st);
    rc = grhip_fft_vcc_work_device(h, noutput_items, h->stage_in.p, h->stage_out.p, st);
    if (rc < 0) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, bytes, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

// ---- pfb_channelizer_ccf ---------------------------------------------------------
int grhip_pfb_channelizer_ccf_create(grhip_pfb_channelizer_ccf **h, unsigned numchans, const float *taps,
                                     size_t ntaps, float oversample_rate, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (numchans < 1 || numchans > 1024) return fail(GRHIP_EINVAL, "numchans must be in [1,1024]");
    if (ntaps && !taps) return fail(GRHIP_EINVAL, "taps is NULL");
    if (!(oversample_rate > 0)) return fail(GRHIP_EINVAL, "oversample rate must be positive");
    double intp = 0;
    double fltp = modf(numchans / oversample_rate, &intp);      // .cc:57-60
    if (fltp != 0.0)
        return fail(GRHIP_EINVAL, "gr_pfb_channelizer: oversample rate must be N/i for i in [1, N]");
    auto *p = new (std::nothrow) grhip_pfb_channelizer_ccf();
    if (!p) return fail(GRHIP_ENOMEM, "alloc");
    p->M = numchans; p->oversample_rate = oversample_rate;
    int rc = p->init_device(device);
    if (!rc) rc = p->set_taps(taps, ntaps);
    if (!rc) {
        p->rate_ratio = (int)rintf(numchans / oversample_rate);                           // .cc:82
        p->idxlut.resize(numchans);
        for (unsigned i = 0; i < numchans; i++)
            p->idxlut[i] = numchans - ((i + p->rate_ratio) % numchans) - 1;                // .cc:85
        p->output_multiple = 1;
        while ((p->output_multiple * p->rate_ratio) % numchans != 0) p->output_multiple++; // .cc:90-92
        rc = p->d_idxlut.reserve(numchans * sizeof(int));
        if (!rc) {
            hipError_t e = hipMemcpy(p->d_idxlut.p, p->idxlut.data(), numchans * sizeof(int), hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "idxlut upload");
        }
        std::vector<float2> dft(2 * (size_t)numchans);
        for (unsigned m = 0; m < numchans; ++m) {
            double ang = 2.0 * M_PI * (double)m / (double)numchans;     // FFTW_BACKWARD: +sign
            dft[m] = make_float2((float)cos(ang), (float)sin(ang));
            dft[numchans + m] = make_float2((float)cos(-ang), (float)sin(-ang));    // forward table (batched-FFT kernels)
        }
        if (!rc) rc = p->d_dft.reserve(2 * (size_t)numchans * sizeof(float2));
        if (!rc) {
            hipError_t e = hipMemcpy(p->d_dft.p, dft.data(), 2 * (size_t)numchans * sizeof(float2), hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "dft upload");
        }
    }
    if (rc) { p->d_ftaps.release(); p->d_idxlut.release(); p->d_dft.release(); p->destroy_base(); delete p; return rc; }
    *h = p;
    return GRHIP_OK;
}

void grhip_pfb_channelizer_ccf_destroy(grhip_pfb_channelizer_ccf *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_ftaps.release(); h->d_idxlut.release(); h->d_dft.release();
    h->d_hier_in.release(); h->d_hier_vec.release();
    h->destroy_base();
    delete h;
}

int grhip_pfb_channelizer_ccf_set_taps(grhip_pfb_channelizer_ccf *h, const float *taps, size_t ntaps)
{
    if (!h || (ntaps && !taps)) return fail(GRHIP_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    int rc = h->bind();
    if (rc) return rc;
    return h->set_taps(taps, ntaps);
}

int grhip_pfb_channelizer_ccf_history(const grhip_pfb_channelizer_ccf *h)
{
    return h ? (int)h->taps_per_filter + 1 : GRHIP_EINVAL;     // .cc:136
}

int grhip_pfb_channelizer_ccf_output_multiple(const grhip_pfb_channelizer_ccf *h)
{
    return h ? h->output_multiple : GRHIP_EINVAL;
}

static long long pfb_valid_outputs(const grhip_pfb_channelizer_ccf *h, int noutput_items, int *toconsume)
{
    // the reference loop emits a vector per iteration while n <= toconsume
    int tc = (int)rintf(noutput_items / h->oversample_rate);
    *toconsume = tc;
    // n_t = 1 + ((t+1)*rr - 1) / M <= tc   <=>   (t+1)*rr - 1 < tc*M
    long long lim = (long long)tc * h->M;                 // need (t+1)*rr <= lim
    long long cnt = lim / h->rate_ratio;
    if (cnt > noutput_items) cnt = noutput_items;
    return cnt;
}

int grhip_pfb_channelizer_ccf_general_work_device(grhip_pfb_channelizer_ccf *h, int noutput_items,
                                                  const void *d_in, size_t stream_stride_items, void *d_out,
                                                  void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if (h->updated) { h->updated = false; return 0; }      // .cc:169-172
    }
    int tc = 0;
    long long nvalid = pfb_valid_outputs(h, noutput_items, &tc);
    PfbArgs a;
    a.M = (int)h->M; a.tpf = (int)h->taps_per_filter; a.rate_ratio = h->rate_ratio;
    a.ftaps = h->d_ftaps.as<float>(); a.idxlut = h->d_idxlut.as<int>(); a.dft = h->d_dft.as<float2>();
    a.in = (const float2 *)d_in; a.stride = (long long)stream_stride_items;
    a.out = (float2 *)d_out; a.nout = nvalid;
    rc = launch_pfb(a, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_pfb_channelizer_ccf_hier_work_device(grhip_pfb_channelizer_ccf *h, int noutput_items, const void *d_in,
                                               void *d_out, size_t out_stride_items, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if (h->updated) { h->updated = false; return 0; }      // .cc:169-172
    }
    if (noutput_items == 0) return 0;
    if ((size_t)noutput_items > out_stride_items) return fail(GRHIP_EINVAL, "out_stride_items smaller than noutput_items");
    hipStream_t st = h->pick(stream);
    int tc = 0;
    const long long nvalid = pfb_valid_outputs(h, noutput_items, &tc);
    PfbArgs a;
    a.M = (int)h->M; a.tpf = (int)h->taps_per_filter; a.rate_ratio = h->rate_ratio;
    a.ftaps = h->d_ftaps.as<float>(); a.idxlut = h->d_idxlut.as<int>(); a.dft = h->d_dft.as<float2>();
    a.in = (const float2 *)d_in; a.stride = 0; a.out = nullptr; a.nout = nvalid;
    a.out_streams = (float2 *)d_out; a.out_stride = (long long)out_stride_items;
    rc = launch_pfb_hier(a, st);
    if (rc != -1) return rc ? rc : noutput_items;
    // no fused kernel for this shape: the three blocks one after the other (stream_to_streams, the channeliser,
    // vector_to_streams = the data movement of stream_to_streams on the output vectors)
    const size_t per = (size_t)h->taps_per_filter + (size_t)tc + 1;        // items per de-interleaved stream incl. history
    if ((rc = h->d_hier_in.reserve(per * h->M * sizeof(float2)))) return rc;
    if ((rc = h->d_hier_vec.reserve((size_t)(nvalid > 0 ? nvalid : 1) * h->M * sizeof(float2)))) return rc;
    const long long items_in = (long long)h->taps_per_filter + tc;          // what the interleaved stream holds per channel
    if ((rc = launch_streams(true, const_cast<void *>(d_in), h->d_hier_in.p, (long long)per, (int)h->M, sizeof(float2), items_in, st)))
        return rc;
    a.in = h->d_hier_in.as<float2>(); a.stride = (long long)per; a.out = h->d_hier_vec.as<float2>();
    a.out_streams = nullptr; a.out_stride = 0;
    if ((rc = launch_pfb(a, st))) return rc;
    if (nvalid > 0 &&
        (rc = launch_streams(true, h->d_hier_vec.p, d_out, (long long)out_stride_items, (int)h->M, sizeof(float2), nvalid, st)))
        return rc;
    return noutput_items;
}

int grhip_pfb_channelizer_ccf_general_work(grhip_pfb_channelizer_ccf *h, int noutput_items,
                                           const void *const *ins, void *out, int *consumed)
{
    if (!h || !ins) return fail(GRHIP_EINVAL, "null argument");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if (h->updated) { h->updated = false; if (consumed) *consumed = 0; return 0; }
    }
    int tc = 0;
    long long nvalid = pfb_valid_outputs(h, noutput_items, &tc);
    if (consumed) *consumed = tc;
    if (noutput_items == 0) return 0;
    // items readable on every stream: history-1 old ones + toconsume new ones
    size_t per = (size_t)h->taps_per_filter + (size_t)tc + 1;
    if ((rc = h->stage_in.reserve(per * h->M * 8))) return rc;
    if ((rc = h->stage_out.reserve((size_t)noutput_items * h->M * 8))) return rc;
    hipStream_t st = h->own_stream;
    size_t copy_items = (size_t)h->taps_per_filter + (size_t)tc;
    for (unsigned j = 0; j < h->M; ++j)
        GRHIP_HIP(hipMemcpyAsync(h->stage_in.as<float2>() + (size_t)j * per, ins[j], copy_items * 8,
                                 hipMemcpyHostToDevice, st));
    PfbArgs a;
    a.M = (int)h->M; a.tpf = (int)h->taps_per_filter; a.rate_ratio = h->rate_ratio;
    a.ftaps = h->d_ftaps.as<float>(); a.idxlut = h->d_idxlut.as<int>(); a.dft = h->d_dft.as<float2>();
    a.in = h->stage_in.as<float2>(); a.stride = (long long)per;
    a.out = h->stage_out.as<float2>(); a.nout = nvalid;
    if ((rc = launch_pfb(a, st))) return rc;
    if (nvalid > 0)
        GRHIP_D2H(h, out, h->stage_out.p, (size_t)nvalid * h->M * 8, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

}  // extern "C"
