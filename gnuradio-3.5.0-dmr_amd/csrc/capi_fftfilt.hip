// capi_fftfilt.hip -- C ABI for gr_fft_filter_ccc.
// filter/gr_fft_filter_ccc.cc:46-128, filter/gri_fft_filter_ccc_generic.cc:63-170:
// overlap-ADD fast convolution with the reference's sizes, taps pre-scaled by 1/fftsize,
// tail carried between blocks and calls.  All blocks of a call are transformed in one
// batched launch of the FFT kernels (fft_kernels.hip); the overlap-add + decimation is a
// gather over the inverse transforms.
#include <cmath>
#include <complex>
#include <vector>

#include "fft_kernels.h"
#include "grhip_internal.h"

using namespace grhip;

struct grhip_fft_filter_ccc : HandleBase {
    int decim = 1, ntaps = 0, fftsize = 0, nsamples = 0;
    std::vector<std::complex<float>> new_taps;
    bool updated = false;
    DevBuf d_xformed, d_tail, d_a, d_b;
    FftPlan plan;              // the fftsize-point transform, both directions (four-step form above 8192 points)
    // fused overlap-save path (ntaps <= FUSED_MAX_TAPS): 4096-point blocks, see fftfilt4096_kernel
    bool fused = false;
    int L = 0, fold = 0;             // full-rate outputs per block (a multiple of the decimation); folded inverse
    DevBuf d_tw4096, d_H4096, d_hist[2];
    int hist_cur = 0;

    int install(const std::complex<float> *taps, size_t n)
    {
        // compute_sizes + set_taps (gri_fft_filter_ccc_generic.cc:63-118)
        ntaps = (int)n;
        fftsize = (int)(2 * pow(2.0, ceil(log((double)ntaps) / log(2.0))));
        nsamples = fftsize - ntaps + 1;
        if (ntaps > (1 << 25) || !FftPlan::size_ok(fftsize))
            return fail(GRHIP_EINVAL, "fft_filter_ccc: %d taps need an FFT of more than 2^26 points", ntaps);
        int rcp = plan.build(fftsize, 1);
        if (rcp) return rcp;
        // forward transform of the scaled, zero-padded taps (double, rounded once)
        const float scale = 1.0 / fftsize;                                              // :76
        std::vector<std::complex<double>> t((size_t)fftsize, std::complex<double>(0, 0));
        for (int i = 0; i < ntaps; ++i)
            t[i] = std::complex<double>((double)(taps[i].real() * scale), (double)(taps[i].imag() * scale));
        host_fft_pow2(t, -1);
        std::vector<float2> H((size_t)fftsize);
        for (int k = 0; k < fftsize; ++k) H[k] = make_float2((float)t[k].real(), (float)t[k].imag());
        fused = ntaps <= OLS_MAX_TAPS && ((OLS_N - (ntaps - 1)) / decim) >= 1;
        if (fused) {
            int rc4 = ols_build((const float *)taps, ntaps, decim, d_tw4096, d_H4096, &L, &fold);
            const size_t hl = (size_t)(ntaps > 1 ? ntaps - 1 : 1);
            if (!rc4) rc4 = d_hist[0].reserve(hl * sizeof(float2));
            if (!rc4) rc4 = d_hist[1].reserve(hl * sizeof(float2));
            if (rc4) return rc4;
            GRHIP_HIP(hipMemset(d_hist[0].p, 0, hl * sizeof(float2)));       // a fresh filter starts from silence
            GRHIP_HIP(hipMemset(d_hist[1].p, 0, hl * sizeof(float2)));
            hist_cur = 0;
        }
        const size_t tail_items = (size_t)(ntaps > 1 ? ntaps - 1 : 1);
        int rc = d_xformed.reserve(H.size() * sizeof(float2));
        if (!rc) rc = d_tail.reserve(tail_items * sizeof(float2));
        if (rc) return rc;
        GRHIP_HIP(hipMemcpy(d_xformed.p, H.data(), H.size() * sizeof(float2), hipMemcpyHostToDevice));
        GRHIP_HIP(hipMemset(d_tail.p, 0, tail_items * sizeof(float2)));                 // tail cleared (:69-71)
        return GRHIP_OK;
    }
    void release_all()
    {
        plan.release(); d_xformed.release(); d_tail.release(); d_a.release(); d_b.release();
        d_tw4096.release(); d_H4096.release(); d_hist[0].release(); d_hist[1].release();
    }
};

extern "C" {

int grhip_fft_filter_ccc_create(grhip_fft_filter_ccc **h, int decimation, const float *taps, size_t ntaps, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (decimation < 1) return fail(GRHIP_EINVAL, "decimation must be >= 1");
    if (!taps || ntaps < 1) return fail(GRHIP_EINVAL, "fft_filter_ccc needs at least one tap");
    auto *f = new (std::nothrow) grhip_fft_filter_ccc();
    if (!f) return fail(GRHIP_ENOMEM, "alloc");
    f->decim = decimation;
    int rc = f->init_device(device);
    if (!rc) rc = f->install((const std::complex<float> *)taps, ntaps);
    if (rc) { f->release_all(); f->destroy_base(); delete f; return rc; }
    *h = f;
    return GRHIP_OK;
}

void grhip_fft_filter_ccc_destroy(grhip_fft_filter_ccc *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->release_all();
    h->destroy_base();
    delete h;
}

int grhip_fft_filter_ccc_set_taps(grhip_fft_filter_ccc *h, const float *taps, size_t ntaps)
{
    if (!h || !taps || ntaps < 1) return fail(GRHIP_EINVAL, "bad argument");
    h->new_taps.assign((const std::complex<float> *)taps, (const std::complex<float> *)taps + ntaps);
    h->updated = true;                                   // gr_fft_filter_ccc.cc:88-93
    return GRHIP_OK;
}

int grhip_fft_filter_ccc_nsamples(const grhip_fft_filter_ccc *h) { return h ? h->nsamples : GRHIP_EINVAL; }
int grhip_fft_filter_ccc_decimation(const grhip_fft_filter_ccc *h) { return h ? h->decim : GRHIP_EINVAL; }

int grhip_fft_filter_ccc_work_device(grhip_fft_filter_ccc *h, int noutput_items, const void *d_in, void *d_out,
                                     void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    if (h->updated) {                                    // .cc:113-118: new sizes, produce nothing this call
        GRHIP_HIP(hipStreamSynchronize(st));
        rc = h->install(h->new_taps.data(), h->new_taps.size());
        if (rc) return rc;
        h->updated = false;
        return 0;
    }
    if (noutput_items == 0) return 0;
    if (noutput_items % h->nsamples)
        return fail(GRHIP_EINVAL, "noutput_items must be a multiple of nsamples (%d)", h->nsamples);
    const long long nin = (long long)noutput_items * h->decim;
    if (h->fused) {
        const float2 *hist = h->d_hist[h->hist_cur].as<float2>();
        float2 *hist_new = h->d_hist[h->hist_cur ^ 1].as<float2>();
        if ((rc = launch_fftfilt4096((const float2 *)d_in, nin, hist, h->ntaps, h->d_tw4096.as<float2>(),
                                     h->d_H4096.as<float2>(), (float2 *)d_out, noutput_items, h->decim, h->L, h->fold, st,
                                     hist_new)))
            return rc;
        h->hist_cur ^= 1;
        return noutput_items;
    }
    const long long nblk = nin / h->nsamples;
    const size_t bytes = (size_t)nblk * h->fftsize * sizeof(float2);
    if ((rc = h->d_a.reserve(bytes))) return rc;
    if ((rc = h->d_b.reserve(bytes))) return rc;
    float2 *A = h->d_a.as<float2>(), *B = h->d_b.as<float2>();
    const int tailsize = h->ntaps - 1;
    if ((rc = launch_fftfilt_pack((const float2 *)d_in, A, h->nsamples, h->fftsize, nblk, st))) return rc;
    if ((rc = h->plan.exec_pow2(1, 0, nullptr, A, B, nblk, st))) return rc;
    if ((rc = launch_fftfilt_mul(B, h->d_xformed.as<float2>(), h->fftsize, nblk, st))) return rc;
    if ((rc = h->plan.exec_pow2(0, 0, nullptr, B, A, nblk, st))) return rc;
    if ((rc = launch_fftfilt_ola(A, h->d_tail.as<float2>(), (float2 *)d_out, noutput_items, h->decim, h->nsamples,
                                 h->fftsize, tailsize, st)))
        return rc;
    if ((rc = launch_fftfilt_tail(A, h->d_tail.as<float2>(), nblk, h->nsamples, h->fftsize, tailsize, st))) return rc;
    return noutput_items;
}

int grhip_fft_filter_ccc_work(grhip_fft_filter_ccc *h, int noutput_items, const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->own_stream;
    if (h->updated || noutput_items == 0) return grhip_fft_filter_ccc_work_device(h, noutput_items, nullptr, nullptr, st);
    const size_t nin = (size_t)noutput_items * h->decim;
    if ((rc = h->stage_in.reserve(nin * 8))) return rc;
    if ((rc = h->stage_out.reserve((size_t)noutput_items * 8))) return rc;
    GRHIP_H2D(h, h->stage_in.p, in, nin * 8, st);
    rc = grhip_fft_filter_ccc_work_device(h, noutput_items, h->stage_in.p, h->stage_out.p, st);
    if (rc < 0) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)noutput_items * 8, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return rc;
}

}  // extern "C"
