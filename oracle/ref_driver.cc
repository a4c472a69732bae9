/*
 * ref_driver.cc -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin extern "C" driver around the pieces of the reference that compile
 * directly from /root/reference with nothing but -I paths (no generated code,
 * no Boost, no stand-in headers):
 *
 *   compiled where they lie (see oracle/Makefile):
 *     gnuradio-core/src/lib/general/gr_fast_atan2f.cc
 *     gnuradio-core/src/lib/general/gr_count_bits.cc
 *     gnuradio-core/src/lib/filter/float_dotprod_sse64.S
 *     gnuradio-core/src/lib/filter/fcomplex_dotprod_sse64.S
 *     gnuradio-core/src/lib/filter/ccomplex_dotprod_sse64.S
 *   header-only, included here:
 *     gnuradio-core/src/lib/filter/gr_rotator.h
 *     gnuradio-core/src/lib/general/gr_math.h   (gr_branchless_clip, gr_binary_slicer)
 *     gnuradio-core/src/lib/filter/interpolator_taps.h (MMSE table)
 *
 * NOT buildable here (need the Python-2 template generators / Boost):
 * gr_fir_XXX{,_generic,_simd}, gri_mmse_fir_interpolator, every gr_block
 * subclass.  The alignment wrapper around the SSE dot products below is this
 * file's own restatement of gr_fir_{ccf,ccc,fff}_simd::set_taps/filter
 * (filter/gr_fir_ccf_simd.cc:71-141, gr_fir_ccc_simd.cc:71-142,
 * gr_fir_fff_simd.cc:69-134).
 *
 * Output: oracle/_ref/libgrref.so (git-ignored; travels to the GPU box as a
 * binary; the reference sources never do).
 */
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <gr_complex.h>
#include <gr_count_bits.h>
#include <gr_math.h>
#include <gr_rotator.h>
#include <float_dotprod_x86.h>
#include <fcomplex_dotprod_x86.h>
#include <ccomplex_dotprod_x86.h>

namespace reftaps {
#include <interpolator_taps.h>
}

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_fast_atan2f_n(const float *y, const float *x, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = gr_fast_atan2f(y[i], x[i]);
}

REF_API void ref_branchless_clip_n(const float *x, float clip, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = gr_branchless_clip(x[i], clip);
}

REF_API void ref_count_bits64_n(const unsigned long long *x, unsigned *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = gr_count_bits64(x[i]);
}

REF_API void ref_binary_slicer_n(const float *x, unsigned char *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = (unsigned char)gr_binary_slicer(x[i]);
}

REF_API void ref_mmse_taps(float *out /* [129][8] */)
{
    for (int s = 0; s <= reftaps::NSTEPS; s++)
        for (int k = 0; k < reftaps::NTAPS; k++) out[s * reftaps::NTAPS + k] = reftaps::taps[s][k];
}

/* phases used by rotate() for outputs 0..n-1, starting from a fresh rotator */
REF_API void ref_rotator_phases(float incr_re, float incr_im, float *phases, size_t n)
{
    gr_rotator r;
    r.set_phase_incr(gr_complex(incr_re, incr_im));
    for (size_t i = 0; i < n; i++) {
        gr_complex z = r.rotate(gr_complex(1.0f, 0.0f));   /* 1*phase == phase exactly */
        phases[2 * i] = z.real();
        phases[2 * i + 1] = z.imag();
    }
}

REF_API void ref_rotator_rotate_n(float incr_re, float incr_im, const float *in, float *out, size_t n)
{
    gr_rotator r;
    r.set_phase_incr(gr_complex(incr_re, incr_im));
    const gr_complex *ci = (const gr_complex *)in;
    gr_complex *co = (gr_complex *)out;
    for (size_t i = 0; i < n; i++) co[i] = r.rotate(ci[i]);
}

/* raw asm entry points */
REF_API float ref_float_dotprod_sse(const float *input, const float *taps, unsigned n4)
{
    return float_dotprod_sse(input, taps, n4);
}
REF_API void ref_fcomplex_dotprod_sse(const float *input, const float *taps, unsigned n2, float *result)
{
    fcomplex_dotprod_sse(input, taps, n2, result);
}
REF_API void ref_ccomplex_dotprod_sse(const float *input, const float *taps, unsigned n2, float *result)
{
    ccomplex_dotprod_sse(input, taps, n2, result);
}

/* ---- SSE FIR, alignment handling restated from gr_fir_*_simd ---------- */
static float *calloc16(size_t nfloats)
{
    void *p = 0;
    if (posix_memalign(&p, 16, (nfloats ? nfloats : 4) * sizeof(float))) return 0;
    memset(p, 0, (nfloats ? nfloats : 4) * sizeof(float));
    return (float *)p;
}

struct sse_fir {
    unsigned ntaps;
    float *aligned[4];
};

/* kind: 0 fff, 1 ccf, 2 ccc.  taps forward order. */
static void sse_fir_build(sse_fir &f, int kind, const float *taps, unsigned ntaps)
{
    f.ntaps = ntaps;
    for (unsigned i = 0; i < 4; i++) {
        if (kind == 2) {
            f.aligned[i] = calloc16((size_t)(1 + (ntaps + i - 1) / 2) * 8);
            for (unsigned j = 0; j < ntaps; j++) {
                f.aligned[i][2 * (j + i)] = taps[2 * (ntaps - 1 - j)];
                f.aligned[i][2 * (j + i) + 1] = taps[2 * (ntaps - 1 - j) + 1];
            }
        } else {
            f.aligned[i] = calloc16((size_t)(1 + (ntaps + i - 1) / 4) * 4);
            for (unsigned j = 0; j < ntaps; j++) f.aligned[i][j + i] = taps[ntaps - 1 - j];
        }
    }
}
static void sse_fir_free(sse_fir &f) { for (int i = 0; i < 4; i++) free(f.aligned[i]); }

static inline float sse_fff_one(const sse_fir &f, const float *input)
{
    if (f.ntaps == 0) return 0.0f;
    const float *ar = (const float *)((uintptr_t)input & ~(uintptr_t)15);
    unsigned al = input - ar;
    return float_dotprod_sse(ar, f.aligned[al], (f.ntaps + al - 1) / 4 + 1);
}
static inline gr_complex sse_ccf_one(const sse_fir &f, const gr_complex *input)
{
    if (f.ntaps == 0) return 0.0f;
    const gr_complex *ar = (const gr_complex *)((uintptr_t)input & ~(uintptr_t)15);
    unsigned al = input - ar;
    float result[2];
    fcomplex_dotprod_sse(f.aligned[al], (const float *)ar, (f.ntaps + al - 1) / 2 + 1, result);
    return gr_complex(result[0], result[1]);
}
static inline gr_complex sse_ccc_one(const sse_fir &f, const gr_complex *input)
{
    if (f.ntaps == 0) return 0.0f;
    const gr_complex *ar = (const gr_complex *)((uintptr_t)input & ~(uintptr_t)15);
    unsigned al = input - ar;
    float result[2];
    ccomplex_dotprod_sse((const float *)ar, f.aligned[al], (f.ntaps + al - 1) / 2 + 1, result);
    return gr_complex(result[0], result[1]);
}

/* `in` must be 16-byte aligned minus nothing in particular: the reference
 * reads below `input` down to the 16-byte boundary, so callers pass buffers
 * with at least 16 bytes of readable slack in front and 32 behind. */
REF_API void ref_fir_fff_sse(const float *taps, unsigned ntaps, const float *in, float *out, size_t n,
                             unsigned decim)
{
    sse_fir f; sse_fir_build(f, 0, taps, ntaps);
    for (size_t i = 0; i < n; i++) out[i] = sse_fff_one(f, in + i * decim);
    sse_fir_free(f);
}
REF_API void ref_fir_ccf_sse(const float *taps, unsigned ntaps, const float *in, float *out, size_t n,
                             unsigned decim)
{
    sse_fir f; sse_fir_build(f, 1, taps, ntaps);
    const gr_complex *ci = (const gr_complex *)in; gr_complex *co = (gr_complex *)out;
    for (size_t i = 0; i < n; i++) co[i] = sse_ccf_one(f, ci + i * decim);
    sse_fir_free(f);
}
REF_API void ref_fir_ccc_sse(const float *taps, unsigned ntaps, const float *in, float *out, size_t n,
                             unsigned decim)
{
    sse_fir f; sse_fir_build(f, 2, taps, ntaps);
    const gr_complex *ci = (const gr_complex *)in; gr_complex *co = (gr_complex *)out;
    for (size_t i = 0; i < n; i++) co[i] = sse_ccc_one(f, ci + i * decim);
    sse_fir_free(f);
}

/* ---- chain: freq_xlating_fir_filter_ccc -> quadrature_demod_cf ---------
 * Block loop bodies restated from
 *   filter/gr_freq_xlating_fir_filter_XXX.cc.t:72-83,116-120
 *   general/gr_quadrature_demod_cf.cc:51-59
 * with the FIR inner loop = reference ccomplex_dotprod_sse, rotator =
 * reference gr_rotator.h, atan = reference gr_fast_atan2f.cc.
 * x: n_in complex samples (history zeros are prepended here).
 * Used as the "reference" CPU baseline by bench.py. */
REF_API size_t ref_chain_xlating_demod(unsigned decim, const float *proto, unsigned ntaps,
                                       double center_freq, double sampling_freq, float gain,
                                       const float *x, size_t n_in, float *y_out, float *demod_out)
{
    size_t n_out = n_in / decim;
    std::vector<gr_complex> ctaps(ntaps);
    const gr_complex *p = (const gr_complex *)proto;
    float fwT0 = 2 * M_PI * center_freq / sampling_freq;
    for (unsigned i = 0; i < ntaps; i++) ctaps[i] = p[i] * exp(gr_complex(0, i * fwT0));
    /* set_taps(gr_reverse(ctaps)) then gr_fir_ccc reverses again: forward taps = reverse(ctaps) */
    std::vector<gr_complex> fwd(ctaps.rbegin(), ctaps.rend());
    sse_fir f; sse_fir_build(f, 2, (const float *)fwd.data(), ntaps);
    gr_rotator r;
    r.set_phase_incr(exp(gr_complex(0, fwT0 * (int)decim)));

    size_t hist = ntaps ? ntaps - 1 : 0;
    /* 16-byte aligned buffer with slack on both sides */
    float *raw = calloc16((n_in + hist) * 2 + 64);
    gr_complex *buf = (gr_complex *)(raw + 16);
    memcpy((void *)(buf + hist), x, n_in * sizeof(gr_complex));
    std::vector<gr_complex> y(n_out + 1);
    y[0] = 0;
    size_t j = 0;
    for (size_t i = 0; i < n_out; i++) {
        y[i + 1] = r.rotate(sse_ccc_one(f, buf + j));
        j += decim;
    }
    for (size_t i = 0; i < n_out; i++) {
        gr_complex product = y[i + 1] * conj(y[i]);
        demod_out[i] = gain * gr_fast_atan2f(imag(product), real(product));
    }
    if (y_out) memcpy(y_out, (const void *)&y[1], n_out * sizeof(gr_complex));
    free(raw); sse_fir_free(f);
    return n_out;
}
