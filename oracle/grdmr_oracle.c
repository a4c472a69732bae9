/*
 * grdmr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one thread) of the GNU Radio 3.5.0 DMR
 * demodulation hot path, written from scratch following the reference's
 * arithmetic step by step.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (libgrhip.so) never links or calls it.
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math) so that
 * every float operation is a single IEEE-754 binary32 round-to-nearest-even
 * operation, exactly like the reference's x86-64 SSE scalar code.
 *
 * Parity pins (see oracle/README.md and tests/test_oracle_*.py):
 *   - fast_atan2f, rotator, branchless_clip, count_bits64, MMSE taps:
 *     bit-exact against the reference's own sources compiled into
 *     oracle/_ref/libgrref.so (fixtures in tests/golden/).
 *   - FIR: qa_gr_fir_fff.cc:58-76 known vectors; reference SSE dot-product asm
 *     (bit-exact on integer-valued data, 1e-5 on float data).
 *   - correlate_access_code: qa_correlate_access_code.py:50-78 (exact).
 *   - clock_recovery_mm_ff: qa_clock_recovery_mm.py:70-102,140-172.
 *   - FFT: qa_fft.py:50-153 32-point vectors (rel 4e-4 as in the reference).
 *   - xlating / quad_demod / pfb block loops: no reference test exists; the
 *     loops are restated literally from the cited lines ("parity pinned at
 *     kernel level only").
 *
 * All citations are relative to /root/reference/.
 * Complex data is interleaved (re, im) float, as gr_complex
 * (gnuradio-core/src/lib/runtime/gr_complex.h:26).
 */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* FIR kernels: gr_fir_XXX_generic                                     */
/* gnuradio-core/src/lib/filter/gr_fir_XXX_generic.cc.t:30-103         */
/* Taps are stored reversed (gr_fir_XXX.h.t:51,65,103-106):            */
/*   d_taps[k] = taps_fwd[ntaps-1-k];  y = sum_k d_taps[k]*input[k]    */
/* ------------------------------------------------------------------ */

/* fff: ACC float, N_UNROLL = 4 (generate_gr_fir_XXX.py:60-66; .cc.t:30-55) */
static float fir_fff_one(const float *dt, unsigned ntaps, const float *in)
{
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    unsigned i = 0, n = (ntaps / 4) * 4;
    for (i = 0; i < n; i += 4) {
        acc0 += dt[i + 0] * in[i + 0];
        acc1 += dt[i + 1] * in[i + 1];
        acc2 += dt[i + 2] * in[i + 2];
        acc3 += dt[i + 3] * in[i + 3];
    }
    for (; i < ntaps; i++)
        acc0 += dt[i] * in[i];
    return (acc0 + acc1 + acc2 + acc3);
}

/* ccf: ACC gr_complex, N_UNROLL = 2 (.cc.t:59-79).  float*complex is
 * (re*t, im*t) (libstdc++ operator*(T, complex<T>)), then complex +=. */
static void fir_ccf_one(const float *dt, unsigned ntaps, const float *in, float *out)
{
    float a0r = 0, a0i = 0, a1r = 0, a1i = 0;
    unsigned i = 0, n = (ntaps / 2) * 2;
    for (i = 0; i < n; i += 2) {
        float pr, pi;
        pr = in[2 * i + 0] * dt[i];     pi = in[2 * i + 1] * dt[i];
        a0r += pr;                      a0i += pi;
        pr = in[2 * i + 2] * dt[i + 1]; pi = in[2 * i + 3] * dt[i + 1];
        a1r += pr;                      a1i += pi;
    }
    for (; i < ntaps; i++) {
        float pr = in[2 * i] * dt[i], pi = in[2 * i + 1] * dt[i];
        a0r += pr; a0i += pi;
    }
    out[0] = a0r + a1r;
    out[1] = a0i + a1i;
}

/* complex<float> product for finite operands: libgcc __mulsc3 computes
 * ac, bd, ad, bc separately, then (ac - bd, ad + bc). */
static inline void cmul(float a, float b, float c, float d, float *re, float *im)
{
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    *re = ac - bd;
    *im = ad + bc;
}

/* ccc: ACC gr_complex, N_UNROLL = 2, taps complex (.cc.t:59-79) */
static void fir_ccc_one(const float *dt, unsigned ntaps, const float *in, float *out)
{
    float a0r = 0, a0i = 0, a1r = 0, a1i = 0;
    unsigned i = 0, n = (ntaps / 2) * 2;
    float pr, pi;
    for (i = 0; i < n; i += 2) {
        cmul(dt[2 * i], dt[2 * i + 1], in[2 * i], in[2 * i + 1], &pr, &pi);
        a0r += pr; a0i += pi;
        cmul(dt[2 * i + 2], dt[2 * i + 3], in[2 * i + 2], in[2 * i + 3], &pr, &pi);
        a1r += pr; a1i += pi;
    }
    for (; i < ntaps; i++) {
        cmul(dt[2 * i], dt[2 * i + 1], in[2 * i], in[2 * i + 1], &pr, &pi);
        a0r += pr; a0i += pi;
    }
    out[0] = a0r + a1r;
    out[1] = a0i + a1i;
}

static float *reversed(const float *taps, unsigned ntaps, unsigned width)
{
    float *dt = (float *)malloc((size_t)(ntaps ? ntaps : 1) * width * sizeof(float));
    for (unsigned k = 0; k < ntaps; k++)
        for (unsigned w = 0; w < width; w++)
            dt[k * width + w] = taps[(ntaps - 1 - k) * width + w];
    return dt;
}

/* filterNdec (.cc.t:92-103): output[i] = filter(&input[i*decimate]).
 * `in` must hold (n-1)*decim + ntaps items.  taps in forward order. */
ORC_API void orc_fir_fff(const float *taps_fwd, unsigned ntaps, const float *in,
                         float *out, size_t n, unsigned decim)
{
    float *dt = reversed(taps_fwd, ntaps, 1);
    for (size_t i = 0; i < n; i++)
        out[i] = fir_fff_one(dt, ntaps, in + i * decim);
    free(dt);
}

ORC_API void orc_fir_ccf(const float *taps_fwd, unsigned ntaps, const float *in,
                         float *out, size_t n, unsigned decim)
{
    float *dt = reversed(taps_fwd, ntaps, 1);
    for (size_t i = 0; i < n; i++)
        fir_ccf_one(dt, ntaps, in + 2 * i * decim, out + 2 * i);
    free(dt);
}

ORC_API void orc_fir_ccc(const float *taps_fwd, unsigned ntaps, const float *in,
                         float *out, size_t n, unsigned decim)
{
    float *dt = reversed(taps_fwd, ntaps, 2);
    for (size_t i = 0; i < n; i++)
        fir_ccc_one(dt, ntaps, in + 2 * i * decim, out + 2 * i);
    free(dt);
}

/* ------------------------------------------------------------------ */
/* gr_rotator  (gnuradio-core/src/lib/filter/gr_rotator.h:29-52)       */
/* ------------------------------------------------------------------ */
typedef struct {
    float pr, pi;   /* d_phase      */
    float ir, ii;   /* d_phase_incr */
    unsigned counter;
} orc_rotator;

ORC_API void orc_rotator_init(orc_rotator *r)
{
    r->pr = 1; r->pi = 0; r->ir = 1; r->ii = 0; r->counter = 0;
}

/* set_phase_incr: incr / abs(incr)  (gr_rotator.h:38); abs = cabsf = hypotf,
 * complex/float divides both parts. */
ORC_API void orc_rotator_set_phase_incr(orc_rotator *r, float re, float im)
{
    float a = hypotf(re, im);
    r->ir = re / a;
    r->ii = im / a;
}

ORC_API void orc_rotator_set_phase(orc_rotator *r, float re, float im)
{
    float a = hypotf(re, im);
    r->pr = re / a;
    r->pi = im / a;
}

/* rotate (gr_rotator.h:40-50) */
static inline void rotator_rotate(orc_rotator *r, float xr, float xi, float *zr, float *zi)
{
    r->counter++;
    cmul(xr, xi, r->pr, r->pi, zr, zi);            /* z = in * d_phase        */
    float nr, ni;
    cmul(r->pr, r->pi, r->ir, r->ii, &nr, &ni);    /* d_phase *= d_phase_incr */
    r->pr = nr; r->pi = ni;
    if ((r->counter % 512) == 0) {
        float a = hypotf(r->pr, r->pi);            /* d_phase /= abs(d_phase) */
        r->pr = r->pr / a;
        r->pi = r->pi / a;
    }
}

ORC_API void orc_rotator_rotate_n(orc_rotator *r, const float *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++)
        rotator_rotate(r, in[2 * i], in[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

/* phase sequence seen by outputs 0..n-1 (the value of d_phase used by rotate) */
ORC_API void orc_rotator_phases(orc_rotator *r, float *phases, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float zr, zi;
        phases[2 * i] = r->pr; phases[2 * i + 1] = r->pi;
        rotator_rotate(r, 0.f, 0.f, &zr, &zi);
    }
}

/* ------------------------------------------------------------------ */
/* gr_freq_xlating_fir_filter_ccc                                      */
/* filter/gr_freq_xlating_fir_filter_XXX.cc.t:72-83 (build), :116-120   */
/* ------------------------------------------------------------------ */
typedef struct {
    unsigned ntaps;
    unsigned decim;
    float *ctaps;        /* composite taps in natural order == d_taps of the
                            gr_fir_ccc after reverse(reverse()) (.cc.t:81 +
                            gr_fir_XXX.h.t:103-106) */
    orc_rotator rot;
} orc_xlating;

/* proto: complex prototype taps (ccc signature: TAP_TYPE = gr_complex) */
ORC_API orc_xlating *orc_xlating_ccc_new(unsigned decim, const float *proto, unsigned ntaps,
                                         double center_freq, double sampling_freq)
{
    orc_xlating *x = (orc_xlating *)calloc(1, sizeof(*x));
    x->ntaps = ntaps; x->decim = decim;
    x->ctaps = (float *)malloc((size_t)(ntaps ? ntaps : 1) * 2 * sizeof(float));
    orc_rotator_init(&x->rot);
    float fwT0 = 2 * M_PI * center_freq / sampling_freq;      /* .cc.t:77 */
    for (unsigned i = 0; i < ntaps; i++) {
        float ang = i * fwT0;                                 /* unsigned*float -> float */
        float complex e = cexpf(CMPLXF(0.0f, ang));           /* std::exp(complex<float>) */
        cmul(proto[2 * i], proto[2 * i + 1], crealf(e), cimagf(e),
             &x->ctaps[2 * i], &x->ctaps[2 * i + 1]);         /* .cc.t:79 */
    }
    float ang = fwT0 * decim;                                 /* float*int -> float, .cc.t:82 */
    float complex e = cexpf(CMPLXF(0.0f, ang));
    orc_rotator_set_phase_incr(&x->rot, crealf(e), cimagf(e));
    return x;
}

ORC_API void orc_xlating_free(orc_xlating *x) { if (x) { free(x->ctaps); free(x); } }
ORC_API const float *orc_xlating_ctaps(const orc_xlating *x) { return x->ctaps; }
ORC_API void orc_xlating_get_rot(const orc_xlating *x, float *five)
{
    five[0] = x->rot.pr; five[1] = x->rot.pi; five[2] = x->rot.ir; five[3] = x->rot.ii;
    five[4] = (float)x->rot.counter;
}

/* work (.cc.t:116-120): in holds (nout-1)*decim + ntaps complex items
 * (history ntaps-1 in front). */
ORC_API void orc_xlating_ccc_work(orc_xlating *x, const float *in, float *out, size_t nout)
{
    size_t j = 0;
    for (size_t i = 0; i < nout; i++) {
        float y[2];
        fir_ccc_one(x->ctaps, x->ntaps, in + 2 * j, y);
        rotator_rotate(&x->rot, y[0], y[1], &out[2 * i], &out[2 * i + 1]);
        j += x->decim;
    }
}

/* ------------------------------------------------------------------ */
/* gr_fast_atan2f  (gnuradio-core/src/lib/general/gr_fast_atan2f.cc)    */
/* The 257-entry table (:38-125) is numeric data: atan_table.inc holds  */
/* the binary32 bit patterns of the reference's decimal literals        */
/* (double literal narrowed to float, as the C initialiser does),       */
/* written by oracle/extract_tables.py.  Pinned against the compiled    */
/* reference on the golden grid (tests/test_oracle_golden.py).          */
/* ------------------------------------------------------------------ */
static float atan_table[257];
static int atan_table_ready = 0;

#include "atan_table.inc"   /* static const uint32_t atan_table_bits[257] */

static void atan_table_init(void)
{
    for (int i = 0; i < 257; i++) {
        uint32_t u = atan_table_bits[i];
        memcpy(&atan_table[i], &u, 4);
    }
    atan_table_ready = 1;
}

ORC_API const float *orc_atan_table(void)
{
    if (!atan_table_ready) atan_table_init();
    return atan_table;
}

/* gr_fast_atan2f.cc:125-198, mixed float/double promotions kept as written */
ORC_API float orc_fast_atan2f(float y, float x)
{
    float x_abs, y_abs, z;
    float alpha, angle, base_angle;
    int index;
    if (!atan_table_ready) atan_table_init();

    if ((y == 0.0) && (x == 0.0))            /* :133 */
        return 0.0;
    y_abs = fabs(y);                          /* :137 */
    x_abs = fabs(x);
    if (y_abs < x_abs)                        /* :140 */
        z = y_abs / x_abs;
    else
        z = x_abs / y_abs;
    if (z < 0.003921569)                      /* :147 double compare */
        base_angle = z;
    else {
        alpha = z * (float)256 - .5;          /* :151 float mul, double sub, narrow */
        index = (int)alpha;
        alpha -= (float)index;
        base_angle = atan_table[index];
        base_angle += (atan_table[index + 1] - atan_table[index]) * alpha;
    }
    if (x_abs > y_abs) {                      /* :161 */
        if (x >= 0.0) {
            if (y >= 0.0) angle = base_angle;
            else          angle = -base_angle;
        } else {
            angle = 3.14159265358979323846;
            if (y >= 0.0) angle -= base_angle;
            else          angle = base_angle - angle;
        }
    } else {
        if (y >= 0.0) {
            angle = 1.57079632679489661923;
            if (x >= 0.0) angle -= base_angle;
            else          angle += base_angle;
        } else {
            angle = -1.57079632679489661923;
            if (x >= 0.0) angle += base_angle;
            else          angle -= base_angle;
        }
    }
    return angle;
}

ORC_API void orc_fast_atan2f_n(const float *y, const float *x, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = orc_fast_atan2f(y[i], x[i]);
}

/* ------------------------------------------------------------------ */
/* gr_quadrature_demod_cf::work (general/gr_quadrature_demod_cf.cc:46-62)*/
/* `in` holds nout+1 complex items (history 2: in[0] is the previous    */
/* item).                                                               */
/* ------------------------------------------------------------------ */
ORC_API void orc_quad_demod_cf(float gain, const float *in, float *out, size_t nout)
{
    in += 2;                                   /* in++ (:54) */
    for (size_t i = 0; i < nout; i++) {
        /* product = in[i] * conj(in[i-1]) (:57); conj negates imag, then
         * the generic complex product */
        float a = in[2 * i], b = in[2 * i + 1];
        float c = in[2 * i - 2], d = -in[2 * i - 1];
        float pr, pi;
        cmul(a, b, c, d, &pr, &pi);
        out[i] = gain * orc_fast_atan2f(pi, pr);   /* :59 */
    }
}

/* ------------------------------------------------------------------ */
/* gri_mmse_fir_interpolator (filter/gri_mmse_fir_interpolator.cc:33-71)*/
/* tap table filter/interpolator_taps.h:6-9 -> mmse_taps.inc (bit        */
/* patterns, [129][8], forward order as in the table).                  */
/* ------------------------------------------------------------------ */
#include "mmse_taps.inc"   /* static const uint32_t mmse_taps_bits[129][8] */
#define MMSE_NTAPS 8
#define MMSE_NSTEPS 128
static float mmse_rev[MMSE_NSTEPS + 1][MMSE_NTAPS];   /* reversed, as gr_fir_fff stores them */
static int mmse_ready = 0;

static void mmse_init(void)
{
    for (int s = 0; s <= MMSE_NSTEPS; s++)
        for (int k = 0; k < MMSE_NTAPS; k++) {
            uint32_t u = mmse_taps_bits[s][MMSE_NTAPS - 1 - k];
            memcpy(&mmse_rev[s][k], &u, 4);
        }
    mmse_ready = 1;
}

ORC_API const float *orc_mmse_taps_reversed(void)
{
    if (!mmse_ready) mmse_init();
    return &mmse_rev[0][0];
}

/* interpolate (:61-71): imu = (int) rint(mu * NSTEPS); 8-tap gr_fir_fff */
ORC_API float orc_mmse_interpolate(const float *input, float mu)
{
    if (!mmse_ready) mmse_init();
    int imu = (int)rint(mu * MMSE_NSTEPS);
    return fir_fff_one(mmse_rev[imu], MMSE_NTAPS, input);
}

/* ------------------------------------------------------------------ */
/* gr_branchless_clip (general/gr_math.h:63-69)                         */
/* ------------------------------------------------------------------ */
ORC_API float orc_branchless_clip(float x, float clip)
{
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5 * x1;
}

/* ------------------------------------------------------------------ */
/* digital_clock_recovery_mm_ff                                         */
/* gr-digital/lib/digital_clock_recovery_mm_ff.cc:48-139                */
/* gr-digital/include/digital_clock_recovery_mm_ff.h:70-92              */
/* ------------------------------------------------------------------ */
typedef struct {
    float mu, omega, min_omega, omega_mid, max_omega;
    float gain_omega, gain_mu, last_sample, omega_relative_limit;
} orc_mm;

ORC_API void orc_mm_set_omega(orc_mm *s, float omega)      /* .h:70-75 */
{
    s->omega = omega;
    s->min_omega = omega * (1.0 - s->omega_relative_limit);
    s->max_omega = omega * (1.0 + s->omega_relative_limit);
    s->omega_mid = 0.5 * (s->min_omega + s->max_omega);
}

/* returns 0 ok, -1 out_of_range (.cc:58-61) */
ORC_API int orc_mm_init(orc_mm *s, float omega, float gain_omega, float mu, float gain_mu,
                        float omega_relative_limit)
{
    if (omega < 1) return -1;
    if (gain_mu < 0 || gain_omega < 0) return -1;
    s->mu = mu; s->gain_omega = gain_omega; s->gain_mu = gain_mu;
    s->last_sample = 0; s->omega_relative_limit = omega_relative_limit;
    orc_mm_set_omega(s, omega);
    return 0;
}

/* forecast (.cc:80-87) */
ORC_API int orc_mm_forecast(const orc_mm *s, int noutput_items)
{
    return (int)ceil((noutput_items * s->omega) + MMSE_NTAPS);
}

static inline float mm_slice(float x) { return x < 0 ? -1.0F : 1.0F; }   /* .cc:89-93 */

/* general_work (.cc:104-139).  Returns number of outputs; *consumed = ii. */
ORC_API int orc_mm_general_work(orc_mm *s, int noutput_items, int ninput_items,
                                const float *in, float *out, int *consumed)
{
    int ii = 0, oo = 0;
    int ni = ninput_items - MMSE_NTAPS;
    float mm_val;
    while (oo < noutput_items && ii < ni) {
        out[oo] = orc_mmse_interpolate(&in[ii], s->mu);
        mm_val = mm_slice(s->last_sample) * out[oo] - mm_slice(out[oo]) * s->last_sample;
        s->last_sample = out[oo];

        s->omega = s->omega + s->gain_omega * mm_val;
        s->omega = s->omega_mid +
                   orc_branchless_clip(s->omega - s->omega_mid, s->omega_relative_limit);
        s->mu = s->mu + s->omega + s->gain_mu * mm_val;

        ii += (int)floor(s->mu);
        s->mu = s->mu - floor(s->mu);
        oo++;
    }
    *consumed = ii;
    return oo;
}

/* ------------------------------------------------------------------ */
/* digital_binary_slicer_fb (gr-digital/lib/digital_binary_slicer_fb.cc */
/* :54-56; general/gr_math.h:82-88)                                     */
/* ------------------------------------------------------------------ */
ORC_API void orc_binary_slicer_fb(const float *in, unsigned char *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = (in[i] >= 0) ? 1 : 0;
}

/* ------------------------------------------------------------------ */
/* pager_slicer_fb (gr-pager/lib/pager_slicer_fb.cc:34-84): 4-level      */
/* slicer behind a one-pole DC tracker.  d_alpha, d_beta, d_avg are       */
/* floats; d_beta = 1.0 - alpha is formed in double and narrowed (:40).   */
/* `avg` carries d_avg across calls.                                      */
/* ------------------------------------------------------------------ */
ORC_API void orc_pager_slicer_fb(float alpha, float *avg, const float *in, unsigned char *out, size_t n)
{
    const float d_alpha = alpha;
    const float d_beta = (float)(1.0 - (double)alpha);
    float d_avg = *avg;
    for (size_t i = 0; i < n; i++) {
        float sample = in[i];
        d_avg = d_avg * d_beta + sample * d_alpha;          /* :52 */
        sample -= d_avg;                                    /* :53 */
        unsigned char decision;
        if (sample > 0) decision = (sample > 2.0) ? 3 : 2;  /* :55-60 */
        else decision = (sample < -2.0) ? 0 : 1;            /* :61-66 */
        out[i] = decision;
    }
    *avg = d_avg;
}

/* ------------------------------------------------------------------ */
/* gr_unpack_k_bits_bb (general/gr_unpack_k_bits_bb.cc:57-74)            */
/* ------------------------------------------------------------------ */
ORC_API void orc_unpack_k_bits_bb(unsigned k, const unsigned char *in, unsigned char *out, size_t noutput_items)
{
    size_t n = 0;
    for (size_t i = 0; i < noutput_items / k; i++) {
        unsigned int t = in[i];
        for (int j = (int)k - 1; j >= 0; j--) out[n++] = (unsigned char)((t >> j) & 0x01);
    }
}

/* ------------------------------------------------------------------ */
/* digital_clock_recovery_mm_cc (gr-digital/lib/digital_clock_recovery_mm_cc.cc:49-215, */
/* include/digital_clock_recovery_mm_cc.h:67-110) with gri_mmse_fir_interpolator_cc     */
/* (filter/gri_mmse_fir_interpolator_cc.cc:33-80: the same tap table through gr_fir_ccf).*/
/* ------------------------------------------------------------------ */
typedef struct {
    float mu, omega, min_omega, omega_mid, max_omega;
    float gain_omega, gain_mu, omega_relative_limit;
    float p_2T[2], p_1T[2], p_0T[2], c_2T[2], c_1T[2], c_0T[2];
} orc_mmcc;

ORC_API size_t orc_mmcc_size(void) { return sizeof(orc_mmcc); }

ORC_API int orc_mmcc_init(orc_mmcc *s, float omega, float gain_omega, float mu, float gain_mu, float rel)
{
    if (omega <= 0.0) return -1;                                   /* .cc:62-63 */
    if (gain_mu < 0 || gain_omega < 0) return -1;                  /* .cc:64-65 */
    memset(s, 0, sizeof(*s));
    s->mu = mu; s->gain_omega = gain_omega; s->gain_mu = gain_mu; s->omega_relative_limit = rel;
    s->omega = omega;                                              /* set_omega, .h:76-81 */
    s->min_omega = omega * (1.0 - rel);
    s->max_omega = omega * (1.0 + rel);
    s->omega_mid = 0.5 * (s->min_omega + s->max_omega);
    return 0;
}

ORC_API int orc_mmcc_forecast(const orc_mmcc *s, int noutput_items)
{
    return (int)ceil((noutput_items * s->omega) + MMSE_NTAPS) + 16;   /* .cc:82-83, FUDGE = 16 */
}

ORC_API float orc_mmcc_mu(const orc_mmcc *s) { return s->mu; }
ORC_API float orc_mmcc_omega(const orc_mmcc *s) { return s->omega; }

/* general_work (.cc:117-215): foptr may be NULL (second loop: clip 1.0 instead of 4.0) */
ORC_API int orc_mmcc_general_work(orc_mmcc *s, int noutput_items, int ninput_items, const float *in, float *out,
                                  float *foptr, int *consumed)
{
    if (!mmse_ready) mmse_init();
    int ii = 0, oo = 0;
    int ni = ninput_items - MMSE_NTAPS - 16;
    const float lim = foptr ? 4.0f : 1.0f;
    while (oo < noutput_items && ii < ni) {
        s->p_2T[0] = s->p_1T[0]; s->p_2T[1] = s->p_1T[1];
        s->p_1T[0] = s->p_0T[0]; s->p_1T[1] = s->p_0T[1];
        int imu = (int)rint(s->mu * MMSE_NSTEPS);
        fir_ccf_one(mmse_rev[imu], MMSE_NTAPS, in + 2 * (size_t)ii, s->p_0T);
        s->c_2T[0] = s->c_1T[0]; s->c_2T[1] = s->c_1T[1];
        s->c_1T[0] = s->c_0T[0]; s->c_1T[1] = s->c_0T[1];
        s->c_0T[0] = s->p_0T[0] > 0 ? 1.0f : 0.0f;                 /* slicer_0deg, .cc:86-96 */
        s->c_0T[1] = s->p_0T[1] > 0 ? 1.0f : 0.0f;
        float xr, xi, yr, yi;
        cmul(s->c_0T[0] - s->c_2T[0], s->c_0T[1] - s->c_2T[1], s->p_1T[0], -s->p_1T[1], &xr, &xi);   /* .cc:147 */
        cmul(s->p_0T[0] - s->p_2T[0], s->p_0T[1] - s->p_2T[1], s->c_1T[0], -s->c_1T[1], &yr, &yi);   /* .cc:148 */
        float mm_val = yr - xr;                                    /* u = y - x; u.real() */
        out[2 * oo] = s->p_0T[0]; out[2 * oo + 1] = s->p_0T[1];
        oo++;
        mm_val = orc_branchless_clip(mm_val, lim);
        s->omega = s->omega + s->gain_omega * mm_val;
        s->omega = s->omega_mid + orc_branchless_clip(s->omega - s->omega_mid, s->omega_relative_limit);
        s->mu = s->mu + s->omega + s->gain_mu * mm_val;
        ii += (int)floor(s->mu);
        s->mu -= floor(s->mu);
        if (foptr) foptr[oo - 1] = mm_val;
        if (ii < 0) ii = 0;
    }
    if (consumed) *consumed = ii > 0 ? ii : 0;
    return oo;
}

/* ------------------------------------------------------------------ */
/* gr_framer_sink_1 (general/gr_framer_sink_1.cc:34-66 state entries,     */
/* 90-190 work; general/gr_framer_sink_1.h:62-98 state, header_ok,       */
/* header_payload).  Messages are appended to caller arrays instead of    */
/* a gr_msg_queue: msg_woff[i] (gr_message arg1), msg_len[i], payload     */
/* bytes back to back in `pool`.  Returns the number of messages added.   */
/* ------------------------------------------------------------------ */
typedef struct {
    int state;                       /* 0 search, 1 have sync, 2 have header */
    unsigned int header;
    int headerbitlen_cnt;
    unsigned char packet[4096];
    unsigned char packet_byte;
    int packet_byte_index;
    int packetlen;
    int packet_whitener_offset;
    int packetlen_cnt;
} orc_framer_state;

ORC_API size_t orc_framer_state_size(void) { return sizeof(orc_framer_state); }

ORC_API void orc_framer_sink_1_init(orc_framer_state *s)
{
    memset(s, 0, sizeof(*s));
    s->state = 0;                    /* enter_search() in the constructor, .cc:84 */
}

ORC_API int orc_framer_sink_1_work(orc_framer_state *s, const unsigned char *in, int noutput_items, int *msg_woff,
                                   int *msg_len, unsigned char *pool, size_t *pool_used)
{
    int count = 0, nmsg = 0;
    while (count < noutput_items) {
        switch (s->state) {
        case 0:
            while (count < noutput_items) {
                if (in[count] & 0x2) {               /* the flagged item is NOT consumed here */
                    s->state = 1; s->header = 0; s->headerbitlen_cnt = 0;
                    break;
                }
                count++;
            }
            break;
        case 1:
            while (count < noutput_items) {
                s->header = (s->header << 1) | (in[count++] & 0x1);
                if (++s->headerbitlen_cnt == 32) {
                    if ((((s->header >> 16) ^ (s->header & 0xffff)) == 0)) {
                        s->state = 2;
                        s->packetlen = (s->header >> 16) & 0x0fff;
                        s->packet_whitener_offset = (s->header >> 28) & 0x000f;
                        s->packetlen_cnt = 0; s->packet_byte = 0; s->packet_byte_index = 0;
                        if (s->packetlen == 0) {
                            msg_woff[nmsg] = s->packet_whitener_offset; msg_len[nmsg] = 0; nmsg++;
                            s->state = 0;
                        }
                    } else {
                        s->state = 0;
                    }
                    break;
                }
            }
            break;
        default:
            while (count < noutput_items) {
                s->packet_byte = (unsigned char)((s->packet_byte << 1) | (in[count++] & 0x1));
                if (s->packet_byte_index++ == 7) {
                    s->packet[s->packetlen_cnt++] = s->packet_byte;
                    s->packet_byte_index = 0;
                    if (s->packetlen_cnt == s->packetlen) {
                        msg_woff[nmsg] = s->packet_whitener_offset; msg_len[nmsg] = s->packetlen_cnt; nmsg++;
                        memcpy(pool + *pool_used, s->packet, (size_t)s->packetlen_cnt);
                        *pool_used += (size_t)s->packetlen_cnt;
                        s->state = 0;
                        break;
                    }
                }
            }
            break;
        }
    }
    return nmsg;
}

/* ------------------------------------------------------------------ */
/* gr_count_bits64 (general/gr_count_bits.cc:75-93)                     */
/* ------------------------------------------------------------------ */
static unsigned count_bits32(unsigned x)
{
    unsigned res = (x & 0x55555555) + ((x >> 1) & 0x55555555);
    res = (res & 0x33333333) + ((res >> 2) & 0x33333333);
    res = (res & 0x0F0F0F0F) + ((res >> 4) & 0x0F0F0F0F);
    res = (res & 0x00FF00FF) + ((res >> 8) & 0x00FF00FF);
    return (res & 0x0000FFFF) + ((res >> 16) & 0x0000FFFF);
}
ORC_API unsigned orc_count_bits64(unsigned long long x)
{
    return count_bits32((unsigned)((x >> 32) & 0xffffffff)) + count_bits32((unsigned)(x & 0xffffffff));
}

/* ------------------------------------------------------------------ */
/* digital_correlate_access_code_bb                                     */
/* gr-digital/lib/digital_correlate_access_code_bb.cc:45-133            */
/* ------------------------------------------------------------------ */
typedef struct {
    unsigned long long access_code, data_reg, flag_reg, flag_bit, mask;
    unsigned threshold;
} orc_corr;

/* set_access_code (:64-85); code is a string of '0'/'1' (LSB of each byte).
 * Returns 0 ok, -1 if longer than 64. */
ORC_API int orc_corr_set_access_code(orc_corr *c, const char *code, unsigned len)
{
    if (len > 64) return -1;
    if (len == 0) { c->mask = 0; c->flag_bit = 0; }   /* reference shifts by 64 (UB); defined as "no bits" here */
    else {
        c->mask = ((~0ULL) >> (64 - len)) << (64 - len);
        c->flag_bit = 1ULL << (64 - len);
    }
    c->access_code = 0;
    for (unsigned i = 0; i < 64; i++) {
        c->access_code <<= 1;
        if (i < len) c->access_code |= code[i] & 1;
    }
    return 0;
}

ORC_API int orc_corr_init(orc_corr *c, const char *code, unsigned len, int threshold)
{
    memset(c, 0, sizeof(*c));
    c->threshold = (unsigned)threshold;
    return orc_corr_set_access_code(c, code, len);
}

/* work (:95-130) */
ORC_API void orc_corr_work(orc_corr *c, const unsigned char *in, unsigned char *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        unsigned t = 0;
        t |= ((c->data_reg >> 63) & 0x1) << 0;
        t |= ((c->flag_reg >> 63) & 0x1) << 1;
        out[i] = (unsigned char)t;
        unsigned long long wrong_bits = (c->data_reg ^ c->access_code) & c->mask;
        unsigned nwrong = orc_count_bits64(wrong_bits);
        int new_flag = (nwrong <= c->threshold);
        c->data_reg = (c->data_reg << 1) | (in[i] & 0x1);
        c->flag_reg = (c->flag_reg << 1);
        if (new_flag) c->flag_reg |= c->flag_bit;
    }
}

/* ------------------------------------------------------------------ */
/* gr_fft_vcc_fftw::work (general/gr_fft_vcc_fftw.cc:55-103)            */
/* The transform itself is FFTW3f in the reference (third-party, not    */
/* under /root/reference; requirement "fftw3f >= 3.0",                   */
/* cmake/Modules/FindFFTW3f.cmake:7).  FFTW's codelet schedule is        */
/* planner-dependent and not restatable; the published definition        */
/* (unnormalised DFT, sign -1 forward / +1 backward) is computed here in */
/* double and rounded once.  Pinned by qa_fft.py 32-pt vectors.          */
/* ------------------------------------------------------------------ */
static void dft_double(const double *xr, const double *xi, double *yr, double *yi,
                       unsigned n, int forward)
{
    /* iterative radix-2 when n is a power of two, O(n^2) otherwise */
    if (n && !(n & (n - 1))) {
        unsigned lg = 0; while ((1u << lg) < n) lg++;
        for (unsigned i = 0; i < n; i++) {
            unsigned r = 0;
            for (unsigned b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
            yr[r] = xr[i]; yi[r] = xi[i];
        }
        double sgn = forward ? -1.0 : 1.0;
        for (unsigned len = 2; len <= n; len <<= 1) {
            unsigned half = len >> 1;
            for (unsigned s = 0; s < n; s += len)
                for (unsigned k = 0; k < half; k++) {
                    double ang = sgn * 2.0 * M_PI * (double)k / (double)len;
                    double wr = cos(ang), wi = sin(ang);
                    double ur = yr[s + k], ui = yi[s + k];
                    double vr = yr[s + k + half] * wr - yi[s + k + half] * wi;
                    double vi = yr[s + k + half] * wi + yi[s + k + half] * wr;
                    yr[s + k] = ur + vr; yi[s + k] = ui + vi;
                    yr[s + k + half] = ur - vr; yi[s + k + half] = ui - vi;
                }
        }
        return;
    }
    for (unsigned k = 0; k < n; k++) {
        double sr = 0, si = 0;
        for (unsigned t = 0; t < n; t++) {
            double ang = (forward ? -1.0 : 1.0) * 2.0 * M_PI * (double)((unsigned long long)k * t % n) / (double)n;
            double wr = cos(ang), wi = sin(ang);
            sr += xr[t] * wr - xi[t] * wi;
            si += xr[t] * wi + xi[t] * wr;
        }
        yr[k] = sr; yi[k] = si;
    }
}

/* window may be NULL (wlen 0).  nvec vectors of fft_size complex. */
ORC_API void orc_fft_vcc(unsigned fft_size, int forward, const float *window, unsigned wlen,
                         int shift, const float *in, float *out, size_t nvec)
{
    double *xr = (double *)malloc(sizeof(double) * fft_size * 4);
    double *xi = xr + fft_size, *yr = xi + fft_size, *yi = yr + fft_size;
    float *dst = (float *)malloc(sizeof(float) * 2 * fft_size);
    for (size_t v = 0; v < nvec; v++) {
        if (wlen) {                                         /* :68-71 in[i]*window[i] */
            for (unsigned i = 0; i < fft_size; i++) {
                dst[2 * i] = in[2 * i] * window[i];
                dst[2 * i + 1] = in[2 * i + 1] * window[i];
            }
        } else if (!forward && shift) {                     /* :74-79 */
            unsigned len = (unsigned)(floor(fft_size / 2.0));
            memcpy(&dst[0], &in[2 * len], sizeof(float) * 2 * (fft_size - len));
            memcpy(&dst[2 * (fft_size - len)], &in[0], sizeof(float) * 2 * len);
        } else {
            memcpy(dst, in, sizeof(float) * 2 * fft_size);  /* :81 */
        }
        for (unsigned i = 0; i < fft_size; i++) { xr[i] = dst[2 * i]; xi[i] = dst[2 * i + 1]; }
        dft_double(xr, xi, yr, yi, fft_size, forward);
        for (unsigned i = 0; i < fft_size; i++) { dst[2 * i] = (float)yr[i]; dst[2 * i + 1] = (float)yi[i]; }
        if (forward && shift) {                             /* :89-93 */
            unsigned len = (unsigned)(ceil(fft_size / 2.0));
            memcpy(&out[0], &dst[2 * len], sizeof(float) * 2 * (fft_size - len));
            memcpy(&out[2 * (fft_size - len)], &dst[0], sizeof(float) * 2 * len);
        } else {
            memcpy(out, dst, sizeof(float) * 2 * fft_size); /* :95 */
        }
        in += 2 * fft_size; out += 2 * fft_size;
    }
    free(xr); free(dst);
}

/* ------------------------------------------------------------------ */
/* gr_fft_filter_ccc / gri_fft_filter_ccc_generic          */
/* filter/gri_fft_filter_ccc_generic.cc:63-170: overlap-ADD fast          */
/* convolution, fftsize = 2 * 2^ceil(log2 ntaps), nsamples = fftsize -    */
/* ntaps + 1, taps pre-scaled by 1/fftsize, tail carried between blocks.  */
/* The transforms are FFTW in the reference (unpinned, see orc_fft_vcc):  */
/* here a double DFT rounded once to float at FFTW's output.              */
/* ------------------------------------------------------------------ */
typedef struct {
    int decim, ntaps, fftsize, nsamples;
    float *xformed;      /* 2*fftsize */
    float *tail;         /* 2*(ntaps-1) */
} orc_fftfilt;

static void fftfilt_xform(int n, int forward, const float *in, float *out)
{
    double *xr = (double *)calloc((size_t)n * 4, sizeof(double));
    double *xi = xr + n, *yr = xi + n, *yi = yr + n;
    for (int i = 0; i < n; i++) { xr[i] = in[2 * i]; xi[i] = in[2 * i + 1]; }
    dft_double(xr, xi, yr, yi, (unsigned)n, forward);
    for (int i = 0; i < n; i++) { out[2 * i] = (float)yr[i]; out[2 * i + 1] = (float)yi[i]; }
    free(xr);
}

ORC_API void orc_fft_filter_free(orc_fftfilt *f)
{
    if (!f) return;
    free(f->xformed); free(f->tail); free(f);
}

ORC_API orc_fftfilt *orc_fft_filter_new(int decimation, const float *taps, unsigned ntaps)
{
    if (decimation < 1 || ntaps < 1) return NULL;
    orc_fftfilt *f = (orc_fftfilt *)calloc(1, sizeof(*f));
    f->decim = decimation; f->ntaps = (int)ntaps;
    f->fftsize = (int)(2 * pow(2.0, ceil(log((double)ntaps) / log(2.0))));      /* :104 */
    f->nsamples = f->fftsize - f->ntaps + 1;                                     /* :105 */
    f->xformed = (float *)calloc(2 * (size_t)f->fftsize, sizeof(float));
    f->tail = (float *)calloc(2 * (size_t)(ntaps > 1 ? ntaps - 1 : 1), sizeof(float));
    float *in = (float *)calloc(2 * (size_t)f->fftsize, sizeof(float));
    float scale = 1.0 / f->fftsize;                                              /* :76 */
    for (unsigned i = 0; i < ntaps; i++) { in[2 * i] = taps[2 * i] * scale; in[2 * i + 1] = taps[2 * i + 1] * scale; }
    fftfilt_xform(f->fftsize, 1, in, f->xformed);
    free(in);
    return f;
}

ORC_API int orc_fft_filter_nsamples(const orc_fftfilt *f) { return f->nsamples; }

/* filter(): nitems outputs from nitems*decimation inputs (a multiple of nsamples) (:121-169) */
ORC_API int orc_fft_filter_filter(orc_fftfilt *f, int nitems, const float *input, float *output)
{
    int dec_ctr = 0, j;
    int ninput_items = nitems * f->decim;
    int tailsize = f->ntaps - 1;
    float *a = (float *)calloc(2 * (size_t)f->fftsize, sizeof(float));
    float *c = (float *)calloc(2 * (size_t)f->fftsize, sizeof(float));
    float *o = (float *)calloc(2 * (size_t)f->fftsize, sizeof(float));
    for (int i = 0; i < ninput_items; i += f->nsamples) {
        memcpy(a, &input[2 * (size_t)i], sizeof(float) * 2 * f->nsamples);
        for (j = f->nsamples; j < f->fftsize; j++) { a[2 * j] = 0; a[2 * j + 1] = 0; }
        fftfilt_xform(f->fftsize, 1, a, o);
        for (j = 0; j < f->fftsize; j++)                                        /* c[j] = a[j] * b[j] */
            cmul(o[2 * j], o[2 * j + 1], f->xformed[2 * j], f->xformed[2 * j + 1], &c[2 * j], &c[2 * j + 1]);
        fftfilt_xform(f->fftsize, 0, c, o);
        for (j = 0; j < tailsize; j++) { o[2 * j] += f->tail[2 * j]; o[2 * j + 1] += f->tail[2 * j + 1]; }
        j = dec_ctr;
        while (j < f->nsamples) {
            *output++ = o[2 * j]; *output++ = o[2 * j + 1];
            j += f->decim;
        }
        dec_ctr = (j - f->nsamples);
        memcpy(f->tail, o + 2 * (size_t)f->nsamples, sizeof(float) * 2 * tailsize);
    }
    free(a); free(c); free(o);
    return nitems;
}

/* ------------------------------------------------------------------ */
/* gr_pfb_channelizer_ccf (filter/gr_pfb_channelizer_ccf.cc:44-200)     */
/* ------------------------------------------------------------------ */
typedef struct {
    unsigned numchans, taps_per_filter;
    float oversample_rate;
    int rate_ratio, output_multiple;
    int *idxlut;
    float *ftaps;     /* [numchans][taps_per_filter] reversed (as gr_fir_ccf stores) */
} orc_pfb;

ORC_API void orc_pfb_free(orc_pfb *p) { if (p) { free(p->idxlut); free(p->ftaps); free(p); } }

/* returns NULL on invalid_argument (:55-60) */
ORC_API orc_pfb *orc_pfb_new(unsigned numchans, const float *taps, unsigned ntaps, float oversample_rate)
{
    double intp = 0;
    double fltp = modf(numchans / oversample_rate, &intp);
    if (fltp != 0.0) return NULL;
    orc_pfb *p = (orc_pfb *)calloc(1, sizeof(*p));
    p->numchans = numchans; p->oversample_rate = oversample_rate;
    /* set_taps (:104-139) */
    p->taps_per_filter = (unsigned)ceil((double)ntaps / (double)numchans);
    size_t tot = (size_t)numchans * p->taps_per_filter;
    float *tmp = (float *)calloc(tot ? tot : 1, sizeof(float));
    memcpy(tmp, taps, sizeof(float) * ntaps);
    p->ftaps = (float *)calloc(tot ? tot : 1, sizeof(float));
    for (unsigned i = 0; i < numchans; i++)
        for (unsigned j = 0; j < p->taps_per_filter; j++) {
            float t = tmp[i + j * numchans];                       /* d_taps[i][j] (:130) */
            p->ftaps[i * p->taps_per_filter + (p->taps_per_filter - 1 - j)] = t;   /* gr_fir reverses */
        }
    free(tmp);
    p->rate_ratio = (int)rintf(numchans / oversample_rate);        /* :82 */
    p->idxlut = (int *)malloc(sizeof(int) * numchans);
    for (unsigned i = 0; i < numchans; i++)
        p->idxlut[i] = numchans - ((i + p->rate_ratio) % numchans) - 1;   /* :85 */
    p->output_multiple = 1;
    while ((p->output_multiple * p->rate_ratio) % numchans != 0) p->output_multiple++;
    return p;
}

ORC_API unsigned orc_pfb_taps_per_filter(const orc_pfb *p) { return p->taps_per_filter; }
ORC_API int orc_pfb_output_multiple(const orc_pfb *p) { return p->output_multiple; }

/* general_work (:160-199).  ins[j] points at stream j INCLUDING history
 * (taps_per_filter+1 - 1 = taps_per_filter old items in front, :136).
 * out: noutput_items vectors of numchans complex.  Returns items consumed
 * per input stream.  The M-point backward DFT is FFTW in the reference;
 * here double DFT rounded once (see orc_fft_vcc note). */
ORC_API int orc_pfb_general_work(orc_pfb *p, int noutput_items, const float *const *ins, float *out)
{
    unsigned M = p->numchans, tpf = p->taps_per_filter;
    double *xr = (double *)malloc(sizeof(double) * M * 4);
    double *xi = xr + M, *yr = xi + M, *yi = yr + M;
    float *inbuf = (float *)calloc(2 * M, sizeof(float));
    int n = 1, i = -1, j = 0, last;
    int toconsume = (int)rintf(noutput_items / p->oversample_rate);
    while (n <= toconsume) {
        j = 0;
        i = (i + p->rate_ratio) % M;
        last = i;
        while (i >= 0) {
            const float *in = ins[j];
            fir_ccf_one(p->ftaps + (size_t)i * tpf, tpf, in + 2 * n, &inbuf[2 * p->idxlut[j]]);
            j++; i--;
        }
        i = M - 1;
        while (i > last) {
            const float *in = ins[j];
            fir_ccf_one(p->ftaps + (size_t)i * tpf, tpf, in + 2 * (n - 1), &inbuf[2 * p->idxlut[j]]);
            j++; i--;
        }
        n += (i + p->rate_ratio) >= (int)M;
        for (unsigned k = 0; k < M; k++) { xr[k] = inbuf[2 * k]; xi[k] = inbuf[2 * k + 1]; }
        dft_double(xr, xi, yr, yi, M, 0);
        for (unsigned k = 0; k < M; k++) { out[2 * k] = (float)yr[k]; out[2 * k + 1] = (float)yi[k]; }
        out += 2 * M;
    }
    free(xr); free(inbuf);
    return toconsume;
}

/* ------------------------------------------------------------------ */
/* gr_pfb_decimator_ccf (filter/gr_pfb_decimator_ccf.cc:77-111 set_taps, */
/* 130-180 work).  ins[s]: stream s with taps_per_filter-1 old items in   */
/* front.  The M-point backward FFT is FFTW in the reference; here the    */
/* double DFT rounded once (see orc_fft_vcc note): rounding unpinned.     */
/* ------------------------------------------------------------------ */
ORC_API unsigned orc_pfb_decimator_taps_per_filter(unsigned decim, unsigned ntaps)
{
    return (unsigned)ceil((double)ntaps / (double)decim);
}

ORC_API void orc_pfb_decimator_ccf_work(unsigned decim, const float *taps, unsigned ntaps, unsigned chan,
                                        const float *const *ins, float *out, int noutput_items)
{
    unsigned M = decim, tpf = orc_pfb_decimator_taps_per_filter(decim, ntaps);
    size_t tot = (size_t)M * tpf;
    float *tmp = (float *)calloc(tot ? tot : 1, sizeof(float));
    float *ft = (float *)calloc(tot ? tot : 1, sizeof(float));
    memcpy(tmp, taps, sizeof(float) * ntaps);
    for (unsigned i = 0; i < M; i++)
        for (unsigned j = 0; j < tpf; j++) ft[i * tpf + (tpf - 1 - j)] = tmp[i + j * M];   /* :99, gr_fir reverses */
    double *xr = (double *)malloc(sizeof(double) * M * 4);
    double *xi = xr + M, *yr = xi + M, *yi = yr + M;
    for (int i = 0; i < noutput_items; i++) {
        for (int j = (int)M - 1; j >= 0; j--) {                      /* :146-160 */
            float f[2];
            fir_ccf_one(ft + (size_t)j * tpf, tpf, ins[M - 1 - j] + 2 * (size_t)i, f);
            xr[j] = f[0]; xi[j] = f[1];
        }
        dft_double(xr, xi, yr, yi, M, 0);                            /* backward, unnormalised (:66, 167) */
        out[2 * i] = (float)yr[chan]; out[2 * i + 1] = (float)yi[chan];   /* :170 */
    }
    free(xr); free(tmp); free(ft);
}

/* ------------------------------------------------------------------ */
/* Chain drivers used by bench.py's cpu_baseline leg ("port") and by    */
/* chain-level parity tests.  They mimic what the scheduler does for a   */
/* whole capture: history zeros in front (runtime/gr_flat_flowgraph.cc   */
/* :150), then one big work() per block.                                */
/* ------------------------------------------------------------------ */

/* xlating(ccc, decim) -> quad_demod.  x: n_in complex (no history; zeros are
 * prepended here).  demod_out gets n_out = n_in/decim floats.  If y_out is
 * non-NULL it receives the xlating output (n_out complex). */
ORC_API size_t orc_chain_xlating_demod(unsigned decim, const float *proto, unsigned ntaps,
                                       double center_freq, double sampling_freq, float gain,
                                       const float *x, size_t n_in, float *y_out, float *demod_out)
{
    size_t n_out = n_in / decim;
    orc_xlating *xl = orc_xlating_ccc_new(decim, proto, ntaps, center_freq, sampling_freq);
    size_t hist = ntaps ? ntaps - 1 : 0;
    float *buf = (float *)calloc((n_in + hist) * 2 + 2, sizeof(float));
    memcpy(buf + 2 * hist, x, n_in * 2 * sizeof(float));
    float *y = (float *)calloc((n_out + 1) * 2, sizeof(float));   /* y[0] = history zero of quad_demod */
    orc_xlating_ccc_work(xl, buf, y + 2, n_out);
    orc_quad_demod_cf(gain, y, demod_out, n_out);
    if (y_out) memcpy(y_out, y + 2, n_out * 2 * sizeof(float));
    free(buf); free(y); orc_xlating_free(xl);
    return n_out;
}

/* M&M over a whole float stream the way the executor would feed it: history
 * is 1 (gr_block default) so no zeros are prepended; all input offered at
 * once.  Returns number of symbols. */
ORC_API int orc_chain_mm(float omega, float gain_omega, float mu, float gain_mu, float rel_limit,
                         const float *in, int n_in, float *out, int out_cap, float *final_state)
{
    orc_mm s;
    if (orc_mm_init(&s, omega, gain_omega, mu, gain_mu, rel_limit)) return -1;
    int consumed = 0;
    int n = orc_mm_general_work(&s, out_cap, n_in, in, out, &consumed);
    if (final_state) {
        final_state[0] = s.mu; final_state[1] = s.omega; final_state[2] = s.last_sample;
        final_state[3] = (float)consumed;
    }
    return n;
}/* ==========================================================================
 * gri_fir_filter_with_buffer_{ccf,ccc,fff}
 *   gnuradio-core/src/lib/filter/gri_fir_filter_with_buffer_XXX.cc.t:30-121
 * The filter owns its delay line: a buffer of 2*ntaps items written twice (d_idx and d_idx + ntaps,
 * .cc.t:65-66), zeroed by set_taps (.cc.t:55-57); one output per `dec` inputs is the dot product of the
 * reversed taps with the last ntaps inputs, accumulated one term after the other in the
 * accumulator type (.cc.t:75-79): ONE accumulator, sequential -- not gr_fir_XXX_generic's unrolled order.
 * kind: 0 fff, 1 ccf, 2 ccc.  The state lives in an opaque object so that calls continue each other.
 * ========================================================================== */
typedef struct {
    int kind;
    unsigned ntaps, idx;
    float *dt;      /* reversed taps (.cc.t:46), complex interleaved for ccc */
    float *buf;     /* 2 * ntaps items */
} orc_fwb;

ORC_API orc_fwb *orc_fwb_create(int kind, const float *taps_fwd, unsigned ntaps)
{
    orc_fwb *f = (orc_fwb *)calloc(1, sizeof(orc_fwb));
    const unsigned tw = kind == 2 ? 2 : 1, iw = kind == 0 ? 1 : 2;
    f->kind = kind; f->ntaps = ntaps; f->idx = 0;
    f->dt = reversed(taps_fwd, ntaps, tw);
    f->buf = (float *)calloc((size_t)2 * (ntaps ? ntaps : 1) * iw, sizeof(float));      /* memset 0, .cc.t:55-57 */
    return f;
}

ORC_API void orc_fwb_destroy(orc_fwb *f)
{
    if (!f) return;
    free(f->dt); free(f->buf); free(f);
}

/* filterNdec (.cc.t:110-121): output[i] = filter(&input[i * decimate], decimate) */
ORC_API void orc_fwb_filterNdec(orc_fwb *f, const float *in, float *out, size_t n, unsigned dec)
{
    const unsigned T = f->ntaps, iw = f->kind == 0 ? 1 : 2;
    for (size_t o = 0; o < n; ++o) {
        for (unsigned i = 0; i < dec; ++i) {                     /* .cc.t:88-94 */
            const float *x = in + (o * dec + i) * iw;
            if (T) {     /* (with no taps the reference writes into a zero-sized buffer; nothing is read back) */
                for (unsigned w = 0; w < iw; ++w) {
                    f->buf[(size_t)f->idx * iw + w] = x[w];
                    f->buf[(size_t)(f->idx + T) * iw + w] = x[w];
                }
            }
            f->idx++;
            if (f->idx >= T) f->idx = 0;
        }
        if (f->kind == 0) {
            float acc = 0;                                        /* .cc.t:96-99 */
            for (unsigned i = 0; i < T; ++i) acc += f->buf[f->idx + i] * f->dt[i];
            out[o] = acc;
        } else if (f->kind == 1) {
            float ar = 0, ai = 0;                                 /* complex<float> * float, then += */
            for (unsigned i = 0; i < T; ++i) {
                const float pr = f->buf[2 * (size_t)(f->idx + i)] * f->dt[i];
                const float pi = f->buf[2 * (size_t)(f->idx + i) + 1] * f->dt[i];
                ar += pr; ai += pi;
            }
            out[2 * o] = ar; out[2 * o + 1] = ai;
        } else {
            float ar = 0, ai = 0;
            for (unsigned i = 0; i < T; ++i) {
                float pr, pi;
                cmul(f->buf[2 * (size_t)(f->idx + i)], f->buf[2 * (size_t)(f->idx + i) + 1], f->dt[2 * i], f->dt[2 * i + 1],
                     &pr, &pi);
                ar += pr; ai += pi;
            }
            out[2 * o] = ar; out[2 * o + 1] = ai;
        }
    }
}


