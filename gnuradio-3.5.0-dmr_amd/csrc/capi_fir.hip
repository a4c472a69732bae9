// capi_fir.hip -- C ABI for gr_fir_filter_XXX, gr_freq_xlating_fir_filter_ccc,
// gr_quadrature_demod_cf and the fused xlating->demod hier block.
#include <cmath>
#include <complex>

#include "fft_kernels.h"
#include "fir_kernels.h"
#include "grhip_internal.h"
#include "mfma_tables.h"
#include "xlating_core.h"

using namespace grhip;
typedef std::complex<float> cf;

namespace grhip {

// ---- tap packing for the tiled kernel ---------------------------------------
// c[k] multiplies x[nD + k]; hp[p*Tq + q] = c[qD + p], zero padded.
static int pack_phase_major(const float *c, int T, int tw, int D, std::vector<float> &hp)
{
    int R = tiled_R();
    int per = (T + D - 1) / D;
    int Tq = ((per + R - 1) / R) * R;
    if (Tq == 0) Tq = R;
    hp.assign(((size_t)D * Tq + R) * tw, 0.f);     // + R taps: the kernel prefetches one iteration ahead
    for (int k = 0; k < T; ++k) {
        int p = k % D, q = k / D;
        for (int w = 0; w < tw; ++w) hp[((size_t)p * Tq + q) * tw + w] = c[(size_t)k * tw + w];
    }
    return Tq;
}

static int upload(DevBuf &b, const void *src, size_t bytes)
{
    int rc = b.reserve(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) GRHIP_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return GRHIP_OK;
}

// ---- operand tables of the matrix-core engine (mfma_tables.h) ---------------------
// real taps c[k] multiplying x[nD + k]; both alignment parities
// fwT0 != 0: freq_xlating's pre-mix form -- the table carries the correction band G of the reference's tap-angle
// quantisation behind A (mfma_tables.h); 0: plain real taps (gr_fir_ccf), G all zero
static int build_mfma_taps(const float *c, int T, int D, DevBuf (&dA)[2], int *kexp, float fwT0 = 0.f)
{
    const int KS = mf::ksteps_inst(D, T);
    *kexp = mf::tap_scale_exp(c, T);
    for (int off = 0; off < 2; ++off) {
        std::vector<uint16_t> A;
        mf::build_A(c, T, D, KS, off, *kexp, A);
        mf::build_G(c, T, D, KS, off, *kexp, fwT0, A);
        int rc = upload(dA[off], A.data(), A.size() * sizeof(uint16_t));
        if (rc) return rc;
    }
    return GRHIP_OK;
}

// ============================================================================
// XlatingCore
// ============================================================================
// the high-decimation direct kernel where it beats (or replaces) the overlap-save engine
static bool hidec_wanted(int decim, int ntaps, bool ctaps, bool have_ols)
{
    if (!hidec_supported(decim, ntaps)) return false;
#ifdef GRHIP_DIAG       // diagnostic builds only: GRHIP_HIDEC=0 / 1 = never / whenever supported (tools/bench_decim.py)
    static int knob = -1;
    if (knob < 0) { const char *e = getenv("GRHIP_HIDEC"); knob = e ? 2 + atoi(e) : 0; }
    if (knob == 2) return false;
    if (knob == 3) return true;
#endif
    if (!have_ols) return true;
    // MACs per input sample (complex taps count twice) against what the engine delivers for the shape
    // (profiles/r02_decim_engines.log, round 2's kernels on both sides): the direct kernel runs at roughly 400 Gsamples/s
    // at 8 MACs, 320-400 at 20, 250 at 40, 130-200 at 80; the engine at 330-400 where its inverse folds (decimation 8, 16),
    // 210-250 elsewhere up to ~500 taps, less for longer filters.
    const double w = (double)ntaps / decim * (ctaps ? 2.0 : 1.0);
    if (decim == 8 || decim == 16) return w <= 16.0;
    if (w <= 32.0) return true;
    return w <= 48.0 && ntaps <= 512 && decim >= 5;
}

// Taps per polyphase component above which the overlap-save engine takes over from the tiled vector kernel (single-stream
// calls).  Measured on MI355X (tools/bench_ols_crossover.py, profiles/r02_ols_crossover.log): the tiled kernel runs at
// about 15000 / (taps per phase) Gsamples/s with real taps and half that with complex taps, the engine (round 2:
// persistent, resident twiddles) at 230-290 whatever the filter.
static int ols_crossover(bool complex_taps, int decim)
{
#ifdef GRHIP_DIAG       // diagnostic builds only: GRHIP_OLS_MIN = threshold for both kinds (0 = engine wherever it can, 9999 = never)
    if (const char *e = getenv("GRHIP_OLS_MIN")) return atoi(e);
#endif
    // measured crossovers (taps per phase): real taps 48 / 60 / 28 at decimation 1 / 2 / 4, complex taps 40 / 40 / 26
    // (the engine's folded inverse makes it fastest at 4, the full-size inverse of decimation 1 comes next)
    if (complex_taps) return decim >= 4 ? 26 : 40;
    return decim >= 4 ? 28 : decim == 2 ? 60 : 48;
}

int XlatingCore::build(int device)
{
    // build_composite_fir (filter/gr_freq_xlating_fir_filter_XXX.cc.t:72-83)
    ntaps = (int)proto.size();
    ctaps.resize(ntaps);
    float fwT0 = 2 * M_PI * center_freq / sampling_freq;
    for (unsigned i = 0; i < (unsigned)ntaps; i++) {
        cf e = std::exp(cf(0, i * fwT0));
        // complex product in libgcc order (ac - bd, ad + bc)
        float a = proto[i].real(), b = proto[i].imag(), c = e.real(), d = e.imag();
        float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
        ctaps[i] = cf(ac - bd, ad + bc);
    }
    // d_r.set_phase_incr(exp(gr_complex(0, fwT0 * decimation()))) ; incr / abs(incr)
    cf inc = std::exp(cf(0, fwT0 * decim));
    float mag = hypotf(inc.real(), inc.imag());
    incr = cf(inc.real() / mag, inc.imag() / mag);
    omega = (double)fwT0;

    // device copies
    // generic kernel wants d_taps order; after reverse(reverse()) that is ctaps itself
    int rc = upload(d_taps_generic, ctaps.data(), sizeof(cf) * ntaps);
    if (rc) return rc;

    bool real_proto = true;
    for (auto &t : proto) if (t.imag() != 0.0f) real_proto = false;
    std::vector<float> hp;
    use_tiled = false; premix = false;
    if (ntaps > 0) {
        if (real_proto) {
            std::vector<float> pr(ntaps);
            for (int i = 0; i < ntaps; ++i) pr[i] = proto[i].real();
            Tq = pack_phase_major(pr.data(), ntaps, 1, decim, hp);
            if (tiled_supported(decim, Tq)) { use_tiled = true; premix = true; }
        }
        if (!use_tiled) {
            Tq = pack_phase_major((const float *)ctaps.data(), ntaps, 2, decim, hp);
            if (tiled_supported(decim, Tq)) use_tiled = true;
        }
    }
    if (use_tiled) {
        rc = upload(d_hp, hp.data(), hp.size() * sizeof(float));
        if (rc) return rc;
        if (premix) {
            // phasor tables of the pre-mix form, computed in double (see fir_tiled.hip)
            const int NT = tiled_NT();
            std::vector<cf> W(tiled_wtab_len()), V(NT + 1), S(tiled_stab_len());
            for (int v = -1; v < (int)W.size() - 1; ++v) {
                double ang = omega * (double)(v - decim);
                W[v + 1] = cf((float)cos(ang), (float)sin(ang));
            }
            for (int i = 0; i < (int)S.size(); ++i) {
                double ang = omega * (double)tiled_load_span() * (double)i;
                S[i] = cf((float)cos(ang), (float)sin(ang));
            }
            for (int j = 0; j < NT; ++j) {
                double ang = -omega * (double)j * (double)decim;
                V[j] = cf((float)cos(ang), (float)sin(ang));
            }
            V[NT] = cf((float)cos(omega * decim), (float)sin(omega * decim));
            rc = upload(d_stab, S.data(), S.size() * sizeof(cf));
            if (rc) return rc;
            rc = upload(d_wtab, W.data(), W.size() * sizeof(cf));
            if (rc) return rc;
            rc = upload(d_vtab, V.data(), V.size() * sizeof(cf));
            if (rc) return rc;
        }
    }
    use_mfma = false;
    if (real_proto && mfma_supported(decim, ntaps) && ntaps / decim >= 24) {
        // pre-mix form on the matrix cores: x'[u] = x[u] e^{jw(u - off)}, real prototype taps
        std::vector<float> pr(ntaps);
        for (int i = 0; i < ntaps; ++i) pr[i] = proto[i].real();
        rc = build_mfma_taps(pr.data(), ntaps, decim, d_mf_A, &mf_kexp, fwT0);
        if (rc) return rc;
        const int KS = mf::ksteps_inst(decim, ntaps);
        for (int off = 0; off < 2; ++off) {
            std::vector<cf> W(2 * mf::THREADS);         // 256 staging lanes (fir_mfma_kernel) or 512 (fir_mfma_rs_kernel)
            for (int t = 0; t < 2 * mf::THREADS; ++t) {
                double ang = omega * (double)(2 * t - off);
                W[t] = cf((float)cos(ang), (float)sin(ang));
            }
            rc = upload(d_mf_wlane[off], W.data(), W.size() * sizeof(cf));
            if (rc) return rc;
        }
        std::vector<cf> S(mf::rounds(decim, KS)), V(mf::NTC + 2);
        for (int i = 0; i < (int)S.size(); ++i) {
            double ang = omega * (double)mf::ROUND * (double)i;
            S[i] = cf((float)cos(ang), (float)sin(ang));
        }
        for (int j = 0; j < (int)V.size(); ++j) {
            double ang = -omega * (double)j * (double)decim;
            V[j] = cf((float)cos(ang), (float)sin(ang));
        }
        mf_wstep[0] = (float)cos(omega); mf_wstep[1] = (float)sin(omega);
        rc = upload(d_mf_stab, S.data(), S.size() * sizeof(cf));
        if (!rc) rc = upload(d_mf_vtab, V.data(), V.size() * sizeof(cf));
        if (rc) return rc;
        use_mfma = true;
    }
    use_ols = false;
    if (ntaps >= 48 && ntaps <= OLS_MAX_TAPS && (OLS_N - (ntaps - 1)) / decim >= 1) {
        // convolution taps h[k] = ctaps[ntaps-1-k] (the composite FIR is set with gr_reverse(ctaps), .cc.t:80)
        std::vector<float> h((size_t)ntaps * 2);
        for (int k = 0; k < ntaps; ++k) {
            h[2 * k] = ctaps[ntaps - 1 - k].real();
            h[2 * k + 1] = ctaps[ntaps - 1 - k].imag();
        }
        rc = ols_build(h.data(), ntaps, decim, d_ols_tw, d_ols_H, &ols_L, &ols_fold);
        if (rc) return rc;
        use_ols = true;
    }
    // single-stream calls only, batched launches stay tiled.  Not gr_fir_filter's crossover: behind the engine this block
    // still needs the rotator-table multiply and (fused form) the stand-alone demodulator as passes of their own, which
    // the tiled kernel does in its epilogue -- the round-1 crossover stays for real prototypes (the matrix-core engine
    // comes first there anyway); complex prototypes cost the tiled kernel twice the FMAs: measured 190 (tiled) against
    // 226 Gsamples/s (engine + rotator pass) at 64 taps per phase, D = 4
    prefer_ols = use_ols && (!use_tiled || ntaps / decim > (real_proto ? 120 : 56));
    // real prototype: pre-mix form, half the FMAs for one more multiply per staged sample -- pays from about 16
    // taps per polyphase component (tools/bench_decim.py)
    hidec_premix = real_proto && ntaps > 0 && ntaps / decim >= 16;
    use_hidec = !use_tiled && hidec_wanted(decim, ntaps, !hidec_premix, use_ols);
    if (for_demod && real_proto && ntaps > 0) {
        // a demodulating handle: whatever engine has the fused demodulator (see xlating_core.h)
        if (use_tiled && premix) prefer_ols = false;
        // (up to 1024 taps: beyond, the f32 direct sum of the pre-mixed products leaves the 1e-5 tolerance -- 1.04e-5 measured
        // at 1200 taps -- and the engine, with its host-side rotator phases, stays)
        else if (!use_tiled && ntaps <= 1024 && hidec_supported(decim, ntaps)) { hidec_premix = true; use_hidec = true; }
    }
    if (use_hidec) {
        std::vector<float> hp2;
        if (hidec_premix) {
            std::vector<float> pr(ntaps), et, vt;
            for (int i = 0; i < ntaps; ++i) pr[i] = proto[i].real();
            hidec_pad_taps(pr.data(), ntaps, 1, decim, hp2);
            hidec_premix_tables(omega, decim, et, vt);
            rc = upload(d_hidec_etab, et.data(), et.size() * sizeof(float));
            if (!rc) rc = upload(d_hidec_vtab, vt.data(), vt.size() * sizeof(float));
            if (rc) return rc;
        } else {
            hidec_pad_taps((const float *)ctaps.data(), ntaps, 2, decim, hp2);
        }
        rc = upload(d_hidec_taps, hp2.data(), hp2.size() * sizeof(float));
        if (rc) return rc;
    }
    reset();
    (void)device;
    return GRHIP_OK;
}

void XlatingCore::reset()
{
    pos = 0;
    if (tab_start != 0) { tab_start = 0; tab_len = 0; }
    if (tab_len == 0) { gen_phase = cf(1.f, 0.f); gen_counter = 0; }
    if (built_incr != incr) {   // a table generated for another increment is useless
        tab_start = 0; tab_len = 0; gen_phase = cf(1.f, 0.f); gen_counter = 0;
        built_incr = incr;
    }
}

// make sure d_rot covers outputs [pos, pos+n); *gtab = device pointer of phase(pos)
int XlatingCore::ensure_rot(long long n, const float2 **gtab, hipStream_t st)
{
    const long long CAP = 1ll << 25;
    // The table is extended in place behind what earlier launches read; only when its front is
    // rewritten (restart, re-anchor) may a launch still in flight on `st` be reading those entries.
    bool rewrites_front = false;
    if (pos < tab_start) { tab_start = 0; tab_len = 0; gen_phase = cf(1.f, 0.f); gen_counter = 0; rewrites_front = true; }
    long long tab_end = tab_start + tab_len;
    if (pos + n > tab_end) {
        if (pos == tab_end && tab_len > 0 && (pos + n - tab_start) > CAP) {
            tab_start = pos; tab_len = 0;   // re-anchor, generator state is already at pos
            rewrites_front = true;
        }
        if (tab_len == 0) rewrites_front = true;
        if (rewrites_front && d_rot.p) GRHIP_HIP(hipStreamSynchronize(st));
        long long need = pos + n - (tab_start + tab_len);
        std::vector<cf> fresh((size_t)need);
        // gr_rotator::rotate (filter/gr_rotator.h:40-50), exact float recurrence
        float pr = gen_phase.real(), pi = gen_phase.imag();
        const float ir = incr.real(), ii = incr.imag();
        unsigned cnt = gen_counter;
        // (runs of steps up to the next multiple of 512 without the counter test inside: the recurrence is a serial chain
        // of one multiply and one add per step, and this loop is what a streaming stand-alone xlating block waits for)
        for (long long i = 0; i < need;) {
            long long run = 512 - (long long)(cnt % 512);
            if (run > need - i) run = need - i;
            cf *dstp = fresh.data() + (size_t)i;
            for (long long r = 0; r < run; ++r) {
                dstp[r] = cf(pr, pi);
                const float ac = pr * ir, bd = pi * ii, ad = pr * ii, bc = pi * ir;
                pr = ac - bd; pi = ad + bc;
            }
            i += run;
            cnt += (unsigned)run;
            if ((cnt % 512) == 0) {
                float a = hypotf(pr, pi);
                pr = pr / a; pi = pi / a;
            }
        }
        gen_phase = cf(pr, pi); gen_counter = cnt;
        size_t new_items = (size_t)(tab_len + need);
        if (new_items * sizeof(cf) > d_rot.cap) {
            DevBuf nb;
            int rc = nb.reserve(new_items * sizeof(cf) * 2);
            if (rc) return rc;
            if (tab_len) GRHIP_HIP(hipMemcpy(nb.p, d_rot.p, (size_t)tab_len * sizeof(cf), hipMemcpyDeviceToDevice));
            d_rot.release();
            d_rot = nb;
        }
        GRHIP_HIP(hipMemcpy(d_rot.as<cf>() + tab_len, fresh.data(), (size_t)need * sizeof(cf),
                            hipMemcpyHostToDevice));
        tab_len += need;
    }
    *gtab = reinterpret_cast<const float2 *>(d_rot.as<cf>() + (pos - tab_start));
    return GRHIP_OK;
}

// rotator phase of output pos-1 (for converting the demodulator carry between the
// frames of the two epilogues at a mode switch): phase(pos) * conj(incr)
int XlatingCore::phase_before_pos(std::complex<float> *g)
{
    const float2 *gtab = nullptr;
    int rc = ensure_rot(1, &gtab, nullptr);
    if (rc) return rc;
    cf ph;
    GRHIP_HIP(hipMemcpy(&ph, gtab, sizeof(ph), hipMemcpyDeviceToHost));
    *g = ph * std::conj(incr);
    return GRHIP_OK;
}

void XlatingCore::release()
{
    d_taps_generic.release(); d_hp.release(); d_wtab.release(); d_stab.release(); d_vtab.release(); d_rot.release();
    for (int i = 0; i < 2; ++i) { d_mf_A[i].release(); d_mf_wlane[i].release(); }
    d_mf_stab.release(); d_mf_vtab.release(); mf_sched.release();
    scratch_y.release(); sched.release(); d_ols_tw.release(); d_ols_H.release(); d_hidec_taps.release();
    d_hidec_etab.release(); d_hidec_vtab.release();
}

// run the FIR + rotator (+ demod) for n_out outputs on device pointers.
// d_in item 0 = input[0] of output 0; n_in readable items.
int XlatingCore::run(int mode, const float2 *d_in, long long n_in, long long n_out, float2 *d_y,
                     float *d_demod, float gain, const float2 *y_prev, float2 *y_last,
                     const float *atan_tab, hipStream_t st, int n_streams, long long x_stride,
                     long long n_lo, long long out_stride)
{
    if (n_out <= 0) return GRHIP_OK;
    const float2 *gtab = nullptr;
    const bool demod = d_demod != nullptr;
    const bool batched = !(n_streams == 1 && n_lo == 0);
    const bool mfma_now = mode_matrix(mode) && use_mfma && (n_streams == 1 || !(x_stride & 1));
    // fused demodulator of the high-decimation direct kernel (pre-mix form): no rotator phases either
    const bool hidec_direct = !mfma_now && demod && mode_fast(mode) && use_hidec && hidec_premix;
    const bool direct = mfma_now ? demod
                                 : (hidec_direct || (demod && mode_fast(mode) && use_tiled && premix && (batched || !prefer_ols)));
    int rc = GRHIP_OK;
    if (!direct) {          // the direct demodulator epilogue needs no rotator phases
        rc = ensure_rot(n_out, &gtab, st);
        if (rc) return rc;
    }
    if (mfma_now) {
        FirMfmaArgs a;
        memset(&a, 0, sizeof(a));
        a.x = d_in; a.x_stride = x_stride; a.n_in = n_in; a.n_lo = n_lo; a.n_out = n_out; a.n_streams = n_streams;
        a.off = (int)((((uintptr_t)d_in) >> 3) & 1);
        a.A = d_mf_A[a.off].p; a.kexp = mf_kexp;
        a.wlane = d_mf_wlane[a.off].as<float2>(); a.wstep = make_float2(mf_wstep[0], mf_wstep[1]);
        a.stab = d_mf_stab.as<float>(); a.vtab = d_mf_vtab.as<float2>(); a.gtab = gtab;
        a.y_out = d_y; a.d_out = d_demod; a.y_stride = out_stride; a.d_stride = out_stride;
        a.gain = gain; a.y_prev = y_prev; a.y_last = y_last; a.atan_tab = atan_tab;
        a.ctaps = d_taps_generic.as<float2>(); a.T = ntaps;
        const uintptr_t o = demod ? (uintptr_t)d_demod : (uintptr_t)d_y;
        a.vec_store = demod ? ((o & 7) == 0 && !(out_stride & 1)) : ((o & 15) == 0 && !(out_stride & 1));
        a.sched = mf_sched.get();
        a.max_wg_per_cu = mf_wg_cap;
        a.max_cus = mf_cu_cap;
        a.omega = omega;
        a.tapq = mode == GRHIP_MODE_FAST_REFTAPS;
        rc = launch_fir_mfma(decim, ntaps, true, demod ? EPI_DEMOD : EPI_ROTATE, a, st);
        if (rc) return rc;
        pos += n_out;
        return GRHIP_OK;
    }
    if (hidec_direct) {
        rc = launch_fir_hidec_demod(d_hidec_taps.as<float>(), ntaps, decim, d_in, batched ? n_in : (n_out - 1) * decim + ntaps,
                                    d_demod, n_out, gain, y_prev, y_last, atan_tab, d_hidec_etab.as<float2>(),
                                    d_hidec_vtab.as<float2>(), st, n_streams, x_stride, out_stride, n_lo, mf_wg_cap, mf_cu_cap);
        if (rc) return rc;
        pos += n_out;
        return GRHIP_OK;
    }
    const bool ols_now = mode_fast(mode) && (prefer_ols || (use_hidec && !use_tiled)) && !batched;
    if (mode_fast(mode) && use_tiled && !ols_now) {
        FirTiledArgs a;
        memset(&a, 0, sizeof(a));
        a.x = d_in; a.x_stride = x_stride; a.n_in = n_in; a.n_lo = n_lo;
        a.hp = d_hp.as<float>(); a.Tq = Tq; a.n_out = n_out;
        a.wtab = d_wtab.as<float2>(); a.stab = d_stab.as<float2>(); a.vtab = d_vtab.as<float2>(); a.gtab = gtab;
        a.gain = gain; a.atan_tab = atan_tab;
        a.sched = sched.get();
        if (direct || !demod) {
            a.y_out = d_y; a.d_out = d_demod;
            a.y_stride = out_stride; a.d_stride = out_stride;
            a.y_prev = y_prev; a.y_last = y_last;
            uintptr_t o = demod ? (uintptr_t)d_demod : (uintptr_t)d_y;
            a.vec_store = (o & 15) == 0 && ((out_stride * (demod ? 4 : 8)) & 15) == 0;
            rc = launch_fir_tiled(decim, !premix, premix, direct ? EPI_DEMOD : EPI_ROTATE, a, n_streams, st);
            if (rc) return rc;
        } else {
            // complex prototype taps: the tiled kernel writes y (with one slot in front of every
            // stream for the demodulator's previous sample), the demodulator is a second kernel
            const long long ys = n_out + 1;
            rc = scratch_y.reserve((size_t)ys * n_streams * sizeof(float2));
            if (rc) return rc;
            float2 *sy = scratch_y.as<float2>();
            if (y_prev) GRHIP_HIP(hipMemcpy2DAsync(sy, ys * sizeof(float2), y_prev, sizeof(float2), sizeof(float2),
                                                   n_streams, hipMemcpyDeviceToDevice, st));
            else GRHIP_HIP(hipMemset2DAsync(sy, ys * sizeof(float2), 0, sizeof(float2), n_streams, st));
            a.y_out = sy + 1; a.y_stride = ys;
            a.vec_store = 0;
            rc = launch_fir_tiled(decim, true, false, EPI_ROTATE, a, n_streams, st);
            if (rc) return rc;
            for (int s = 0; s < n_streams; ++s) {
                rc = launch_quad_demod(sy + s * ys, d_demod + s * out_stride, n_out, gain, atan_tab, st);
                if (rc) return rc;
            }
            if (y_last) GRHIP_HIP(hipMemcpy2DAsync(y_last, sizeof(float2), sy + n_out, ys * sizeof(float2), sizeof(float2),
                                                   n_streams, hipMemcpyDeviceToDevice, st));
        }
    } else if (ols_now) {
        // overlap-save FIR (history in front of d_in = the engine's previous samples), rotator table
        // multiply, stand-alone demodulator
        float2 *y = d_y;
        if (demod) {
            rc = scratch_y.reserve((size_t)(n_out + 1) * sizeof(float2));
            if (rc) return rc;
            y = scratch_y.as<float2>() + 1;
            GRHIP_HIP(hipMemcpyAsync(scratch_y.p, y_prev, sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
        if (use_hidec) {
            rc = launch_fir_hidec(!hidec_premix, d_hidec_taps.as<float>(), ntaps, decim, d_in, (n_out - 1) * decim + ntaps, y,
                                  n_out, gtab, st,           // rotator multiply inside
                                  hidec_premix ? d_hidec_etab.as<float2>() : nullptr,
                                  hidec_premix ? d_hidec_vtab.as<float2>() : nullptr);
        } else {
            rc = launch_fftfilt4096(d_in + (ntaps - 1), (n_out - 1) * decim + 1, d_in, ntaps, d_ols_tw.as<float2>(),
                                    d_ols_H.as<float2>(), y, n_out, decim, ols_L, ols_fold, st);
            if (!rc) rc = launch_rotate(y, gtab, n_out, st);
        }
        if (rc) return rc;
        if (demod) {
            rc = launch_quad_demod(scratch_y.as<float2>(), d_demod, n_out, gain, atan_tab, st);
            if (rc) return rc;
            GRHIP_HIP(hipMemcpyAsync(y_last, y + (n_out - 1), sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
    } else {
        // generic order (bit-exact) FIR + rotate; the demodulator in the same kernel where there is one (that kernel also
        // takes several streams and synthesises a fresh capture's history), else as a second kernel over y
        if (demod) {
            rc = launch_fir_generic_demod(d_taps_generic.as<float>(), ntaps, d_in, d_demod, n_out, decim, gtab, gain, atan_tab,
                                          y_prev, y_last, st, n_streams, x_stride, out_stride, n_lo);
            if (rc == GRHIP_OK) { pos += n_out; return GRHIP_OK; }
            if (rc != 1) return rc;
        }
        if (n_streams != 1 || n_lo != 0)
            return fail(GRHIP_EINVAL, "generic-order path: this shape runs one stream with explicit history");
        float2 *y = d_y;
        if (demod) {
            rc = scratch_y.reserve((size_t)(n_out + 1) * sizeof(float2));
            if (rc) return rc;
            y = scratch_y.as<float2>() + 1;
            GRHIP_HIP(hipMemcpyAsync(scratch_y.p, y_prev, sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
        rc = launch_fir_generic(FIR_CCC, d_taps_generic.as<float>(), ntaps, d_in, y, n_out, decim, gtab, st);
        if (rc) return rc;
        if (demod) {
            rc = launch_quad_demod(scratch_y.as<float2>(), d_demod, n_out, gain, atan_tab, st);
            if (rc) return rc;
            GRHIP_HIP(hipMemcpyAsync(y_last, y + (n_out - 1), sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
    }
    pos += n_out;
    return GRHIP_OK;
}

}  // namespace grhip

// ============================================================================
// gr_fir_filter_XXX
// ============================================================================
struct grhip_fir_filter : HandleBase {
    FirKind kind;
    int decim = 1;
    int mode = GRHIP_MODE_FAST;
    std::vector<float> taps;        // current forward taps (floats; x2 for ccc)
    std::vector<float> new_taps;    // latched by set_taps
    bool updated = false;
    int ntaps = 0;
    DevBuf d_taps_rev, d_hp;
    SchedBuf sched;
    int Tq = 0;
    bool use_tiled = false;
    // FAST mode for the shapes the tiled kernel does not take (decimation other than 1/2/4, more
    // than 1024 taps): the overlap-save engine of gr_fft_filter_ccc (fft_kernels.hip), any decimation
    bool use_ols = false, prefer_ols = false;
    int ols_L = 0, ols_fold = 0;
    DevBuf d_ols_tw, d_ols_H;
    // ... and the high-decimation direct kernel where a polyphase component has few taps (fir_kernels.hip)
    bool use_hidec = false;
    DevBuf d_hidec_taps;
    // ... and the matrix-core engine for long real-tap filters on complex data (fir_mfma.hip)
    bool use_mfma = false;
    int mf_kexp = 0;
    DevBuf d_mf_A[2];
    SchedBuf mf_sched;

    int tw() const { return kind == FIR_CCC ? 2 : 1; }
    size_t in_item() const { return kind == FIR_FFF ? 4 : 8; }
    size_t out_item() const { return kind == FIR_FFF ? 4 : 8; }

    int install(const std::vector<float> &t)
    {
        taps = t;
        ntaps = (int)(taps.size() / tw());
        std::vector<float> rev(taps.size());
        for (int k = 0; k < ntaps; ++k)
            for (int w = 0; w < tw(); ++w) rev[(size_t)k * tw() + w] = taps[(size_t)(ntaps - 1 - k) * tw() + w];
        int rc = upload(d_taps_rev, rev.data(), rev.size() * sizeof(float));
        if (rc) return rc;
        use_tiled = false;
        if (ntaps > 0 && (kind != FIR_FFF || decim <= 2)) {
            // float data runs the complex kernel on overlapped pairs at twice the decimation
            const int dk = kind == FIR_FFF ? 2 * decim : decim;
            std::vector<float> hp;
            Tq = pack_phase_major(rev.data(), ntaps, tw(), dk, hp);
            if (tiled_supported(dk, Tq)) {
                rc = upload(d_hp, hp.data(), hp.size() * sizeof(float));
                if (rc) return rc;
                use_tiled = true;
            }
        }
        use_mfma = false;
        if (kind == FIR_CCF && mfma_supported(decim, ntaps) && ntaps / decim >= 24) {
            rc = build_mfma_taps(rev.data(), ntaps, decim, d_mf_A, &mf_kexp);
            if (rc) return rc;
            use_mfma = true;
        }
        use_ols = false;
        if (ntaps >= 48 && ntaps <= OLS_MAX_TAPS && (OLS_N - (ntaps - 1)) / decim >= 1) {
            std::vector<float> ct((size_t)ntaps * 2);
            for (int k = 0; k < ntaps; ++k) {
                ct[2 * k] = kind == FIR_CCC ? taps[2 * k] : taps[k];
                ct[2 * k + 1] = kind == FIR_CCC ? taps[2 * k + 1] : 0.f;
            }
            rc = ols_build(ct.data(), ntaps, decim, d_ols_tw, d_ols_H, &ols_L, &ols_fold);
            if (rc) return rc;
            use_ols = true;
        }
        // (crossover between the tiled kernel and the overlap-save engine: ols_crossover above)
        use_hidec = !use_tiled && kind != FIR_FFF && hidec_wanted(decim, ntaps, kind == FIR_CCC, use_ols);
        if (use_hidec) {
            std::vector<float> hp2;
            hidec_pad_taps(rev.data(), ntaps, tw(), decim, hp2);
            rc = upload(d_hidec_taps, hp2.data(), hp2.size() * sizeof(float));
            if (rc) return rc;
        }
        // (float data: the tiled kernel's float-pair mode runs at about 50000 / taps Gsamples/s, the real-data engine at
        // 250-290: the engine where the float-pair mode does not reach, and from 176 taps per phase on)
        prefer_ols = use_ols && (!use_tiled || ntaps / decim > (kind == FIR_FFF ? 176 : ols_crossover(kind == FIR_CCC, decim)));
        return GRHIP_OK;
    }

    int run(const void *d_in, void *d_out, long long n, int dec, hipStream_t st)
    {
        if (n <= 0) return GRHIP_OK;
        if (mode_matrix(mode) && use_mfma && dec == decim) {
            FirMfmaArgs a;
            memset(&a, 0, sizeof(a));
            a.x = (const float2 *)d_in; a.n_in = (n - 1) * dec + ntaps; a.n_out = n; a.n_streams = 1;
            a.off = (int)((((uintptr_t)d_in) >> 3) & 1);
            a.A = d_mf_A[a.off].p; a.kexp = mf_kexp;
            a.y_out = (float2 *)d_out;
            a.vec_store = (((uintptr_t)d_out) & 15) == 0;
            a.sched = mf_sched.get();
            return launch_fir_mfma(dec, ntaps, false, EPI_NONE, a, st);
        }
        if (mode_fast(mode) && use_tiled && !prefer_ols && dec == decim && (kind != FIR_FFF || n >= 2)) {
            FirTiledArgs a;
            memset(&a, 0, sizeof(a));
            a.x = (const float2 *)d_in; a.n_in = (n - 1) * dec + ntaps;
            a.hp = d_hp.as<float>(); a.Tq = Tq; a.n_out = n;
            a.y_out = (float2 *)d_out;
            a.vec_store = (((uintptr_t)d_out) & 15) == 0;
            a.sched = sched.get();
            if (kind != FIR_FFF) return launch_fir_tiled(dec, kind == FIR_CCC, false, EPI_NONE, a, 1, st);
            // gr_fir_fff: output pairs (y[2j], y[2j+1]); an odd last output goes through the generic kernel
            a.fpair = dec;
            a.n_out = n / 2;
            int rc = launch_fir_tiled(2 * dec, false, false, EPI_NONE, a, 1, st);
            if (rc || !(n & 1)) return rc;
            return launch_fir_generic(kind, d_taps_rev.as<float>(), ntaps, (const float *)d_in + (n - 1) * dec,
                                      (float *)d_out + (n - 1), 1, dec, nullptr, st);
        }
        if (mode_fast(mode) && use_hidec && dec == decim)
            return launch_fir_hidec(kind == FIR_CCC, d_hidec_taps.as<float>(), ntaps, dec, (const float2 *)d_in,
                                    (n - 1) * dec + ntaps, (float2 *)d_out, n, nullptr, st);
        if (mode_fast(mode) && use_ols && (prefer_ols || !use_tiled) && dec == decim) {
            // y[n] = sum_k taps[k] x[nD + ntaps-1-k]: the ntaps-1 history items in front of d_in are the
            // engine's "previous call" samples, the rest is the stream
            if (kind == FIR_FFF) {
                const float *xf = (const float *)d_in;
                return launch_fftfilt4096_real(xf + (ntaps - 1), (n - 1) * dec + 1, xf, ntaps, d_ols_tw.as<float2>(),
                                               d_ols_H.as<float2>(), (float *)d_out, n, dec, ols_L, ols_fold, st);
            }
            const float2 *x = (const float2 *)d_in;
            // (the scheduler guarantees (n-1)*dec + ntaps items: nothing past the last needed sample is read)
            return launch_fftfilt4096(x + (ntaps - 1), (n - 1) * dec + 1, x, ntaps, d_ols_tw.as<float2>(),
                                      d_ols_H.as<float2>(), (float2 *)d_out, n, dec, ols_L, ols_fold, st);
        }
        return launch_fir_generic(kind, d_taps_rev.as<float>(), ntaps, d_in, d_out, n, dec, nullptr, st);
    }
};

extern "C" {

int grhip_fir_filter_create(grhip_fir_filter **h, const char *kind, int decimation, const float *taps,
                            size_t ntaps, int device)
{
    if (!h || !kind) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    FirKind k;
    if (!strcmp(kind, "ccf")) k = FIR_CCF;
    else if (!strcmp(kind, "fff")) k = FIR_FFF;
    else if (!strcmp(kind, "ccc")) k = FIR_CCC;
    else return fail(GRHIP_EINVAL, "unknown FIR kind '%s'", kind);
    if (decimation < 1) return fail(GRHIP_EINVAL, "decimation must be >= 1");
    if (ntaps && !taps) return fail(GRHIP_EINVAL, "taps is NULL");
    grhip_fir_filter *f = new (std::nothrow) grhip_fir_filter();
    if (!f) return fail(GRHIP_ENOMEM, "alloc");
    f->kind = k; f->decim = decimation; f->mode = default_mode();
    int rc = f->init_device(device);
    if (!rc) rc = f->install(std::vector<float>(taps, taps + ntaps * f->tw()));
    if (rc) { f->d_taps_rev.release(); f->d_hp.release(); f->sched.release(); f->d_ols_tw.release(); f->d_ols_H.release(); f->d_hidec_taps.release(); f->d_mf_A[0].release(); f->d_mf_A[1].release(); f->mf_sched.release(); f->destroy_base(); delete f; return rc; }
    *h = f;
    return GRHIP_OK;
}

void grhip_fir_filter_destroy(grhip_fir_filter *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->d_taps_rev.release(); h->d_hp.release(); h->sched.release(); h->d_ols_tw.release(); h->d_ols_H.release(); h->d_hidec_taps.release();
    h->d_mf_A[0].release(); h->d_mf_A[1].release(); h->mf_sched.release();
    h->destroy_base();
    delete h;
}

int grhip_fir_filter_set_taps(grhip_fir_filter *h, const float *taps, size_t ntaps)
{
    if (!h || (ntaps && !taps)) return fail(GRHIP_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    h->new_taps.assign(taps, taps + ntaps * h->tw());   // d_new_taps (.cc.t:59-64)
    h->updated = true;
    return GRHIP_OK;
}

int grhip_fir_filter_set_mode(grhip_fir_filter *h, int mode)
{
    if (!h || !mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode");
    h->mode = mode;
    return GRHIP_OK;
}

int grhip_fir_filter_history(const grhip_fir_filter *h)
{
    // set_history(d_fir->ntaps()) (.cc.t:51,77); gr_block history is at least 1
    return h ? (h->ntaps > 0 ? h->ntaps : 1) : GRHIP_EINVAL;
}

int grhip_fir_filter_decimation(const grhip_fir_filter *h) { return h ? h->decim : GRHIP_EINVAL; }

// `st`: the stream this work call will use.  install() rewrites the tap buffers with blocking copies on
// the null stream, which do not wait for the handle's non-blocking streams: an earlier *_work_device
// launch may still be reading them, so the streams are drained first.
static int fir_apply_update(grhip_fir_filter *h, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    if (!h->updated) return 0;
    GRHIP_HIP(hipStreamSynchronize(st));
    if (st != h->own_stream) GRHIP_HIP(hipStreamSynchronize(h->own_stream));
    int rc = h->install(h->new_taps);
    if (rc) return rc;
    h->updated = false;
    return 1;
}

int grhip_fir_filter_work_device(grhip_fir_filter *h, int noutput_items, const void *d_in, void *d_out,
                                 void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = fir_apply_update(h, h->pick(stream));
    if (rc < 0) return rc;
    if (rc == 1) return 0;   // history requirements may have changed (.cc.t:74-79)
    rc = h->run(d_in, d_out, noutput_items, h->decim, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_fir_filter_work(grhip_fir_filter *h, int noutput_items, const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = fir_apply_update(h, h->own_stream);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
    if (noutput_items == 0) return 0;
    long long n = noutput_items;
    size_t n_in = (size_t)((n - 1) * h->decim + (h->ntaps > 0 ? h->ntaps : 0));
    if (n_in == 0) n_in = 1;
    if ((rc = h->stage_in.reserve(n_in * h->in_item() + 16))) return rc;
    if ((rc = h->stage_out.reserve((size_t)n * h->out_item()))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n_in * h->in_item(), st);
    if ((rc = h->run(h->stage_in.p, h->stage_out.p, n, h->decim, st))) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)n * h->out_item(), st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

int grhip_fir_filterNdec(grhip_fir_filter *h, void *output, const void *input, unsigned long n,
                         unsigned decimate)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (decimate < 1) return fail(GRHIP_EINVAL, "decimate must be >= 1");
    int rc = h->bind();
    if (rc) return rc;
    if (n == 0) return GRHIP_OK;
    size_t n_in = (size_t)((n - 1) * decimate + h->ntaps);
    if (n_in == 0) n_in = 1;
    if ((rc = h->stage_in.reserve(n_in * h->in_item() + 16))) return rc;
    if ((rc = h->stage_out.reserve((size_t)n * h->out_item()))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, input, n_in * h->in_item(), st);
    if ((rc = h->run(h->stage_in.p, h->stage_out.p, (long long)n, (int)decimate, st))) return rc;
    GRHIP_D2H(h, output, h->stage_out.p, (size_t)n * h->out_item(), st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return GRHIP_OK;
}

}  // extern "C"

// ============================================================================
// gri_fir_filter_with_buffer_{ccf,ccc,fff}  (second half)
//   filter/gri_fir_filter_with_buffer_XXX.h.t:44-126, .cc.t:30-121
// The reference object owns its delay line (2 * ntaps items, zeroed by set_taps) and is called with NEW
// samples only: filterNdec(out, in, n, dec) consumes n * dec items.  Here the delay line is the last
// ntaps - 1 items, kept in HBM; a call lays [delay line | new items] out once and runs an engine of the FIR
// family over it: out[o] = sum_k rev[k] * both[(dec - 1) + dec * o + k].  GENERIC mode uses the
// reference's own accumulation order (one accumulator, term after term): bit-exact.
// ============================================================================
struct grhip_fir_filter_with_buffer : HandleBase {
    FirKind kind = FIR_CCF;
    int mode = GRHIP_MODE_FAST;
    std::vector<float> taps;            // forward taps (x2 floats for ccc)
    int ntaps = 0;
    grhip_fir_filter *inner = nullptr;  // engines of the FIR family at the decimation last used
    unsigned inner_dec = 0;
    DevBuf d_hist, d_both;
    size_t item() const { return kind == FIR_FFF ? 4 : 8; }
    int tw() const { return kind == FIR_CCC ? 2 : 1; }
    const char *kind_name() const { return kind == FIR_FFF ? "fff" : kind == FIR_CCF ? "ccf" : "ccc"; }
    int ensure_inner(unsigned dec)
    {
        if (inner && inner_dec == dec) return GRHIP_OK;
        if (inner) grhip_fir_filter_destroy(inner);
        inner = nullptr;
        int rc = grhip_fir_filter_create(&inner, kind_name(), (int)dec, taps.data(), (size_t)ntaps, device);
        if (rc) return rc;
        inner_dec = dec;
        return GRHIP_OK;
    }
    int reset_line()
    {
        const size_t bytes = (size_t)(ntaps > 1 ? ntaps - 1 : 1) * item();
        int rc = d_hist.reserve(bytes);
        if (rc) return rc;
        GRHIP_HIP(hipMemset(d_hist.p, 0, bytes));                       // memset(d_buffer, 0), .cc.t:55-57
        return GRHIP_OK;
    }
};

extern "C" {

int grhip_fir_filter_with_buffer_create(grhip_fir_filter_with_buffer **h, const char *kind, const float *taps,
                                        size_t ntaps, int device)
{
    if (!h || !kind) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    FirKind k;
    if (!strcmp(kind, "ccf")) k = FIR_CCF;
    else if (!strcmp(kind, "fff")) k = FIR_FFF;
    else if (!strcmp(kind, "ccc")) k = FIR_CCC;
    else return fail(GRHIP_EINVAL, "unknown FIR kind '%s'", kind);
    if (ntaps && !taps) return fail(GRHIP_EINVAL, "taps is NULL");
    auto *f = new (std::nothrow) grhip_fir_filter_with_buffer();
    if (!f) return fail(GRHIP_ENOMEM, "alloc");
    f->kind = k; f->mode = default_mode();
    int rc = f->init_device(device);
    if (!rc) {
        f->taps.assign(taps, taps + ntaps * f->tw());
        f->ntaps = (int)ntaps;
        rc = f->reset_line();
    }
    if (!rc) rc = f->ensure_inner(1);
    if (rc) { grhip_fir_filter_with_buffer_destroy(f); return rc; }
    *h = f;
    return GRHIP_OK;
}

void grhip_fir_filter_with_buffer_destroy(grhip_fir_filter_with_buffer *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->inner) grhip_fir_filter_destroy(h->inner);
    h->d_hist.release(); h->d_both.release();
    h->destroy_base();
    delete h;
}

// set_taps (.cc.t:44-59): new taps, delay line zeroed, takes effect at once (this is a kernel-level object,
// not a block: there is no "returns 0 once")
int grhip_fir_filter_with_buffer_set_taps(grhip_fir_filter_with_buffer *h, const float *taps, size_t ntaps)
{
    if (!h || (ntaps && !taps)) return fail(GRHIP_EINVAL, "null argument");
    int rc = h->bind();
    if (rc) return rc;
    GRHIP_HIP(hipStreamSynchronize(h->own_stream));
    h->taps.assign(taps, taps + ntaps * h->tw());
    h->ntaps = (int)ntaps;
    if (h->inner) { grhip_fir_filter_destroy(h->inner); h->inner = nullptr; h->inner_dec = 0; }
    rc = h->reset_line();
    if (!rc) rc = h->ensure_inner(1);
    return rc;
}

int grhip_fir_filter_with_buffer_set_mode(grhip_fir_filter_with_buffer *h, int mode)
{
    if (!h || !mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode");
    h->mode = mode;
    return GRHIP_OK;
}

int grhip_fir_filter_with_buffer_ntaps(const grhip_fir_filter_with_buffer *h) { return h ? h->ntaps : GRHIP_EINVAL; }

int grhip_fir_filter_with_buffer_filterNdec_device(grhip_fir_filter_with_buffer *h, void *d_output, const void *d_input,
                                                   unsigned long n, unsigned long decimate, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (decimate < 1) return fail(GRHIP_EINVAL, "decimate must be >= 1");
    if (n == 0) return GRHIP_OK;
    if (!d_output || !d_input) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    const size_t it = h->item(), T = (size_t)h->ntaps, nin = (size_t)n * decimate;
    if (T == 0) {                                   // no taps: every output is the empty sum
        GRHIP_HIP(hipMemsetAsync(d_output, 0, (size_t)n * it, st));
        return GRHIP_OK;
    }
    const size_t H = T - 1;
    if ((rc = h->d_both.reserve((H + nin) * it + 64))) return rc;
    unsigned char *both = h->d_both.as<unsigned char>();
    if (H) GRHIP_HIP(hipMemcpyAsync(both, h->d_hist.p, H * it, hipMemcpyDeviceToDevice, st));
    GRHIP_HIP(hipMemcpyAsync(both + H * it, d_input, nin * it, hipMemcpyDeviceToDevice, st));
    const void *first = both + (decimate - 1) * it;         // window of output 0 ends at new item decimate - 1
    if (!mode_fast(h->mode)) {
        if ((rc = h->ensure_inner(h->inner_dec ? h->inner_dec : 1))) return rc;      // (its reversed taps on the device)
        rc = launch_fir_generic(h->kind, h->inner->d_taps_rev.as<float>(), (int)T, first, d_output, (long long)n, (int)decimate,
                                nullptr, st, true);
    } else {
        if ((rc = h->ensure_inner((unsigned)decimate))) return rc;
        h->inner->mode = h->mode;
        rc = h->inner->run(first, d_output, (long long)n, (int)decimate, st);
    }
    if (rc) return rc;
    if (H) GRHIP_HIP(hipMemcpyAsync(h->d_hist.p, both + nin * it, H * it, hipMemcpyDeviceToDevice, st));   // the last ntaps - 1 items
    return GRHIP_OK;
}

int grhip_fir_filter_with_buffer_filterNdec(grhip_fir_filter_with_buffer *h, void *output, const void *input, unsigned long n,
                                            unsigned long decimate)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (decimate < 1) return fail(GRHIP_EINVAL, "decimate must be >= 1");
    if (n == 0) return GRHIP_OK;
    if (!output || !input) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    const size_t it = h->item(), nin = (size_t)n * decimate;
    if ((rc = h->stage_in.reserve(nin * it + 16))) return rc;
    if ((rc = h->stage_out.reserve((size_t)n * it))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, input, nin * it, st);
    rc = grhip_fir_filter_with_buffer_filterNdec_device(h, h->stage_out.p, h->stage_in.p, n, decimate, st);
    if (rc) return rc;
    GRHIP_D2H(h, output, h->stage_out.p, (size_t)n * it, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return GRHIP_OK;
}

}  // extern "C"

// ============================================================================
// gr_freq_xlating_fir_filter_ccc, gr_quadrature_demod_cf, fused hier block
// ============================================================================
struct grhip_freq_xlating_fir_filter_ccc : HandleBase {
    XlatingCore core;
    int mode = GRHIP_MODE_FAST;
    // latched by the setters (.cc.t:86-98)
    std::vector<cf> new_proto; double new_center_freq = 0; bool updated = false;
};

struct grhip_quadrature_demod_cf : HandleBase {
    float gain = 1.f;
    const DeviceTables *tabs = nullptr;
};

struct grhip_xlating_demod : HandleBase {
    XlatingCore core;
    int mode = GRHIP_MODE_FAST;
    float gain = 1.f;
    const DeviceTables *tabs = nullptr;
    DevBuf ystate;   // float2[4]: [0],[1] ping-pong carry of the demodulator's previous
                     // sample, [2] constant zero (the history item of a fresh block)
    int cur = 0;
    bool fresh = true;
    bool carry_direct = false;   // frame of the carried sample (fir_kernels.h, EPI_DEMOD)
};

static int xlating_args_ok(int decimation, const float *taps, size_t ntaps, double sampling_freq)
{
    if (decimation < 1) return fail(GRHIP_EINVAL, "decimation must be >= 1");
    if (ntaps && !taps) return fail(GRHIP_EINVAL, "taps is NULL");
    if (sampling_freq == 0.0) return fail(GRHIP_EINVAL, "sampling_freq is 0");
    return GRHIP_OK;
}

extern "C" {

int grhip_freq_xlating_fir_filter_ccc_create(grhip_freq_xlating_fir_filter_ccc **h, int decimation,
                                             const float *taps, size_t ntaps, double center_freq,
                                             double sampling_freq, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    int rc = xlating_args_ok(decimation, taps, ntaps, sampling_freq);
    if (rc) return rc;
    auto *x = new (std::nothrow) grhip_freq_xlating_fir_filter_ccc();
    if (!x) return fail(GRHIP_ENOMEM, "alloc");
    x->mode = default_mode();
    rc = x->init_device(device);
    if (!rc) {
        x->core.decim = decimation;
        x->core.proto.assign((const cf *)taps, (const cf *)taps + ntaps);
        x->core.center_freq = center_freq; x->core.sampling_freq = sampling_freq;
        rc = x->core.build(device);
    }
    if (rc) { x->core.release(); x->destroy_base(); delete x; return rc; }
    *h = x;
    return GRHIP_OK;
}

void grhip_freq_xlating_fir_filter_ccc_destroy(grhip_freq_xlating_fir_filter_ccc *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->core.release();
    h->destroy_base();
    delete h;
}

int grhip_freq_xlating_fir_filter_ccc_set_center_freq(grhip_freq_xlating_fir_filter_ccc *h, double center_freq)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    if (!h->updated) h->new_proto = h->core.proto;
    h->new_center_freq = center_freq;
    h->updated = true;
    return GRHIP_OK;
}

int grhip_freq_xlating_fir_filter_ccc_set_taps(grhip_freq_xlating_fir_filter_ccc *h, const float *taps,
                                              size_t ntaps)
{
    if (!h || (ntaps && !taps)) return fail(GRHIP_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    if (!h->updated) h->new_center_freq = h->core.center_freq;
    h->new_proto.assign((const cf *)taps, (const cf *)taps + ntaps);
    h->updated = true;
    return GRHIP_OK;
}

int grhip_freq_xlating_fir_filter_ccc_set_mode(grhip_freq_xlating_fir_filter_ccc *h, int mode)
{
    if (!h || !mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode");
    h->mode = mode;
    return GRHIP_OK;
}

int grhip_freq_xlating_fir_filter_ccc_history(const grhip_freq_xlating_fir_filter_ccc *h)
{
    return h ? (h->core.ntaps > 0 ? h->core.ntaps : 1) : GRHIP_EINVAL;
}

int grhip_freq_xlating_fir_filter_ccc_reset(grhip_freq_xlating_fir_filter_ccc *h)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    h->core.reset();
    return GRHIP_OK;
}

static int xl_apply_update(grhip_freq_xlating_fir_filter_ccc *h, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    if (!h->updated) return 0;
    GRHIP_HIP(hipStreamSynchronize(st));         // see fir_apply_update
    if (st != h->own_stream) GRHIP_HIP(hipStreamSynchronize(h->own_stream));
    // work(): set_history; build_composite_fir(); d_updated = false; return 0 (.cc.t:109-114).
    // NB the reference keeps the rotator's d_phase/d_counter and only replaces
    // d_phase_incr; XlatingCore::rebuild_keep_phase does the same.
    h->core.proto = h->new_proto;
    h->core.center_freq = h->new_center_freq;
    int rc = h->core.rebuild_keep_phase(h->device);
    if (rc) return rc;
    h->updated = false;
    return 1;
}

int grhip_freq_xlating_fir_filter_ccc_work_device(grhip_freq_xlating_fir_filter_ccc *h, int noutput_items,
                                                  const void *d_in, void *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = xl_apply_update(h, h->pick(stream));
    if (rc < 0) return rc;
    if (rc == 1) return 0;
    long long n = noutput_items;
    long long n_in = n > 0 ? (n - 1) * h->core.decim + h->core.ntaps : 0;
    rc = h->core.run(h->mode, (const float2 *)d_in, n_in, n, (float2 *)d_out, nullptr, 0.f, nullptr, nullptr,
                     nullptr, h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_freq_xlating_fir_filter_ccc_work(grhip_freq_xlating_fir_filter_ccc *h, int noutput_items,
                                           const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = xl_apply_update(h, h->own_stream);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
    if (noutput_items == 0) return 0;
    long long n = noutput_items;
    size_t n_in = (size_t)((n - 1) * h->core.decim + h->core.ntaps);
    if (n_in == 0) n_in = 1;
    if ((rc = h->stage_in.reserve(n_in * 8 + 16))) return rc;
    if ((rc = h->stage_out.reserve((size_t)n * 8))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n_in * 8, st);
    rc = h->core.run(h->mode, h->stage_in.as<float2>(), (long long)n_in, n, h->stage_out.as<float2>(), nullptr,
                     0.f, nullptr, nullptr, nullptr, st);
    if (rc) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)n * 8, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

// ---- quadrature_demod_cf -----------------------------------------------------
int grhip_quadrature_demod_cf_create(grhip_quadrature_demod_cf **h, float gain, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    auto *q = new (std::nothrow) grhip_quadrature_demod_cf();
    if (!q) return fail(GRHIP_ENOMEM, "alloc");
    q->gain = gain;
    int rc = q->init_device(device);
    if (!rc) rc = get_device_tables(device, &q->tabs);
    if (rc) { q->destroy_base(); delete q; return rc; }
    *h = q;
    return GRHIP_OK;
}

void grhip_quadrature_demod_cf_destroy(grhip_quadrature_demod_cf *h)
{
    if (!h) return;
    h->destroy_base();
    delete h;
}

int grhip_quadrature_demod_cf_work_device(grhip_quadrature_demod_cf *h, int noutput_items, const void *d_in,
                                          void *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    rc = launch_quad_demod((const float2 *)d_in, (float *)d_out, noutput_items, h->gain, h->tabs->atan_tab,
                           h->pick(stream));
    return rc ? rc : noutput_items;
}

int grhip_quadrature_demod_cf_work(grhip_quadrature_demod_cf *h, int noutput_items, const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    size_t n = (size_t)noutput_items;
    if ((rc = h->stage_in.reserve((n + 1) * 8))) return rc;
    if ((rc = h->stage_out.reserve(n * 4))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, (n + 1) * 8, st);
    if ((rc = launch_quad_demod(h->stage_in.as<float2>(), h->stage_out.as<float>(), (long long)n, h->gain,
                                h->tabs->atan_tab, st)))
        return rc;
    GRHIP_D2H(h, out, h->stage_out.p, n * 4, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

// ---- fused xlating -> demod ----------------------------------------------------
int grhip_xlating_demod_create(grhip_xlating_demod **h, int decimation, const float *taps, size_t ntaps,
                               double center_freq, double sampling_freq, float gain, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    int rc = xlating_args_ok(decimation, taps, ntaps, sampling_freq);
    if (rc) return rc;
    auto *x = new (std::nothrow) grhip_xlating_demod();
    if (!x) return fail(GRHIP_ENOMEM, "alloc");
    x->mode = default_mode();
    x->gain = gain;
    rc = x->init_device(device);
    if (!rc) rc = get_device_tables(device, &x->tabs);
    if (!rc) {
        x->core.decim = decimation;
        x->core.for_demod = true;
        x->core.proto.assign((const cf *)taps, (const cf *)taps + ntaps);
        x->core.center_freq = center_freq; x->core.sampling_freq = sampling_freq;
        rc = x->core.build(device);
    }
    if (!rc) rc = x->ystate.reserve(4 * sizeof(float2));
    if (!rc) { hipError_t e = hipMemset(x->ystate.p, 0, 4 * sizeof(float2)); if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "memset"); }
    if (rc) { x->core.release(); x->ystate.release(); x->destroy_base(); delete x; return rc; }
    *h = x;
    return GRHIP_OK;
}

void grhip_xlating_demod_destroy(grhip_xlating_demod *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->core.release(); h->ystate.release();
    h->destroy_base();
    delete h;
}

int grhip_xlating_demod_set_mode(grhip_xlating_demod *h, int mode)
{
    if (!h || !mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode");
    h->mode = mode;
    return GRHIP_OK;
}

int grhip_xlating_demod_reset(grhip_xlating_demod *h)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    int rc = h->bind();
    if (rc) return rc;
    h->core.reset();     // host-side only: no device work, no synchronisation
    h->fresh = true;
    return GRHIP_OK;
}

int grhip_xlating_demod_history(const grhip_xlating_demod *h)
{
    return h ? (h->core.ntaps > 0 ? h->core.ntaps : 1) : GRHIP_EINVAL;
}

int grhip_xlating_demod_work_device(grhip_xlating_demod *h, int noutput_items, const void *d_in, void *d_out,
                                    void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    long long n = noutput_items;
    long long n_in = (n - 1) * h->core.decim + h->core.ntaps;
    float2 *ys = h->ystate.as<float2>();
    const bool direct = h->core.demod_is_direct(h->mode, true);
    if (!h->fresh && direct != h->carry_direct) {
        // mode switch in mid-stream: move the one-sample carry into the other epilogue's frame
        cf g, y;
        if ((rc = h->core.phase_before_pos(&g))) return rc;
        GRHIP_HIP(hipStreamSynchronize(h->pick(stream)));
        GRHIP_HIP(hipMemcpy(&y, ys + h->cur, sizeof(y), hipMemcpyDeviceToHost));
        y = direct ? y * std::conj(g) : y * g;
        GRHIP_HIP(hipMemcpy(ys + h->cur, &y, sizeof(y), hipMemcpyHostToDevice));
    }
    h->carry_direct = direct;
    rc = h->core.run(h->mode, (const float2 *)d_in, n_in, n, nullptr, (float *)d_out, h->gain,
                     h->fresh ? ys + 2 : ys + h->cur, ys + (h->cur ^ 1), h->tabs->atan_tab, h->pick(stream));
    if (rc) return rc;
    h->cur ^= 1;
    h->fresh = false;
    return noutput_items;
}

int grhip_xlating_demod_run_captures_device(grhip_xlating_demod *h, int n_streams, size_t n_samples,
                                            const void *d_in, size_t in_stride_items, void *d_out,
                                            size_t out_stride_items, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (n_streams < 1) return fail(GRHIP_EINVAL, "n_streams must be >= 1");
    int rc = h->bind();
    if (rc) return rc;
    const long long n_out = (long long)(n_samples / (size_t)h->core.decim);
    if (n_out <= 0) return GRHIP_OK;
    // (GRHIP_MODE_GENERIC: the fused generic-order kernel takes batches at decimation 1 / 2 / 4; run() says so otherwise)
    if (mode_fast(h->mode) && !(h->core.use_tiled || (mode_matrix(h->mode) && h->core.use_mfma) ||
                                (h->core.use_hidec && h->core.hidec_premix)))
        return fail(GRHIP_EINVAL, "run_captures needs a batched engine (FAST mode, supported decimation)");
    if (h->core.tab_start != 0 && !h->core.demod_is_direct(h->mode, true, true))
        return fail(GRHIP_EINVAL, "handle has streamed past its cached rotator table; use a fresh handle");
    const long long hist = h->core.ntaps > 0 ? h->core.ntaps - 1 : 0;
    const long long keep = h->core.pos;
    h->core.pos = 0;                                  // every capture starts at rotator phase 1
    rc = h->core.run(h->mode, (const float2 *)d_in - hist, hist + (long long)n_samples, n_out, nullptr,
                     (float *)d_out, h->gain, nullptr, nullptr, h->tabs->atan_tab, h->pick(stream), n_streams,
                     (long long)in_stride_items, hist, (long long)out_stride_items);
    h->core.pos = keep;
    return rc;
}

int grhip_xlating_demod_work(grhip_xlating_demod *h, int noutput_items, const void *in, void *out)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    if (noutput_items == 0) return 0;
    int rc = h->bind();
    if (rc) return rc;
    long long n = noutput_items;
    size_t n_in = (size_t)((n - 1) * h->core.decim + h->core.ntaps);
    if (n_in == 0) n_in = 1;
    if ((rc = h->stage_in.reserve(n_in * 8 + 16))) return rc;
    if ((rc = h->stage_out.reserve((size_t)n * 4))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, n_in * 8, st);
    rc = grhip_xlating_demod_work_device(h, noutput_items, h->stage_in.p, h->stage_out.p, st);
    if (rc < 0) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)n * 4, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

}  // extern "C"
