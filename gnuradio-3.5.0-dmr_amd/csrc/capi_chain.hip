// capi_chain.hip -- the full DMR chain as one device-resident pipeline over a
// batch of independent captures:
//   freq_xlating_fir_filter_ccc -> quadrature_demod_cf (one fused kernel)
//   -> clock_recovery_mm_ff (one wavefront per stream)
//   -> binary_slicer_fb -> correlate_access_code_bb (one fused kernel)
// Streams are independent units; batching them is what fills the
// device for the serial M&M stage.
#include <algorithm>
#include <complex>

#include "digital_kernels.h"
#include "fir_kernels.h"
#include "grhip_internal.h"
#include "xlating_core.h"

using namespace grhip;

namespace grhip {
int mm_init_state(MMState &s, float omega, float gain_omega, float mu, float gain_mu, float rel);
int corr_set_code(CorrParams &p, unsigned long long &flag_bit, const char *code, size_t len);
}

struct grhip_dmr_chain : HandleBase {
    XlatingCore core;
    float gain = 1.f;
    MMState mm_init;
    CorrParams cp;
    int S = 1;
    size_t max_samples = 0, max_out = 0;
    const DeviceTables *tabs = nullptr;
    DevBuf d_demod, d_soft, d_mm, d_mm_init, d_counts, d_ystate, d_corr, d_scratch;
    size_t out_stride = 0;
    int mode = GRHIP_MODE_FAST;
    // FAST modes: the capture is processed in PIPE_CHUNKS time slices; the clock recovery of slice c (second
    // stream) runs beside the FIR of slice c+1
#ifndef GRHIP_PIPE_CHUNKS
#define GRHIP_PIPE_CHUNKS 32
#endif
#ifndef GRHIP_CHAIN_WGCAP
#define GRHIP_CHAIN_WGCAP 1
#endif
    // clock recovery with eight captures per wave (mm_rows_kernel): 2048 captures are 256 waves, one per SIMD of 64 CUs,
    // and the FIR keeps two workgroups on each of the other CUs; from GRHIP_MM_ROWS_MIN captures on (below, one wave per
    // capture has the shorter pass -- 0.145 against 0.18 us per symbol -- and there are SIMDs enough; same box, one wave per
    // capture / eight captures per wave: 1600 captures 313 / 317 Gsamples/s, 2048: 272 / 396; profiles/r02_chain_batch_sizes.log)
#ifndef GRHIP_MM_ROWS_MIN
#define GRHIP_MM_ROWS_MIN 1600
#endif
#ifndef GRHIP_CHAIN_WGCAP_ROWS
#define GRHIP_CHAIN_WGCAP_ROWS 0
#endif
    // ... and the two kernels get CUs of their own through two streams with CU masks: the loop's waves on GRHIP_MM_CUS CUs
    // (mask bits go round the XCDs, so every XCD gives the same number), the FIR on the others.  On shared CUs the FIR's
    // waves delay every pass of the loop (+40 %); a wave of the loop wants a SIMD to itself (2048 captures, same box:
    // no masks 72.4 ms, 48 CUs 63.0, 56: 63.1, 64: 60.7, 96: 72.3 -- the FIR then lacks CUs)
#ifndef GRHIP_MM_CUS
#define GRHIP_MM_CUS 64
#endif
    // Thirty-two captures per wave (mm_pairs_kernel, the FIFO in registers): 2048 captures are 64 waves, one per SIMD of
    // 16 CUs -- the loop's share is then the SIMDs its waves fill, rounded up to whole CUs per XCD.
#ifndef GRHIP_MM_BIG_FORM
#define GRHIP_MM_BIG_FORM 8           // the form from GRHIP_MM_ROWS_MIN captures on: 8 or 32 captures per wave
#endif
    hipStream_t st_mm8 = nullptr, st_fir8 = nullptr;
    int fir8_cus = 0, mm8_cus = 0, ncu = 0;
    int captures_per_wave = 0;        // 0: by batch size; 1 / 8 / 32: forced (grhip_dmr_chain_set_captures_per_wave)
    size_t max_symbols = 0;           // 0: none (grhip_dmr_chain_set_max_symbols)
    int mm_form() const { return captures_per_wave ? captures_per_wave : (S >= GRHIP_MM_ROWS_MIN ? GRHIP_MM_BIG_FORM : 1); }
    int mm_cus_wanted() const
    {
        if (mm_form() != 32) return GRHIP_MM_CUS;
        const int waves = (S + 31) / 32;
        return std::max(8, ((waves + 3) / 4 + 7) / 8 * 8);
    }
    // the two masked streams for a loop on `cus` CUs and the FIR on the others (made again when the share changes)
    void ensure_masks(int cus)
    {
        if (cus == mm8_cus && st_mm8 && st_fir8) return;
        if (st_mm8) (void)hipStreamDestroy(st_mm8);
        if (st_fir8) (void)hipStreamDestroy(st_fir8);
        st_mm8 = st_fir8 = nullptr; mm8_cus = fir8_cus = 0;
        if (cus <= 0 || ncu < 4 * cus || ncu > 1024) return;
        uint32_t m_mm[32] = {}, m_fir[32] = {};
        for (int i = 0; i < ncu; ++i) (i < cus ? m_mm : m_fir)[i / 32] |= 1u << (i % 32);
        const uint32_t words = (uint32_t)((ncu + 31) / 32);
        if (hipExtStreamCreateWithCUMask(&st_mm8, words, m_mm) != hipSuccess) st_mm8 = nullptr;
        if (st_mm8 && hipExtStreamCreateWithCUMask(&st_fir8, words, m_fir) != hipSuccess) st_fir8 = nullptr;
        if (st_mm8 && !st_fir8) { (void)hipStreamDestroy(st_mm8); st_mm8 = nullptr; }
        if (st_fir8) { fir8_cus = ncu - cus; mm8_cus = cus; }
        (void)hipGetLastError();
    }
    static constexpr int PIPE_CHUNKS = GRHIP_PIPE_CHUNKS;
    // 4FSK tail (grhip_dmr_chain_set_four_level): pager_slicer_fb -> unpack_k_bits(2) in front of the correlator
    bool four_level = false;
    float pager_alpha = 0.f;
    DevBuf d_sym, d_dibits, d_avg, d_nbits2;
    // ... in time slices too (GRHIP_CHAIN_SLICED_TAIL=1): the four-level slicer of slice c (third stream) beside the clock
    // recovery of slice c + 1.  It reads the symbol totals slice c left, so every slice's totals are kept (d_cnt_hist; the
    // loop of slice c + 1 writes its own), and carries its position per capture (d_pos).  Built, bit-exact, and OFF: the
    // slicer is one wave per capture that issues on every cycle it gets, and 2048 of them per slice beside the FIR take a
    // third of the issue slots of the CUs they land on -- the FIR's static tile split then waits for its slowest CU:
    // 2048 captures 91.7 ms against 54.7 with the slicer behind the last slice (profiles/r03_notes.md).
#ifndef GRHIP_CHAIN_SLICED_TAIL
#define GRHIP_CHAIN_SLICED_TAIL 0
#endif
    DevBuf d_cnt_hist, d_pos;
    hipStream_t st2 = nullptr, st3 = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_tail = nullptr, ev_fir[PIPE_CHUNKS] = {}, ev_mm[PIPE_CHUNKS] = {};
};

extern "C" {

int grhip_dmr_chain_create(grhip_dmr_chain **h, const grhip_dmr_chain_params *p, int n_streams,
                           size_t max_samples_per_stream, int device)
{
    if (!h || !p) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (n_streams < 1) return fail(GRHIP_EINVAL, "n_streams must be >= 1");
    if (p->decimation < 1) return fail(GRHIP_EINVAL, "decimation must be >= 1");
    if (p->ntaps && !p->taps) return fail(GRHIP_EINVAL, "taps is NULL");
    MMState ms;
    int rc = mm_init_state(ms, p->omega, p->gain_omega, p->mu, p->gain_mu, p->omega_relative_limit);
    if (rc) return rc;
    CorrParams cp;
    memset(&cp, 0, sizeof(cp));
    unsigned long long fb = 0;
    rc = corr_set_code(cp, fb, p->access_code, p->access_code_len);
    if (rc) return rc;
    cp.threshold = (unsigned)p->threshold;

    auto *c = new (std::nothrow) grhip_dmr_chain();
    if (!c) return fail(GRHIP_ENOMEM, "alloc");
    c->gain = p->demod_gain; c->mm_init = ms; c->cp = cp; c->S = n_streams;
    c->max_samples = max_samples_per_stream;
    c->max_out = max_samples_per_stream / p->decimation;
    c->out_stride = ((c->max_out + 63) / 64) * 64 + 64;     // 16-byte aligned rows
    rc = c->init_device(device);
    if (!rc) rc = get_device_tables(device, &c->tabs);
    if (!rc) {
        c->core.decim = p->decimation;
        c->core.proto.assign((const std::complex<float> *)p->taps, (const std::complex<float> *)p->taps + p->ntaps);
        c->core.center_freq = p->center_freq; c->core.sampling_freq = p->sampling_freq;
        c->core.for_demod = true;           // decimations other than 1/2/4: the direct kernel with the fused demodulator
        rc = c->core.build(device);
    }
    if (!rc && !c->core.use_tiled && !c->core.use_mfma && !(c->core.use_hidec && c->core.hidec_premix))
        rc = fail(GRHIP_EINVAL, "dmr_chain needs a decimation/tap count a batched FIR engine supports");
    if (!rc) {
        hipError_t e = hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->st3, hipStreamNonBlocking);
        if (e == hipSuccess && GRHIP_MM_CUS > 0) {
            (void)hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, device);
            c->ensure_masks(c->mm_cus_wanted());
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_end, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming);
        for (int i = 0; i < grhip_dmr_chain::PIPE_CHUNKS && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&c->ev_fir[i], hipEventDisableTiming);
        for (int i = 0; i < grhip_dmr_chain::PIPE_CHUNKS && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&c->ev_mm[i], hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "stream / event creation: %s", hipGetErrorString(e));
    }
    size_t S = (size_t)n_streams;
    if (!rc) rc = c->d_demod.reserve(S * c->out_stride * 4);
    if (!rc) rc = c->d_soft.reserve(S * c->out_stride * 4);
    if (!rc) rc = c->d_mm.reserve(S * sizeof(MMState));
    if (!rc) rc = c->d_mm_init.reserve(S * sizeof(MMState));
    if (!rc) rc = c->d_counts.reserve(S * 2 * sizeof(int));
    if (!rc) rc = c->d_cnt_hist.reserve((size_t)grhip_dmr_chain::PIPE_CHUNKS * S * 2 * sizeof(int));
    if (!rc) rc = c->d_ystate.reserve(S * 2 * sizeof(float2));
    if (!rc) rc = c->d_corr.reserve(S * sizeof(CorrState));
    if (!rc) {
        std::vector<MMState> init(S, ms);
        hipError_t e = hipMemcpy(c->d_mm_init.p, init.data(), S * sizeof(MMState), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(GRHIP_ERUNTIME, "state upload");
    }
    if (rc) { grhip_dmr_chain_destroy(c); return rc; }
    *h = c;
    return GRHIP_OK;
}

void grhip_dmr_chain_destroy(grhip_dmr_chain *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->core.release();
    h->d_demod.release(); h->d_soft.release(); h->d_mm.release(); h->d_mm_init.release();
    h->d_counts.release(); h->d_ystate.release(); h->d_corr.release(); h->d_scratch.release();
    h->d_sym.release(); h->d_dibits.release(); h->d_avg.release(); h->d_nbits2.release();
    h->d_cnt_hist.release(); h->d_pos.release();
    if (h->st2) (void)hipStreamDestroy(h->st2);
    if (h->st3) (void)hipStreamDestroy(h->st3);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    for (auto &e : h->ev_mm) if (e) (void)hipEventDestroy(e);
    if (h->st_mm8) (void)hipStreamDestroy(h->st_mm8);
    if (h->st_fir8) (void)hipStreamDestroy(h->st_fir8);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    for (auto &e : h->ev_fir) if (e) (void)hipEventDestroy(e);
    h->destroy_base();
    delete h;
}

int grhip_dmr_chain_run_device(grhip_dmr_chain *h, const void *d_in, size_t n_samples,
                               size_t stream_stride_items, unsigned char *d_bits, size_t bits_stride,
                               int *d_nbits, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (n_samples > h->max_samples) return fail(GRHIP_EINVAL, "n_samples exceeds max_samples_per_stream");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    const size_t S = (size_t)h->S;
    const long long n_out = (long long)(n_samples / h->core.decim);
    const int mm_nout = (int)(h->max_symbols && (long long)h->max_symbols < n_out ? (long long)h->max_symbols : n_out);
    if (n_out <= 0) { GRHIP_HIP(hipMemsetAsync(d_nbits, 0, S * sizeof(int), st)); return GRHIP_OK; }
    if ((size_t)n_out * (h->four_level ? 2 : 1) > bits_stride) return fail(GRHIP_EINVAL, "bits_stride too small");

    // fresh block state for every capture
    h->core.reset();
    GRHIP_HIP(hipMemsetAsync(h->d_ystate.p, 0, S * 2 * sizeof(float2), st));
    GRHIP_HIP(hipMemcpyAsync(h->d_mm.p, h->d_mm_init.p, S * sizeof(MMState), hipMemcpyDeviceToDevice, st));
    GRHIP_HIP(hipMemsetAsync(h->d_corr.p, 0, S * sizeof(CorrState), st));

    // 1) xlating + quad demod; the ntaps-1 history zeros are synthesised by n_lo
    const long long hist = h->core.ntaps > 0 ? h->core.ntaps - 1 : 0;
    const float2 *x = (const float2 *)d_in - hist;
    float2 *ys = h->d_ystate.as<float2>();
    hipStream_t st_mm = st;         // the stream the clock recovery and the correlator run on
    bool sliced_tail = false;       // the four-level slicer has run beside the clock recovery, slice by slice
    // GRHIP_MODE_GENERIC takes the same time-sliced route where its fused kernel takes batches (decimation 1 / 2 / 4): the
    // whole chain bit-exact, the clock recovery beside a FIR that is five times the fast one -- on the same CUs (no masked
    // streams: the vector-bound FIR wants every CU, and the loop is a tenth of its time)
    const bool generic_sliced = !mode_fast(h->mode) && generic_demod_batch_ok(h->core.ntaps, h->core.decim, x, (long long)stream_stride_items);
    if (generic_sliced) {
        const float2 *gt = nullptr;                  // the rotator's phases for the whole capture, once (kept across runs)
        rc = h->core.ensure_rot(n_out, &gt, st);
        if (rc) return rc;
    }
    if (mode_fast(h->mode) || generic_sliced) {
        // FIR + demodulator in time slices on `st`; the clock recovery of a slice starts on the second stream as
        // soon as that slice is written and continues from where the previous slice left it (mm_kernel's resume
        // mode: same recurrence, same results, whatever the slicing).  The serial loop (one wavefront per capture,
        // ~0.15 us per symbol) is the long pole: with the FIR beside it, a batch costs little more than it alone.
        // The FIR keeps to one workgroup per CU here, so that the four clock-recovery waves of a CU find room.
        // (up to PIPE_CHUNKS slices of at least 8192 outputs: 16 -> 32 slices +2.4 % at 2048 captures, +1 % at 1024, same box)
        const int NC = n_out >= 64 * 1024 ? (int)std::min<long long>(grhip_dmr_chain::PIPE_CHUNKS, n_out / 8192) : 1;
        const long long Lc = ((n_out + NC - 1) / NC + 63) / 64 * 64;       // slice length in outputs (rows stay 16-byte aligned)
        const int nsl = (int)((n_out + Lc - 1) / Lc);                      // slices that hold outputs (<= NC)
        GRHIP_HIP(hipMemsetAsync(h->d_counts.p, 0, S * 2 * sizeof(int), st));
        // the totals slice c leaves: d_cnt_hist[c] (the last slice's: d_counts, where the stages behind read them)
        auto totals = [&](int c) { return c == nsl - 1 ? h->d_counts.as<int>() : h->d_cnt_hist.as<int>() + (size_t)c * S * 2; };
        sliced_tail = GRHIP_CHAIN_SLICED_TAIL && h->four_level && nsl > 1;
        if (sliced_tail) {
            GRHIP_HIP(hipMemsetAsync(h->d_avg.p, 0, S * sizeof(float), st));                     // d_avg = 0 (pager_slicer_fb.cc:40)
            GRHIP_HIP(hipMemsetAsync(h->d_pos.p, 0, S * sizeof(int), st));
        }
        // (ring of 1024 samples while four waves per CU of the loop's share hold the batch, else 512: see mm_rows_kernel)
        // (without the masked streams -- their creation failed -- the loop and the FIR share every CU: the batch-size rule then
        // keeps one wave per capture; a forced 8 runs with the small ring beside a FIR held to one workgroup per CU)
        const int form = h->mm_form();
        if (form != 1 && h->ncu) h->ensure_masks(h->mm_cus_wanted());
        const bool masks = h->st_mm8 && h->st_fir8 && !generic_sliced;
        const bool want_rows = form != 1 && (masks || h->captures_per_wave == 8);
        // (launch_mm's `rows`: 32 = thirty-two captures per wave; else the ring of the eight-captures form)
        const int rows = !want_rows ? 0 : form == 32 && masks ? 32 : (masks && (h->S + 7) / 8 <= 4 * GRHIP_MM_CUS ? 1024 : 512);
        // eight captures per wave: the loop and the FIR on CUs of their own (two masked streams)
        const bool split = rows && NC > 1 && masks;
        hipStream_t st_side = split ? h->st_mm8 : h->st2;
        hipStream_t st_fir = split ? h->st_fir8 : st;
        GRHIP_HIP(hipEventRecord(h->ev_begin, st));
        GRHIP_HIP(hipStreamWaitEvent(st_side, h->ev_begin, 0));
        if (split) GRHIP_HIP(hipStreamWaitEvent(st_fir, h->ev_begin, 0));
        if (sliced_tail) GRHIP_HIP(hipStreamWaitEvent(h->st3, h->ev_begin, 0));
        st_mm = NC > 1 ? st_side : st;
        h->core.mf_wg_cap = NC > 1 ? (split ? GRHIP_CHAIN_WGCAP_ROWS : GRHIP_CHAIN_WGCAP) : 0;
        h->core.mf_cu_cap = split ? h->fir8_cus : 0;
        for (int c = 0; c < NC; ++c) {
            const long long o0 = (long long)c * Lc;
            if (o0 >= n_out) break;
            const long long len = std::min(Lc, n_out - o0);
            const float2 *xc = x + o0 * h->core.decim;            // oldest history item of the slice
            const float2 *yp = (c & 1) ? ys + S : ys;             // carry of the demodulator's previous sample
            float2 *yl = (c & 1) ? ys : ys + S;
            rc = h->core.run(h->mode, xc, hist + len * h->core.decim, len, nullptr, h->d_demod.as<float>() + o0,
                             h->gain, yp, yl, h->tabs->atan_tab, st_fir, h->S, (long long)stream_stride_items,
                             c == 0 ? hist : 0, (long long)h->out_stride);
            if (rc) { h->core.mf_wg_cap = 0; h->core.mf_cu_cap = 0; return rc; }
            if (NC > 1) {
                GRHIP_HIP(hipEventRecord(h->ev_fir[c], st_fir));
                GRHIP_HIP(hipStreamWaitEvent(st_side, h->ev_fir[c], 0));
            }
            // 2) M&M clock recovery, one wavefront per stream, over what has been demodulated so far
            rc = launch_mm(h->d_mm.as<MMState>(), h->S, mm_nout, (int)(o0 + len), h->d_demod.as<float>(),
                           (long long)h->out_stride, h->d_soft.as<float>(), (long long)h->out_stride,
                           c == 0 ? h->d_counts.as<int>() : totals(c - 1), h->tabs->mmse_rev, st_mm, 1, rows,
                           nsl > 1 ? totals(c) : nullptr);
            if (rc) { h->core.mf_wg_cap = 0; h->core.mf_cu_cap = 0; return rc; }
            if (sliced_tail) {
                // 3') 4FSK, in slices: the symbols this slice's clock recovery has added, on the third stream
                GRHIP_HIP(hipEventRecord(h->ev_mm[c], st_mm));
                GRHIP_HIP(hipStreamWaitEvent(h->st3, h->ev_mm[c], 0));
                rc = launch_pager_slicer(h->d_avg.as<float>(), h->S, h->pager_alpha, 1.0f - h->pager_alpha, h->d_soft.as<float>(),
                                         (long long)h->out_stride, h->d_sym.as<unsigned char>(), (long long)h->out_stride, n_out,
                                         h->st3, totals(c), 2, h->d_pos.as<int>());
                if (rc) { h->core.mf_wg_cap = 0; h->core.mf_cu_cap = 0; return rc; }
            }
        }
        h->core.mf_cu_cap = 0;
        if (split) {
            // what follows the loop (slicers, unpack, correlator) has the whole device again: on the loop's masked stream
            // 2048 single-wave workgroups shared 64 CUs (pager slicer 12.4 ms instead of 2.x, correlator 2.0 instead of 0.4)
            GRHIP_HIP(hipEventRecord(h->ev_end, st_side));
            GRHIP_HIP(hipStreamWaitEvent(st, h->ev_end, 0));
            st_mm = st;
        }
        h->core.mf_wg_cap = 0;
    } else {
        // generic order: all captures in one launch where the fused generic-order kernel takes the shape (decimation
        // 1 / 2 / 4), else explicit zero history in a scratch row, one stream at a time
        h->core.reset();
        const bool batched_generic = h->core.run(GRHIP_MODE_GENERIC, x, hist + (long long)n_samples, n_out, nullptr,
                                                 h->d_demod.as<float>(), h->gain, ys, ys + S, h->tabs->atan_tab, st, h->S,
                                                 (long long)stream_stride_items, hist, (long long)h->out_stride) == GRHIP_OK;
        if (!batched_generic) rc = h->d_scratch.reserve((size_t)(hist + (long long)n_samples) * sizeof(float2));
        if (rc) return rc;
        for (size_t s = 0; s < S && !batched_generic; ++s) {
            GRHIP_HIP(hipMemsetAsync(h->d_scratch.p, 0, (size_t)hist * sizeof(float2), st));
            GRHIP_HIP(hipMemcpyAsync(h->d_scratch.as<float2>() + hist, (const float2 *)d_in + s * stream_stride_items,
                                     n_samples * sizeof(float2), hipMemcpyDeviceToDevice, st));
            h->core.reset();
            rc = h->core.run(GRHIP_MODE_GENERIC, h->d_scratch.as<float2>(), hist + (long long)n_samples, n_out,
                             nullptr, h->d_demod.as<float>() + s * h->out_stride, h->gain, ys + s, ys + S + s,
                             h->tabs->atan_tab, st);
            if (rc) return rc;
        }
        // 2) M&M clock recovery, one wavefront per stream
        rc = launch_mm(h->d_mm.as<MMState>(), h->S, mm_nout, (int)n_out, h->d_demod.as<float>(),
                       (long long)h->out_stride, h->d_soft.as<float>(), (long long)h->out_stride,
                       h->d_counts.as<int>(), h->tabs->mmse_rev, st, 0, h->mm_form() == 32 ? 32 : h->mm_form() == 8 ? 512 : 0);
        if (rc) return rc;
    }

    // symbols a capture is expected to produce at most (the nominal clock and a margin): sizes the grids of the stages
    // behind the clock recovery, which read the true counts on the device and walk longer streams in strides
    const long long n_expect = (long long)((double)n_out / (double)h->mm_init.omega_mid * 1.125) + 4096;
    if (h->four_level) {
        // 3') 4FSK: DC-tracking four-level slicer, dibit unpack, correlator on the bit stream (2 items per symbol)
        if (sliced_tail) {
            GRHIP_HIP(hipEventRecord(h->ev_tail, h->st3));
            GRHIP_HIP(hipStreamWaitEvent(st_mm, h->ev_tail, 0));
        } else {
            GRHIP_HIP(hipMemsetAsync(h->d_avg.p, 0, S * sizeof(float), st_mm));                  // d_avg = 0 (pager_slicer_fb.cc:40)
            rc = launch_pager_slicer(h->d_avg.as<float>(), h->S, h->pager_alpha, 1.0f - h->pager_alpha, h->d_soft.as<float>(),
                                     (long long)h->out_stride, h->d_sym.as<unsigned char>(), (long long)h->out_stride, n_out, st_mm,
                                     h->d_counts.as<int>(), 2);
            if (rc) return rc;
        }
        rc = launch_unpack_k_bits_streams(2, h->S, h->d_sym.as<unsigned char>(), (long long)h->out_stride,
                                          h->d_dibits.as<unsigned char>(), 2 * (long long)h->out_stride, n_out,
                                          h->d_counts.as<int>(), 2, h->d_nbits2.as<int>(), 1, st_mm);
        if (rc) return rc;
        rc = launch_correlate(h->cp, h->d_corr.as<CorrState>(), h->S, h->d_dibits.as<unsigned char>(), nullptr,
                              2 * (long long)h->out_stride, d_bits, (long long)bits_stride, 2 * n_out, h->d_nbits2.as<int>(), 1,
                              st_mm, 2 * n_expect);
        if (rc) return rc;
        GRHIP_HIP(hipMemcpyAsync(d_nbits, h->d_nbits2.p, S * sizeof(int), hipMemcpyDeviceToDevice, st_mm));
        if (st_mm != st) {
            GRHIP_HIP(hipEventRecord(h->ev_end, st_mm));
            GRHIP_HIP(hipStreamWaitEvent(st, h->ev_end, 0));
        }
        return GRHIP_OK;
    }
    // 3) slicer + access-code correlator on the symbols each stream produced
    rc = launch_correlate(h->cp, h->d_corr.as<CorrState>(), h->S, nullptr, h->d_soft.as<float>(),
                          (long long)h->out_stride, d_bits, (long long)bits_stride, n_out,
                          h->d_counts.as<int>(), 2, st_mm, n_expect);
    if (rc) return rc;
    GRHIP_HIP(hipMemcpy2DAsync(d_nbits, sizeof(int), h->d_counts.p, 2 * sizeof(int), sizeof(int), S,
                               hipMemcpyDeviceToDevice, st_mm));
    if (st_mm != st) {          // the caller's stream continues when the second one is done
        GRHIP_HIP(hipEventRecord(h->ev_end, st_mm));
        GRHIP_HIP(hipStreamWaitEvent(st, h->ev_end, 0));
    }
    return GRHIP_OK;
}

int grhip_dmr_chain_set_mode(grhip_dmr_chain *h, int mode)
{
    if (!h || !mode_valid(mode)) return fail(GRHIP_EINVAL, "bad mode");
    h->mode = mode;
    return GRHIP_OK;
}

int grhip_dmr_chain_set_captures_per_wave(grhip_dmr_chain *h, int captures)
{
    if (!h || !(captures == 0 || captures == 1 || captures == 8 || captures == 32))
        return fail(GRHIP_EINVAL, "captures per wave: 0 (by batch size), 1, 8 or 32");
    h->captures_per_wave = captures;
    return GRHIP_OK;
}

int grhip_dmr_chain_set_max_symbols(grhip_dmr_chain *h, size_t max_symbols)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    h->max_symbols = max_symbols;
    return GRHIP_OK;
}

int grhip_dmr_chain_set_four_level(grhip_dmr_chain *h, int enable, float pager_alpha)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    int rc = h->bind();
    if (rc) return rc;
    if (enable) {
        const size_t S = (size_t)h->S;
        if ((rc = h->d_sym.reserve(S * h->out_stride))) return rc;
        if ((rc = h->d_dibits.reserve(S * 2 * h->out_stride))) return rc;
        if ((rc = h->d_avg.reserve(S * sizeof(float)))) return rc;
        if ((rc = h->d_pos.reserve(S * sizeof(int)))) return rc;
        if ((rc = h->d_nbits2.reserve(S * sizeof(int)))) return rc;
    }
    h->four_level = enable != 0;
    h->pager_alpha = pager_alpha;
    return GRHIP_OK;
}

int grhip_dmr_chain_intermediate(grhip_dmr_chain *h, int which, void **d_ptr, size_t *stride)
{
    if (!h || !d_ptr || !stride) return fail(GRHIP_EINVAL, "null argument");
    *stride = h->out_stride;
    if (which == 0) *d_ptr = h->d_demod.p;
    else if (which == 1) *d_ptr = h->d_soft.p;
    else if (which == 2 && h->four_level) *d_ptr = h->d_sym.p;
    else return fail(GRHIP_EINVAL, "which must be 0, 1 (or 2 with the 4FSK tail)");
    return GRHIP_OK;
}

}  // extern "C"
