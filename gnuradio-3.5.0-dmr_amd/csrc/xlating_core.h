// xlating_core.h -- host-side state of gr_freq_xlating_fir_filter_ccc shared by
// the plain block handle, the fused xlating->demod block and the DMR chain.
//
// Reference: filter/gr_freq_xlating_fir_filter_XXX.cc.t:46-123, gr_rotator.h:29-52.
//
// The rotator phase is a serial float recurrence that is data independent
// (SURVEY F4/H1): its value for output k depends only on (phase_incr, k).  It is
// generated once on the host with the exact recurrence, kept in HBM as a table
// (8 bytes per output), extended on demand while a stream runs and reused by
// every capture that starts from a fresh block (reset()).
#pragma once
#include <complex>
#include <cstdlib>
#include <vector>

#include "grhip_internal.h"

namespace grhip {

struct XlatingCore {
    // parameters
    int decim = 1;
    std::vector<std::complex<float>> proto;
    double center_freq = 0, sampling_freq = 1;

    // set before build(): the handle only ever demodulates (xlating_demod, dmr_chain).  Its FAST dispatch then prefers the
    // engines with a fused demodulator -- they need no rotator phases, whose exact recurrence is computed on the host at
    // ~3 ns per output and bounds a STREAMING block on any other path (measured: 1-8 Gsamples/s against 200-320)
    bool for_demod = false;
    // built by build()
    int ntaps = 0;
    std::vector<std::complex<float>> ctaps;     // composite taps, reference arithmetic
    std::complex<float> incr{1.f, 0.f};         // normalised d_phase_incr
    double omega = 0;                           // (double)(float)fwT0
    DevBuf d_taps_generic;                      // ctaps (d_taps order of the inner gr_fir_ccc)
    DevBuf d_hp, d_wtab, d_stab, d_vtab;        // tiled kernel operands
    int Tq = 0;
    bool use_tiled = false, premix = false;
    // FAST mode for the shapes the tiled kernel does not take (other decimations, long prototypes):
    // overlap-save engine (fft_kernels.hip) + rotator table multiply
    bool use_ols = false, prefer_ols = false, use_hidec = false;
    int ols_L = 0, ols_fold = 0;
    DevBuf d_ols_tw, d_ols_H, d_hidec_taps, d_hidec_etab, d_hidec_vtab;
    bool hidec_premix = false;
    // FAST mode, real prototype, decimation 2 / 4, up to ~260 taps: matrix-core engine (fir_mfma.hip)
    bool use_mfma = false;
    int mf_kexp = 0;
    float mf_wstep[2] = {1.f, 0.f};
    DevBuf d_mf_A[2], d_mf_wlane[2], d_mf_stab, d_mf_vtab;     // [alignment parity]
    SchedBuf mf_sched;
    int mf_cu_cap = 0;                          // FirMfmaArgs::max_cus of the next launches (the chain's masked FIR stream)
    int mf_wg_cap = 0;                          // FirMfmaArgs::max_wg_per_cu of the next launches (the chain's pipeline sets 1)
    DevBuf scratch_y;
    SchedBuf sched;                             // tile queue of the tiled kernel (one launch at a time per handle)

    // rotator table: phases of outputs [tab_start, tab_start+tab_len)
    long long pos = 0;                          // outputs produced since construction/reset
    long long tab_start = 0, tab_len = 0;
    std::complex<float> gen_phase{1.f, 0.f};    // generator state at tab_start+tab_len
    unsigned gen_counter = 0;
    std::complex<float> built_incr{0.f, 0.f};
    DevBuf d_rot;

    int build(int device);
    // set_center_freq/set_taps path: new taps and increment, rotator phase and
    // counter carry on (only set_phase_incr is called, .cc.t:82)
    int rebuild_keep_phase(int device)
    {
        std::complex<float> ph(1.f, 0.f);
        unsigned cnt = 0;
        long long tab_end = tab_start + tab_len;
        if (pos == tab_end) { ph = gen_phase; cnt = gen_counter; }
        else if (pos >= tab_start && pos < tab_end) {
            hipError_t e = hipMemcpy(&ph, d_rot.as<std::complex<float>>() + (pos - tab_start), sizeof(ph),
                                     hipMemcpyDeviceToHost);
            if (e != hipSuccess) return fail(GRHIP_ERUNTIME, "rotator state readback failed");
            cnt = gen_counter - (unsigned)(tab_end - pos);
        }
        long long keep_pos = pos;
        int rc = build(device);          // resets pos and the table
        if (rc) return rc;
        pos = keep_pos; tab_start = keep_pos; tab_len = 0;
        gen_phase = ph; gen_counter = cnt;
        return GRHIP_OK;
    }
    void reset();
    int ensure_rot(long long n, const float2 **gtab, hipStream_t st);
    int phase_before_pos(std::complex<float> *g);
    // fused demodulator on the pre-mixed accumulators (EPI_DEMOD): FAST mode, real prototype taps
    // (single-stream calls of a long filter go through the overlap-save engine instead; batched
    // launches -- several streams, or history supplied by range check -- always take the tiled kernel)
    bool demod_is_direct(int mode, bool demod, bool batched = false) const
    {
        if (demod && mode_matrix(mode) && use_mfma) return true;
        if (demod && mode_fast(mode) && use_hidec && hidec_premix) return true;     // the direct kernel's fused demodulator
        return demod && mode_fast(mode) && use_tiled && premix && (batched || !prefer_ols);
    }
    // d_in item 0 = input[0] of output 0 (oldest history item); items with index
    // < n_lo or >= n_in read as zero.  n_streams > 1: stream s at d_in + s*x_stride,
    // outputs at d_y/d_demod + s*out_stride, y_prev[s] / y_last[s].
    int run(int mode, const float2 *d_in, long long n_in, long long n_out, float2 *d_y, float *d_demod,
            float gain, const float2 *y_prev, float2 *y_last, const float *atan_tab, hipStream_t st,
            int n_streams = 1, long long x_stride = 0, long long n_lo = 0, long long out_stride = 0);
    void release();
};

}  // namespace grhip
