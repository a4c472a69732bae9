// fir_kernels.h -- launchers of the FIR-family kernels (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace grhip {

enum FirKind { FIR_FFF = 0, FIR_CCF = 1, FIR_CCC = 2 };

// ---- (A) generic-order kernel: bit-exact gr_fir_XXX_generic -----------------
// taps_rev: device, d_taps order (reversed forward taps); complex interleaved
// for CCC.  in: device, item 0 = input[0] of output 0.  Any decimation.
// epilogue: if gtab != nullptr (complex kinds) out[n] = rotate(out[n], gtab[n])
// with the reference's unfused complex product (gr_rotator.h:43).
// seq: one accumulator, terms in order (gri_fir_filter_with_buffer_XXX.cc.t:75-79) instead of the unrolled order
int launch_fir_generic(FirKind kind, const float *taps_rev, int ntaps, const void *in, void *out,
                       long long n_out, int decim, const float2 *gtab, hipStream_t st, bool seq = false);
// gr_fir_ccc_generic + rotator (gtab) + gr_quadrature_demod_cf fused, bit-exact (decimation 1 / 2 / 4, the tiled kernels'
// shapes, y_prev != y_last); returns 1 where there is no such kernel: the caller runs the two kernels instead
// n_streams > 1 (or n_lo > 0): stream s at in + s * x_stride (x_stride even), its first n_lo items are history zeros that
// are never read, outputs at d + s * out_stride, carries y_prev[s] / y_last[s]; the streams may start on an 8-byte boundary
bool generic_demod_batch_ok(int ntaps, int decim, const void *in, long long x_stride);
int launch_fir_generic_demod(const float *taps_rev, int ntaps, const void *in, float *d, long long n_out, int decim,
                             const float2 *gtab, float gain, const float *atan_tab, const float2 *y_prev, float2 *y_last,
                             hipStream_t st, int n_streams = 1, long long x_stride = 0, long long out_stride = 0,
                             long long n_lo = 0);

// ---- (B) tiled fast kernel (complex data) ------------------------------------
struct FirTiledArgs {
    const float2 *x;        // stream 0; item 0 = input[0] of output 0
    long long x_stride;     // items between streams
    long long n_in;         // items at index >= n_in read as 0
    long long n_lo;         // items at index <  n_lo read as 0 (never dereferenced)
    const float *hp;        // phase-major padded taps [D][Tq] (x2 floats if complex)
    int Tq;                 // taps per phase, multiple of R
    long long n_out;        // outputs per stream
    const float2 *wtab;     // PREMIX: wtab[v+1] = e^{jw(v-D)}, v = -1..511 (tiled_wtab_len entries)
    const float2 *stab;     // PREMIX: stab[i] = e^{jw 512 i}, i < tiled_stab_len()
    const float2 *vtab;     // PREMIX: vtab[j] = e^{-jw j D}, j < NT; vtab[NT] = e^{+jwD} (boundary output)
    int n_streams;          // filled in by launch_fir_tiled
    const float2 *gtab;     // EPI 1/2: rotator phase per output of this call (stream-independent)
    float2 *y_out;          // EPI 0/1
    long long y_stride;
    float *d_out;           // EPI 2/3
    long long d_stride;
    float gain;             // EPI 2
    const float2 *y_prev;   // EPI 2: [n_streams] y[-1] carried in from the previous call
    float2 *y_last;         // EPI 2: [n_streams] receives the last y of this call (distinct buffer)
    const float *atan_tab;  // EPI 2
    int vec_store;          // 1 if output rows are 16-byte aligned
    int fpair;              // 0: complex items; 1/2: float-pair mode of gr_fir_fff with that decimation
    unsigned *sched;        // 2 zero-initialised counters owned by the caller (tile queue), or null: static split
};

// EPI_DEMOD (pre-mix form only): the demodulator works on the pre-mixed accumulators
// directly.  y[n] conj(y[n-1]) = acc[n] conj(acc[n-1]) * (v[j] conj(v[j-1])) * (g[n] conj(g[n-1]))
// and the last two factors are e^{-jwD} and the rotator step e^{+jwD}(1 + O(1e-7)), so neither
// the phase correction nor the rotator table is needed; the carry (y_prev / y_last) is kept
// in the frame of the un-rotated composite FIR output, acc * v.
enum { EPI_NONE = 0, EPI_ROTATE = 1, EPI_ROTATE_DEMOD = 2 /* retired: rotate, then the stand-alone demodulator */, EPI_DEMOD = 3 };

// returns GRHIP_OK or <0 ; `decim` must be one of tiled_supported_decim().
bool tiled_supported(int decim, int ntaps_padded_per_phase);
int tiled_R();                      // outputs per lane
int tiled_NT();                     // outputs per workgroup tile
int tiled_wtab_len();
int tiled_stab_len();
int tiled_load_span();              // samples covered by one round of 16-byte loads of a workgroup
int launch_fir_tiled(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a,
                     int n_streams, hipStream_t st);

// ---- (B2) matrix-core engine (fir_mfma.hip): real taps, complex data, decimation 2 / 4, up to ~260 taps ----
struct FirMfmaArgs {
    const float2 *x;        // stream 0; item 0 = input[0] of output 0
    long long x_stride;     // items between streams (even)
    long long n_in;         // items at index >= n_in read as 0
    long long n_lo;         // items at index <  n_lo read as 0 (never dereferenced)
    long long n_out;        // outputs per stream
    int n_streams;
    int off;                // (address of x / 8) & 1: parity of the stream's 16-byte alignment
    const void *A;          // mf::build_A() table for that parity
    int kexp;               // taps were scaled by 2^kexp
    const float2 *wlane;    // PREMIX: [256] e^{jw(2t - off)}
    float2 wstep;           // PREMIX: e^{jw}
    const float *stab;      // PREMIX: [rounds] e^{jw 512 i} (re, im)
    const float2 *vtab;     // PREMIX: [NTC] e^{-jw j D}
    const float2 *gtab;     // EPI_ROTATE: rotator phase per output of this call
    float2 *y_out;          // EPI_NONE / EPI_ROTATE
    long long y_stride;
    float *d_out;           // EPI_DEMOD
    long long d_stride;
    float gain;
    const float2 *y_prev;   // EPI_DEMOD: [n_streams] carry in (null: zeros)
    float2 *y_last;         // EPI_DEMOD: [n_streams] carry out (null: none)
    const float2 *ctaps;    // EPI_DEMOD with y_last: composite taps c[i] (multiplying x[nD + i]), T of them
    int T;
    const float *atan_tab;
    int vec_store;          // (unused: outputs go through buffer stores, dword alignment suffices)
    unsigned *sched;        // tile queue counters (as FirTiledArgs::sched), or null
    int max_wg_per_cu;      // 0: as many as fit (2); 1: leave half of every CU to a kernel running beside this one
    int max_cus;            // 0: the device's; else the CUs the stream may use (a stream with a CU mask)
    double omega;           // PREMIX: the angle step (read by the attribution probe of diagnostic builds only)
    int tapq;               // PREMIX: 1 = also the correction band of the reference's tap-angle quantisation (GRHIP_MODE_FAST_REFTAPS)
};
bool mfma_supported(int decim, int ntaps);
int launch_fir_mfma(int decim, int ntaps, bool premix, int epi, const FirMfmaArgs &a, hipStream_t st);

// high-decimation direct form (FAST mode): c[k] multiplies x[n*decim + k] (complex interleaved if ctaps), given
// as hidec_pad_taps() lays them out; x item 0 = input[0] of output 0, items >= n_in read as zero; optional rotator table
bool hidec_supported(int decim, int ntaps);
int hidec_outputs_per_tile(int decim, int ntaps);
void hidec_pad_taps(const float *taps_corr, int ntaps, int tw, int decim, std::vector<float> &out);
void hidec_premix_tables(double omega, int decim, std::vector<float> &etab, std::vector<float> &vtab);
// etab / vtab (both or neither): pre-mix form for freq_xlating with a real prototype -- taps_padded are then the REAL
// prototype taps and the result is the band-pass sum (times gtab)
// fused xlating -> quadrature demodulator of the pre-mix form (real prototype): d_out[n], carry in / out in the composite frame
int launch_fir_hidec_demod(const float *taps_padded, int ntaps, int decim, const float2 *x, long long n_in, float *d_out,
                           long long n_out, float gain, const float2 *y_prev, float2 *y_last, const float *atan_tab,
                           const float2 *etab, const float2 *vtab, hipStream_t st, int n_streams = 1, long long x_stride = 0,
                           long long d_stride = 0, long long n_lo = 0, int max_wg_per_cu = 0, int max_cus = 0);
int launch_fir_hidec(bool ctaps, const float *taps_padded, int ntaps, int decim, const float2 *x, long long n_in, float2 *y,
                     long long n_out, const float2 *gtab, hipStream_t st, const float2 *etab = nullptr,
                     const float2 *vtab = nullptr);

// in-place rotator multiply with a phase table (gr_rotator.h:43)
int launch_rotate(float2 *y, const float2 *gtab, long long n, hipStream_t st);

// standalone quadrature demod: in has 1 history item in front
int launch_quad_demod(const float2 *in, float *out, long long n_out, float gain,
                      const float *atan_tab, hipStream_t st);

}  // namespace grhip
