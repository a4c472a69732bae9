// fft_kernels.h -- launchers for gr_fft_vcc and gr_pfb_channelizer_ccf (internal).
#pragma once
#include "grhip_internal.h"
#include <hip/hip_runtime.h>

namespace grhip {

// One workgroup per vector, radix-4/radix-2 Stockham in LDS.
// twiddle: device float2[N], twiddle[k] = exp(-2*pi*i*k/N) (forward sign; the
// backward transform conjugates on the fly).  window: device float[N] or null.
// shift semantics as gr_fft_vcc_fftw::work (general/gr_fft_vcc_fftw.cc:68-96).
int launch_fft(int N, int forward, int shift, const float *window, const float2 *twiddle,
               const float2 *in, float2 *out, long long nvec, hipStream_t st);
bool fft_size_supported(int N);

// A transform of ANY size the reference's gri_fft_complex accepts (general/gri_fft.cc:97-123), on top of launch_fft:
// fft_any.hip.  `forward` is fixed at build time for the Bluestein kind (its transformed chirp depends on the sign);
// exec_pow2 takes either direction (gr_fft_filter_ccc needs both of one power-of-two size).
struct FftPlan {
    enum Kind { NATIVE, DIRECT, FOURSTEP, BLUESTEIN };
    int N = 0, forward = 1;
    Kind kind = NATIVE;
    int N1 = 0, N2 = 0;         // FOURSTEP: N = N1 N2
    int L = 0;                  // BLUESTEIN: convolution length (power of two >= 2N - 1)
    FftPlan *sub = nullptr;     // BLUESTEIN: the plan of size L
    DevBuf d_tw, d_tw2, d_thi, d_tlo, d_chirp, d_B, d_s1, d_s2;
    static bool size_ok(long long N);
    int build(int N, int forward);
    void release();
    int exec(int shift, const float *window, const float2 *in, float2 *out, long long nvec, hipStream_t st);
    int exec_pow2(int fwd, int shift, const float *window, const float2 *in, float2 *out, long long nvec, hipStream_t st);
};

}  // namespace grhip
#include <complex>
namespace grhip {
// in-place radix-2 transform on the host, in double (set-up work: transformed taps, Bluestein's chirp); sign -1 = forward
void host_fft_pow2(std::vector<std::complex<double>> &a, int sign);

struct PfbArgs {
    int M;              // numchans
    int tpf;            // taps per filter
    int rate_ratio;     // (int)rintf(M / oversample_rate)
    const float *ftaps; // [M][tpf] reversed taps (as gr_fir_ccf stores them)
    const int *idxlut;  // [M]
    const float2 *dft;  // [M] exp(+2*pi*i*m/M), followed by [M] exp(-2*pi*i*m/M) (the forward table launch_fft takes)
    const float2 *in;   // stream j at in + j*stride, item 0 = oldest history item
    long long stride;
    float2 *out;        // [nout][M]
    long long nout;
    // launch_pfb_hier only: `in` is the single interleaved stream (stream j's item m = in[m M + j], tpf M history items in front)
    float2 *out_streams = nullptr;      // channel k's stream at out_streams + k * out_stride
    long long out_stride = 0;
    // pfb_os1_kernel's view of one sub-sequence of an integer-oversampled channeliser (filled in by launch_pfb)
    int sub_r = 0, sub_os = 1, sub_q = 0, sub_last = 0;
    long long in_items = 0;             // readable items per stream
};
int launch_pfb(const PfbArgs &a, hipStream_t st);
// blks2's hier block in one pass (stream_to_streams -> pfb -> vector_to_streams); GRHIP_OK, < 0 on error, or -1 when
// the shape has no fused kernel (the caller then runs the three blocks)
int launch_pfb_hier(const PfbArgs &a, hipStream_t st);

// gr_fft_filter_ccc, fused overlap-save on 4096-point blocks (ntaps <= 2049)
constexpr int OLS_N = 4096, OLS_MAX_TAPS = 2049;
// twiddle table and transformed taps (scaled by 1/4096) for launch_fftfilt4096; taps = ntaps complex floats
// (interleaved), h[k] multiplies x[i-k].  L = outputs per block at full rate, a multiple of `decim`.
int ols_build(const float *taps_cplx, int ntaps, int decim, DevBuf &d_tw, DevBuf &d_H, int *L, int *fold);
// hist_new (optional): receives the last ntaps-1 items of (hist ++ in), the history of the next call
int launch_fftfilt4096(const float2 *in, long long nin, const float2 *hist, int ntaps, const float2 *twiddle,
                       const float2 *H, float2 *out, long long nout, int decim, int L, int fold, hipStream_t st,
                       float2 *hist_new = nullptr);
int launch_fftfilt4096_real(const float *in, long long nin, const float *hist, int ntaps, const float2 *twiddle,
                            const float2 *H, float *out, long long nout, int decim, int L, int fold, hipStream_t st);
int launch_fftfilt_hist(const float2 *in, long long nin, const float2 *hist_old, float2 *hist_new, int hlen, hipStream_t st);
// gr_fft_filter_ccc helpers (overlap-add around launch_fft)
int launch_fftfilt_pack(const float2 *in, float2 *blocks, int nsamples, int fftsize, long long nblk, hipStream_t st);
int launch_fftfilt_mul(float2 *blocks, const float2 *xformed, int fftsize, long long nblk, hipStream_t st);
int launch_fftfilt_ola(const float2 *blocks, const float2 *tail, float2 *out, long long nitems, int decim, int nsamples,
                       int fftsize, int tailsize, hipStream_t st);
int launch_fftfilt_tail(const float2 *blocks, float2 *tail, long long nblk, int nsamples, int fftsize, int tailsize,
                        hipStream_t st);

}  // namespace grhip
