// fft_any.hip -- gr_fft_vcc / gri_fft_complex for EVERY size the reference accepts (round 3).
//
// gri_fft_complex hands any fft_size > 0 to FFTW (general/gri_fft.cc:97-123); the radix-16 register
// kernels of fft_kernels.hip take powers of two up to 8192.  FftPlan closes the gap on top of them:
//   * DIRECT     N <= 128 that is not a power of two: the N x N DFT itself, one lane per bin, the
//                vectors and the N-entry twiddle table in LDS (8 N flop per sample: cheap at these N);
//   * FOURSTEP   powers of two above 8192 (up to 2^26): N = N1 N2, three transposes (window /
//                ifft-shift folded into the first, the W_N^{n2 k1} twiddles into the second, the
//                fft-shift into the third) around two batched launches of the register kernels;
//   * BLUESTEIN  everything else: the DFT as a convolution with a chirp,
//                e^{s 2 pi i nk/N} = c[n] c[k] conj(c[k-n]),  c[n] = e^{s pi i n^2 / N},
//                carried out with power-of-two transforms of L >= 2N - 1 points (native or four-step);
//                chirp and transformed chirp come from the host in double (n^2 reduced mod 2N in
//                integers, so the phase keeps its accuracy at any N).
// Parity: the reference's result is FFTW's, planner dependent and not in the tree; like the
// power-of-two kernels these paths are checked against a float64 DFT at 1e-6 log2 N of the spectrum's
// peak (tests/test_gpu_fft_pfb.py).
#include <cmath>
#include <complex>
#include <vector>

#include "device_math.h"
#include "fft_kernels.h"
#include "grhip_internal.h"

namespace grhip {

namespace {

constexpr int TLO_BITS = 13, TLO = 1 << TLO_BITS;      // two-level twiddle table: W_N^m = T_hi[m >> 13] T_lo[m & 8191]

// ---- direct DFT, N <= 256 -------------------------------------------------------------------------
template <bool FWD>
__global__ void __launch_bounds__(256)
dft_direct_kernel(int N, int vpw, int ishift, int oshift, const float *__restrict__ window, const float2 *__restrict__ tw,
                  const float2 *__restrict__ in, float2 *__restrict__ out, long long nvec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *T = reinterpret_cast<float2 *>(smem);          // [N] e^{-2 pi i k / N}
    float2 *X = T + N;                                     // [vpw][N]
    const int t = threadIdx.x;
    const long long v0 = (long long)blockIdx.x * vpw;
    for (int i = t; i < N; i += 256) T[i] = tw[i];
    const int h = N / 2;                                   // floor(N / 2): both shifts move by it
    for (int i = t; i < vpw * N; i += 256) {
        const int v = i / N, n = i - v * N;
        float2 x = make_float2(0.f, 0.f);
        if (v0 + v < nvec) {
            int src = n;
            if (ishift) { src = n + h; if (src >= N) src -= N; }
            x = in[(v0 + v) * N + src];
            if (window) { const float wv = window[n]; x.x *= wv; x.y *= wv; }
        }
        X[i] = x;
    }
    __syncthreads();
    const int v = t / N, k = t - v * N;
    if (v < vpw && v0 + v < nvec) {
        const float2 *xv = X + v * N;
        float ax = 0.f, ay = 0.f;
        int m = 0;
        for (int n = 0; n < N; ++n) {
            float2 w = T[m];
            if (!FWD) w.y = -w.y;
            const float2 x = xv[n];
            ax = __builtin_fmaf(x.x, w.x, ax); ax = __builtin_fmaf(-x.y, w.y, ax);
            ay = __builtin_fmaf(x.x, w.y, ay); ay = __builtin_fmaf(x.y, w.x, ay);
            m += k; if (m >= N) m -= N;
        }
        int dst = k;
        if (oshift) { dst = k + h; if (dst >= N) dst -= N; }
        out[(v0 + v) * N + dst] = make_float2(ax, ay);
    }
}

// ---- batched transposes of the four-step form ------------------------------------------------------
// out[v][c][r] = f(in[v][r][c]), R rows x C columns per vector, both multiples of 32.
// MODE 0: input side   -- window[n] or the ifft-shift (n = r C + c is the sample index)
// MODE 1: twiddle      -- times W_N^{r c} (conjugated for the backward transform)
// MODE 2: output side  -- the bin index is c R + r; fft-shift on the way out
template <int MODE, bool FWD>
__global__ void __launch_bounds__(256)
transpose_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int R, int C, long long nvec, int shift,
                 const float *__restrict__ window, const float2 *__restrict__ thi, const float2 *__restrict__ tlo)
{
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int tiles_c = C >> 5;
    const int tr = (int)(blockIdx.x / tiles_c), tc = (int)(blockIdx.x - (unsigned)tr * tiles_c);
    const long long N = (long long)R * C;
    for (long long v = blockIdx.y; v < nvec; v += gridDim.y) {
        const float2 *x = in + v * N;
        float2 *y = out + v * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = tr * 32 + ty + 8 * i, c = tc * 32 + tx;
            long long n = (long long)r * C + c;
            float2 val;
            if (MODE == 0) {
                if (window) {
                    val = x[n];
                    const float wv = window[n];
                    val.x *= wv; val.y *= wv;
                } else {
                    if (shift) { n += N / 2; if (n >= N) n -= N; }
                    val = x[n];
                }
            } else {
                val = x[n];
                if (MODE == 1) {
                    const long long m = (long long)r * c;
                    float2 w = cmul_fma(thi[m >> TLO_BITS], tlo[m & (TLO - 1)]);
                    if (!FWD) w.y = -w.y;
                    val = cmul_fma(val, w);
                }
            }
            tile[ty + 8 * i][tx] = val;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tc * 32 + ty + 8 * i, r = tr * 32 + tx;
            long long k = (long long)c * R + r;
            if (MODE == 2 && shift) { k += N / 2; if (k >= N) k -= N; }
            y[k] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

// ---- Bluestein ---------------------------------------------------------------------------------------
// a[v][i] = x[v][src(i)] window[i] c[i]  (i < N), 0 (N <= i < L)
__global__ void __launch_bounds__(256)
blu_pre_kernel(const float2 *__restrict__ in, float2 *__restrict__ a, int N, int L, long long nvec, int ishift,
               const float *__restrict__ window, const float2 *__restrict__ chirp)
{
    const int h = N / 2;
    for (long long v = blockIdx.y; v < nvec; v += gridDim.y) {
        const float2 *x = in + v * (long long)N;
        float2 *o = a + v * (long long)L;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
            float2 r = make_float2(0.f, 0.f);
            if (i < N) {
                int src = i;
                if (ishift) { src = i + h; if (src >= N) src -= N; }
                float2 xv = x[src];
                if (window) { const float wv = window[i]; xv.x *= wv; xv.y *= wv; }
                r = cmul_fma(xv, chirp[i]);
            }
            o[i] = r;
        }
    }
}

__global__ void __launch_bounds__(256)
blu_mul_kernel(float2 *__restrict__ a, const float2 *__restrict__ B, int L, long long nvec)
{
    for (long long v = blockIdx.y; v < nvec; v += gridDim.y) {
        float2 *o = a + v * (long long)L;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) o[i] = cmul_fma(o[i], B[i]);
    }
}

// out[v][dst(k)] = c[k] s[v][k], k < N
__global__ void __launch_bounds__(256)
blu_post_kernel(const float2 *__restrict__ s, float2 *__restrict__ out, int N, int L, long long nvec, int oshift,
                const float2 *__restrict__ chirp)
{
    const int h = N / 2;
    for (long long v = blockIdx.y; v < nvec; v += gridDim.y) {
        const float2 *x = s + v * (long long)L;
        float2 *o = out + v * (long long)N;
        for (int k = blockIdx.x * 256 + threadIdx.x; k < N; k += gridDim.x * 256) {
            int dst = k;
            if (oshift) { dst = k + h; if (dst >= N) dst -= N; }
            o[dst] = cmul_fma(x[k], chirp[k]);
        }
    }
}

int up(DevBuf &b, const void *src, size_t bytes)
{
    int rc = b.reserve(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) GRHIP_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return GRHIP_OK;
}

void fwd_table(int N, std::vector<float2> &tw)
{
    tw.resize((size_t)N);
    for (int k = 0; k < N; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)N;
        tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
}

unsigned grid_y(long long nvec) { return (unsigned)(nvec > 4096 ? 4096 : nvec); }

}  // namespace

// in-place radix-2 transform of the host (double), n a power of two; sign -1 = forward
void host_fft_pow2(std::vector<std::complex<double>> &a, int sign)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        // twiddles of this stage straight from cos / sin (no recurrence: accuracy over speed, set-up only)
        std::vector<std::complex<double>> w(len / 2);
        for (size_t k = 0; k < len / 2; ++k) {
            const double ang = sign * 2.0 * M_PI * (double)k / (double)len;
            w[k] = std::complex<double>(cos(ang), sin(ang));
        }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w[k];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

bool FftPlan::size_ok(long long N) { return N >= 1 && N <= (1ll << 26) && (((N & (N - 1)) == 0) || N <= (1ll << 25)); }

int FftPlan::build(int n, int fwd)
{
    release();
    N = n; forward = fwd ? 1 : 0;
    if (n < 1) return fail(GRHIP_ERANGE, "gri_fftw: invalid fft_size");
    if (!size_ok(n)) return fail(GRHIP_EINVAL, "fft_size %d: more than 2^26 points (2^25 when not a power of two)", n);
    const bool pow2 = (n & (n - 1)) == 0;
    std::vector<float2> tw;
    if (pow2 && n <= 8192) {
        kind = NATIVE;
        fwd_table(n, tw);
        return up(d_tw, tw.data(), tw.size() * sizeof(float2));
    }
    if (!pow2 && n <= 128) {
        kind = DIRECT;
        fwd_table(n, tw);
        return up(d_tw, tw.data(), tw.size() * sizeof(float2));
    }
    if (pow2) {
        kind = FOURSTEP;
        int lg = 0;
        while ((1 << lg) < n) ++lg;
        N1 = 1 << ((lg + 1) / 2); N2 = n / N1;
        fwd_table(N1, tw);
        int rc = up(d_tw, tw.data(), tw.size() * sizeof(float2));
        if (rc) return rc;
        fwd_table(N2, tw);
        if ((rc = up(d_tw2, tw.data(), tw.size() * sizeof(float2)))) return rc;
        std::vector<float2> hi((size_t)(n >> TLO_BITS)), lo((size_t)TLO);
        for (size_t i = 0; i < hi.size(); ++i) {
            const double ang = -2.0 * M_PI * (double)(i << TLO_BITS) / (double)n;
            hi[i] = make_float2((float)cos(ang), (float)sin(ang));
        }
        for (size_t i = 0; i < lo.size(); ++i) {
            const double ang = -2.0 * M_PI * (double)i / (double)n;
            lo[i] = make_float2((float)cos(ang), (float)sin(ang));
        }
        if ((rc = up(d_thi, hi.data(), hi.size() * sizeof(float2)))) return rc;
        return up(d_tlo, lo.data(), lo.size() * sizeof(float2));
    }
    kind = BLUESTEIN;
    L = 1;
    while (L < 2 * n - 1) L <<= 1;
    sub = new (std::nothrow) FftPlan();
    if (!sub) return fail(GRHIP_ENOMEM, "alloc");
    int rc = sub->build(L, 1);
    if (rc) return rc;
    // chirp c[i] = e^{s pi i i^2 / N}, s = -1 forward: i^2 mod 2N in integers
    const double s = forward ? -1.0 : 1.0;
    std::vector<float2> c((size_t)n);
    std::vector<std::complex<double>> b((size_t)L, std::complex<double>(0, 0));
    for (long long i = 0; i < n; ++i) {
        const long long q = (i * i) % (2ll * n);
        const double ang = s * M_PI * (double)q / (double)n;
        c[(size_t)i] = make_float2((float)cos(ang), (float)sin(ang));
        const std::complex<double> bc(cos(ang), -sin(ang));          // conj(c[i]) = b[i] = b[-i]
        b[(size_t)i] = bc;
        if (i) b[(size_t)(L - i)] = bc;
    }
    host_fft_pow2(b, -1);
    std::vector<float2> B((size_t)L);
    for (int i = 0; i < L; ++i) B[(size_t)i] = make_float2((float)(b[(size_t)i].real() / L), (float)(b[(size_t)i].imag() / L));
    if ((rc = up(d_chirp, c.data(), c.size() * sizeof(float2)))) return rc;
    return up(d_B, B.data(), B.size() * sizeof(float2));
}

void FftPlan::release()
{
    d_tw.release(); d_tw2.release(); d_thi.release(); d_tlo.release(); d_chirp.release(); d_B.release();
    d_s1.release(); d_s2.release();
    if (sub) { sub->release(); delete sub; sub = nullptr; }
}

// one power-of-two transform of the plan's size, either direction (NATIVE / FOURSTEP only)
int FftPlan::exec_pow2(int fwd, int shift, const float *window, const float2 *in, float2 *out, long long nvec, hipStream_t st)
{
    if (kind == NATIVE) return launch_fft(N, fwd, shift, window, d_tw.as<float2>(), in, out, nvec, st);
    if (kind != FOURSTEP) return fail(GRHIP_EINVAL, "fft plan: not a power of two");
    long long chunk = (1ll << 25) / N;
    if (chunk < 1) chunk = 1;
    if (chunk > nvec) chunk = nvec;
    int rc = d_s1.reserve((size_t)chunk * N * sizeof(float2));
    if (!rc) rc = d_s2.reserve((size_t)chunk * N * sizeof(float2));
    if (rc) return rc;
    float2 *S1 = d_s1.as<float2>(), *S2 = d_s2.as<float2>();
    const float2 *thi = d_thi.as<float2>(), *tlo = d_tlo.as<float2>();
    for (long long v0 = 0; v0 < nvec; v0 += chunk) {
        const long long nv = nvec - v0 < chunk ? nvec - v0 : chunk;
        const float2 *x = in + v0 * N;
        float2 *y = out + v0 * N;
        const dim3 g((unsigned)((N1 / 32) * (N2 / 32)), grid_y(nv));
        // a: A[n1][n2] -> At[n2][n1]   (R = N1 rows, C = N2 columns); window / ifft-shift (the reference applies
        //    the shift only without a window and only backward: gr_fft_vcc_fftw.cc:68-83)
        hipLaunchKernelGGL((transpose_kernel<0, true>), g, dim3(256), 0, st, x, S1, N1, N2, nv, (!fwd && shift) ? 1 : 0, window,
                           thi, tlo);
        // b: N2 rows of N1 points
        if ((rc = launch_fft(N1, fwd, 0, nullptr, d_tw.as<float2>(), S1, S1, nv * N2, st))) return rc;
        // c: Bt[n2][k1] W_N^{n2 k1} -> C[k1][n2]   (R = N2, C = N1)
        if (fwd) hipLaunchKernelGGL((transpose_kernel<1, true>), g, dim3(256), 0, st, S1, S2, N2, N1, nv, 0, nullptr, thi, tlo);
        else hipLaunchKernelGGL((transpose_kernel<1, false>), g, dim3(256), 0, st, S1, S2, N2, N1, nv, 0, nullptr, thi, tlo);
        // d: N1 rows of N2 points
        if ((rc = launch_fft(N2, fwd, 0, nullptr, d_tw2.as<float2>(), S2, S2, nv * N1, st))) return rc;
        // e: D[k1][k2] -> X[k1 + N1 k2]   (R = N1, C = N2); fft-shift forward only (.cc:89-96)
        hipLaunchKernelGGL((transpose_kernel<2, true>), g, dim3(256), 0, st, S2, y, N1, N2, nv, (fwd && shift) ? 1 : 0, nullptr,
                           thi, tlo);
        GRHIP_HIP(hipGetLastError());
    }
    return GRHIP_OK;
}

int FftPlan::exec(int shift, const float *window, const float2 *in, float2 *out, long long nvec, hipStream_t st)
{
    if (nvec <= 0) return GRHIP_OK;
    if (kind == NATIVE || kind == FOURSTEP) return exec_pow2(forward, shift, window, in, out, nvec, st);
    const int ishift = (!forward && shift && !window) ? 1 : 0, oshift = (forward && shift) ? 1 : 0;
    if (kind == DIRECT) {
        const int vpw = 256 / N > 0 ? 256 / N : 1;
        const size_t lds = (size_t)(vpw + 1) * N * sizeof(float2);
        const unsigned grid = (unsigned)((nvec + vpw - 1) / vpw);
        if (forward) hipLaunchKernelGGL(dft_direct_kernel<true>, dim3(grid), dim3(256), lds, st, N, vpw, ishift, oshift, window,
                                        d_tw.as<float2>(), in, out, nvec);
        else hipLaunchKernelGGL(dft_direct_kernel<false>, dim3(grid), dim3(256), lds, st, N, vpw, ishift, oshift, window,
                                d_tw.as<float2>(), in, out, nvec);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
    // BLUESTEIN
    long long chunk = (1ll << 25) / L;
    if (chunk < 1) chunk = 1;
    if (chunk > nvec) chunk = nvec;
    int rc = d_s1.reserve((size_t)chunk * L * sizeof(float2));
    if (rc) return rc;
    float2 *S = d_s1.as<float2>();
    const unsigned gx = (unsigned)((L + 255) / 256 > 1024 ? 1024 : (L + 255) / 256);
    for (long long v0 = 0; v0 < nvec; v0 += chunk) {
        const long long nv = nvec - v0 < chunk ? nvec - v0 : chunk;
        const dim3 g(gx, grid_y(nv));
        hipLaunchKernelGGL(blu_pre_kernel, g, dim3(256), 0, st, in + v0 * N, S, N, L, nv, ishift, window, d_chirp.as<float2>());
        if ((rc = sub->exec_pow2(1, 0, nullptr, S, S, nv, st))) return rc;
        hipLaunchKernelGGL(blu_mul_kernel, g, dim3(256), 0, st, S, d_B.as<float2>(), L, nv);
        if ((rc = sub->exec_pow2(0, 0, nullptr, S, S, nv, st))) return rc;
        hipLaunchKernelGGL(blu_post_kernel, g, dim3(256), 0, st, S, out + v0 * N, N, L, nv, oshift, d_chirp.as<float2>());
        GRHIP_HIP(hipGetLastError());
    }
    return GRHIP_OK;
}

}  // namespace grhip
