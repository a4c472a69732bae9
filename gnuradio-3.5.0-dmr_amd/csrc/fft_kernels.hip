// fft_kernels.hip -- gfx950 kernels for gr_fft_vcc and gr_pfb_channelizer_ccf.
#include "fft_kernels.h"

#include "device_math.h"
#include "grhip_internal.h"

namespace grhip {

// ===========================================================================
// gr_fft_vcc (general/gr_fft_vcc_fftw.cc:55-103; transform = FFTW3f c2c,
// unnormalised, general/gri_fft.cc:119-123).
// One 256-lane workgroup per vector.  The vector is loaded once from HBM
// (window / ifftshift folded into the load), transformed by radix-4 Stockham
// passes (a final radix-2 pass when log2 N is odd) ping-ponging between two LDS
// buffers, and stored once (fftshift folded into the store): 16 B of HBM
// traffic per sample.  Twiddles come from a table computed in double on the
// host.
// ===========================================================================
bool fft_size_supported(int N) { return N >= 1 && N <= 8192 && (N & (N - 1)) == 0; }

template <bool FWD>
__device__ __forceinline__ float2 tw(const float2 *__restrict__ table, int m)
{
    float2 w = table[m];
    if (!FWD) w.y = -w.y;
    return w;
}

template <bool FWD>
__global__ void __launch_bounds__(256)
fft_kernel(int N, int shift, const float *__restrict__ window, const float2 *__restrict__ twiddle,
           const float2 *__restrict__ in, float2 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = (float2 *)smem;
    float2 *B = A + N;
    const int t = threadIdx.x;
    const float2 *__restrict__ x = in + (long long)blockIdx.x * N;
    float2 *__restrict__ y = out + (long long)blockIdx.x * N;

    // ---- load (gr_fft_vcc_fftw.cc:68-83)
    if (window) {
        for (int i = t; i < N; i += 256) {
            float2 v = x[i];
            float w = window[i];
            A[i] = make_float2(v.x * w, v.y * w);
        }
    } else if (!FWD && shift) {
        const int len = N / 2;                 // floor(N/2.0); dst[k] = in[(k+len) mod N]
        for (int i = t; i < N; i += 256) {
            int src = i + len; if (src >= N) src -= N;
            A[i] = x[src];
        }
    } else {
        for (int i = t; i < N; i += 256) A[i] = x[i];
    }
    __syncthreads();

    // ---- Stockham passes
    float2 *src = A, *dst = B;
    int p = 1;
    const int T4 = N >> 2;
    while (p * 4 <= N) {
        const int tstep = N / (4 * p);         // twiddle index step
        for (int i = t; i < T4; i += 256) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const int m = k * tstep;
            float2 u0 = src[i];
            float2 u1 = src[i + T4];
            float2 u2 = src[i + 2 * T4];
            float2 u3 = src[i + 3 * T4];
            if (p > 1) {
                u1 = cmul_fma(u1, tw<FWD>(twiddle, m));
                u2 = cmul_fma(u2, tw<FWD>(twiddle, 2 * m));
                u3 = cmul_fma(u3, tw<FWD>(twiddle, 3 * m));
            }
            float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
            float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
            float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
            float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
            // multiply by -i (forward) or +i (backward)
            float2 v3 = FWD ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x);
            dst[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
            dst[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
            dst[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
            dst[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
        p <<= 2;
    }
    if (p < N) {                                // one radix-2 pass, p == N/2
        const int T2 = N >> 1;
        const int tstep = N / (2 * p);
        for (int i = t; i < T2; i += 256) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 1) + k;
            float2 u0 = src[i];
            float2 u1 = src[i + T2];
            if (p > 1) u1 = cmul_fma(u1, tw<FWD>(twiddle, k * tstep));
            dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
    }

    // ---- store (gr_fft_vcc_fftw.cc:89-96)
    if (FWD && shift) {
        const int len = (N + 1) / 2;            // ceil(N/2.0); out[k] = fft[(k+len) mod N]
        for (int i = t; i < N; i += 256) {
            int s = i + len; if (s >= N) s -= N;
            y[i] = src[s];
        }
    } else {
        for (int i = t; i < N; i += 256) y[i] = src[i];
    }
}

int launch_fft(int N, int forward, int shift, const float *window, const float2 *twiddle, const float2 *in,
               float2 *out, long long nvec, hipStream_t st)
{
    if (nvec <= 0) return GRHIP_OK;
    if (!fft_size_supported(N)) return fail(GRHIP_EINVAL, "fft size %d not supported on device", N);
    size_t lds = (size_t)N * 2 * sizeof(float2);
    static size_t cfg_f = 0, cfg_b = 0;
    if (forward) {
        if (lds > 48 * 1024 && lds > cfg_f) {
            GRHIP_HIP(hipFuncSetAttribute((const void *)fft_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cfg_f = lds;
        }
        hipLaunchKernelGGL(fft_kernel<true>, dim3((unsigned)nvec), dim3(256), lds, st, N, shift, window,
                           twiddle, in, out);
    } else {
        if (lds > 48 * 1024 && lds > cfg_b) {
            GRHIP_HIP(hipFuncSetAttribute((const void *)fft_kernel<false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cfg_b = lds;
        }
        hipLaunchKernelGGL(fft_kernel<false>, dim3((unsigned)nvec), dim3(256), lds, st, N, shift, window,
                           twiddle, in, out);
    }
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_pfb_channelizer_ccf::general_work (filter/gr_pfb_channelizer_ccf.cc:160-199)
// The commutator state of the reference loop has a closed form in the output
// index t:  c = (t+1)*rate_ratio - 1,  last = c mod M,  n = 1 + c div M
//   stream j <= last : filter last-j     on &in_j[n]
//   stream j >  last : filter M+last-j   on &in_j[n-1]
// result -> IFFT slot idxlut[j]; out[t][k] = sum_s slot[s] * exp(+2 pi i s k / M).
// Lane (j, ty) filters stream j for output vector t (generic gr_fir_ccf order,
// unfused: the filter part is bit-exact), the M lanes of a row then each
// produce one bin of the M-point backward DFT from LDS.
// ===========================================================================
__global__ void __launch_bounds__(1024)
pfb_kernel(const PfbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *slots = (float2 *)smem;             // [blockDim.y][M]
    const int M = a.M, tpf = a.tpf;
    const int j = threadIdx.x, ty = threadIdx.y;
    const long long t = (long long)blockIdx.x * blockDim.y + ty;
    const bool active = t < a.nout;

    if (active) {
        const long long c = (t + 1) * (long long)a.rate_ratio - 1;
        const int last = (int)(c % M);
        const long long n = 1 + c / M;
        int filt; long long pos;
        if (j <= last) { filt = last - j; pos = n; }
        else           { filt = M + last - j; pos = n - 1; }
        const float *__restrict__ dt = a.ftaps + (size_t)filt * tpf;
        const float2 *__restrict__ x = a.in + (long long)j * a.stride + pos;
        // gr_fir_ccf_generic::filter (2 complex accumulators, .cc.t:59-79)
        float a0r = 0, a0i = 0, a1r = 0, a1i = 0;
        int i = 0, nn = (tpf / 2) * 2;
        for (i = 0; i < nn; i += 2) {
            float2 v0 = x[i], v1 = x[i + 1];
            float t0 = dt[i], t1 = dt[i + 1];
            float pr = v0.x * t0, pi = v0.y * t0;
            a0r += pr; a0i += pi;
            pr = v1.x * t1; pi = v1.y * t1;
            a1r += pr; a1i += pi;
        }
        for (; i < tpf; i++) {
            float2 v0 = x[i];
            float t0 = dt[i];
            float pr = v0.x * t0, pi = v0.y * t0;
            a0r += pr; a0i += pi;
        }
        slots[ty * M + a.idxlut[j]] = make_float2(a0r + a1r, a0i + a1i);
    }
    __syncthreads();
    if (active) {
        const int k = j;
        const float2 *row = slots + ty * M;
        float2 acc = make_float2(0.f, 0.f);
        int ph = 0;                              // (s*k) mod M
        for (int s = 0; s < M; ++s) {
            float2 w = a.dft[ph];
            float2 v = row[s];
            acc.x = __builtin_fmaf(v.x, w.x, acc.x);
            acc.x = __builtin_fmaf(-v.y, w.y, acc.x);
            acc.y = __builtin_fmaf(v.x, w.y, acc.y);
            acc.y = __builtin_fmaf(v.y, w.x, acc.y);
            ph += k; if (ph >= M) ph -= M;
        }
        a.out[t * M + k] = acc;
    }
}

int launch_pfb(const PfbArgs &a, hipStream_t st)
{
    if (a.nout <= 0) return GRHIP_OK;
    if (a.M < 1 || a.M > 1024) return fail(GRHIP_EINVAL, "numchans %d not supported on device", a.M);
    int ty = 256 / a.M; if (ty < 1) ty = 1;
    dim3 block(a.M, ty);
    dim3 grid((unsigned)((a.nout + ty - 1) / ty));
    size_t lds = (size_t)a.M * ty * sizeof(float2);
    hipLaunchKernelGGL(pfb_kernel, grid, block, lds, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

}  // namespace grhip
