// fft_kernels.hip -- gfx950 kernels for gr_fft_vcc and gr_pfb_channelizer_ccf.
#include "fft_kernels.h"

#include <cmath>
#include <vector>

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
    const int t = threadIdx.x, nthr = blockDim.x;      // N/4 lanes (64..256): small transforms get more workgroups per CU
    const float2 *__restrict__ x = in + (long long)blockIdx.x * N;
    float2 *__restrict__ y = out + (long long)blockIdx.x * N;

    // ---- load (gr_fft_vcc_fftw.cc:68-83)
    if (window) {
        for (int i = t; i < N; i += nthr) {
            float2 v = x[i];
            float w = window[i];
            A[i] = make_float2(v.x * w, v.y * w);
        }
    } else if (!FWD && shift) {
        const int len = N / 2;                 // floor(N/2.0); dst[k] = in[(k+len) mod N]
        for (int i = t; i < N; i += nthr) {
            int src = i + len; if (src >= N) src -= N;
            A[i] = x[src];
        }
    } else {
        for (int i = t; i < N; i += nthr) A[i] = x[i];
    }
    __syncthreads();

    // ---- Stockham passes
    float2 *src = A, *dst = B;
    int p = 1;
    const int T4 = N >> 2;
    while (p * 4 <= N) {
        const int tstep = N / (4 * p);         // twiddle index step
        for (int i = t; i < T4; i += nthr) {
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
        for (int i = t; i < T2; i += nthr) {
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
        for (int i = t; i < N; i += nthr) {
            int s = i + len; if (s >= N) s -= N;
            y[i] = src[s];
        }
    } else {
        for (int i = t; i < N; i += nthr) y[i] = src[i];
    }
}

// ---------------------------------------------------------------------------
// N = 4096 (config 3): three radix-16 Stockham passes, one butterfly per lane and pass.
// The 16 points of a butterfly live in registers, so the first pass reads HBM directly
// (coalesced: point q of lane t is x[t + 256 q]) and the last writes HBM directly; LDS
// only carries the two exchanges in between (one padded 34 KB buffer, conflict-free for
// both the stride-16 writes and the stride-1 reads), against six LDS round trips and six
// barriers of the radix-4 kernel.  A radix-16 butterfly is 4 x radix-4, seven constant
// twiddles, 4 x radix-4.  External twiddles come from the same double-precision table.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// complex product in two packed instructions (device_math.h)
__device__ __forceinline__ float2 cmul2(float2 a, float2 b)
{
    const f32x2_t r = cmul_pk(f32x2_t{a.x, a.y}, f32x2_t{b.x, b.y});
    return make_float2(r.x, r.y);
}

// a + (-i) b and a + (+i) b in one packed instruction: the halves of b are routed with op_sel, the sign is an operand modifier
__device__ __forceinline__ f32x2_t add_mi(f32x2_t a, f32x2_t b)
{
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));     // (a.x + b.y, a.y - b.x)
    return r;
}
__device__ __forceinline__ f32x2_t add_pi(f32x2_t a, f32x2_t b)
{
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));     // (a.x - b.y, a.y + b.x)
    return r;
}

// DFT of 4 points in place (forward: W4 = -i): eight packed additions
template <bool FWD>
__device__ __forceinline__ void radix4(f32x2_t &a, f32x2_t &b, f32x2_t &c, f32x2_t &d)
{
    const f32x2_t s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
    a = s0 + s1;
    c = s0 - s1;
    b = FWD ? add_mi(d0, d1) : add_pi(d0, d1);          // d0 -/+ i d1
    d = FWD ? add_pi(d0, d1) : add_mi(d0, d1);
}
template <bool FWD>
__device__ __forceinline__ void radix4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    f32x2_t A{a.x, a.y}, B{b.x, b.y}, C{c.x, c.y}, D{d.x, d.y};
    radix4<FWD>(A, B, C, D);
    a = make_float2(A.x, A.y); b = make_float2(B.x, B.y); c = make_float2(C.x, C.y); d = make_float2(D.x, D.y);
}

// v[m] <- sum_n v[n] W16^{nm}, n = c + 4d, m = r + 4s
template <bool FWD>
__device__ __forceinline__ void dft16(f32x2_t (&v)[16])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) radix4<FWD>(v[c], v[c + 4], v[c + 8], v[c + 12]);    // v[c + 4r] = a[c][r]
    const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    const float sg = FWD ? -1.f : 1.f;
    const f32x2_t w1{C1, sg * S1}, w2{H, sg * H}, w3{S1, sg * C1}, w6{-H, sg * H}, w9{-C1, -sg * S1};
    // a[c][r] *= W16^{c r}
    v[1 + 4] = cmul_pk(v[1 + 4], w1);
    v[1 + 8] = cmul_pk(v[1 + 8], w2);
    v[1 + 12] = cmul_pk(v[1 + 12], w3);
    v[2 + 4] = cmul_pk(v[2 + 4], w2);
    v[2 + 8] = FWD ? f32x2_t{v[2 + 8].y, -v[2 + 8].x} : f32x2_t{-v[2 + 8].y, v[2 + 8].x};   // W16^4 = -/+ i
    v[2 + 12] = cmul_pk(v[2 + 12], w6);
    v[3 + 4] = cmul_pk(v[3 + 4], w3);
    v[3 + 8] = cmul_pk(v[3 + 8], w6);
    v[3 + 12] = cmul_pk(v[3 + 12], w9);
#pragma unroll
    for (int r = 0; r < 4; ++r) radix4<FWD>(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);  // v[4r + s] = X[r + 4s]
    // un-permute: X[m], m = r + 4s, sits at 4r + s
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s2 = r + 1; s2 < 4; ++s2) {
            const f32x2_t tmp = v[4 * r + s2];
            v[4 * r + s2] = v[4 * s2 + r];
            v[4 * s2 + r] = tmp;
        }
}
template <bool FWD>
__device__ __forceinline__ void dft16(float2 (&v)[16])
{
    f32x2_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = f32x2_t{v[i].x, v[i].y};
    dft16<FWD>(u);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_float2(u[i].x, u[i].y);
}

// MODE bit 0: window on the way in; bit 1: shift (forward: fftshift on the way out; backward and no window: ifftshift on
// the way in -- both are "point q <-> point q ^ 8" of the same lane, a renaming of registers).
// Persistent workgroups (four per CU) walk the vectors: the twiddles of the last pass (W_N^{t q}: fifteen per lane) stay in
// registers and those of the middle pass (W_256^{k q}: 256 values) in LDS for the whole launch instead of being fetched
// per vector, and the next vector's sixteen points are requested as soon as the first pass has left its registers, so that
// HBM latency runs under passes 2 and 3 and the stores.
// (the window's sixteen values per lane are resident too: three workgroups per CU then, four otherwise)
constexpr int fft4096_wg_per_cu(int mode) { return (mode & 1) ? 3 : 4; }
template <bool FWD, int MODE>
__global__ void __launch_bounds__(256, fft4096_wg_per_cu(MODE))
fft4096_kernel(const float *__restrict__ window, const float2 *__restrict__ twiddle,
               const float2 *__restrict__ in, float2 *__restrict__ out, int nvec)
{
    constexpr int N = 4096;
    __shared__ f32x2_t S[N + N / 16];
    __shared__ f32x2_t W2[16 * 17];                         // [k][q], rows skewed by one slot: 16 rows on 16 bank pairs
    // slot of element i is i + (i >> 4) (one pad slot per 16).  Written out per access pattern -- 17 t + m, (t + (t >> 4)) +
    // 272 q, (j + (j >> 4)) + 17 m -- so that each pattern is one base register and immediate offsets
    const int t = threadIdx.x;
    constexpr bool WIN = MODE & 1, SHIFT = MODE & 2;
    constexpr int SW_IN = (!FWD && SHIFT && !WIN) ? 8 : 0, SW_OUT = (FWD && SHIFT) ? 8 : 0;

    f32x2_t w3[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float2 w = tw<FWD>(twiddle, t * q); w3[q] = f32x2_t{w.x, w.y}; }
    {
        const float2 w = tw<FWD>(twiddle, 16 * (t >> 4) * (t & 15));
        W2[(t >> 4) * 17 + (t & 15)] = f32x2_t{w.x, w.y};              // visible after the loop's first barrier
    }
    float wn[16];
    if (WIN) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wn[q] = window[t + 256 * q];
    }

    // vectors through raw buffer descriptors (wave-uniform base in SGPRs, one lane offset, the point's stride as the
    // instruction's scalar offset): no 64-bit address registers beside the thirty-two points in flight
    typedef unsigned int fft_u32x2 __attribute__((ext_vector_type(2)));
    f32x2_t pre[16];
    auto request = [&](int vec) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + (long long)vec * N), 0, N * 8, 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) {      // dst[k] = in[(k + N/2) mod N] when shifting
            // (bit-cast the whole vector: clang's __builtin_bit_cast of a vector ELEMENT reads element 0)
            pre[q] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, 8 * t, 2048 * (q ^ SW_IN), 0));
        }
    };
    int vec = blockIdx.x;
    if (vec < nvec) request(vec);
    for (; vec < nvec; vec += gridDim.x) {
        f32x2_t v[16];
        // ---- pass 1 (p = 1): points straight from HBM (gr_fft_vcc_fftw.cc:68-83)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = WIN ? pre[q] * wn[q] : pre[q];
        dft16<FWD>(v);
#pragma unroll
        for (int m = 0; m < 16; ++m) S[17 * t + m] = v[m];
        __syncthreads();
#ifndef GRHIP_FFT_NOPRE
        if (vec + (int)gridDim.x < nvec) request(vec + gridDim.x);
#endif

        // ---- pass 2 (p = 16): twiddle W_256^{k q} = W_N^{16 k q}
        {
            const int k = t & 15;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = S[t + (t >> 4) + 272 * q];
            __syncthreads();                               // every lane has read before anyone writes
            // (two table reads at a time: all fifteen hoisted would cost thirty registers beside the points in flight)
#pragma unroll
            for (int q0 = 0; q0 < 16; q0 += 2) {
#pragma unroll
                for (int q = q0 ? q0 : 1; q < q0 + 2; ++q) v[q] = cmul_pk(v[q], W2[k * 17 + q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            dft16<FWD>(v);
            const int j = (t - k) * 16 + k;
#pragma unroll
            for (int m = 0; m < 16; ++m) S[j + (j >> 4) + 17 * m] = v[m];
            __syncthreads();
        }

        // ---- pass 3 (p = 256): twiddle W_N^{t q}; results straight to HBM (gr_fft_vcc_fftw.cc:89-96)
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = S[t + (t >> 4) + 272 * q];
            __syncthreads();                               // S belongs to the next vector from here
#pragma unroll
            for (int q = 1; q < 16; ++q) v[q] = cmul_pk(v[q], w3[q]);
            dft16<FWD>(v);
#ifdef GRHIP_FFT_NOPRE
            if (vec + (int)gridDim.x < nvec) request(vec + gridDim.x);
#endif
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + (long long)vec * N, 0, N * 8, 0x00020000);
#pragma unroll
            for (int m = 0; m < 16; ++m)        // out[k] = fft[(k + N/2) mod N] when shifting
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fft_u32x2, v[m]), r, 8 * t, 2048 * (m ^ SW_OUT), 0);
        }
    }
}

static int fft_num_cus()
{
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 0;
        n_cus = n > 0 ? n : 256;
    }
    return n_cus;
}

template <bool FWD>
static void launch_fft4096(int shift, const float *window, const float2 *twiddle, const float2 *in, float2 *out,
                           long long nvec, hipStream_t st)
{
    // at most 2^31 - 1 vectors per launch (8 TB of samples): the caller's sizes are far below
    const int nv = (int)nvec;
    const int mode = (window ? 1 : 0) | (shift ? 2 : 0);
    const long long cap = (long long)fft4096_wg_per_cu(mode) * fft_num_cus();
    const dim3 grid((unsigned)(nvec < cap ? nvec : cap));
    switch (mode) {
    case 0: hipLaunchKernelGGL((fft4096_kernel<FWD, 0>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 1: hipLaunchKernelGGL((fft4096_kernel<FWD, 1>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 2: hipLaunchKernelGGL((fft4096_kernel<FWD, 2>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    default: hipLaunchKernelGGL((fft4096_kernel<FWD, 3>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    }
}

static int launch_fft8192(int forward, int shift, const float *window, const float2 *twiddle, const float2 *in, float2 *out,
                         long long nvec, hipStream_t st);
static int launch_fft16x(int N, int forward, int shift, const float *window, const float2 *twiddle, const float2 *in,
                         float2 *out, long long nvec, hipStream_t st);

int launch_fft(int N, int forward, int shift, const float *window, const float2 *twiddle, const float2 *in,
               float2 *out, long long nvec, hipStream_t st)
{
    if (nvec <= 0) return GRHIP_OK;
    if (!fft_size_supported(N)) return fail(GRHIP_EINVAL, "fft size %d not supported on device", N);
    if (N == 8192) return launch_fft8192(forward, shift, window, twiddle, in, out, nvec, st);
    if (N >= 32 && N <= 2048) return launch_fft16x(N, forward, shift, window, twiddle, in, out, nvec, st);
    if (N == 4096) {
        if (nvec > 0x7fffffffLL) return fail(GRHIP_EINVAL, "fft: too many vectors in one call");
        if (forward) launch_fft4096<true>(shift, window, twiddle, in, out, nvec, st);
        else launch_fft4096<false>(shift, window, twiddle, in, out, nvec, st);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
    size_t lds = (size_t)N * 2 * sizeof(float2);
    const int nthr = N / 4 >= 256 ? 256 : (N / 4 <= 64 ? 64 : N / 4);
    static size_t cfg_f = 0, cfg_b = 0;
    if (forward) {
        if (lds > 48 * 1024 && lds > cfg_f) {
            GRHIP_HIP(hipFuncSetAttribute((const void *)fft_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cfg_f = lds;
        }
        hipLaunchKernelGGL(fft_kernel<true>, dim3((unsigned)nvec), dim3(nthr), lds, st, N, shift, window,
                           twiddle, in, out);
    } else {
        if (lds > 48 * 1024 && lds > cfg_b) {
            GRHIP_HIP(hipFuncSetAttribute((const void *)fft_kernel<false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cfg_b = lds;
        }
        hipLaunchKernelGGL(fft_kernel<false>, dim3((unsigned)nvec), dim3(nthr), lds, st, N, shift, window,
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

// ---------------------------------------------------------------------------
// Fast path of the channeliser: oversample_rate 1 (rate_ratio == M), M a power of two
// <= 16.  Then output vector t applies filter M-1-j to stream j at in_j[t+1 ...] and
// the result goes to IFFT slot M-1-j, for every t: M independent real-tap FIRs
// followed by an M-point backward DFT.
//   * wave j of a 64*M-lane workgroup owns stream j for a tile of 512 output vectors:
//     it stages its own 512+tpf samples in LDS (coalesced 16-byte loads, one pad slot
//     per 8 samples -> conflict-free ds_read_b64) and runs the FIR with 8 outputs per
//     lane, an 8-deep register window and the (wave-uniform) taps in SGPRs -- the same
//     inner structure as fir_tiled_kernel;
//   * the filtered samples go back to LDS transposed (slot-major), one barrier, and
//     each lane then finishes one output vector with an in-register radix-2 FFT and
//     writes M contiguous complex values.
// 16 B of HBM traffic per input sample, HBM-bound (about 9 flop/B).
// ---------------------------------------------------------------------------
typedef const float __attribute__((address_space(4))) *pfb_cfloat_p;

// Persistent workgroups walk the tiles.  NT = padded taps per filter / 8: for NT in 1..4 the wave's taps are read once
// per launch and stay in SGPRs, the FIR is straight-line code and the tile's 512 + 8 NT samples per stream are nine
// 64-lane rounds of range-checked buffer loads (no branches); NT = 0 takes any length with the taps read in the loop.
// IL (round 3): the hier block's form (blks2impl/pfb_channelizer.py:25-75: stream_to_streams -> pfb -> vector_to_streams) in
// ONE pass -- the input is the single interleaved stream (stream j's item m is x[m M + j]; the tile's samples are one
// contiguous stretch, fetched by the whole workgroup with 16-byte loads and dealt to the streams' LDS rows), the output
// is M streams (a.out_streams + k a.out_stride; a lane's bin k goes to stream k, 64 consecutive items per wave and store).
// 16 B of HBM traffic per sample instead of the 48 of the three blocks one after the other; same arithmetic, same results.
template <int M, int NT, bool IL = false>
__global__ void __launch_bounds__(64 * M) __attribute__((amdgpu_waves_per_eu(M == 8 && NT > 0 ? 6 : 1)))       // (M = 8, resident taps: 40 KB of LDS -> three per CU)
pfb_os1_kernel(const PfbArgs a, long long ntiles)
{
    constexpr int R = 8, TT = 64 * R;                  // output vectors per tile
    constexpr bool POW2 = (M & (M - 1)) == 0;          // otherwise (M = 3, 5, 6, ...): direct M x M DFT per output vector
    constexpr int LOGM = M == 1 ? 0 : M == 2 ? 1 : M <= 4 ? 2 : M <= 8 ? 3 : 4;
    static_assert(M >= 2 && M <= 16, "2 <= M <= 16");
    typedef float pfb_f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int pfb_u32x4 __attribute__((ext_vector_type(4)));
    typedef float pfb_f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tpfp = NT ? R * NT : (a.tpf + R - 1) / R * R;    // taps padded to a multiple of R (zeros)
    const int XS = (TT + tpfp + R) + (TT + tpfp + R) / R + 1;   // slots per stream, padded
    const int SS = TT + TT / R + 1;                    // slots per IFFT input row, padded
    pfb_f32x2 *xs = (pfb_f32x2 *)smem;                 // [M][XS]
    pfb_f32x2 *sl = xs;                                // [M][SS], SS < XS: reuses the sample buffer once every
                                                       // wave is done with its FIR (40 KB instead of 77 KB: 3 workgroups per CU)
    const int t = threadIdx.x, ln = t & 63;
    const int j = __builtin_amdgcn_readfirstlane(t >> 6);      // wave = stream: descriptors and taps stay scalar
    // Oversampled by an integer factor os = M / rate_ratio (round 3): output vectors t = os u + r, r fixed per launch, are a
    // critically sampled channeliser of their own -- the commutator's `last` is the same for all of them
    // (last_r = ((r + 1) rate_ratio - 1) mod M), stream j goes through filter (last_r - j) mod M at item u + q_r + (j <= last_r)
    // into IFFT slot idxlut[j], one item further per u -- so the launcher runs this kernel os times.  os = 1: r = 0,
    // last = M - 1, q = 0: filter and slot M - 1 - j, item u + 1, as before.
    const int sub_last = a.sub_last, sub_os = a.sub_os;
    const int fj = j <= sub_last ? sub_last - j : M + sub_last - j;
    const int oj = a.sub_q + (j <= sub_last ? 1 : 0);
    const int slot_j = sub_os == 1 ? M - 1 - j : __builtin_amdgcn_readfirstlane(a.idxlut[j]);
    const pfb_cfloat_p taps = (pfb_cfloat_p)(a.ftaps) + (size_t)fj * a.tpf;
    const pfb_cfloat_p dft = (pfb_cfloat_p)(a.dft);
    float hres[NT ? R * NT : 1];
    if (NT) {
#pragma unroll
        for (int i = 0; i < R * NT; ++i) hres[i] = i < a.tpf ? taps[i] : 0.f;
    }
    // stream j: tpf history items + nout new ones; past that the range check returns zeros (and moves no bytes)
    const __amdgpu_buffer_rsrc_t xr = IL
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.in), 0, (int)((a.nout + a.tpf) * 8 * M), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.in + (long long)j * a.stride), 0,
                                            (int)(a.in_items * 8), 0x00020000);
    constexpr int OOB = (int)0xfffffff0;
    const int tot = TT + tpfp;
    pfb_f32x2 *dst = xs + (size_t)j * XS;
    const pfb_f32x2 *xp = xs + (size_t)j * XS + ln * R + ln;      // slot of m = 8 ln

    // (NT > 0) the next tile's samples are requested as soon as this tile's have left their registers for LDS: HBM latency
    // runs under the FIR, the DFT and the stores
    // (NT = 0: up to 256 taps per filter -- twelve rounds, the ones past the tile skipped -- and the wave's taps in LDS
    // behind the sample area, read at a wave-uniform address: in order with the sample reads, where scalar loads in the
    // loop shared a counter that can only be waited to zero)
    constexpr int NR = NT ? (TT + R * NT + 63) / 64 : 12;
    pfb_f32x2 pv[IL ? 1 : NR];
    // IL: the tile's (TT + tpfp) M interleaved samples, two per lane and round of the whole workgroup
    constexpr int NRI = IL ? ((NT ? TT + R * NT : TT + 256) * M + 128 * M - 1) / (128 * M) : 1;
    pfb_f32x4 pvi[NRI];
    auto request = [&](long long tile_) __attribute__((always_inline)) {
        if (IL) {
            const long long b0 = (tile_ * TT + 1) * (8ll * M) + 16 * t;       // (< 2^32: the launcher checks the stream's size)
#pragma unroll
            for (int i = 0; i < NRI; ++i) {
                const int sidx = 2 * (t + 64 * M * i);
                if (NT || sidx < tot * M)
                    pvi[i] = __builtin_bit_cast(pfb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 xr, sidx < tot * M ? (int)(unsigned)(b0 + 16ll * 64 * M * i) : OOB, 0, 0));
            }
            return;
        }
        const int vb = (int)((tile_ * TT + oj) * 8) + 8 * ln;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (NT || 64 * i < tot) {
                const int m = ln + 64 * i;
                pv[i] = __builtin_bit_cast(pfb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, m < tot ? vb + 512 * i : OOB, 0, 0));
            }
        }
    };
    float *tlj = reinterpret_cast<float *>(smem + (size_t)M * XS * sizeof(float2)) + (size_t)j * tpfp;
    if (!NT) {
        for (int q = ln; q < tpfp; q += 64) tlj[q] = q < a.tpf ? taps[q] : 0.f;       // (wave-private: ordered by the first barrier)
    }
    if ((long long)blockIdx.x < ntiles) request(blockIdx.x);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long t0 = tile * TT;
        // ---- stage stream j: items in_j[t0+1 .. t0+TT+tpfp]
        if (IL) {
            // interleaved sample s of the tile is item s / M of stream s mod M: every lane deals its two samples to their rows
#pragma unroll
            for (int i = 0; i < NRI; ++i) {
                const int sidx = 2 * (t + 64 * M * i);
                if (!NT && sidx >= tot * M) break;
                if (sidx < tot * M) {
                    const int m0 = sidx / M, j0 = sidx - m0 * M;
                    xs[(size_t)j0 * XS + m0 + (m0 >> 3)] = pfb_f32x2{pvi[i][0], pvi[i][1]};
                    const int m1 = (sidx + 1) / M, j1 = sidx + 1 - m1 * M;
                    xs[(size_t)j1 * XS + m1 + (m1 >> 3)] = pfb_f32x2{pvi[i][2], pvi[i][3]};
                }
            }
            if (tile + gridDim.x < ntiles) request(tile + gridDim.x);
            __syncthreads();
        } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (!NT && 64 * i >= tot) break;
            const int m = ln + 64 * i;
            if (m < tot) dst[m + (m >> 3)] = pv[i];
        }
        if (tile + gridDim.x < ntiles) request(tile + gridDim.x);
        // (wave-private region: no workgroup barrier needed; the compiler's own waits order the LDS stores and reads of a lane,
        // the wave barrier keeps it from moving one lane's reads above another lane's stores)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- FIR, 8 output vectors per lane: an 8-deep register window slides over the samples,
        // (re, im) pairs as 2-vectors: one v_pk_fma_f32 per tap and output
        pfb_f32x2 accv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) accv[r] = (pfb_f32x2){0.f, 0.f};
        {
            pfb_f32x2 w[R];
#pragma unroll
            for (int q = 0; q < R; ++q) w[q] = xp[q];
            if (NT) {
#pragma unroll
                for (int q0 = 0; q0 < R * NT; q0 += R) {
                    const int nxt = q0 + R + (q0 >> 3) + 1;
#pragma unroll
                    for (int qq = 0; qq < R; ++qq) {
                        const float h = hres[q0 + qq];
                        const pfb_f32x2 hv = (pfb_f32x2){h, h};
#pragma unroll
                        for (int r = 0; r < R; ++r) accv[r] = __builtin_elementwise_fma(hv, w[(qq + r) & (R - 1)], accv[r]);
                        w[qq] = xp[nxt + qq];
                    }
                }
            } else {
                for (int q0 = 0; q0 < tpfp; q0 += R) {
                    const int nxt = q0 + R + (q0 >> 3) + 1;
                    const pfb_f32x4 ha = *reinterpret_cast<const pfb_f32x4 *>(tlj + q0), hb = *reinterpret_cast<const pfb_f32x4 *>(tlj + q0 + 4);
#pragma unroll
                    for (int qq = 0; qq < R; ++qq) {
                        const float h = qq < 4 ? ha[qq] : hb[qq - 4];
                        const pfb_f32x2 hv = (pfb_f32x2){h, h};
#pragma unroll
                        for (int r = 0; r < R; ++r) accv[r] = __builtin_elementwise_fma(hv, w[(qq + r) & (R - 1)], accv[r]);
                        w[qq] = xp[nxt + qq];
                    }
                }
            }
        }
        // ---- to IFFT slot M-1-j, transposed: sl[slot][t_local]
        __syncthreads();                               // sl aliases xs
        {
            pfb_f32x2 *row = sl + (size_t)slot_j * SS + ln * R + ln;
#pragma unroll
            for (int r = 0; r < R; ++r) row[r] = accv[r];
        }
        __syncthreads();

        // ---- M-point backward DFT (unnormalised), one output vector per lane.  M = 8 (one vector per lane and tile): a
        // lane's vector is 64 contiguous bytes, so storing it directly would touch 64 separate 64-byte segments per
        // instruction; the wave's 64 vectors go through its own (by then free) row of LDS instead and leave as
        // 1 KB-contiguous 16-byte stores.
        constexpr bool COAL = M == 8 && !IL;
        constexpr int PIECES = M / 2 > 0 ? M / 2 : 1;         // 16-byte pieces per vector
        for (int tl = t; tl < TT; tl += 64 * M) {
            const long long tt = t0 + tl;
            if (!COAL && tt >= a.nout) continue;
            float2 v[M];
            if (POW2) {
            // bit-reversed load, then radix-2 decimation-in-time stages
#pragma unroll
            for (int s = 0; s < M; ++s) {
                int rv = 0;
#pragma unroll
                for (int bit = 0; bit < LOGM; ++bit)
                    if (s & (1 << bit)) rv |= (M >> 1) >> bit;
                const pfb_f32x2 u = sl[(size_t)s * SS + tl + (tl >> 3)];
                v[rv] = make_float2(u.x, u.y);
            }
            // (canonical loop bounds everywhere: anything the compiler cannot unroll turns v[] into
            // a scratch array)
#pragma unroll
            for (int stg = 0; stg < LOGM; ++stg) {
                const int len = 2 << stg;
                const int half = len >> 1, step = M / len;
#pragma unroll
                for (int s0 = 0; s0 < M; s0 += len) {
#pragma unroll
                    for (int k = 0; k < half; ++k) {
                        const float wr = dft[2 * (k * step)], wi = dft[2 * (k * step) + 1];   // e^{+2 pi i k/len}
                        const float2 u = v[s0 + k], q = v[s0 + k + half];
                        const float2 tw = (k == 0) ? q : make_float2(__builtin_fmaf(q.x, wr, -(q.y * wi)),
                                                                     __builtin_fmaf(q.x, wi, q.y * wr));
                        v[s0 + k] = make_float2(u.x + tw.x, u.y + tw.y);
                        v[s0 + k + half] = make_float2(u.x - tw.x, u.y - tw.y);
                    }
                }
            }
            } else {
                // out[k] = sum_s slot[s] e^{+2 pi i s k / M}: M^2 complex multiply-adds, the M table entries wave-uniform
                float2 u[M];
#pragma unroll
                for (int s = 0; s < M; ++s) { const pfb_f32x2 q = sl[(size_t)s * SS + tl + (tl >> 3)]; u[s] = make_float2(q.x, q.y); }
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    float2 acc = u[0];
#pragma unroll
                    for (int s = 1; s < M; ++s) {
                        const int ph = (s * k) % M;
                        const float wr = dft[2 * ph], wi = dft[2 * ph + 1];
                        acc.x = __builtin_fmaf(u[s].x, wr, acc.x);
                        acc.x = __builtin_fmaf(-u[s].y, wi, acc.x);
                        acc.y = __builtin_fmaf(u[s].x, wi, acc.y);
                        acc.y = __builtin_fmaf(u[s].y, wr, acc.y);
                    }
                    v[k] = acc;
                }
            }
            if (COAL && sub_os == 1) {
                __syncthreads();                       // every wave has read its vectors' slots: the rows are free
                pfb_f32x2 *sc = dst;                   // this wave's row, (M + 1) slots per vector
#pragma unroll
                for (int k = 0; k < M; ++k) sc[ln * (M + 1) + k] = (pfb_f32x2){v[k].x, v[k].y};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // the wave's 64 vectors through a descriptor of their own: vectors past nout are out of its range
                const long long tw0 = t0 + __builtin_amdgcn_readfirstlane(tl - ln);
                long long left = (a.nout - tw0) * (long long)(8 * M);
                left = left < 0 ? 0 : (left > 64 * 8 * M ? 64 * 8 * M : left);
                const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out + tw0 * M, 0, (int)left, 0x00020000);
#pragma unroll
                for (int k = 0; k < PIECES; ++k) {
                    const int idx = 64 * k + ln, vec = idx / PIECES, pc = idx % PIECES;
                    const pfb_f32x2 p0 = sc[vec * (M + 1) + 2 * pc], p1 = sc[vec * (M + 1) + 2 * pc + 1];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pfb_u32x4, pfb_f32x4{p0.x, p0.y, p1.x, p1.y}), orr, 16 * idx, 0, 0);
                }
                continue;
            }
            if (IL) {
#pragma unroll
                for (int k = 0; k < M; ++k) a.out_streams[(long long)k * a.out_stride + tt] = v[k];
                continue;
            }
            if (COAL && tt >= a.nout) continue;
            float2 *o = a.out + (tt * sub_os + a.sub_r) * M;
            if (M % 2 == 0) {
                float4 *o4 = reinterpret_cast<float4 *>(o);
#pragma unroll
                for (int k = 0; k + 1 < M; k += 2) o4[k >> 1] = make_float4(v[k].x, v[k].y, v[k + 1].x, v[k + 1].y);
            } else {
#pragma unroll
                for (int k = 0; k < M; ++k) o[k] = v[k];          // (an odd vector is not a whole number of 16-byte pieces)
            }
        }
        if (M != 8 || IL || sub_os != 1) __syncthreads();   // sl (= xs) belongs to the next tile's samples from here
    }
}

template <int M, int NT, bool IL = false>
static int launch_pfb_os1_nt(const PfbArgs &a, size_t lds, hipStream_t st)
{
    const int TT = 512;
    static size_t cfg = 0;
    if (lds > 48 * 1024 && lds > cfg) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)pfb_os1_kernel<M, NT, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cfg = lds;
    }
    const long long ntiles = (a.nout + TT - 1) / TT;
    int per_cu = 0;                                    // resident workgroups per CU (registers and LDS both count)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)pfb_os1_kernel<M, NT, IL>, 64 * M, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    const long long cap = (long long)per_cu * fft_num_cus();
    hipLaunchKernelGGL((pfb_os1_kernel<M, NT, IL>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(64 * M), lds, st, a, ntiles);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// the hier block's fused form (interleaved stream in, M streams out): oversample rate 1, M = 2 / 4 / 8 / 16, up to 32 taps
// per filter resident in SGPRs or any length from LDS; -1 = not this shape (the caller runs the three blocks one after the other)
template <int M>
static int launch_pfb_os1_il(const PfbArgs &a, hipStream_t st)
{
    const int R = 8, TT = 512;
    const int tpfp = (a.tpf + R - 1) / R * R;
    const int XS = (TT + tpfp + R) + (TT + tpfp + R) / R + 1;
    const size_t lds = (size_t)M * XS * sizeof(float2) + (tpfp / R > 4 ? (size_t)M * tpfp * sizeof(float) : 0);
    if (lds > 150 * 1024 || tpfp > 256) return -1;
    if ((a.nout + a.tpf) * 8 * M > 0x7fffffffLL) return -1;        // the interleaved stream's buffer descriptor counts bytes in 32 bits
    switch (tpfp / R) {
    case 1: return launch_pfb_os1_nt<M, 1, true>(a, lds, st);
    case 2: return launch_pfb_os1_nt<M, 2, true>(a, lds, st);
    case 3: return launch_pfb_os1_nt<M, 3, true>(a, lds, st);
    case 4: return launch_pfb_os1_nt<M, 4, true>(a, lds, st);
    default: return launch_pfb_os1_nt<M, 0, true>(a, lds, st);
    }
}

int launch_pfb_hier(const PfbArgs &a_in, hipStream_t st)
{
    PfbArgs a = a_in;
    a.sub_r = 0; a.sub_os = 1; a.sub_q = 0; a.sub_last = a.M - 1; a.in_items = a.nout + a.tpf;
    if (a.nout <= 0) return GRHIP_OK;
    if (a.rate_ratio != a.M || (((uintptr_t)a.in) & 15) || (((uintptr_t)a.out_streams) & 7)) return -1;
    switch (a.M) {
    case 2: return launch_pfb_os1_il<2>(a, st);
    case 4: return launch_pfb_os1_il<4>(a, st);
    case 8: return launch_pfb_os1_il<8>(a, st);
    case 16: return launch_pfb_os1_il<16>(a, st);
    }
    return -1;
}

template <int M>
static int launch_pfb_os1(const PfbArgs &a, hipStream_t st)
{
    const int R = 8, TT = 512;
    const int tpfp = (a.tpf + R - 1) / R * R;
    const int XS = (TT + tpfp + R) + (TT + tpfp + R) / R + 1;
    const size_t lds = (size_t)M * XS * sizeof(float2) + (tpfp / R > 4 ? (size_t)M * tpfp * sizeof(float) : 0);
    if (lds > 150 * 1024 || tpfp > 256) return -1;
    if ((a.nout + a.tpf) * 8 > 0xffffffffLL) return -1;         // the stream's buffer descriptor counts bytes in 32 bits
    switch (tpfp / R) {
    case 1: return launch_pfb_os1_nt<M, 1>(a, lds, st);
    case 2: return launch_pfb_os1_nt<M, 2>(a, lds, st);
    case 3: return launch_pfb_os1_nt<M, 3>(a, lds, st);
    case 4: return launch_pfb_os1_nt<M, 4>(a, lds, st);
    default: return launch_pfb_os1_nt<M, 0>(a, lds, st);
    }
}

// General channeliser, M <= 64: the mapping of pfb_kernel turned round.  There, neighbouring lanes were neighbouring STREAMS
// (addresses a whole stream apart: every 8-byte load its own memory transaction; 0.05-0.10 of the HBM peak); here a wave is
// 64 consecutive output vectors of ONE stream (consecutive addresses, the taps' window sliding through L1), ty walks the
// streams, and the same lanes then produce the bins of their vector from LDS.  Filter part in the reference's generic
// order, unfused (bit-exact), as before.
__global__ void __launch_bounds__(1024)
pfb_rows_kernel(const PfbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *slots = (float2 *)smem;             // [64][M + 1]
    const int M = a.M, tpf = a.tpf, MP = M + 1;
    const int tx = threadIdx.x, ty = threadIdx.y, YS = blockDim.y;
    const long long t = (long long)blockIdx.x * 64 + tx;
    const bool active = t < a.nout;
    long long c = 0, n = 0;
    int last = 0;
    if (active) {
        c = (t + 1) * (long long)a.rate_ratio - 1;
        last = (int)(c % M);
        n = 1 + c / M;
    }
    for (int j = ty; j < M; j += YS) {
        if (active) {
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
            slots[tx * MP + a.idxlut[j]] = make_float2(a0r + a1r, a0i + a1i);
        }
    }
    __syncthreads();
    if (active) {
        const float2 *row = slots + tx * MP;
        for (int k = ty; k < M; k += YS) {
            float2 acc = make_float2(0.f, 0.f);
            int ph = 0;                              // (s*k) mod M
            for (int sidx = 0; sidx < M; ++sidx) {
                float2 w = a.dft[ph];
                float2 v = row[sidx];
                acc.x = __builtin_fmaf(v.x, w.x, acc.x);
                acc.x = __builtin_fmaf(-v.y, w.y, acc.x);
                acc.y = __builtin_fmaf(v.x, w.y, acc.y);
                acc.y = __builtin_fmaf(v.y, w.x, acc.y);
                ph += k; if (ph >= M) ph -= M;
            }
            a.out[t * M + k] = acc;
        }
    }
}

template <int M>
static int launch_pfb_os1_any(const PfbArgs &a, hipStream_t st)
{
    const int R = 8, TT = 512;
    const int tpfp = (a.tpf + R - 1) / R * R;
    const int XS = (TT + tpfp + R) + (TT + tpfp + R) / R + 1;
    const size_t lds = (size_t)M * XS * sizeof(float2) + (size_t)M * tpfp * sizeof(float);
    if (lds > 150 * 1024 || tpfp > 256) return -1;
    if ((a.nout + a.tpf) * 8 > 0xffffffffLL) return -1;
    return launch_pfb_os1_nt<M, 0>(a, lds, st);
}

// ---------------------------------------------------------------------------
// Channel counts 32 / 64 / 128 at oversample_rate 1: the M polyphase FIRs in one kernel, the M-point backward DFTs in
// the batched FFT kernel (fft16x_kernel, in place on the output).  A 256-lane workgroup takes a tile of TT = 8192 / M output
// vectors; wave w runs streams w, w + 4, ... one after the other -- its own TT + tpf samples staged in a wave-private
// LDS row (odd slot stride: conflict-free), R = TT / 64 consecutive outputs per lane with an R-deep register window, the
// taps at a wave-uniform LDS address, the same fma order as pfb_os1_kernel -- and drops the filtered samples into the
// tile's [t][slot] array in LDS; after one barrier the tile leaves as whole output vectors (M x 8 contiguous bytes per
// t, 16-byte stores).  24 B of HBM traffic per sample more than the fused kernels (the vectors are written, read and
// written again), against pfb_rows_kernel's strided accesses: M = 32 went from 38 to 150+ Gsamples/s.
// ---------------------------------------------------------------------------
template <int R, int M, int TP>       // TP: taps per filter padded to a compile-time count (0: any length, taps and window read in the loop)
__global__ void __launch_bounds__(256)
pfb_fir_t_kernel(const PfbArgs a)
{
    constexpr int TT = 64 * R;
    constexpr bool FUSE = M <= 64;                     // the DFT in this kernel too (one / two lanes per output vector)
    constexpr int SPW = M / 4;                         // streams per wave
    constexpr int NLM = R + 2;                         // 64-lane load rounds per stream: TT + tpfp + R <= 64 (R + 2) samples
    typedef float pfb_f32x2 __attribute__((ext_vector_type(2)));
    typedef float pfb_f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tpf = a.tpf;
    const int tpfp = TP ? TP : (tpf + R - 1) / R * R;  // taps padded to a multiple of R (zeros)
    constexpr int SW = M + 1;                          // slots per output vector in LDS (odd)
    const int NS = TT + tpfp + R;                      // samples a stream stages per tile
    const int XSW = NS + (R > 1 ? NS / R : 0) + 1;     // their slots: one pad slot per R samples (odd lane stride)
    const int t = threadIdx.x, ln = t & 63, w = t >> 6;
    pfb_f32x2 *sl = reinterpret_cast<pfb_f32x2 *>(smem);
    pfb_f32x2 *xs = sl + (size_t)TT * SW + (size_t)w * XSW;
    float *tl = reinterpret_cast<float *>(sl + (size_t)TT * SW + (size_t)4 * XSW) + (size_t)w * tpfp;
    const long long items = a.nout + tpf;              // readable items of a stream (item 0 = oldest history item)

    // persistent workgroups walk the tiles; all of the wave's samples of a tile are requested at once, and a stream's share
    // of the NEXT tile as soon as its registers have been emptied into LDS: HBM latency runs under the FIRs and the stores
    pfb_f32x2 pv[SPW][NLM];
    auto request = [&](long long tile_, int js) __attribute__((always_inline)) {
        const float2 *src = a.in + (long long)(w + 4 * js) * a.stride;
#pragma unroll
        for (int i = 0; i < NLM; ++i) {
            const int m = ln + 64 * i;
            const long long g = tile_ * TT + 1 + m;
            pv[js][i] = pfb_f32x2{0.f, 0.f};
            if (m < NS && g < items) { const float2 q = src[g]; pv[js][i] = pfb_f32x2{q.x, q.y}; }
        }
    };
    const long long ntiles = (a.nout + TT - 1) / TT;
    if ((long long)blockIdx.x < ntiles) {
#pragma unroll
        for (int js = 0; js < SPW; ++js) request(blockIdx.x, js);
    }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long t0 = tile * TT;
    // (M = 64 / 128: a lane holds 128 / 192 registers of samples; with a second tile's loads live beside the FIR and the DFT it
    // runs out of them: one tile per workgroup)
    const bool more = M < 64 && tile + gridDim.x < ntiles;
#pragma unroll
    for (int js = 0; js < SPW; ++js) {
        const int j = w + 4 * js;
        const float *taps = a.ftaps + (size_t)(M - 1 - j) * tpf;
#pragma unroll
        for (int i = 0; i < NLM; ++i) {
            const int m = ln + 64 * i;
            if (m < NS) xs[m + (R > 1 ? m / R : 0)] = pv[js][i];
        }
        if (more) request(tile + gridDim.x, js);
        for (int q = ln; q < tpfp; q += 64) tl[q] = q < tpf ? taps[q] : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const pfb_f32x2 *xp = xs + ln * R + (R > 1 ? ln : 0);          // slot of sample R ln
        pfb_f32x2 acc[R], win[R];
        if (TP) {
            // everything the lane needs in registers first (one LDS wait), then straight-line FMAs in the same order
            pfb_f32x2 xv[R + (TP ? TP : 1) - 1];
            float hq[TP ? TP : 1];
#pragma unroll
            for (int i = 0; i < R + TP - 1; ++i) xv[i] = xp[i + (R > 1 ? i / R : 0)];
#pragma unroll
            for (int q = 0; q < TP; ++q) hq[q] = tl[q];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = pfb_f32x2{0.f, 0.f};
#pragma unroll
            for (int q = 0; q < TP; ++q) {
                const pfb_f32x2 hv = {hq[q], hq[q]};
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = __builtin_elementwise_fma(hv, xv[r + q], acc[r]);
            }
        } else {
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r] = pfb_f32x2{0.f, 0.f}; win[r] = xp[r]; }
        for (int q0 = 0; q0 < tpfp; q0 += R) {
            const int nxt = q0 + R + (R > 1 ? q0 / R + 1 : 0);          // slot offset of sample R ln + q0 + R
#pragma unroll
            for (int qq = 0; qq < R; ++qq) {
                const float h = tl[q0 + qq];
                const pfb_f32x2 hv = {h, h};
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = __builtin_elementwise_fma(hv, win[(qq + r) % R], acc[r]);
                win[qq] = xp[nxt + qq];
            }
        }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) sl[(size_t)(ln * R + r) * SW + (M - 1 - j)] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                               // the staging row belongs to the next stream
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if (FUSE) {
        // the backward DFT of the tile's output vectors in registers (no second kernel, no intermediate in HBM).  M = 32:
        // one vector per lane; M = 64: two lanes per vector -- each the 32-point DFT of the even / odd slots (bit-reversed
        // load, radix-2 decimation-in-time stages as in pfb_os1_kernel), then the last radix-2 stage across the lane pair:
        // the odd lane turns its half by e^{+2 pi i k / 64}, the two swap through a quad permute, the even lane keeps
        // E + T = outputs 0..31, the odd one E - T = outputs 32..63 -- and back into the vector's own LDS row.
        const pfb_cfloat_p dft = (pfb_cfloat_p)(a.dft);
        constexpr int LPV = M / 32, LOGP = 5, P = 32;
        static_assert(!FUSE || TT * LPV == 256, "every lane has its share of a vector");
        const int half = t & (LPV - 1);
        pfb_f32x2 *row = sl + (size_t)(t / LPV) * SW;
        float2 v[P];
#pragma unroll
        for (int s_ = 0; s_ < P; ++s_) {
            int rv = 0;
#pragma unroll
            for (int bit = 0; bit < LOGP; ++bit)
                if (s_ & (1 << bit)) rv |= (P >> 1) >> bit;
            const pfb_f32x2 u = row[LPV * s_ + half];
            v[rv] = make_float2(u.x, u.y);
        }
#pragma unroll
        for (int stg = 0; stg < LOGP; ++stg) {
            const int len = 2 << stg;
            const int hl = len >> 1, step = M / len;
#pragma unroll
            for (int s0 = 0; s0 < P; s0 += len) {
#pragma unroll
                for (int k = 0; k < hl; ++k) {
                    const float wr = dft[2 * (k * step)], wi = dft[2 * (k * step) + 1];   // e^{+2 pi i k/len}
                    const float2 u = v[s0 + k], q = v[s0 + k + hl];
                    const float2 tw = (k == 0) ? q : make_float2(__builtin_fmaf(q.x, wr, -(q.y * wi)),
                                                                 __builtin_fmaf(q.x, wi, q.y * wr));
                    v[s0 + k] = make_float2(u.x + tw.x, u.y + tw.y);
                    v[s0 + k + hl] = make_float2(u.x - tw.x, u.y - tw.y);
                }
            }
        }
        // radix-2 stages across the lanes of a vector: lane `half` holds the P-point DFT of the slots = half (mod LPV).
        // Stage with partner lane ^ XR combines two DFTs of L points into one of 2 L: the upper lane of the pair (bit XR of
        // `half` set) turns its values by e^{+2 pi i (k + off) / (2 L)}, the two swap through a quad permute, the lower lane
        // keeps A + T (outputs off + k), the upper one A - T (outputs off + k + L), off = the block of L outputs the lanes hold.
        auto cross = [&](const bool upper, const int tw_index_base, const int tw_step, const bool swap2) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int ti = tw_index_base + tw_step * k;            // index into the M-entry table e^{+2 pi i m / M}
                const float wr = dft[2 * ti], wi = dft[2 * ti + 1];
                const float2 q = v[k];
                const float2 mine = upper ? make_float2(__builtin_fmaf(q.x, wr, -(q.y * wi)), __builtin_fmaf(q.x, wi, q.y * wr)) : q;
                float ox, oy;
                if (swap2) {
                    ox = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine.x), 0x4E, 0xf, 0xf, true));
                    oy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine.y), 0x4E, 0xf, 0xf, true));
                } else {
                    ox = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine.x), 0xB1, 0xf, 0xf, true));
                    oy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine.y), 0xB1, 0xf, 0xf, true));
                }
                v[k] = upper ? make_float2(ox - mine.x, oy - mine.y) : make_float2(mine.x + ox, mine.y + oy);
            }
        };
        int oslot = 0;                                                 // first output slot of the lane's 32
        if (LPV == 2) {
            cross(half != 0, 0, 1, false);                             // 32 + 32 -> 64: twiddle e^{+2 pi i k / 64}
            oslot = P * half;
        }
        // (four lanes per vector at M = 128 -- two such stages, partners lane ^ 2 then lane ^ 1 -- were built and passed the
        // tests, but with 192 registers of samples per lane the kernel ran at 91 Gsamples/s against 110 for the two-kernel form)
        // (a vector's row belongs to its lane(s), which sit side by side in one wave: no barrier between their reads and writes)
#pragma unroll
        for (int k = 0; k < P; ++k) row[oslot + k] = pfb_f32x2{v[k].x, v[k].y};
        __syncthreads();
    }
    // the tile as whole output vectors: 16-byte piece p of vector tl holds slots 2p, 2p + 1
    constexpr int PCS = M / 2;
    for (int idx = t; idx < TT * PCS; idx += 256) {
        const int row = idx / PCS, pc = idx - row * PCS;
        if (t0 + row >= a.nout) break;
        const pfb_f32x2 p0 = sl[(size_t)row * SW + 2 * pc], p1 = sl[(size_t)row * SW + 2 * pc + 1];
        *reinterpret_cast<pfb_f32x4 *>(a.out + (t0 + row) * M + 2 * pc) = pfb_f32x4{p0.x, p0.y, p1.x, p1.y};
    }
    __syncthreads();                                   // sl belongs to the next tile from here
    }
}

template <int R, int M, int TP>
static int launch_pfb_fir_t_tp(const PfbArgs &a, hipStream_t st)
{
    constexpr int TT = 64 * R;
    const int tpfp = TP ? TP : (a.tpf + R - 1) / R * R;
    const int NS = TT + tpfp + R;
    if (NS > 64 * (R + 2)) return -1;                  // (filters beyond ~120 taps per channel: the general kernel)
    const int XSW = NS + (R > 1 ? NS / R : 0) + 1;
    const size_t lds = ((size_t)TT * (M + 1) + (size_t)4 * XSW) * sizeof(float2) + (size_t)4 * tpfp * sizeof(float);
    if (lds > 150 * 1024) return -1;
    if (lds > 48 * 1024)
        GRHIP_HIP(hipFuncSetAttribute((const void *)pfb_fir_t_kernel<R, M, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long ntiles = (a.nout + TT - 1) / TT;
    const long long cap = M < 64 ? (long long)fft_num_cus() * (lds > 80 * 1024 ? 1 : 2) : ntiles;      // (M = 64 / 128: one tile per workgroup)
    hipLaunchKernelGGL((pfb_fir_t_kernel<R, M, TP>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), lds, st, a);
    GRHIP_HIP(hipGetLastError());
    if (M <= 64) return GRHIP_OK;                      // (the kernel has done the DFT)
    // the M-point backward DFT of every output vector, in place (unnormalised, as fftw's)
    return launch_fft(M, 0, 0, nullptr, a.dft + M, a.out, a.out, a.nout, st);
}

template <int R, int M>
static int launch_pfb_fir_t(const PfbArgs &a, hipStream_t st)
{
    // taps per filter up to 16 / 32: the FIR of a stream entirely in registers
    if (a.tpf <= 16) return launch_pfb_fir_t_tp<R, M, 16>(a, st);
    if (a.tpf <= 32) return launch_pfb_fir_t_tp<R, M, 32>(a, st);
    return launch_pfb_fir_t_tp<R, M, 0>(a, st);
}

int launch_pfb(const PfbArgs &a_in, hipStream_t st)
{
    PfbArgs a = a_in;
    a.sub_r = 0; a.sub_os = 1; a.sub_q = 0; a.sub_last = a.M - 1; a.in_items = a.nout + a.tpf;
    if (a.nout <= 0) return GRHIP_OK;
    if (a.M < 1 || a.M > 1024) return fail(GRHIP_EINVAL, "numchans %d not supported on device", a.M);
    // integer oversampling (os = M / rate_ratio) on the fast kernel's channel counts: os launches of pfb_os1_kernel, one per
    // residue r of the output index (see the kernel); 0.08-0.13 of the HBM peak on pfb_rows_kernel before
    if (a.rate_ratio > 0 && a.rate_ratio < a.M && a.M % a.rate_ratio == 0 && a.M <= 16 && (((uintptr_t)a.out) & 15) == 0) {
        const int os = a.M / a.rate_ratio;
        int rc = GRHIP_OK;
        for (int r = 0; r < os && rc == GRHIP_OK; ++r) {
            PfbArgs b = a;
            b.sub_r = r; b.sub_os = os;
            const long long c0 = (long long)(r + 1) * a.rate_ratio - 1;
            b.sub_last = (int)(c0 % a.M); b.sub_q = (int)(c0 / a.M);
            b.nout = (a.nout - r + os - 1) / os;
            b.in_items = a.tpf + (a.nout * a.rate_ratio + a.M - 1) / a.M;
            if (b.nout <= 0) continue;
            rc = -1;
            switch (a.M) {
            case 2: rc = launch_pfb_os1<2>(b, st); break;
            case 4: rc = launch_pfb_os1<4>(b, st); break;
            case 8: rc = launch_pfb_os1<8>(b, st); break;
            case 16: rc = launch_pfb_os1<16>(b, st); break;
            case 3: rc = launch_pfb_os1_any<3>(b, st); break;
            case 5: rc = launch_pfb_os1_any<5>(b, st); break;
            case 6: rc = launch_pfb_os1_any<6>(b, st); break;
            case 7: rc = launch_pfb_os1_any<7>(b, st); break;
            case 9: rc = launch_pfb_os1_any<9>(b, st); break;
            case 10: rc = launch_pfb_os1_any<10>(b, st); break;
            case 11: rc = launch_pfb_os1_any<11>(b, st); break;
            case 12: rc = launch_pfb_os1_any<12>(b, st); break;
            case 13: rc = launch_pfb_os1_any<13>(b, st); break;
            case 14: rc = launch_pfb_os1_any<14>(b, st); break;
            case 15: rc = launch_pfb_os1_any<15>(b, st); break;
            }
            if (rc == -1 && r > 0) return fail(GRHIP_ERUNTIME, "pfb: sub-sequence launch refused after the first");
        }
        if (rc != -1) return rc;
    }
    if (a.rate_ratio == a.M && (((uintptr_t)a.out) & 15) == 0) {
        int rc = -1;
        switch (a.M) {
        case 2: rc = launch_pfb_os1<2>(a, st); break;
        case 4: rc = launch_pfb_os1<4>(a, st); break;
        case 8: rc = launch_pfb_os1<8>(a, st); break;
        case 16: rc = launch_pfb_os1<16>(a, st); break;
        // channel counts that are not a power of two: the same kernel with a direct DFT (taps in LDS, any filter length)
        case 3: rc = launch_pfb_os1_any<3>(a, st); break;
        case 5: rc = launch_pfb_os1_any<5>(a, st); break;
        case 6: rc = launch_pfb_os1_any<6>(a, st); break;
        case 7: rc = launch_pfb_os1_any<7>(a, st); break;
        case 9: rc = launch_pfb_os1_any<9>(a, st); break;
        case 10: rc = launch_pfb_os1_any<10>(a, st); break;
        case 11: rc = launch_pfb_os1_any<11>(a, st); break;
        case 12: rc = launch_pfb_os1_any<12>(a, st); break;
        case 13: rc = launch_pfb_os1_any<13>(a, st); break;
        case 14: rc = launch_pfb_os1_any<14>(a, st); break;
        case 15: rc = launch_pfb_os1_any<15>(a, st); break;
        }
        if (rc != -1) return rc;
        // 32 / 64 / 128 channels: polyphase FIRs + batched FFT
        if (a.M == 32) rc = launch_pfb_fir_t<4, 32>(a, st);
        else if (a.M == 64) rc = launch_pfb_fir_t<2, 64>(a, st);
        else if (a.M == 128) rc = launch_pfb_fir_t<1, 128>(a, st);
        if (rc != -1) return rc;
    }
    if (a.M <= 64) {
        int ys = a.M < 16 ? a.M : 16;
        dim3 block(64, ys);
        dim3 grid((unsigned)((a.nout + 63) / 64));
        size_t lds = (size_t)64 * (a.M + 1) * sizeof(float2);
        hipLaunchKernelGGL(pfb_rows_kernel, grid, block, lds, st, a);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
    int ty = 256 / a.M; if (ty < 1) ty = 1;
    dim3 block(a.M, ty);
    dim3 grid((unsigned)((a.nout + ty - 1) / ty));
    size_t lds = (size_t)a.M * ty * sizeof(float2);
    hipLaunchKernelGGL(pfb_kernel, grid, block, lds, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_fft_filter_ccc, fused: one workgroup = one 4096-sample block of overlap-SAVE fast
// convolution.  Forward radix-16 FFT (first pass straight from HBM), spectrum times the
// transformed taps in registers, inverse radix-16 FFT -- its first pass starts from the
// registers the forward transform ended in, because "point q of lane t is index
// t + 256 q" is both the output layout of a last Stockham pass and the input layout of a
// first one -- and the last pass stores the valid, decimated outputs straight to HBM.
// 8 B in + 8/D B out per sample, four LDS exchanges per block, no intermediate in HBM.
// Block b produces full-rate outputs [bL, bL+L), L = 4096 - (ntaps-1) rounded down to a
// multiple of the decimation, from inputs [bL-(ntaps-1), bL+L); inputs before the call come
// from `hist` (the last ntaps-1 items of the previous call), inputs past nin read as zero.
// The reference block is overlap-ADD with its own transform size; both are the same
// convolution to within transform rounding (its FFTs are FFTW: unpinned anyway).
// ===========================================================================
// a * conj(b) in two packed instructions
__device__ __forceinline__ f32x2_t cmul_conj_pk(f32x2_t a, f32x2_t b)
{
    f32x2_t t, r;
    // t = (a.y * b.y, a.y * b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    // r = (fma(a.x, b.x, t.x), fma(a.x, -b.y, t.y)) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

// exchange + pass 2 + exchange + pass 3 of a 4096-point transform whose pass 1 the caller has done (dft16 on v).
// w3 = the last pass's twiddles of the FORWARD transform, W_N^{t q}, resident in registers; W2 = the middle pass's,
// W_N^{16 k q} at [k * 17 + q] in LDS; the backward transform multiplies by their conjugates.
template <bool FWD>
__device__ __forceinline__ void fft4096_mid_passes(f32x2_t (&v)[16], f32x2_t *S, const f32x2_t *W2, const f32x2_t (&w3)[16], int t)
{
#pragma unroll
    for (int m = 0; m < 16; ++m) S[17 * t + m] = v[m];
    __syncthreads();
    {
        const int k = t & 15;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = S[t + (t >> 4) + 272 * q];
        __syncthreads();
#pragma unroll
        for (int q0 = 0; q0 < 16; q0 += 2) {
#pragma unroll
            for (int q = q0 ? q0 : 1; q < q0 + 2; ++q) v[q] = FWD ? cmul_pk(v[q], W2[k * 17 + q]) : cmul_conj_pk(v[q], W2[k * 17 + q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        dft16<FWD>(v);
        const int j = (t - k) * 16 + k;
#pragma unroll
        for (int m = 0; m < 16; ++m) S[j + (j >> 4) + 17 * m] = v[m];
        __syncthreads();
    }
    {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = S[t + (t >> 4) + 272 * q];
        __syncthreads();                    // S is reused by the next transform
#pragma unroll
        for (int q = 1; q < 16; ++q) v[q] = FWD ? cmul_pk(v[q], w3[q]) : cmul_conj_pk(v[q], w3[q]);
        dft16<FWD>(v);
    }
}

// ===========================================================================
// gr_fft_vcc, N = 32 ... 2048: the radix-16 register kernel with N / 16 lanes per vector (a 256-lane workgroup carries
// 4096 / N vectors at a time): two radix-16 passes (one below 256 points) and one pass of radix R3 = N / 256 (N / 16
// below 256 points; none for 256) in which a lane does 16 / R3 butterflies on the sixteen points it holds.  Every pass reads point q of lane l at l + (N/16) q of its
// vector (pass 1: straight from HBM), so the three passes share one index pattern; the last writes HBM directly.
// Persistent workgroups, next group's points requested one group ahead, middle-pass twiddles in LDS, last-pass twiddles
// in registers -- as fft4096_kernel.  (The radix-4 LDS kernel these sizes used before fetched three twiddles per
// butterfly and pass from global memory: 0.27-0.50 of the HBM peak.)
// ===========================================================================
template <bool FWD>
__device__ __forceinline__ void radix2(f32x2_t &a, f32x2_t &b)
{
    const f32x2_t s = a + b, d = a - b;
    a = s; b = d;
}

// v[m] <- sum_n v[n] W8^{nm} (natural order in, natural order out)
template <bool FWD>
__device__ __forceinline__ void dft8(f32x2_t &x0, f32x2_t &x1, f32x2_t &x2, f32x2_t &x3, f32x2_t &x4, f32x2_t &x5, f32x2_t &x6, f32x2_t &x7)
{
    // n = n0 + 2 n1 (n0 < 2, n1 < 4): radix-4 over n1 for even and odd n, twiddle W8^{n0 k1}, radix-2 over n0
    f32x2_t e0 = x0, e1 = x2, e2 = x4, e3 = x6, o0 = x1, o1 = x3, o2 = x5, o3 = x7;
    radix4<FWD>(e0, e1, e2, e3);        // E[k1]
    radix4<FWD>(o0, o1, o2, o3);        // O[k1]
    const float H = 0.70710678118654752f, sg = FWD ? -1.f : 1.f;
    o1 = cmul_pk(o1, f32x2_t{H, sg * H});                               // W8^1
    o2 = FWD ? f32x2_t{o2.y, -o2.x} : f32x2_t{-o2.y, o2.x};            // W8^2 = -/+ i
    o3 = cmul_pk(o3, f32x2_t{-H, sg * H});                              // W8^3
    x0 = e0 + o0; x4 = e0 - o0;
    x1 = e1 + o1; x5 = e1 - o1;
    x2 = e2 + o2; x6 = e2 - o2;
    x3 = e3 + o3; x7 = e3 - o3;
}

template <int N, bool FWD, int MODE>
__global__ void __launch_bounds__(256, 3)
fft16x_kernel(const float *__restrict__ window, const float2 *__restrict__ twiddle, const float2 *__restrict__ in,
              float2 *__restrict__ out, int nvec)
{
    constexpr int LPV = N / 16;                 // lanes per vector
    constexpr int VPG = 256 / LPV;              // vectors per workgroup step
    constexpr bool TWO = N >= 256;              // two radix-16 passes (N = 32, 64, 128: one, then the radix N / 16 pass)
    constexpr int P3 = TWO ? 256 : 16;          // transform length done before the last pass
    constexpr int R3 = N / P3;                  // radix of the last pass (1: none)
    constexpr int NB3 = R3 > 1 ? 16 / R3 : 0;   // its butterflies per lane
    constexpr int VS = N + N / 16;              // LDS slots per vector (one pad slot per 16)
    __shared__ f32x2_t S[VPG * VS];
    __shared__ f32x2_t W2[16 * 17];
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    constexpr bool WIN = MODE & 1, SHIFT = MODE & 2;
    constexpr int SW_IN = (!FWD && SHIFT && !WIN) ? 8 : 0;
    constexpr bool SHIFT_OUT = FWD && SHIFT;
    const int t = threadIdx.x;
    const int l = t % LPV, vl = t / LPV;        // lane inside its vector, vector inside the group
    f32x2_t *Sv = S + vl * VS;

    {   // middle pass: W_256^{k q} = table[(N / 256) k q]
        const float2 w = twiddle[(N / 256) * (t >> 4) * (t & 15)];
        W2[(t >> 4) * 17 + (t & 15)] = f32x2_t{w.x, w.y};
    }
    // third pass: butterfly b of the lane is i = l + LPV b, k = i mod 256, twiddles W_N^{k m}, m = 1 .. R3 - 1 (forward sign)
    f32x2_t w3[NB3 > 0 ? NB3 * (R3 - 1) : 1];
    if (R3 > 1) {
#pragma unroll
        for (int b = 0; b < NB3; ++b)
#pragma unroll
            for (int m = 1; m < R3; ++m) {
                const int k = (l + LPV * b) & (P3 - 1);
                const float2 w = twiddle[(k * m) & (N - 1)];
                w3[b * (R3 - 1) + m - 1] = f32x2_t{w.x, w.y};
            }
    }
    float wn[WIN ? 16 : 1];
    if (WIN) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wn[q] = window[l + LPV * q];
    }
    const long long ngroups = ((long long)nvec + VPG - 1) / VPG;
    f32x2_t pre[16];
    auto request = [&](long long grp) __attribute__((always_inline)) {
        const long long vec = grp * VPG + vl;
        // (a whole group through one descriptor: vectors past nvec are out of its range)
        const long long left = ((long long)nvec - grp * VPG) * (long long)(N * 8);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + grp * VPG * (long long)N), 0,
                                                                          (int)(left < (long long)VPG * N * 8 ? left : (long long)VPG * N * 8), 0x00020000);
        (void)vec;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            pre[q] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, 8 * (vl * N + l), 8 * LPV * (q ^ SW_IN), 0));
    };
    long long grp = blockIdx.x;
    if (grp < ngroups) request(grp);
    for (; grp < ngroups; grp += gridDim.x) {
        f32x2_t v[16];
        // ---- pass 1 (p = 1)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = WIN ? pre[q] * wn[q] : pre[q];
        dft16<FWD>(v);
#pragma unroll
        for (int m = 0; m < 16; ++m) Sv[17 * l + m] = v[m];
        __syncthreads();
        if (grp + gridDim.x < ngroups) request(grp + gridDim.x);
        // ---- pass 2 (p = 16)
        if (TWO) {
            const int k = l & 15;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = Sv[l + LPV * q + (LPV >= 16 ? (l >> 4) + (LPV / 16) * q : (LPV * q) / 16)];
            __syncthreads();
#pragma unroll
            for (int q0 = 0; q0 < 16; q0 += 2) {
#pragma unroll
                for (int q = q0 ? q0 : 1; q < q0 + 2; ++q) v[q] = FWD ? cmul_pk(v[q], W2[k * 17 + q]) : cmul_conj_pk(v[q], W2[k * 17 + q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            dft16<FWD>(v);
        }
        const long long vec = grp * VPG + vl;
        const long long left = ((long long)nvec - grp * VPG) * (long long)(N * 8);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(out + grp * VPG * (long long)N, 0,
                                                                           (int)(left < (long long)VPG * N * 8 ? left : (long long)VPG * N * 8), 0x00020000);
        (void)vec;
        const int j2 = (l - (l & 15)) * 16 + (l & 15);          // pass 2 writes element j2 + 16 m
        if (TWO && R3 == 1) {
            // N = 256: pass 2 is the last one; a shift by N / 2 = 128 = 16 * 8 is m ^ 8
#pragma unroll
            for (int m = 0; m < 16; ++m)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v[m]), yr, 8 * (vl * N + j2), 8 * 16 * (m ^ (SHIFT_OUT ? 8 : 0)), 0);
            continue;
        }
        if (TWO) {
#pragma unroll
            for (int m = 0; m < 16; ++m) Sv[j2 + (j2 >> 4) + 17 * m] = v[m];
            __syncthreads();
        }
        // ---- last pass (p = P3, radix R3): butterfly b works on points q = b + NB3 m
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = Sv[l + LPV * q + (LPV >= 16 ? (l >> 4) + (LPV / 16) * q : (LPV * q) / 16)];
        __syncthreads();                                        // S belongs to the next group from here
#pragma unroll
        for (int b = 0; b < NB3; ++b) {
#pragma unroll
            for (int m = 1; m < R3; ++m) {
                f32x2_t &x = v[b + NB3 * m];
                x = FWD ? cmul_pk(x, w3[b * (R3 - 1) + m - 1]) : cmul_conj_pk(x, w3[b * (R3 - 1) + m - 1]);
            }
            if (R3 == 2) radix2<FWD>(v[b], v[b + NB3]);
            if (R3 == 4) radix4<FWD>(v[b], v[b + NB3], v[b + 2 * NB3], v[b + 3 * NB3]);
            if (R3 == 8) dft8<FWD>(v[b], v[b + NB3], v[b + 2 * NB3], v[b + 3 * NB3], v[b + 4 * NB3], v[b + 5 * NB3], v[b + 6 * NB3], v[b + 7 * NB3]);
            // output m of the butterfly is element j + P3 m, i = l + LPV b, k = i mod P3, j = (i - k) R3 + k
            const int i = l + LPV * b, k = i & (P3 - 1), j = (i - k) * R3 + k;
#pragma unroll
            for (int m = 0; m < R3; ++m)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v[b + NB3 * m]), yr, 8 * (vl * N + j),
                                                      8 * P3 * (m ^ (SHIFT_OUT ? R3 / 2 : 0)), 0);
        }
    }
}

template <int N, bool FWD>
static void launch_fft16x_t(int shift, const float *window, const float2 *twiddle, const float2 *in, float2 *out,
                            long long nvec, hipStream_t st)
{
    constexpr int VPG = 4096 / N;
    const long long ngroups = (nvec + VPG - 1) / VPG;
    const long long cap = 3LL * fft_num_cus();
    const dim3 grid((unsigned)(ngroups < cap ? ngroups : cap));
    const int nv = (int)nvec;
    switch ((window ? 1 : 0) | (shift ? 2 : 0)) {
    case 0: hipLaunchKernelGGL((fft16x_kernel<N, FWD, 0>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 1: hipLaunchKernelGGL((fft16x_kernel<N, FWD, 1>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 2: hipLaunchKernelGGL((fft16x_kernel<N, FWD, 2>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    default: hipLaunchKernelGGL((fft16x_kernel<N, FWD, 3>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    }
}

static int launch_fft16x(int N, int forward, int shift, const float *window, const float2 *twiddle, const float2 *in,
                         float2 *out, long long nvec, hipStream_t st)
{
    if (nvec > 0x7fffffffLL) return fail(GRHIP_EINVAL, "fft: too many vectors in one call");
#define GRHIP_FFT16X(NN) do { if (forward) launch_fft16x_t<NN, true>(shift, window, twiddle, in, out, nvec, st); \
                              else launch_fft16x_t<NN, false>(shift, window, twiddle, in, out, nvec, st); } while (0)
    switch (N) {
    case 32: GRHIP_FFT16X(32); break;
    case 64: GRHIP_FFT16X(64); break;
    case 128: GRHIP_FFT16X(128); break;
    case 256: GRHIP_FFT16X(256); break;
    case 512: GRHIP_FFT16X(512); break;
    case 1024: GRHIP_FFT16X(1024); break;
    default: GRHIP_FFT16X(2048); break;
    }
#undef GRHIP_FFT16X
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_fft_vcc, N = 8192: two interleaved 4096-point transforms (even and odd samples: decimation in time) on the
// radix-16 register machinery above, and one radix-2 combine on the way out: X[k] = E[k] + W^k O[k],
// X[k + 4096] = E[k] - W^k O[k].  A lane loads sample pairs (x[2j], x[2j+1]) with 16-byte loads; the even half goes through
// its three passes, then the odd half (the same LDS buffer, 37 KB: three workgroups per CU instead of the one that the
// 128 KB ping-pong buffers of the radix-4 kernel allowed: 0.12 of the HBM peak).  Persistent workgroups, the sub-transform
// twiddles (table entries 2m) and the sixteen combine twiddles W^{t + 256 m} of the lane resident.
// MODE as fft4096_kernel: bit 0 window, bit 1 shift (a shift by N/2 swaps the two output halves / is q ^ 8 on the way in).
// ===========================================================================
template <bool FWD, int MODE>
__global__ void __launch_bounds__(256, (MODE & 1) ? 2 : 3)      // (the window's values in flight beside the points: two per CU)
fft8192_kernel(const float *__restrict__ window, const float2 *__restrict__ twiddle, const float2 *__restrict__ in,
               float2 *__restrict__ out, int nvec)
{
    constexpr int N = 8192, H = 4096;
    __shared__ f32x2_t S[H + H / 16];
    __shared__ f32x2_t W2[16 * 17];
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    constexpr bool WIN = MODE & 1, SHIFT = MODE & 2;
    constexpr int SW_IN = (!FWD && SHIFT && !WIN) ? 8 : 0;
    constexpr bool SWAP_OUT = FWD && SHIFT;
    const int t = threadIdx.x;
    f32x2_t w3[16], wc[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float2 w = twiddle[2 * t * q]; w3[q] = f32x2_t{w.x, w.y}; }      // W_4096^{t q}, forward sign
#pragma unroll
    for (int m = 0; m < 16; ++m) { const float2 w = twiddle[t + 256 * m]; wc[m] = f32x2_t{w.x, w.y}; }    // W_8192^{t + 256 m}
    {
        const float2 w = twiddle[2 * 16 * (t >> 4) * (t & 15)];
        W2[(t >> 4) * 17 + (t & 15)] = f32x2_t{w.x, w.y};              // visible after the first barrier of the loop
    }
    for (int vec = blockIdx.x; vec < nvec; vec += gridDim.x) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in + (long long)vec * N), 0, N * 8, 0x00020000);
        f32x2_t ve[16], vo[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {       // samples 2j, 2j + 1, j = t + 256 (q ^ SW_IN)
            const f32x4_t u = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, 16 * t, 4096 * (q ^ SW_IN), 0));
            ve[q] = f32x2_t{u[0], u[1]};
            vo[q] = f32x2_t{u[2], u[3]};
        }
        if (WIN) {
            const float2 *wp = reinterpret_cast<const float2 *>(window);
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float2 ww = wp[t + 256 * q]; ve[q] = ve[q] * ww.x; vo[q] = vo[q] * ww.y; }
        }
        dft16<FWD>(ve);
        fft4096_mid_passes<FWD>(ve, S, W2, w3, t);          // ve[m] = E[t + 256 m]
        dft16<FWD>(vo);
        fft4096_mid_passes<FWD>(vo, S, W2, w3, t);          // vo[m] = O[t + 256 m]
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(out + (long long)vec * N, 0, N * 8, 0x00020000);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const f32x2_t p = FWD ? cmul_pk(vo[m], wc[m]) : cmul_conj_pk(vo[m], wc[m]);
            const f32x2_t lo = ve[m] + p, hi = ve[m] - p;   // X[k], X[k + 4096], k = t + 256 m
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, SWAP_OUT ? hi : lo), yr, 8 * t, 2048 * m, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, SWAP_OUT ? lo : hi), yr, 8 * t + 8 * H, 2048 * m, 0);
        }
    }
}

template <bool FWD>
static void launch_fft8192_t(int shift, const float *window, const float2 *twiddle, const float2 *in, float2 *out,
                             long long nvec, hipStream_t st)
{
    const long long cap = (window ? 2LL : 3LL) * fft_num_cus();
    const dim3 grid((unsigned)(nvec < cap ? nvec : cap));
    const int nv = (int)nvec;
    switch ((window ? 1 : 0) | (shift ? 2 : 0)) {
    case 0: hipLaunchKernelGGL((fft8192_kernel<FWD, 0>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 1: hipLaunchKernelGGL((fft8192_kernel<FWD, 1>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    case 2: hipLaunchKernelGGL((fft8192_kernel<FWD, 2>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    default: hipLaunchKernelGGL((fft8192_kernel<FWD, 3>), grid, dim3(256), 0, st, window, twiddle, in, out, nv); break;
    }
}

static int launch_fft8192(int forward, int shift, const float *window, const float2 *twiddle, const float2 *in, float2 *out,
                          long long nvec, hipStream_t st)
{
    if (nvec > 0x7fffffffLL) return fail(GRHIP_EINVAL, "fft: too many vectors in one call");
    if (forward) launch_fft8192_t<true>(shift, window, twiddle, in, out, nvec, st);
    else launch_fft8192_t<false>(shift, window, twiddle, in, out, nvec, st);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// REAL: float items in and out (gr_fir_fff shapes the tiled kernel does not take): the imaginary
// half of the transform idles, still far ahead of the one-output-per-lane generic kernel.
// FOLD = log2(decimation) for decimations 2, 4, 8, 16 (0: any decimation, full-size inverse).  Keeping every
// D-th output of the block, starting at its first valid one (offset ntaps-1: that shift is folded into H on the
// host), is the (4096/D)-point inverse transform of the spectrum folded D times: the inverse costs 1/D-th.
// Persistent workgroups (three per CU) walk the blocks as fft4096_kernel walks its vectors: twiddles resident, the next
// block's points requested under this block's backward transform.
template <bool REAL, int FOLD>
__global__ void __launch_bounds__(256, 3)
fftfilt4096_kernel(const void *__restrict__ in_v, long long nin, const void *__restrict__ hist_v, int ntaps,
                   const float2 *__restrict__ twiddle, const float2 *__restrict__ H, void *__restrict__ out_v,
                   long long nout, int decim, int L, long long nblk, void *__restrict__ hist_new_v)
{
    constexpr int N = 4096;
    __shared__ f32x2_t S[N + N / 16];
    __shared__ f32x2_t W2[16 * 17];
    typedef unsigned int ols_u32x2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    // the history of the NEXT call (the last ntaps - 1 items of history ++ stream; its own buffer, the caller flips the
    // two): the last workgroup writes it on its way in (a kernel of its own cost 5 us per call)
    if (hist_new_v && blockIdx.x == gridDim.x - 1) {
        const int hlen = ntaps - 1;
        for (int j = t; j < hlen; j += 256) {
            const long long i = nin - hlen + j;          // index into the stream, negative: still in the old history
            if (REAL) ((float *)hist_new_v)[j] = i >= 0 ? ((const float *)in_v)[i] : ((const float *)hist_v)[i + hlen];
            else ((float2 *)hist_new_v)[j] = i >= 0 ? ((const float2 *)in_v)[i] : ((const float2 *)hist_v)[i + hlen];
        }
    }
    f32x2_t w3[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float2 w = twiddle[t * q]; w3[q] = f32x2_t{w.x, w.y}; }
    {
        const float2 w = twiddle[16 * (t >> 4) * (t & 15)];
        W2[(t >> 4) * 17 + (t & 15)] = f32x2_t{w.x, w.y};          // visible after the first barrier of the loop
    }
    // the stream through a raw buffer descriptor: items past nin (and, as unsigned offsets, before 0) read as zero
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(in_v), 0, (int)(nin * (REAL ? 4 : 8)), 0x00020000);
    f32x2_t pre[16];
    auto request = [&](long long b) __attribute__((always_inline)) {
        const long long base = b * L - (ntaps - 1);          // stream index of block position 0
        // Offsets: the hardware's range check looks at vector offset + immediate, not at the scalar offset, and a NEGATIVE
        // vector offset whose immediate brings it back into range does not come out as the in-range load it is (measured:
        // the stream's first sample read as zero).  So the whole offset goes into the VGPR, and the positions before the
        // stream (blocks with b L < ntaps - 1: one or two per call) get an explicit out-of-range
        // offset through a select (in every block: it also keeps the compiler from splitting the sum into register + immediate).
        int tq = t;
        asm volatile("" : "+v"(tq));                         // (recomputed per block: hoisted out of the loop, the sixteen offsets spill)
        const int vo = ((int)base + tq) * (REAL ? 4 : 8);
        const int first = base < 0 ? (int)-base : 0;         // block positions below this one lie before the stream
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int off = (tq + 256 * q >= first) ? vo + 256 * (REAL ? 4 : 8) * q : 0x7ffffff0;
            if (REAL) pre[q] = f32x2_t{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0)), 0.f};
            else pre[q] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0));
        }
    };
    long long b = blockIdx.x;
    if (b < nblk) request(b);
    for (; b < nblk; b += gridDim.x) {
        f32x2_t v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = pre[q];
        if (b * L < ntaps - 1) {
            // positions before the call come from the history (block 0, and the next ones too while b L < ntaps - 1):
            // block position p is history item b L + p, and every position past the history's end is out of its
            // descriptor's range (zero) -- as the stream's loads returned zero for the positions before the stream: the
            // sum of the two is the block, without a branch per lane
            const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(hist_v), 0, (ntaps - 1) * (REAL ? 4 : 8), 0x00020000);
            const int hb = ((int)(b * L) + t) * (REAL ? 4 : 8);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (REAL) v[q].x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr, hb + 256 * 4 * q, 0, 0));
                else v[q] = v[q] + __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(hr, hb + 256 * 8 * q, 0, 0));
            }
        }
        // the lane's sixteen bins of H travel (from L2) in the registers the block's points have just left, under the
        // forward transform; the next block's points take the same registers under the backward one
        f32x2_t Hr[16];
        const __amdgpu_buffer_rsrc_t Hd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(H), 0, N * 8, 0x00020000);
#pragma unroll
        for (int m = 0; m < 16; ++m) Hr[m] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(Hd, 8 * t, 2048 * m, 0));
        dft16<true>(v);
        fft4096_mid_passes<true>(v, S, W2, w3, t);           // v[m] = X[t + 256 m]
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = cmul_pk(v[m], Hr[m]);
        __builtin_amdgcn_sched_barrier(0);                  // (the requests must not be hoisted over the bins of H: same registers)
        if (FOLD == 0 && b + gridDim.x < nblk) request(b + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        if (FOLD > 0) {
            constexpr int DD = 1 << FOLD, NP = N >> FOLD, MP = 16 >> FOLD;     // decimation, inverse size, bins per lane
            int tf = t;
            asm volatile("" : "+v"(tf));                                    // (index arithmetic per block, not hoisted and spilled)
            // bin t + 256 m folds onto t + 256 (m mod MP): inside the lane
            f32x2_t *A = S, *B = S + NP;
            const __amdgpu_buffer_rsrc_t Td = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(twiddle), 0, N * 8, 0x00020000);
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                f32x2_t f = v[mp];
#pragma unroll
                for (int a = 1; a < DD; ++a) f = f + v[mp + a * MP];
                A[tf + 256 * mp] = f;
            }
            __syncthreads();
            if (b + gridDim.x < nblk) request(b + gridDim.x);       // (the spectrum has left its registers)
            __builtin_amdgcn_sched_barrier(0);
            // (4096/D)-point backward transform, radix-4 Stockham passes (+ one radix-2) between the halves of S
            f32x2_t *src = A, *dst = B;
            int p = 1;
            constexpr int T4 = NP >> 2;
            while (p * 4 <= NP) {
                const int tstep = (NP / (4 * p)) << FOLD;              // step in the 4096-entry twiddle table
#pragma unroll 1
                for (int i = tf; i < T4; i += 256) {
                    const int k = i & (p - 1);
                    const int j = ((i - k) << 2) + k;
                    const int m = k * tstep;
                    f32x2_t u0 = src[i], u1 = src[i + T4], u2 = src[i + 2 * T4], u3 = src[i + 3 * T4];
                    if (p > 1) {
                        u1 = cmul_conj_pk(u1, __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(Td, 8 * m, 0, 0)));
                        u2 = cmul_conj_pk(u2, __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(Td, 16 * m, 0, 0)));
                        u3 = cmul_conj_pk(u3, __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(Td, 24 * m, 0, 0)));
                    }
                    radix4<false>(u0, u1, u2, u3);
                    dst[j] = u0; dst[j + p] = u1; dst[j + 2 * p] = u2; dst[j + 3 * p] = u3;
                }
                __syncthreads();
                f32x2_t *tmp = src; src = dst; dst = tmp;
                p <<= 2;
            }
            if (p < NP) {                                             // one radix-2 pass, p == NP/2
                constexpr int T2 = NP >> 1;
#pragma unroll 1
                for (int i = tf; i < T2; i += 256) {
                    f32x2_t u0 = src[i], u1 = src[i + T2];
                    if (p > 1) u1 = cmul_conj_pk(u1, __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(Td, 8 * ((i & (p - 1)) << FOLD), 0, 0)));
                    dst[i] = u0 + u1;
                    dst[i + p] = u0 - u1;
                }
                __syncthreads();
                f32x2_t *tmp = src; src = dst; dst = tmp;
            }
            const int nvalid = L >> FOLD;                             // outputs of this block
#pragma unroll 1
            for (int r = tf; r < nvalid; r += 256) {
                const long long n = b * nvalid + r;
                if (n < nout) {
                    if (REAL) ((float *)out_v)[n] = src[r].x;
                    else ((f32x2_t *)out_v)[n] = src[r];
                }
            }
            __syncthreads();                                          // S belongs to the next block from here
            continue;
        }
        dft16<false>(v);                                     // inverse pass 1 on the same registers
        fft4096_mid_passes<false>(v, S, W2, w3, t);          // v[m] = z[t + 256 m]
        if (decim == 1) {
            // point t + 256 m is output b L + j, j = t + 256 m - (ntaps - 1), when 0 <= j < L: through a descriptor of
            // the output stream (outputs past nout are out of its range), offsets of the other points out of range too
            const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(out_v, 0, (int)(nout * (REAL ? 4 : 8)), 0x00020000);
            const int j0 = t - (ntaps - 1);
            const int ob = (int)((b * L + j0) * (REAL ? 4 : 8));
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int j = j0 + 256 * m;
                const int off = (j >= 0 && j < L) ? ob + 256 * (REAL ? 4 : 8) * m : (int)0xfffffff0;
                if (REAL) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[m].x), orr, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ols_u32x2, v[m]), orr, off, 0, 0);
            }
            continue;
        }
#pragma unroll 1
        for (int m = 0; m < 16; ++m) {                       // any other decimation (rare here: 2, 4, 8, 16 fold)
            const int j = t + 256 * m - (ntaps - 1);         // offset of this point inside the block's outputs
            if (j >= 0 && j < L && (j % decim) == 0) {
                const long long n = (b * L + j) / decim;
                if (n < nout) {
                    if (REAL) ((float *)out_v)[n] = v[m].x;
                    else ((f32x2_t *)out_v)[n] = v[m];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256)
fftfilt_hist_kernel(const float2 *__restrict__ in, long long nin, const float2 *__restrict__ hist_old,
                    float2 *__restrict__ hist_new, int hlen)
{
    // the last hlen items of (hist_old ++ in)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hlen) return;
    const long long i = nin - hlen + j;                  // index into `in`, negative: still in the old history
    hist_new[j] = i >= 0 ? in[i] : hist_old[i + hlen];
}

int ols_build(const float *taps_cplx, int ntaps, int decim, DevBuf &d_tw, DevBuf &d_H, int *L, int *fold)
{
    if (ntaps < 1 || ntaps > OLS_MAX_TAPS || decim < 1 || (OLS_N - (ntaps - 1)) / decim < 1)
        return fail(GRHIP_EINVAL, "overlap-save engine: %d taps / decimation %d not supported", ntaps, decim);
    *L = ((OLS_N - (ntaps - 1)) / decim) * decim;
    // decimations 2, 4, 8, 16: folded spectrum + small inverse (see fftfilt4096_kernel)
    *fold = decim == 2 ? 1 : decim == 4 ? 2 : decim == 8 ? 3 : decim == 16 ? 4 : 0;
    std::vector<float2> tw((size_t)OLS_N), H((size_t)OLS_N);
    std::vector<double> cs((size_t)OLS_N), sn((size_t)OLS_N);
    for (int k = 0; k < OLS_N; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)OLS_N;
        cs[k] = cos(ang); sn[k] = sin(ang);
        tw[k] = make_float2((float)cs[k], (float)sn[k]);
    }
    const double sc = 1.0 / OLS_N;
    for (int k = 0; k < OLS_N; ++k) {
        double ar = 0, ai = 0;
        for (int i = 0; i < ntaps; ++i) {
            const int m = (int)(((long long)k * i) & (OLS_N - 1));
            const double tr = taps_cplx[2 * i], ti = taps_cplx[2 * i + 1];
            ar += tr * cs[m] - ti * sn[m];
            ai += tr * sn[m] + ti * cs[m];
        }
        if (*fold) {        // advance the block by ntaps-1 samples: the first valid output becomes output 0
            const int m = (int)(((long long)k * (ntaps - 1)) & (OLS_N - 1));
            const double cr = cs[m], ci = -sn[m];               // e^{+j 2 pi k (ntaps-1) / N}
            const double r2 = ar * cr - ai * ci, i2 = ar * ci + ai * cr;
            ar = r2; ai = i2;
        }
        H[k] = make_float2((float)(ar * sc), (float)(ai * sc));
    }
    int rc = d_tw.reserve(tw.size() * sizeof(float2));
    if (!rc) rc = d_H.reserve(H.size() * sizeof(float2));
    if (rc) return rc;
    GRHIP_HIP(hipMemcpy(d_tw.p, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    GRHIP_HIP(hipMemcpy(d_H.p, H.data(), H.size() * sizeof(float2), hipMemcpyHostToDevice));
    return GRHIP_OK;
}

static int launch_fftfilt_hist_any(bool real, const void *in, long long nin, const void *hist_old, void *hist_new, int hlen, hipStream_t st);

template <bool REAL>
static int launch_fftfilt4096_t(const void *in, long long nin, const void *hist, int ntaps, const float2 *twiddle,
                                const float2 *H, void *out, long long nout, int decim, int L, int fold, hipStream_t st,
                                void *hist_new)
{
    if (nout <= 0) return GRHIP_OK;
    // The kernel addresses a launch's input and output through 32-bit byte offsets: a longer call goes out in pieces of
    // whole blocks (blocks are independent: the history of a later piece is the stream itself).
    constexpr long long ISZ = REAL ? 4 : 8;
    const long long max_blocks = (0x7fff0000LL / ISZ - 2 * OLS_N) / L;
    const long long nblk_all = (nin + L - 1) / L;
    const long long cap = 3LL * fft_num_cus();
    for (long long b0 = 0; b0 < nblk_all; b0 += max_blocks) {
        const long long nb = nblk_all - b0 < max_blocks ? nblk_all - b0 : max_blocks;
        const long long i0 = b0 * L, o0 = b0 * (L / decim);
        const void *in_c = (const char *)in + i0 * ISZ;
        const void *hist_c = b0 == 0 ? hist : (const void *)((const char *)in_c - (long long)(ntaps - 1) * ISZ);
        void *out_c = (char *)out + o0 * ISZ;
        const bool last = b0 + nb >= nblk_all;
        // (a piece in the middle reads on into the next one: its last block needs up to 4096 items from its start)
        const long long nin_rd = last || nin - i0 < nb * (long long)L + OLS_N ? nin - i0 : nb * (long long)L + OLS_N;
        const long long nout_c = last ? nout - o0 : nb * (L / decim);
        const unsigned grid = (unsigned)(nb < cap ? nb : cap);
        void *hn = last ? hist_new : nullptr;
        if (hn && b0 != 0) {
            // the next call's history is defined against the whole call: written by a piece only when the piece is the call
            const int rc = launch_fftfilt_hist_any(REAL, in, nin, hist, hn, ntaps - 1, st);
            if (rc) return rc;
            hn = nullptr;
        }
#define GRHIP_OLS_LAUNCH(F) hipLaunchKernelGGL((fftfilt4096_kernel<REAL, F>), dim3(grid), dim3(256), 0, st, in_c, nin_rd, hist_c, ntaps, \
                                               twiddle, H, out_c, nout_c, decim, L, nb, hn)
        switch (fold) {
        case 1: GRHIP_OLS_LAUNCH(1); break;
        case 2: GRHIP_OLS_LAUNCH(2); break;
        case 3: GRHIP_OLS_LAUNCH(3); break;
        case 4: GRHIP_OLS_LAUNCH(4); break;
        default: GRHIP_OLS_LAUNCH(0); break;
        }
#undef GRHIP_OLS_LAUNCH
        GRHIP_HIP(hipGetLastError());
    }
    return GRHIP_OK;
}

int launch_fftfilt4096(const float2 *in, long long nin, const float2 *hist, int ntaps, const float2 *twiddle,
                       const float2 *H, float2 *out, long long nout, int decim, int L, int fold, hipStream_t st,
                       float2 *hist_new)
{
    return launch_fftfilt4096_t<false>(in, nin, hist, ntaps, twiddle, H, out, nout, decim, L, fold, st, hist_new);
}

int launch_fftfilt4096_real(const float *in, long long nin, const float *hist, int ntaps, const float2 *twiddle,
                            const float2 *H, float *out, long long nout, int decim, int L, int fold, hipStream_t st)
{
    return launch_fftfilt4096_t<true>(in, nin, hist, ntaps, twiddle, H, out, nout, decim, L, fold, st, nullptr);
}

static int launch_fftfilt_hist_any(bool real, const void *in, long long nin, const void *hist_old, void *hist_new, int hlen, hipStream_t st)
{
    if (real) return fail(GRHIP_EINVAL, "overlap-save engine: no history buffer for real data");
    return launch_fftfilt_hist((const float2 *)in, nin, (const float2 *)hist_old, (float2 *)hist_new, hlen, st);
}

int launch_fftfilt_hist(const float2 *in, long long nin, const float2 *hist_old, float2 *hist_new, int hlen, hipStream_t st)
{
    if (hlen <= 0) return GRHIP_OK;
    hipLaunchKernelGGL(fftfilt_hist_kernel, dim3((hlen + 255) / 256), dim3(256), 0, st, in, nin, hist_old, hist_new, hlen);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_fft_filter_ccc (filter/gri_fft_filter_ccc_generic.cc:121-169): the pieces around the
// two transforms.  Blocks are independent except for the tail that block b adds into the
// head of block b+1, so all blocks of a call are transformed in one batched launch and the
// overlap-add + decimation is a gather.
// ===========================================================================
__global__ void __launch_bounds__(256)
fftfilt_pack_kernel(const float2 *__restrict__ in, float2 *__restrict__ blocks, int nsamples, int fftsize, long long total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long b = i / fftsize;
        const int j = (int)(i - b * fftsize);
        blocks[i] = j < nsamples ? in[b * nsamples + j] : make_float2(0.f, 0.f);      // :128-131
    }
}

__global__ void __launch_bounds__(256)
fftfilt_mul_kernel(float2 *__restrict__ blocks, const float2 *__restrict__ xformed, int fftsize, long long total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) blocks[i] = cmul_ref(blocks[i], xformed[i % fftsize]);   // :139-141
}

__global__ void __launch_bounds__(256)
fftfilt_ola_kernel(const float2 *__restrict__ blocks, const float2 *__restrict__ tail, float2 *__restrict__ out,
                   long long nitems, int decim, int nsamples, int fftsize, int tailsize)
{
    long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; o < nitems; o += stride) {
        const long long i = o * decim;                      // position in the full-rate overlap-added sequence
        const long long b = i / nsamples;
        const int j = (int)(i - b * nsamples);
        float2 v = blocks[b * fftsize + j];
        if (j < tailsize) {                                 // :147-148 outbuf[j] += d_tail[j]
            const float2 t = b > 0 ? blocks[(b - 1) * fftsize + nsamples + j] : tail[j];
            v.x += t.x; v.y += t.y;
        }
        out[o] = v;
    }
}

__global__ void __launch_bounds__(256)
fftfilt_tail_kernel(const float2 *__restrict__ blocks, float2 *__restrict__ tail, long long nblk, int nsamples,
                    int fftsize, int tailsize)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < tailsize) tail[j] = blocks[(nblk - 1) * fftsize + nsamples + j];            // :160-161
}

static unsigned grid_for(long long n)
{
    long long b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

int launch_fftfilt_pack(const float2 *in, float2 *blocks, int nsamples, int fftsize, long long nblk, hipStream_t st)
{
    const long long total = nblk * fftsize;
    hipLaunchKernelGGL(fftfilt_pack_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, blocks, nsamples, fftsize, total);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_fftfilt_mul(float2 *blocks, const float2 *xformed, int fftsize, long long nblk, hipStream_t st)
{
    const long long total = nblk * fftsize;
    hipLaunchKernelGGL(fftfilt_mul_kernel, dim3(grid_for(total)), dim3(256), 0, st, blocks, xformed, fftsize, total);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_fftfilt_ola(const float2 *blocks, const float2 *tail, float2 *out, long long nitems, int decim, int nsamples,
                       int fftsize, int tailsize, hipStream_t st)
{
    hipLaunchKernelGGL(fftfilt_ola_kernel, dim3(grid_for(nitems)), dim3(256), 0, st, blocks, tail, out, nitems, decim,
                       nsamples, fftsize, tailsize);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_fftfilt_tail(const float2 *blocks, float2 *tail, long long nblk, int nsamples, int fftsize, int tailsize,
                        hipStream_t st)
{
    if (tailsize <= 0) return GRHIP_OK;
    hipLaunchKernelGGL(fftfilt_tail_kernel, dim3((tailsize + 255) / 256), dim3(256), 0, st, blocks, tail, nblk,
                       nsamples, fftsize, tailsize);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

}  // namespace grhip
