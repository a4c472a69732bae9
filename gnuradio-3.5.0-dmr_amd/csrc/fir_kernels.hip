// fir_kernels.hip -- gfx950 kernels for the FIR family of the DMR hot path:
//   gr_fir_{fff,ccf,ccc} (filterN/filterNdec), gr_fir_filter_XXX,
//   gr_freq_xlating_fir_filter_ccc (+ gr_rotator), gr_quadrature_demod_cf.
//
// Two kernel families:
//  (A) fir_generic_kernel  -- one output per lane, summation order and unfused
//      arithmetic of gr_fir_XXX_generic (filter/gr_fir_XXX_generic.cc.t:30-79):
//      bit-exact parity path, any taps / decimation.
//  (B) fir_tiled_kernel    -- the throughput path.  A 256-lane workgroup owns a
//      tile of 256*R consecutive outputs.  The input tile is staged once from
//      HBM with coalesced 16-byte loads into LDS, de-interleaved into the D
//      polyphase components (x_p[m] = x[mD+p]) with one pad slot per R samples
//      so that lane stride is R+1 (odd) 8-byte slots: conflict-free
//      ds_read_b64.  Each lane keeps R complex accumulators and an R-deep
//      sliding window of samples in VGPRs: one LDS read feeds R MACs.  Taps are
//      wave-uniform and arrive through scalar loads (SGPRs), phase-major.
//      VALU-bound by design; no MFMA (vector x scalar work).
//      Epilogues: rotator multiply (xlating) and fused quadrature demod.
#include "fir_kernels.h"

#include <cstdlib>

#include "device_math.h"
#include "grhip_internal.h"

namespace grhip {

// ===========================================================================
// (A) generic-order kernel
// ===========================================================================
template <int KIND>
__global__ void __launch_bounds__(256)
fir_generic_kernel(const float *__restrict__ taps_rev, int ntaps, const float *__restrict__ in,
                   float *__restrict__ out, long long n_out, int decim,
                   const float2 *__restrict__ gtab)
{
    extern __shared__ __attribute__((aligned(16))) float s_taps[];
    const int tw = (KIND == FIR_CCC) ? 2 : 1;
    for (int i = threadIdx.x; i < ntaps * tw; i += blockDim.x) s_taps[i] = taps_rev[i];
    __syncthreads();
    long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) return;

    if (KIND == FIR_FFF) {
        // N_UNROLL 4, float accumulators (.cc.t:30-55)
        const float *x = in + n * decim;
        float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        int i = 0, nn = (ntaps / 4) * 4;
        for (i = 0; i < nn; i += 4) {
            acc0 += s_taps[i + 0] * x[i + 0];
            acc1 += s_taps[i + 1] * x[i + 1];
            acc2 += s_taps[i + 2] * x[i + 2];
            acc3 += s_taps[i + 3] * x[i + 3];
        }
        for (; i < ntaps; i++) acc0 += s_taps[i] * x[i];
        out[n] = (acc0 + acc1 + acc2 + acc3);
    } else {
        // N_UNROLL 2, complex accumulators (.cc.t:59-79)
        const float2 *x = (const float2 *)in + n * decim;
        float a0r = 0, a0i = 0, a1r = 0, a1i = 0;
        int i = 0, nn = (ntaps / 2) * 2;
        if (KIND == FIR_CCF) {
            for (i = 0; i < nn; i += 2) {
                float2 v0 = x[i], v1 = x[i + 1];
                float t0 = s_taps[i], t1 = s_taps[i + 1];
                float pr = v0.x * t0, pi = v0.y * t0;
                a0r += pr; a0i += pi;
                pr = v1.x * t1; pi = v1.y * t1;
                a1r += pr; a1i += pi;
            }
            for (; i < ntaps; i++) {
                float2 v0 = x[i];
                float t0 = s_taps[i];
                float pr = v0.x * t0, pi = v0.y * t0;
                a0r += pr; a0i += pi;
            }
        } else {
            const float2 *tc = (const float2 *)s_taps;
            for (i = 0; i < nn; i += 2) {
                float2 p0 = cmul_ref(tc[i], x[i]);
                a0r += p0.x; a0i += p0.y;
                float2 p1 = cmul_ref(tc[i + 1], x[i + 1]);
                a1r += p1.x; a1i += p1.y;
            }
            for (; i < ntaps; i++) {
                float2 p0 = cmul_ref(tc[i], x[i]);
                a0r += p0.x; a0i += p0.y;
            }
        }
        float2 y = make_float2(a0r + a1r, a0i + a1i);
        if (gtab) y = cmul_ref(y, gtab[n]);     // gr_rotator::rotate: z = in * d_phase
        ((float2 *)out)[n] = y;
    }
}

int launch_fir_generic(FirKind kind, const float *taps_rev, int ntaps, const void *in, void *out,
                       long long n_out, int decim, const float2 *gtab, hipStream_t st)
{
    if (n_out <= 0) return GRHIP_OK;
    size_t sh = (size_t)(ntaps > 0 ? ntaps : 1) * (kind == FIR_CCC ? 8 : 4);
    if (sh > 160 * 1024 - 256) return fail(GRHIP_EINVAL, "generic FIR: %d taps exceed LDS", ntaps);
    dim3 grid((unsigned)((n_out + 255) / 256)), block(256);
    switch (kind) {
    case FIR_FFF:
        if (sh > 64 * 1024)
            GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_kernel<FIR_FFF>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(fir_generic_kernel<FIR_FFF>, grid, block, sh, st, taps_rev, ntaps,
                           (const float *)in, (float *)out, n_out, decim, gtab);
        break;
    case FIR_CCF:
        if (sh > 64 * 1024)
            GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_kernel<FIR_CCF>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(fir_generic_kernel<FIR_CCF>, grid, block, sh, st, taps_rev, ntaps,
                           (const float *)in, (float *)out, n_out, decim, gtab);
        break;
    default:
        if (sh > 64 * 1024)
            GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_kernel<FIR_CCC>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(fir_generic_kernel<FIR_CCC>, grid, block, sh, st, taps_rev, ntaps,
                           (const float *)in, (float *)out, n_out, decim, gtab);
        break;
    }
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// (B) tiled kernel
// ===========================================================================
constexpr int TILED_R = 8;
constexpr int TILED_LOGR = 3;
constexpr int TILED_THREADS = 256;
constexpr int TILED_NT = TILED_THREADS * TILED_R;

int tiled_R() { return TILED_R; }
int tiled_NT() { return TILED_NT; }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// LDS geometry (in float2 slots).  mm = m + R where m is the polyphase sample
// index relative to the tile's first real output (m = -1 is the boundary
// output's first sample).  slot(mm) = mm + mm/R.
__host__ __device__ inline int tiled_phase_stride(int Tq)
{
    int MM = TILED_NT + Tq + 2 * TILED_R;
    return MM + (MM >> TILED_LOGR) + 1;
}
__host__ inline size_t tiled_lds_bytes(int D, int Tq)
{
    return (size_t)D * tiled_phase_stride(Tq) * sizeof(float2) + (TILED_THREADS + 8) * sizeof(float2);
}

bool tiled_supported(int decim, int Tq)
{
    if (!(decim == 1 || decim == 2 || decim == 4 || decim == 8)) return false;
    if (Tq <= 0 || (Tq % TILED_R) != 0) return false;
    return tiled_lds_bytes(decim, Tq) <= 80 * 1024;   // two workgroups per CU
}

template <int D, bool CTAPS, bool PREMIX, int EPI>
__global__ void __launch_bounds__(TILED_THREADS, 2) fir_tiled_kernel(const FirTiledArgs a)
{
    constexpr int R = TILED_R, LOGR = TILED_LOGR, NT = TILED_NT;
    constexpr int LOGD = ilog2(D);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *xs = (float2 *)smem;

    const int t = threadIdx.x;
    const int s = blockIdx.y;
    const long long n0 = (long long)blockIdx.x * NT;
    const float2 *__restrict__ x = a.x + (long long)s * a.x_stride;
    const int Tq = a.Tq;
    const int PS = tiled_phase_stride(Tq);
    float2 *red = xs + (size_t)D * PS;          // [TILED_THREADS + 8] exchange area

    // ---------------- stage the input tile into LDS ------------------------
    // local sample index u = 0 is global sample g0 = (n0-1)*D  (m = -1, p = 0)
    const long long g0 = (n0 - 1) * D;
    const int Lu = (NT + Tq) * D;
    {
        // align pair starts to 16 bytes of the global address
        const long long unit0 = (long long)(((unsigned long long)(uintptr_t)x) >> 3) + g0;
        const int off = (int)(unit0 & 1);
        for (int u = -off + 2 * t; u < Lu; u += 2 * TILED_THREADS) {
            const long long g = g0 + u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.ablate & 1) {
                v = make_float4(1.f, 0.5f, 0.25f, 0.125f);
            } else if (g >= a.n_lo && g + 1 < a.n_in) {
                v = *reinterpret_cast<const float4 *>(x + g);
            } else {
                if (g >= a.n_lo && g < a.n_in) { float2 e = x[g]; v.x = e.x; v.y = e.y; }
                if (g + 1 >= a.n_lo && g + 1 < a.n_in) { float2 e = x[g + 1]; v.z = e.x; v.w = e.y; }
            }
            float2 e0 = make_float2(v.x, v.y), e1 = make_float2(v.z, v.w);
            if (PREMIX) {
                if (u >= 0) e0 = cmul_fma(e0, a.wtab[u]);
                if (u + 1 < Lu) e1 = cmul_fma(e1, a.wtab[u + 1]);
            }
            if (u >= 0) {
                int mm = (u >> LOGD) - 1 + R, p = u & (D - 1);
                xs[p * PS + mm + (mm >> LOGR)] = e0;
            }
            if (u + 1 < Lu) {
                int u1 = u + 1;
                int mm = (u1 >> LOGD) - 1 + R, p = u1 & (D - 1);
                xs[p * PS + mm + (mm >> LOGR)] = e1;
            }
        }
    }
    __syncthreads();

    // ---------------- boundary output y[n0-1] (fused demod only) -----------
    float2 yb = make_float2(0.f, 0.f);
    if (EPI == EPI_ROTATE_DEMOD) {
        if (blockIdx.x == 0) {
            yb = a.y_prev[s];
        } else {
            float2 part = make_float2(0.f, 0.f);
            for (int k = t; k < Tq * D; k += TILED_THREADS) {
                int p = k & (D - 1), q = k >> LOGD;
                int mm = R - 1 + q;
                float2 xv = xs[p * PS + mm + (mm >> LOGR)];
                if (CTAPS) {
                    float2 h = reinterpret_cast<const float2 *>(a.hp)[p * Tq + q];
                    part.x = __builtin_fmaf(h.x, xv.x, part.x);
                    part.x = __builtin_fmaf(-h.y, xv.y, part.x);
                    part.y = __builtin_fmaf(h.x, xv.y, part.y);
                    part.y = __builtin_fmaf(h.y, xv.x, part.y);
                } else {
                    float h = a.hp[p * Tq + q];
                    part.x = __builtin_fmaf(h, xv.x, part.x);
                    part.y = __builtin_fmaf(h, xv.y, part.y);
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                part.x += __shfl_xor(part.x, o);
                part.y += __shfl_xor(part.y, o);
            }
            if ((t & 63) == 0) red[TILED_THREADS + (t >> 6)] = part;
            __syncthreads();
            float2 r0 = red[TILED_THREADS + 0], r1 = red[TILED_THREADS + 1];
            float2 r2 = red[TILED_THREADS + 2], r3 = red[TILED_THREADS + 3];
            yb = make_float2((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y));
            if (PREMIX) yb = cmul_fma(yb, a.vtab[0]);
            yb = cmul_ref(yb, a.gtab[n0 - 1]);
        }
    }

    // ---------------- main loop: R outputs per lane -------------------------
    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);

    const int lane_base = (t + 1) * R + (t + 1);   // slot of mm = (t+1)R
    for (int p = 0; p < ((a.ablate & 2) ? 0 : D); ++p) {
        const float2 *xp = xs + p * PS + lane_base;
        float2 w[R];
#pragma unroll
        for (int j = 0; j < R; ++j) w[j] = xp[j];
        for (int q0 = 0; q0 < Tq; q0 += R) {
            const int nxt = q0 + R + (q0 >> LOGR) + 1;   // slot offset of sample j = q0+R (+qq)
#pragma unroll
            for (int qq = 0; qq < R; ++qq) {
                if (CTAPS) {
                    const float2 h = reinterpret_cast<const float2 *>(a.hp)[p * Tq + q0 + qq];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float2 xv = w[(qq + r) & (R - 1)];
                        acc[r].x = __builtin_fmaf(h.x, xv.x, acc[r].x);
                        acc[r].x = __builtin_fmaf(-h.y, xv.y, acc[r].x);
                        acc[r].y = __builtin_fmaf(h.x, xv.y, acc[r].y);
                        acc[r].y = __builtin_fmaf(h.y, xv.x, acc[r].y);
                    }
                } else {
                    const float h = a.hp[p * Tq + q0 + qq];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float2 xv = w[(qq + r) & (R - 1)];
                        acc[r].x = __builtin_fmaf(h, xv.x, acc[r].x);
                        acc[r].y = __builtin_fmaf(h, xv.y, acc[r].y);
                    }
                }
                w[qq] = xp[nxt + qq];
            }
        }
    }

    // ---------------- epilogue ----------------------------------------------
    const long long nl = n0 + (long long)t * R;     // first output of this lane
    if (PREMIX) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = cmul_fma(acc[r], a.vtab[1 + t * R + r]);
    }
    if (EPI >= EPI_ROTATE) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (nl + r < a.n_out) acc[r] = cmul_ref(acc[r], a.gtab[nl + r]);
    }

    if (EPI != EPI_ROTATE_DEMOD) {
        float2 *__restrict__ y = a.y_out + (long long)s * a.y_stride;
        if (a.vec_store && nl + R <= a.n_out) {
            float4 *dst = reinterpret_cast<float4 *>(y + nl);
#pragma unroll
            for (int r = 0; r < R; r += 2)
                dst[r >> 1] = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (nl + r < a.n_out) y[nl + r] = acc[r];
        }
    } else {
        // previous output for r = 0 comes from the neighbouring lane
        red[t] = acc[R - 1];
        __syncthreads();
        float2 prev = (t == 0) ? yb : red[t - 1];
        float d[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            d[r] = (a.ablate & 4) ? acc[r].x + prev.y : quad_demod_one(acc[r], prev, a.gain, a.atan_tab);
            prev = acc[r];
        }
        float *__restrict__ o = a.d_out + (long long)s * a.d_stride;
        if (a.vec_store && nl + R <= a.n_out) {
            float4 *dst = reinterpret_cast<float4 *>(o + nl);
#pragma unroll
            for (int r = 0; r < R; r += 4) dst[r >> 2] = make_float4(d[r], d[r + 1], d[r + 2], d[r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (nl + r < a.n_out) o[nl + r] = d[r];
        }
        // carry for the next call: the last output of the stream
        const long long last = a.n_out - 1;
        if (last >= nl && last < nl + R) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (nl + r == last) a.y_last[s] = acc[r];
        }
    }
}

template <int D, bool CTAPS, bool PREMIX, int EPI>
static int launch_tiled_inst(const FirTiledArgs &a, int n_streams, hipStream_t st)
{
    size_t lds = tiled_lds_bytes(D, a.Tq);
    auto kern = fir_tiled_kernel<D, CTAPS, PREMIX, EPI>;
    static size_t configured = 0;   // per instantiation
    if (lds > 48 * 1024 && lds > configured) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        configured = lds;
    }
    dim3 grid((unsigned)((a.n_out + TILED_NT - 1) / TILED_NT), (unsigned)n_streams), block(TILED_THREADS);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int D>
static int launch_tiled_d(bool ctaps, bool premix, int epi, const FirTiledArgs &a, int ns, hipStream_t st)
{
    if (ctaps) {
        switch (epi) {
        case EPI_NONE: return launch_tiled_inst<D, true, false, EPI_NONE>(a, ns, st);
        case EPI_ROTATE: return launch_tiled_inst<D, true, false, EPI_ROTATE>(a, ns, st);
        default: return launch_tiled_inst<D, true, false, EPI_ROTATE_DEMOD>(a, ns, st);
        }
    }
    if (premix) {
        switch (epi) {
        case EPI_ROTATE: return launch_tiled_inst<D, false, true, EPI_ROTATE>(a, ns, st);
        case EPI_ROTATE_DEMOD: return launch_tiled_inst<D, false, true, EPI_ROTATE_DEMOD>(a, ns, st);
        default: return fail(GRHIP_EINVAL, "premix needs a rotate epilogue");
        }
    }
    if (epi != EPI_NONE) return fail(GRHIP_EINVAL, "real taps without premix have no rotator");
    return launch_tiled_inst<D, false, false, EPI_NONE>(a, ns, st);
}

static int launch_fir_tiled_impl(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a,
                                 int n_streams, hipStream_t st);

int launch_fir_tiled(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a, int n_streams,
                     hipStream_t st)
{
    if (a.n_out <= 0 || n_streams <= 0) return GRHIP_OK;
    if (!tiled_supported(decim, a.Tq)) return fail(GRHIP_EINVAL, "tiled FIR: unsupported shape");
    static int ablate = -1;
    if (ablate < 0) { const char *e = getenv("GRHIP_ABLATE"); ablate = e ? atoi(e) : 0; }
    if (ablate) { FirTiledArgs b = a; b.ablate = ablate; return launch_fir_tiled_impl(decim, ctaps, premix, epi, b, n_streams, st); }
    return launch_fir_tiled_impl(decim, ctaps, premix, epi, a, n_streams, st);
}

static int launch_fir_tiled_impl(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a,
                                 int n_streams, hipStream_t st)
{
    switch (decim) {
    case 1: return launch_tiled_d<1>(ctaps, premix, epi, a, n_streams, st);
    case 2: return launch_tiled_d<2>(ctaps, premix, epi, a, n_streams, st);
    case 4: return launch_tiled_d<4>(ctaps, premix, epi, a, n_streams, st);
    case 8: return launch_tiled_d<8>(ctaps, premix, epi, a, n_streams, st);
    }
    return fail(GRHIP_EINVAL, "tiled FIR: unsupported decimation %d", decim);
}

// ===========================================================================
// standalone quadrature demod (general/gr_quadrature_demod_cf.cc:46-62)
// HBM-bound: 8 B in + 4 B out per item.  4 outputs per lane.
// ===========================================================================
__global__ void __launch_bounds__(256)
quad_demod_kernel(const float2 *__restrict__ in, float *__restrict__ out, long long n_out, float gain,
                  const float *__restrict__ tab)
{
    long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n_out) return;
    // in[0] is the history item: output i uses in[i+1] and in[i]
    float2 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (i0 + k <= n_out) ? in[i0 + k] : make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n_out) out[i0 + k] = quad_demod_one(v[k + 1], v[k], gain, tab);
}

int launch_quad_demod(const float2 *in, float *out, long long n_out, float gain, const float *atan_tab,
                      hipStream_t st)
{
    if (n_out <= 0) return GRHIP_OK;
    long long lanes = (n_out + 3) / 4;
    dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    hipLaunchKernelGGL(quad_demod_kernel, grid, block, 0, st, in, out, n_out, gain, atan_tab);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

}  // namespace grhip
