// fir_tiled.hip -- the throughput kernel of the FIR family (complex data):
// gr_fir_ccf / gr_fir_ccc decimating FIR, the composite FIR + rotator of
// gr_freq_xlating_fir_filter_ccc and the fused quadrature demodulator.
//
// Shape of the work: y[n] = sum_k c[k] x[nD+k] is a vector x scalar recurrence
// (wave-uniform taps against per-lane data), VALU-bound at 256 taps,
// so the design goal is to keep the vector FMA pipes issuing while HBM traffic,
// LDS staging and the epilogue hide underneath.  No MFMA.
//
//  * Persistent 256-lane workgroups (2 per CU) walk tiles of NT = 256*R
//    consecutive outputs, over all streams of the launch.
//  * The input tile is fetched from HBM with coalesced 16-byte loads into
//    REGISTERS one tile ahead: the loads of tile i+1 are issued before the MAC
//    loop of tile i and land while it runs (branch-free, so no wait at a join).
//  * The tile is then written to LDS de-interleaved into its D polyphase
//    components (x_p[m] = x[mD+p]) with one pad slot per R samples, so that the
//    lane stride is R+1 (odd) 8-byte slots: conflict-free ds_read_b64.
//  * Each lane keeps R complex accumulators and an R-deep sliding window of
//    samples in VGPRs: one LDS read feeds R complex MACs (v_pk_fma_f32 with the
//    tap as the SGPR operand).  Taps and the staging phasor steps are read
//    through the CONSTANT address space so that they stay scalar loads (s_load)
//    even though the persistent loop also stores to global memory; the taps of
//    the next 8 steps are requested one iteration ahead.
//  * Real prototype taps (the usual low-pass) take the pre-mix form of the
//    frequency translation: x'[u] = x[u] * e^{jwu} at staging, real-tap MACs
//    (half the flops of complex taps), per-output phase correction in the
//    epilogue.  W[u] is built from a 513-entry lane table times a wave-uniform
//    step e^{jw 512 i}, so staging reads nothing but the samples.
//  * Epilogues: rotator multiply with the exact-recurrence phase table
//    (8 B / output) and the fused quadrature demodulator.  The demodulator's
//    predecessor of a tile's first output is recomputed by the whole workgroup
//    (tree-order sum) instead of being exchanged between workgroups.
#include <cstdlib>

#include "device_math.h"
#include "fir_kernels.h"
#include "grhip_internal.h"

namespace grhip {

constexpr int TILED_R = 8;
constexpr int TILED_LOGR = 3;
constexpr int TILED_THREADS = 256;
constexpr int TILED_NT = TILED_THREADS * TILED_R;
constexpr int TILED_NI = 18;                 // 16-byte loads per lane per tile (upper bound)
constexpr int TILED_LDS_LIMIT = 80 * 1024;   // two workgroups per CU

int tiled_R() { return TILED_R; }
int tiled_NT() { return TILED_NT; }
int tiled_wtab_len() { return 2 * TILED_THREADS + 1; }
int tiled_stab_len() { return TILED_NI; }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// read-only, wave-uniform operands: constant address space => s_load
typedef const float __attribute__((address_space(4))) *cfloat_p;

// LDS geometry (float2 slots).  mm = m + R where m is the polyphase sample index
// relative to the tile's first real output (m = -1: first sample of the boundary
// output).  slot(mm) = mm + mm/R.
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
    if (!(decim == 1 || decim == 2 || decim == 4)) return false;
    if (Tq <= 0 || (Tq % TILED_R) != 0) return false;
    if ((TILED_NT + Tq) * decim > TILED_NI * 2 * TILED_THREADS) return false;
    return tiled_lds_bytes(decim, Tq) <= (size_t)TILED_LDS_LIMIT;
}

template <int D, bool CTAPS, bool PREMIX, int EPI>
__global__ void __launch_bounds__(TILED_THREADS, 2) fir_tiled_kernel(const FirTiledArgs a)
{
    constexpr int R = TILED_R, LOGR = TILED_LOGR, NT = TILED_NT, NI = TILED_NI;
    constexpr int LOGD = ilog2(D);
    constexpr int TW = CTAPS ? 2 : 1;               // floats per tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *xs = (float2 *)smem;

    const int t = threadIdx.x;
    const int Tq = a.Tq;
    const int PS = tiled_phase_stride(Tq);
    float2 *red = xs + (size_t)D * PS;              // [TILED_THREADS + 8] exchange area
    const int Lu = (NT + Tq) * D;                   // samples per tile, u = 0 <-> (n0-1)*D
    const long long tiles_per_stream = (a.n_out + NT - 1) / NT;
    const long long total_tiles = tiles_per_stream * a.n_streams;
    const cfloat_p hp = (cfloat_p)a.hp;             // phase-major taps, padded by R entries
    const cfloat_p stab = (cfloat_p)a.stab;

    // lane constants of the pre-mix phasor: W[v] = e^{jw(v-D)}, v = 2t-1, 2t, 2t+1
    // (table entry v+1 holds W[v], v = -1 .. 511)
    float2 wA = make_float2(1.f, 0.f), wB = wA, wC = wA;
    if (PREMIX) {
        wA = a.wtab[2 * t];
        wB = a.wtab[2 * t + 1];
        wC = a.wtab[2 * t + 2];
    }

    float4 pf[NI];

    // tile -> stream pointer, first global sample, load parity; returns true when
    // every 16-byte pair of the tile lies inside [n_lo, n_in) (all but the first
    // and last tiles of a stream)
    auto tile_geom = [&](long long tile, const float2 *&x, long long &g0, int &off) {
        const long long s = tile / tiles_per_stream;
        const long long n0 = (tile - s * tiles_per_stream) * NT;
        x = a.x + s * a.x_stride;
        g0 = (n0 - 1) * D;
        const long long unit0 = (long long)(((unsigned long long)(uintptr_t)x) >> 3) + g0;
        off = (int)(unit0 & 1);                     // pair starts on a 16-byte boundary
        return (g0 - 1 >= a.n_lo) && (g0 + Lu + 1 < a.n_in);
    };

    // ---- issue the HBM loads of one tile into registers (branch-free) ----------
    auto fetch = [&](long long tile) {
        const float2 *x; long long g0; int off;
        const bool inside = tile_geom(tile, x, g0, off);
        const float4 *base = reinterpret_cast<const float4 *>(x + (g0 - off + 2 * t));
        if (inside && !(a.ablate & 1)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int u = -off + 2 * t + 2 * TILED_THREADS * i;
                // beyond the tile (only the last i): re-read the lane's first pair
                pf[i] = base[(u < Lu) ? TILED_THREADS * i : 0];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int u = -off + 2 * t + 2 * TILED_THREADS * i;
                const long long g = g0 + u;
                const bool full = (u < Lu) && (g >= a.n_lo) && (g + 1 < a.n_in) && !(a.ablate & 1);
                const float4 *src = full ? base + TILED_THREADS * i : reinterpret_cast<const float4 *>(a.hp);
                pf[i] = *src;
            }
        }
    };

    // ---- registers -> LDS, de-interleaved, pre-mixed ----------------------------
    auto stage = [&](long long tile) {
        const float2 *x; long long g0; int off;
        const bool inside = tile_geom(tile, x, g0, off) && !(a.ablate & 1);
        const float2 w0l = off ? wA : wB, w1l = off ? wB : wC;
        const int ub = -off + 2 * t;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = ub + 2 * TILED_THREADS * i;
            if (u < Lu) {
                float4 v = pf[i];
                if (!inside) {          // rare: patch pairs that straddle the stream's ends
                    const long long g = g0 + u;
                    if (!((g >= a.n_lo) && (g + 1 < a.n_in)) || (a.ablate & 1)) {
                        v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (!(a.ablate & 1)) {
                            if (g >= a.n_lo && g < a.n_in) { float2 e = x[g]; v.x = e.x; v.y = e.y; }
                            if (g + 1 >= a.n_lo && g + 1 < a.n_in) { float2 e = x[g + 1]; v.z = e.x; v.w = e.y; }
                        }
                    }
                }
                float2 e0 = make_float2(v.x, v.y), e1 = make_float2(v.z, v.w);
                if (PREMIX) {
                    const float2 si = make_float2(stab[2 * i], stab[2 * i + 1]);   // uniform e^{jw 512 i}
                    e0 = cmul_fma(e0, cmul_fma(w0l, si));
                    e1 = cmul_fma(e1, cmul_fma(w1l, si));
                }
                if (u >= 0) {
                    const int mm = (u >> LOGD) - 1 + R, p = u & (D - 1);
                    xs[p * PS + mm + (mm >> LOGR)] = e0;
                }
                if (u + 1 < Lu) {
                    const int u1 = u + 1;
                    const int mm = (u1 >> LOGD) - 1 + R, p = u1 & (D - 1);
                    xs[p * PS + mm + (mm >> LOGR)] = e1;
                }
            }
        }
    };

    long long tile = blockIdx.x;
    if (tile < total_tiles) fetch(tile);

    for (; tile < total_tiles; tile += gridDim.x) {
        const long long s = tile / tiles_per_stream;
        const long long bidx = tile - s * tiles_per_stream;
        const long long n0 = bidx * NT;

        stage(tile);
        __syncthreads();
        // next tile's HBM traffic flies under this tile's MAC loop
        if (tile + gridDim.x < total_tiles) fetch(tile + gridDim.x);

        // ---------------- boundary output y[n0-1] (fused demod only) -----------
        float2 yb = make_float2(0.f, 0.f);
        if (EPI == EPI_ROTATE_DEMOD) {
            if (bidx == 0) {
                yb = a.y_prev ? a.y_prev[s] : make_float2(0.f, 0.f);
            } else {
                float2 part = make_float2(0.f, 0.f);
                for (int k = t; k < Tq * D; k += TILED_THREADS) {
                    const int p = k & (D - 1), q = k >> LOGD;
                    const int mm = R - 1 + q;
                    const float2 xv = xs[p * PS + mm + (mm >> LOGR)];
                    if (CTAPS) {
                        const float2 h = reinterpret_cast<const float2 *>(a.hp)[p * Tq + q];
                        part.x = __builtin_fmaf(h.x, xv.x, part.x);
                        part.x = __builtin_fmaf(-h.y, xv.y, part.x);
                        part.y = __builtin_fmaf(h.x, xv.y, part.y);
                        part.y = __builtin_fmaf(h.y, xv.x, part.y);
                    } else {
                        const float h = a.hp[p * Tq + q];
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
                const float2 r0 = red[TILED_THREADS + 0], r1 = red[TILED_THREADS + 1];
                const float2 r2 = red[TILED_THREADS + 2], r3 = red[TILED_THREADS + 3];
                yb = make_float2((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y));
                if (PREMIX) yb = cmul_fma(yb, a.vtab[NT]);
                yb = cmul_ref(yb, a.gtab[n0 - 1]);
            }
        }

        // ---------------- MAC loop: R outputs per lane -----------------------------
        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);

        const int lane_base = (t + 1) * R + (t + 1);       // slot of mm = (t+1)R
        const int nph = (a.ablate & 2) ? 0 : D;
        for (int p = 0; p < nph; ++p) {
            const float2 *xp = xs + p * PS + lane_base;
            const cfloat_p tp = hp + (size_t)p * Tq * TW;
            float2 w[R];
#pragma unroll
            for (int j = 0; j < R; ++j) w[j] = xp[j];
            float hn[R * TW];
#pragma unroll
            for (int k = 0; k < R * TW; ++k) hn[k] = tp[k];
            for (int q0 = 0; q0 < Tq; q0 += R) {
                float h[R * TW];
#pragma unroll
                for (int k = 0; k < R * TW; ++k) h[k] = hn[k];
                // request the taps of the next 8 steps now (the table is padded by R
                // taps, so the read past the last step of the last phase stays in bounds)
#pragma unroll
                for (int k = 0; k < R * TW; ++k) hn[k] = tp[(q0 + R) * TW + k];
                const int nxt = q0 + R + (q0 >> LOGR) + 1;   // slot offset of sample j = q0+R (+qq)
#pragma unroll
                for (int qq = 0; qq < R; ++qq) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float2 xv = w[(qq + r) & (R - 1)];
                        if (CTAPS) {
                            const float hr = h[2 * qq], hi = h[2 * qq + 1];
                            acc[r].x = __builtin_fmaf(hr, xv.x, acc[r].x);
                            acc[r].x = __builtin_fmaf(-hi, xv.y, acc[r].x);
                            acc[r].y = __builtin_fmaf(hr, xv.y, acc[r].y);
                            acc[r].y = __builtin_fmaf(hi, xv.x, acc[r].y);
                        } else {
                            acc[r].x = __builtin_fmaf(h[qq], xv.x, acc[r].x);
                            acc[r].y = __builtin_fmaf(h[qq], xv.y, acc[r].y);
                        }
                    }
                    w[qq] = xp[nxt + qq];
                }
            }
        }

        // ---------------- epilogue ---------------------------------------------------
        const long long nl = n0 + (long long)t * R;          // first output of this lane
        if (PREMIX) {
            const float4 *vv = reinterpret_cast<const float4 *>(a.vtab + t * R);   // 64-byte aligned
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const float4 v2 = vv[r >> 1];
                acc[r] = cmul_fma(acc[r], make_float2(v2.x, v2.y));
                acc[r + 1] = cmul_fma(acc[r + 1], make_float2(v2.z, v2.w));
            }
        }
        if (EPI >= EPI_ROTATE) {
            if (nl + R <= a.n_out) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = cmul_ref(acc[r], a.gtab[nl + r]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (nl + r < a.n_out) acc[r] = cmul_ref(acc[r], a.gtab[nl + r]);
            }
        }

        if (EPI != EPI_ROTATE_DEMOD) {
            float2 *__restrict__ y = a.y_out + s * a.y_stride;
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
            float *__restrict__ o = a.d_out + s * a.d_stride;
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
            if (a.y_last && last >= nl && last < nl + R) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (nl + r == last) a.y_last[s] = acc[r];
            }
        }
        __syncthreads();        // xs / red are rewritten by the next tile
    }
}

static int g_num_cus = 0;

template <int D, bool CTAPS, bool PREMIX, int EPI>
static int launch_tiled_inst(const FirTiledArgs &a, hipStream_t st)
{
    size_t lds = tiled_lds_bytes(D, a.Tq);
    auto kern = fir_tiled_kernel<D, CTAPS, PREMIX, EPI>;
    static size_t configured = 0;   // per instantiation
    if (lds > 48 * 1024 && lds > configured) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        configured = lds;
    }
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        GRHIP_HIP(hipGetDevice(&dev));
        GRHIP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_num_cus = n > 0 ? n : 256;
    }
    const long long tiles = ((a.n_out + TILED_NT - 1) / TILED_NT) * a.n_streams;
    long long grid = 2ll * g_num_cus;                 // persistent: two workgroups per CU
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TILED_THREADS), lds, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int D>
static int launch_tiled_d(bool ctaps, bool premix, int epi, const FirTiledArgs &a, hipStream_t st)
{
    if (ctaps) {
        switch (epi) {
        case EPI_NONE: return launch_tiled_inst<D, true, false, EPI_NONE>(a, st);
        case EPI_ROTATE: return launch_tiled_inst<D, true, false, EPI_ROTATE>(a, st);
        default: return launch_tiled_inst<D, true, false, EPI_ROTATE_DEMOD>(a, st);
        }
    }
    if (premix) {
        switch (epi) {
        case EPI_ROTATE: return launch_tiled_inst<D, false, true, EPI_ROTATE>(a, st);
        case EPI_ROTATE_DEMOD: return launch_tiled_inst<D, false, true, EPI_ROTATE_DEMOD>(a, st);
        default: return fail(GRHIP_EINVAL, "premix needs a rotate epilogue");
        }
    }
    if (epi != EPI_NONE) return fail(GRHIP_EINVAL, "real taps without premix have no rotator");
    return launch_tiled_inst<D, false, false, EPI_NONE>(a, st);
}

int launch_fir_tiled(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a_in, int n_streams,
                     hipStream_t st)
{
    if (a_in.n_out <= 0 || n_streams <= 0) return GRHIP_OK;
    if (!tiled_supported(decim, a_in.Tq)) return fail(GRHIP_EINVAL, "tiled FIR: unsupported shape");
    FirTiledArgs a = a_in;
    a.n_streams = n_streams;
    static int ablate = -1;
    if (ablate < 0) { const char *e = getenv("GRHIP_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    switch (decim) {
    case 1: return launch_tiled_d<1>(ctaps, premix, epi, a, st);
    case 2: return launch_tiled_d<2>(ctaps, premix, epi, a, st);
    case 4: return launch_tiled_d<4>(ctaps, premix, epi, a, st);
    }
    return fail(GRHIP_EINVAL, "tiled FIR: unsupported decimation %d", decim);
}

}  // namespace grhip
