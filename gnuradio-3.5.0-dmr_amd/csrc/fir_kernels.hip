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

#include <vector>

#include <cstdlib>

#include "device_math.h"
#include "grhip_internal.h"

namespace grhip {

// ===========================================================================
// (A) generic-order kernel
// ===========================================================================
// SEQ: ONE accumulator, terms added one after the other -- the order of
// gri_fir_filter_with_buffer_XXX::filter (filter/gri_fir_filter_with_buffer_XXX.cc.t:75-79)
template <int KIND, bool SEQ = false>
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

    if (SEQ) {
        if (KIND == FIR_FFF) {
            const float *x = in + n * decim;
            float acc = 0;
            for (int i = 0; i < ntaps; i++) acc += x[i] * s_taps[i];
            out[n] = acc;
        } else {
            const float2 *x = (const float2 *)in + n * decim;
            float ar = 0, ai = 0;
            if (KIND == FIR_CCF) {
                for (int i = 0; i < ntaps; i++) {
                    const float2 v = x[i];
                    const float t = s_taps[i];
                    const float pr = v.x * t, pi = v.y * t;
                    ar += pr; ai += pi;
                }
            } else {
                const float2 *tc = (const float2 *)s_taps;
                for (int i = 0; i < ntaps; i++) {
                    const float2 p = cmul_ref(x[i], tc[i]);
                    ar += p.x; ai += p.y;
                }
            }
            ((float2 *)out)[n] = make_float2(ar, ai);
        }
        return;
    }
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

// ---------------------------------------------------------------------------
// (A') the same arithmetic as a tiled kernel (round 3): GRHIP_MODE_GENERIC is the only mode whose bit decisions are
// the reference's, so it gets a throughput.  gr_fir_ccf_generic / gr_fir_ccc_generic (filter/gr_fir_XXX_generic.cc.t:59-79):
// two complex accumulators, even taps into the first, odd taps into the second, every product and every sum a single
// IEEE operation, the two added at the end -- kept term by term; what changes is where the operands come from.
//   * a 256-lane workgroup takes a tile of 1024 outputs (four per lane, 256 apart: neighbouring lanes are neighbouring
//     outputs); the tile's (1023 D + ntaps) samples are staged ONCE in LDS, de-interleaved into the D polyphase rows
//     (sample u -> row u mod D, slot u / D), so that the lanes' reads of x[n D + i] are consecutive 8-byte slots
//     (conflict-free) whatever the decimation; fir_generic_kernel's lanes read global memory at a stride of 8 D bytes,
//     once per tap and output (L1-bound: 31 Gsamples/s);
//   * per pair of taps: one 16-byte broadcast read of the two taps, eight sample reads, and the reference's operations as
//     packed instructions on (re, im): v_pk_mul_f32 x 2, v_pk_add_f32 x 2 per complex-tap term (v_pk_mul + v_pk_add with
//     real taps) -- unfused, so 256 complex taps at D = 4 cost 256 packed instructions per input sample: the ceiling is
//     the vector pipes' non-FMA rate, about 130 Gsamples/s of input.
// Bit-exactness is what tests/test_gpu_fir.py's generic-order cases check, on every shape they hold.
// ---------------------------------------------------------------------------
constexpr int GT_R = 4, GT_T = 256, GT_NT = GT_R * GT_T;

template <int KIND>
__global__ void __launch_bounds__(GT_T)
fir_generic_tiled_kernel(const float *__restrict__ taps_rev, int ntaps, const float2 *__restrict__ in, long long n_in,
                         float2 *__restrict__ out, long long n_out, int decim, const float2 *__restrict__ gtab)
{
    static_assert(KIND == FIR_CCF || KIND == FIR_CCC, "complex data");
    typedef float gf2 __attribute__((ext_vector_type(2)));
    typedef float gf4 __attribute__((ext_vector_type(4)));
    typedef unsigned int gu4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int D = decim, t = threadIdx.x;
    const int per_row = GT_NT + (ntaps + D - 1) / D + 2;            // slots per polyphase row
    gf2 *xs = reinterpret_cast<gf2 *>(gsm);                           // [D][per_row]
    float *tp = reinterpret_cast<float *>(gsm + (((size_t)D * per_row * 8 + 15) & ~(size_t)15));   // taps (16-byte aligned: read in pairs)
    const int tw = KIND == FIR_CCC ? 2 : 1;
    const int ntp = ntaps & ~1;                                       // the pairs; an odd last tap goes to the first accumulator
    for (int i = t; i < ntaps * tw; i += GT_T) tp[i] = taps_rev[i];
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in), 0,
                                                                        (int)(n_in * 8 > 0x7ffffff0ll ? 0x7ffffff0ll : n_in * 8), 0x00020000);
    const long long ntiles = (n_out + GT_NT - 1) / GT_NT;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long n0 = tile * GT_NT;
        const long long u0 = n0 * D;                                  // first sample of the tile
        const int span = (GT_NT - 1) * D + ntaps;                     // samples the tile touches
        __syncthreads();                                              // the previous tile's reads are done (and the taps are in)
        // ---- stage: 16-byte loads (two samples per lane), out-of-range lanes read zeros
        for (int m = 2 * t; m < span; m += 2 * GT_T) {
            const long long u = u0 + m;
            const gu4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, u * 8 < 0x7ffffff0ll ? (int)(u * 8) : 0x7ffffff0, 0, 0);
            const gf4 f = __builtin_bit_cast(gf4, v);
            const int r0 = m % D, c0 = m / D;
            xs[(size_t)r0 * per_row + c0] = gf2{f[0], f[1]};
            const int m1 = m + 1, r1 = m1 % D, c1 = m1 / D;
            if (m1 < span) xs[(size_t)r1 * per_row + c1] = gf2{f[2], f[3]};
        }
        __syncthreads();
        gf2 a0[GT_R], a1[GT_R];
#pragma unroll
        for (int r = 0; r < GT_R; ++r) { a0[r] = gf2{0.f, 0.f}; a1[r] = gf2{0.f, 0.f}; }
        int ph = 0, co = 0;                                           // i mod D, i / D
        for (int i = 0; i < ntp; i += 2) {
            const gf2 *x0 = xs + (size_t)ph * per_row + co + t;
            int ph1 = ph + 1, co1 = co;
            if (ph1 == D) { ph1 = 0; ++co1; }
            const gf2 *x1 = xs + (size_t)ph1 * per_row + co1 + t;
            if (KIND == FIR_CCC) {
                const gf4 tt = *reinterpret_cast<const gf4 *>(tp + 2 * i);     // taps i, i + 1 (wave-uniform address)
                const gf2 t0{tt[0], tt[1]}, t1{tt[2], tt[3]};
#pragma unroll
                for (int r = 0; r < GT_R; ++r) {
                    const gf2 v0 = x0[r * GT_T], v1 = x1[r * GT_T];
                    gf2 m1, m2;
                    // d_taps[i] * input[i] as __mulsc3 does it for finite operands: (tr xr - ti xi, tr xi + ti xr)
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(m1) : "v"(t0), "v"(v0));
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m2) : "v"(t0), "v"(v0));
                    a0[r] = a0[r] + (m1 + m2);
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(m1) : "v"(t1), "v"(v1));
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m2) : "v"(t1), "v"(v1));
                    a1[r] = a1[r] + (m1 + m2);
                }
            } else {
                const gf2 tt = *reinterpret_cast<const gf2 *>(tp + i);
#pragma unroll
                for (int r = 0; r < GT_R; ++r) {
                    const gf2 v0 = x0[r * GT_T], v1 = x1[r * GT_T];
                    a0[r] = a0[r] + v0 * gf2{tt[0], tt[0]};
                    a1[r] = a1[r] + v1 * gf2{tt[1], tt[1]};
                }
            }
            ph = ph1 + 1; co = co1;
            if (ph == D) { ph = 0; ++co; }
        }
        if (ntaps & 1) {                                              // .cc.t:72-73: for (; i < ntaps; i++) acc0 += ...
            const gf2 *x0 = xs + (size_t)ph * per_row + co + t;
#pragma unroll
            for (int r = 0; r < GT_R; ++r) {
                const gf2 v0 = x0[r * GT_T];
                if (KIND == FIR_CCC) {
                    const gf2 t0{tp[2 * ntp], tp[2 * ntp + 1]};
                    gf2 m1, m2;
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(m1) : "v"(t0), "v"(v0));
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m2) : "v"(t0), "v"(v0));
                    a0[r] = a0[r] + (m1 + m2);
                } else {
                    a0[r] = a0[r] + v0 * gf2{tp[ntp], tp[ntp]};
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GT_R; ++r) {
            const long long n = n0 + r * GT_T + t;
            if (n < n_out) {
                const gf2 y = a0[r] + a1[r];
                float2 o = make_float2(y[0], y[1]);
                if (gtab) o = cmul_ref(o, gtab[n]);                   // gr_rotator::rotate: z = in * d_phase
                out[n] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// (A'') the tiled form with the samples in registers (decimation 1 / 2 / 4).  In (A') every term reads its sample from LDS:
// one 8-byte read per four packed instructions, from four SIMDs, is all the LDS pipe of a CU delivers (128 bytes per
// cycle) -- the vector pipes wait for it half the time.  Here a lane's four outputs are NEIGHBOURS (n, n+1, n+2, n+3), and
// x[(n + 1) D + i] = x[n D + i + D]: the sample output n+1 needs for tap i is the one output n needs for tap i + D.  So the
// lane keeps, per polyphase row, a window of four samples in registers; a tap uses the four of its row (one per output) and
// then replaces the oldest by ONE new read: a quarter of the LDS traffic, and that read is consumed D taps (16 D packed
// instructions) later.  The window's registers rotate with period 4 in i / D, so the tap loop is unrolled by 4 D taps
// and every register name is static; taps beyond ntaps are skipped by wave-uniform tests (a padding tap would add
// +0 * x terms: wrong for -0 sums and non-finite samples).  The taps themselves are wave-uniform: scalar loads, handed to
// the packed multiplies as their one scalar operand.  LDS rows are split by slot mod 4 (lane t's slots 4t + j, j fixed,
// are consecutive 8-byte words: conflict-free reads).  The arithmetic is (A')'s term by term: even taps into the first
// accumulator, odd taps into the second, in increasing i, every operation unfused.
// ---------------------------------------------------------------------------
// DEMOD: gr_quadrature_demod_cf (quad_demod_one: the reference's operations) in the epilogue instead of the store of y --
// d[n] needs y[n - 1]: a lane has it for three of its four outputs, gets the fourth from its neighbour through LDS, and a
// tile starts four outputs early (its lane 0 repeats the previous tile's last four and stores nothing), so no tile waits
// for another; d[0] takes the block's carried sample (*y_prev), the lane that holds y[n_out - 1] leaves it in *y_last.
#ifndef GRHIP_GW_T
#define GRHIP_GW_T 256                // lanes of a workgroup of the window kernel (A/B on the batch: 256 / 128 / 64 lanes 108.2 / 106.9 / 103.8 Gsamples/s)
#endif
constexpr int GW_T = GRHIP_GW_T, GW_NT = GT_R * GW_T;
struct GenericDemodArgs { float *d; float gain; const float *atan_tab; const float2 *y_prev; float2 *y_last; };
// Several streams in one launch (the multi-capture entries; DEMOD only): stream s reads in + s * x_stride, its first n_lo
// items are the zeros a fresh flowgraph's history holds (never read: the pointer may lie before the capture), its outputs
// go to d + s * out_stride, its carried samples are y_prev[s] / y_last[s].  a_shift (0 / 1): the streams start a_shift items
// behind a 16-byte boundary -- the staging loads stay aligned, item u of a stream is item u + a_shift of the aligned row.
struct GenericBatch { int n_streams; long long x_stride, out_stride; int n_lo, a_shift; };

// Where a block's taps come from, A/B on one box (cfg2, one 10 M-sample capture, profiles/r03_generic_ab.log): scalar loads at
// the block's start 91.4 Gsamples/s; requested a block ahead 88.5 (the wait counter SMEM shares with LDS cannot skip a load
// in flight, so the block's first sample wait becomes a wait for everything, and 32 more scalar registers are moved per
// block); from LDS at a wave-uniform address 84.4 (one more LDS read per tap).  The scalar pair as the multiplies' operand
// costs nothing: copied to a vector register pair first, the batch of 64 captures runs at 98.6 against 109.5
// (profiles/r03_generic_batch_ab.log).
#ifndef GRHIP_GW_TAPS_LDS
#define GRHIP_GW_TAPS_LDS 0
#endif
#ifndef GRHIP_GW_TAP_PREFETCH
#define GRHIP_GW_TAP_PREFETCH 0
#endif
#ifndef GRHIP_GW_TAPS_VGPR
#define GRHIP_GW_TAPS_VGPR 0          // (A/B) the scalar-loaded tap copied to a vector register pair in front of its sixteen instructions
#endif
template <int KIND, int D, bool DEMOD>
__global__ void __launch_bounds__(GW_T)
fir_generic_win_kernel(const float *__restrict__ taps_rev, int ntaps, const float2 *__restrict__ in, long long n_in,
                       float2 *__restrict__ out, long long n_out, const float2 *__restrict__ gtab, const GenericDemodArgs dm,
                       const GenericBatch gb)
{
    static_assert(KIND == FIR_CCF || KIND == FIR_CCC, "complex data");
    constexpr int R = GT_R, UB = R * D;
    typedef float gf2 __attribute__((ext_vector_type(2)));
    typedef float gf4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int t = threadIdx.x;
    const int kmax = (ntaps + D - 1) / D;                             // taps per polyphase row, at most
    const int sub_len = GW_T + (kmax + R - 1) / R + 2;                // words per (row, slot mod 4)
    const int per_row = R * sub_len;
    gf2 *xs = reinterpret_cast<gf2 *>(gsm);                           // [D][R][sub_len]
    float *tp = reinterpret_cast<float *>(gsm + (size_t)D * per_row * 8 + GW_T * 8);     // (GRHIP_GW_TAPS_LDS) behind rows and s_last
    if (GRHIP_GW_TAPS_LDS)
        for (int i = t; i < ntaps * (KIND == FIR_CCC ? 2 : 1); i += GW_T) tp[i] = taps_rev[i];
    constexpr int NS = DEMOD ? GW_NT - R : GW_NT;                     // new outputs per tile
    const long long ntiles = (n_out + NS - 1) / NS;
    const int ash = gb.a_shift;
    const long long lim = n_in + ash;                                 // items of the aligned row that exist
    for (long long blk = blockIdx.x; blk < ntiles * gb.n_streams; blk += gridDim.x) {
        const long long strm = blk / ntiles, tile = blk - strm * ntiles;
        const float2 *row = in + strm * gb.x_stride - ash;            // 16-byte aligned
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(row), 0,
                                                                            (int)(lim * 8 > 0x7ffffff0ll ? 0x7ffffff0ll : lim * 8), 0x00020000);
        const long long n0 = tile * NS - (DEMOD ? R : 0);             // (DEMOD, first tile: outputs -4 .. -1 read zeros, unused)
        const long long u0 = n0 * D;                                  // first sample of the tile
        const int span = (GW_NT - 1) * D + ntaps;                     // samples the tile touches
        __syncthreads();                                              // the previous tile's reads are done
        // aligned pairs of items of the row: e = u + a_shift, even.  Items before the row or the stream's n_lo are zeros,
        // items behind its end too; a pair that straddles the end is read as one item
        const long long e0 = (u0 + ash) & ~1ll;
        for (int mm = 2 * t; mm < span + 2; mm += 2 * GW_T) {
            const long long e = e0 + mm;
            gf4 f{0.f, 0.f, 0.f, 0.f};
            // (only items that exist are touched: the row's first n_lo + a_shift items may lie outside the allocation)
            const bool i0 = e >= gb.n_lo + ash && e < lim, i1 = e + 1 >= gb.n_lo + ash && e + 1 < lim;
            if (e * 8 < 0x7ffffff0ll - 16) {
                if (i0 && i1) {
                    f = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(e * 8), 0, 0));
                } else if (i0) {
                    const gf2 g = __builtin_bit_cast(gf2, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)(e * 8), 0, 0));
                    f[0] = g[0]; f[1] = g[1];
                } else if (i1) {
                    const gf2 g = __builtin_bit_cast(gf2, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)(e * 8 + 8), 0, 0));
                    f[2] = g[0]; f[3] = g[1];
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const long long u = e + q - ash;                      // the stream's item
                const int m = (int)(u - u0);                          // its place in the tile
                if (m >= 0 && m < span) {
                    xs[(m % D) * per_row + ((m / D) % R) * sub_len + (m / D) / R] = gf2{f[2 * q], f[2 * q + 1]};
                }
            }
        }
        __syncthreads();
        gf2 acc[2][R], w[D][R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[0][r] = gf2{0.f, 0.f}; acc[1][r] = gf2{0.f, 0.f}; }
        const gf2 *base = xs + t;
#pragma unroll
        for (int ph = 0; ph < D; ++ph)
#pragma unroll
            for (int j = 0; j < R; ++j) w[ph][j] = base[ph * per_row + j * sub_len];       // slot 4 t + j of row ph
        // one tap: its four terms, then the row's next sample (slot 4 t + k + 4, k = ib / D + kk) over the one no later tap
        // of this row uses
        auto tap_step = [&](int ib, int j, gf2 ctap, float rtap) __attribute__((always_inline)) {
            const int ph = j % D, kk = j / D;
            if (KIND == FIR_CCC) {
                // d_taps[i] * input[i] as __mulsc3 does it for finite operands: (tr xr - ti xi, tr xi + ti xr), then acc += it.
                // The four outputs' sixteen instructions in ONE block, producers four instructions ahead of their consumers
                // (left to the scheduler, each sum followed its two products directly: the wave waited out the packed
                // pipeline's latency sixteen times per tap -- SQ_WAIT_INST_ANY 43 % of the wave cycles)
                static_assert(R == 4, "four outputs per lane");
                const gf2 v0 = w[ph][kk % R], v1 = w[ph][(1 + kk) % R], v2 = w[ph][(2 + kk) % R], v3 = w[ph][(3 + kk) % R];
                gf2 p0, p1, p2, p3, q0, q1, q2, q3;
                asm("v_pk_mul_f32 %0, %12, %13 op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %4, %12, %13 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
                    "v_pk_mul_f32 %1, %12, %14 op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %5, %12, %14 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
                    "v_pk_mul_f32 %2, %12, %15 op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %6, %12, %15 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
                    "v_pk_mul_f32 %3, %12, %16 op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %7, %12, %16 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
                    "v_pk_add_f32 %0, %0, %4\n\t"
                    "v_pk_add_f32 %1, %1, %5\n\t"
                    "v_pk_add_f32 %2, %2, %6\n\t"
                    "v_pk_add_f32 %3, %3, %7\n\t"
                    "v_pk_add_f32 %8, %8, %0\n\t"
                    "v_pk_add_f32 %9, %9, %1\n\t"
                    "v_pk_add_f32 %10, %10, %2\n\t"
                    "v_pk_add_f32 %11, %11, %3"
                    : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3),
                      "+v"(acc[j & 1][0]), "+v"(acc[j & 1][1]), "+v"(acc[j & 1][2]), "+v"(acc[j & 1][3])
#if GRHIP_GW_TAPS_LDS || GRHIP_GW_TAPS_VGPR
                    : "v"(ctap), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#else
                    : "s"(ctap), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#endif
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[j & 1][r] = acc[j & 1][r] + w[ph][(r + kk) % R] * gf2{rtap, rtap};
            }
            w[ph][kk] = base[ph * per_row + kk * sub_len + ib / UB + 1];
        };
        int ib = 0;
        // whole blocks: a block's taps in one scalar load, requested a block ahead (at the block's start the wave would sit
        // out the scalar cache's latency once per block)
        gf2 ct[UB], cn[UB];
        float rt[UB], rn[UB];
        const float *tsrc = GRHIP_GW_TAPS_LDS ? tp : taps_rev;
        auto load_taps = [&](int at, gf2 (&c)[UB], float (&r)[UB]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                if (KIND == FIR_CCC) c[j] = *reinterpret_cast<const gf2 *>(tsrc + 2 * (at + j));
                else r[j] = tsrc[at + j];
            }
        };
        if (GRHIP_GW_TAP_PREFETCH && !GRHIP_GW_TAPS_LDS) {
            if (UB <= ntaps) load_taps(0, ct, rt);
            for (; ib + UB <= ntaps; ib += UB) {
                if (ib + 2 * UB <= ntaps) load_taps(ib + UB, cn, rn);
#pragma unroll
                for (int j = 0; j < UB; ++j) tap_step(ib, j, KIND == FIR_CCC ? ct[j] : gf2{0.f, 0.f}, KIND == FIR_CCC ? 0.f : rt[j]);
#pragma unroll
                for (int j = 0; j < UB; ++j) { ct[j] = cn[j]; rt[j] = rn[j]; }
            }
        } else {
            for (; ib + UB <= ntaps; ib += UB) {
                load_taps(ib, ct, rt);
#pragma unroll
                for (int j = 0; j < UB; ++j) tap_step(ib, j, KIND == FIR_CCC ? ct[j] : gf2{0.f, 0.f}, KIND == FIR_CCC ? 0.f : rt[j]);
            }
        }
        if (ib < ntaps) {                                             // the last, partial block (wave-uniform tests)
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                if (ib + j < ntaps) {
                    if (KIND == FIR_CCC) tap_step(ib, j, *reinterpret_cast<const gf2 *>(tsrc + 2 * (ib + j)), 0.f);
                    else tap_step(ib, j, gf2{0.f, 0.f}, tsrc[ib + j]);
                }
            }
        }
        const long long nb = n0 + (long long)R * t;
        if (!DEMOD) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long n = nb + r;
                if (n < n_out) {
                    const gf2 y = acc[0][r] + acc[1][r];
                    float2 o = make_float2(y[0], y[1]);
                    if (gtab) o = cmul_ref(o, gtab[n]);               // gr_rotator::rotate: z = in * d_phase
                    out[n] = o;
                }
            }
        } else {
            float2 y[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long n = nb + r;
                const gf2 a = acc[0][r] + acc[1][r];
                y[r] = make_float2(a[0], a[1]);
                if (gtab && n >= 0 && n < n_out) y[r] = cmul_ref(y[r], gtab[n]);
            }
            float2 *s_last = reinterpret_cast<float2 *>(gsm + (size_t)D * per_row * 8);        // [GW_T], behind the sample rows
            s_last[t] = y[R - 1];
            __syncthreads();
            if (t > 0) {
                float2 prev = s_last[t - 1];
                float *dd = dm.d + strm * gb.out_stride;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const long long n = nb + r;
                    if (n == 0) prev = dm.y_prev ? dm.y_prev[strm] : make_float2(0.f, 0.f);
                    if (n >= 0 && n < n_out) dd[n] = quad_demod_one(y[r], prev, dm.gain, dm.atan_tab);
                    if (n == n_out - 1 && dm.y_last) dm.y_last[strm] = y[r];
                    prev = y[r];
                }
            }
        }
    }
}

static int g_gt_cus = 0;
static const bool g_generic_no_window = getenv("GRHIP_GENERIC_NO_WINDOW") != nullptr;      // (A/B: the (A') kernel on every shape)
template <int KIND>
static int launch_generic_tiled(const float *taps_rev, int ntaps, const void *in, void *out, long long n_out, int decim,
                                const float2 *gtab, hipStream_t st)
{
    const int per_row = GT_NT + (ntaps + decim - 1) / decim + 2;
    const size_t lds = (((size_t)decim * per_row * 8 + 15) & ~(size_t)15) + (size_t)(ntaps + 1) * (KIND == FIR_CCC ? 8 : 4);
    static size_t cfg = 0;
    if (lds > 64 * 1024 && lds > cfg) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_tiled_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cfg = lds;
    }
    if (g_gt_cus == 0) {
        int dev = 0, n = 0;
        GRHIP_HIP(hipGetDevice(&dev));
        GRHIP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_gt_cus = n > 0 ? n : 256;
    }
    const long long ntiles = (n_out + GT_NT - 1) / GT_NT;
    long long per_cu = (long long)(160 * 1024) / (long long)(lds + 512);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    long long grid = per_cu * g_gt_cus;
    if (grid > ntiles) grid = ntiles;
    const long long n_in = (n_out - 1) * decim + ntaps;               // what the caller guarantees readable
    hipLaunchKernelGGL(fir_generic_tiled_kernel<KIND>, dim3((unsigned)grid), dim3(GT_T), lds, st, taps_rev, ntaps,
                       (const float2 *)in, n_in, (float2 *)out, n_out, decim, gtab);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int KIND, int D, bool DEMOD = false>
static int launch_generic_win(const float *taps_rev, int ntaps, const void *in, void *out, long long n_out, const float2 *gtab,
                              hipStream_t st, const GenericDemodArgs dm = GenericDemodArgs{},
                              const GenericBatch gb = GenericBatch{1, 0, 0, 0, 0}, long long n_in_arg = -1)
{
    const int kmax = (ntaps + D - 1) / D;
    const size_t lds = (size_t)D * GT_R * (GW_T + (kmax + GT_R - 1) / GT_R + 2) * 8 + GW_T * 8 + (GRHIP_GW_TAPS_LDS ? (size_t)ntaps * 8 : 0);
    static size_t cfg = 0;
    if (lds > 64 * 1024 && lds > cfg) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_win_kernel<KIND, D, DEMOD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cfg = lds;
    }
    // one workgroup per tile (nothing is set up per workgroup: the taps are scalar loads): the dispatcher hands a CU its
    // next tile when one is done, so a stream of a few tiles per CU -- 10 M samples are 9.5 -- does not wait for the
    // workgroups that drew one tile more (a persistent grid of 4 per CU: 80 against 84.5 Gsamples/s on one 10 M-sample capture)
    const long long ns = DEMOD ? GW_NT - GT_R : GW_NT;
    const long long ntiles = (n_out + ns - 1) / ns * gb.n_streams;
    long long grid = ntiles < (1ll << 20) ? ntiles : (1ll << 20);
    const long long n_in = n_in_arg >= 0 ? n_in_arg : (n_out - 1) * D + ntaps;     // what the caller guarantees readable
    hipLaunchKernelGGL((fir_generic_win_kernel<KIND, D, DEMOD>), dim3((unsigned)grid), dim3(GW_T), lds, st, taps_rev, ntaps,
                       (const float2 *)in, n_in, (float2 *)out, n_out, gtab, dm, gb);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int KIND, bool SEQ>
static int launch_generic_inst(const float *taps_rev, int ntaps, const void *in, void *out, long long n_out, int decim,
                               size_t sh, hipStream_t st)
{
    dim3 grid((unsigned)((n_out + 255) / 256)), block(256);
    if (sh > 64 * 1024)
        GRHIP_HIP(hipFuncSetAttribute((const void *)fir_generic_kernel<KIND, SEQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL((fir_generic_kernel<KIND, SEQ>), grid, block, sh, st, taps_rev, ntaps, (const float *)in, (float *)out, n_out,
                       decim, (const float2 *)nullptr);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// the tiled kernels' shapes (complex data, a tile's worth of outputs, a decimation and a tap count whose tile fits LDS;
// 16-byte loads: the stream on an 8-byte boundary is served by the range check only if it starts on a 16-byte one)
static bool generic_tiled_ok(FirKind kind, int ntaps, const void *in, long long n_out, int decim)
{
    return kind != FIR_FFF && ntaps >= 8 && n_out >= 2 * GT_NT && decim >= 1 && decim <= 16 &&
           (size_t)decim * (GT_NT + ntaps / decim + 3) * 8 + (size_t)ntaps * 8 + 64 <= 150 * 1024 && (((uintptr_t)in) & 15) == 0 &&
           ((n_out - 1) * decim + ntaps) * 8 < 0x7ffffff0ll;
}

// gr_fir_ccc_generic + rotator + gr_quadrature_demod_cf in one kernel (bit-exact); returns 1 where the shape has no such
// kernel (the caller then runs the FIR and the demodulator as two)
// the shapes launch_fir_generic_demod takes as a batch (several streams, or history zeros to synthesise)
bool generic_demod_batch_ok(int ntaps, int decim, const void *in, long long x_stride)
{
    const void *row0 = (const float2 *)in - ((((uintptr_t)in) & 15) ? 1 : 0);
    return !g_generic_no_window && (decim == 1 || decim == 2 || decim == 4) && !(x_stride & 1) &&
           generic_tiled_ok(FIR_CCC, ntaps, row0, 2 * GT_NT, decim);
}

int launch_fir_generic_demod(const float *taps_rev, int ntaps, const void *in, float *d, long long n_out, int decim,
                             const float2 *gtab, float gain, const float *atan_tab, const float2 *y_prev, float2 *y_last,
                             hipStream_t st, int n_streams, long long x_stride, long long out_stride, long long n_lo)
{
    if (n_out <= 0 || n_streams <= 0) return GRHIP_OK;
    // the aligned row every stream's items sit in: one item further back where the stream starts on an 8-byte boundary
    const int a_shift = (((uintptr_t)in) & 15) ? 1 : 0;
    const void *row0 = (const float2 *)in - a_shift;
    // (one stream with its history in front and less than two tiles of outputs: the two-kernel path; a batch of any length)
    const bool batch = n_streams > 1 || n_lo > 0;
    if (g_generic_no_window || !(decim == 1 || decim == 2 || decim == 4) ||
        !generic_tiled_ok(FIR_CCC, ntaps, row0, batch && n_out < 2 * GT_NT ? 2 * GT_NT : n_out, decim) ||
        ((n_out - 1) * decim + ntaps) * 8 >= 0x7ffffff0ll ||
        (y_prev && (const void *)y_prev == (const void *)y_last) || (n_streams > 1 && (x_stride & 1)) || n_lo < 0 || n_lo > 0x7fffffff)
        return 1;
    const GenericDemodArgs dm{d, gain, atan_tab, y_prev, y_last};
    const GenericBatch gb{n_streams, n_streams > 1 ? x_stride : 0, n_streams > 1 ? out_stride : 0, (int)n_lo, a_shift};
    const long long n_in = (n_out - 1) * decim + ntaps;
    if (decim == 4) return launch_generic_win<FIR_CCC, 4, true>(taps_rev, ntaps, in, nullptr, n_out, gtab, st, dm, gb, n_in);
    if (decim == 2) return launch_generic_win<FIR_CCC, 2, true>(taps_rev, ntaps, in, nullptr, n_out, gtab, st, dm, gb, n_in);
    return launch_generic_win<FIR_CCC, 1, true>(taps_rev, ntaps, in, nullptr, n_out, gtab, st, dm, gb, n_in);
}

int launch_fir_generic(FirKind kind, const float *taps_rev, int ntaps, const void *in, void *out,
                       long long n_out, int decim, const float2 *gtab, hipStream_t st, bool seq)
{
    if (n_out <= 0) return GRHIP_OK;
    size_t sh = (size_t)(ntaps > 0 ? ntaps : 1) * (kind == FIR_CCC ? 8 : 4);
    if (sh > 160 * 1024 - 256) return fail(GRHIP_EINVAL, "generic FIR: %d taps exceed LDS", ntaps);
    if (seq) {
        if (gtab) return fail(GRHIP_EINVAL, "sequential-order FIR has no rotator epilogue");
        switch (kind) {
        case FIR_FFF: return launch_generic_inst<FIR_FFF, true>(taps_rev, ntaps, in, out, n_out, decim, sh, st);
        case FIR_CCF: return launch_generic_inst<FIR_CCF, true>(taps_rev, ntaps, in, out, n_out, decim, sh, st);
        default: return launch_generic_inst<FIR_CCC, true>(taps_rev, ntaps, in, out, n_out, decim, sh, st);
        }
    }
    // the tiled form of the same arithmetic wherever it applies: complex data, a tile's worth of outputs, a decimation
    // and a tap count whose tile fits LDS (16-byte loads: the stream on an 8-byte boundary is served by the range check
    // only if it starts on a 16-byte one)
    if (generic_tiled_ok(kind, ntaps, in, n_out, decim)) {
        const bool ccc = kind == FIR_CCC;
        if (!g_generic_no_window) {
            if (decim == 4) return ccc ? launch_generic_win<FIR_CCC, 4>(taps_rev, ntaps, in, out, n_out, gtab, st)
                                       : launch_generic_win<FIR_CCF, 4>(taps_rev, ntaps, in, out, n_out, gtab, st);
            if (decim == 2) return ccc ? launch_generic_win<FIR_CCC, 2>(taps_rev, ntaps, in, out, n_out, gtab, st)
                                       : launch_generic_win<FIR_CCF, 2>(taps_rev, ntaps, in, out, n_out, gtab, st);
            if (decim == 1) return ccc ? launch_generic_win<FIR_CCC, 1>(taps_rev, ntaps, in, out, n_out, gtab, st)
                                       : launch_generic_win<FIR_CCF, 1>(taps_rev, ntaps, in, out, n_out, gtab, st);
        }
        return ccc ? launch_generic_tiled<FIR_CCC>(taps_rev, ntaps, in, out, n_out, decim, gtab, st)
                   : launch_generic_tiled<FIR_CCF>(taps_rev, ntaps, in, out, n_out, decim, gtab, st);
    }
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
// standalone quadrature demod (general/gr_quadrature_demod_cf.cc:46-62)
// HBM-bound: 8 B in + 4 B out per item.  4 outputs per lane.
// ===========================================================================
__global__ void __launch_bounds__(256)
quad_demod_kernel(const float2 *__restrict__ in, float *__restrict__ out, long long n_out, float gain,
                  const float *__restrict__ tab, int vec)
{
    // arctangent table as (tab[k], tab[k+1]) pairs in LDS: one 8-byte read per interpolation
    __shared__ float2 s_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_tab[i] = make_float2(tab[i], tab[i + 1]);
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n_out; i0 += stride) {
        // in[0] is the history item: output i uses in[i+1] and in[i]
        float2 v[5];
        if (vec && i0 + 4 <= n_out) {
            // 16-byte loads and one 16-byte store per lane (rows are 16-byte aligned)
            const float4 a = *reinterpret_cast<const float4 *>(in + i0);
            const float4 b = *reinterpret_cast<const float4 *>(in + i0 + 2);
            v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w);
            v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
            v[4] = in[i0 + 4];
            float4 o;
            o.x = quad_demod_one(v[1], v[0], gain, s_tab);
            o.y = quad_demod_one(v[2], v[1], gain, s_tab);
            o.z = quad_demod_one(v[3], v[2], gain, s_tab);
            o.w = quad_demod_one(v[4], v[3], gain, s_tab);
            *reinterpret_cast<float4 *>(out + i0) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = (i0 + k <= n_out) ? in[i0 + k] : make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < n_out) out[i0 + k] = quad_demod_one(v[k + 1], v[k], gain, s_tab);
        }
    }
}

// y[n] = y[n] * phase[n] with the reference's unfused complex product (gr_rotator.h:43)
__global__ void __launch_bounds__(256)
rotate_kernel(float2 *__restrict__ y, const float2 *__restrict__ gtab, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = cmul_ref(y[i], gtab[i]);
}

int launch_rotate(float2 *y, const float2 *gtab, long long n, hipStream_t st)
{
    if (n <= 0) return GRHIP_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, gtab, n);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_quad_demod(const float2 *in, float *out, long long n_out, float gain, const float *atan_tab,
                      hipStream_t st)
{
    if (n_out <= 0) return GRHIP_OK;
    long long lanes = (n_out + 3) / 4;
    long long blocks = (lanes + 255) / 256;
    if (blocks > 8192) blocks = 8192;          // grid-stride: 32 workgroups per CU are plenty, the table is loaded once each
    dim3 grid((unsigned)blocks), block(256);
    const int vec = ((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(quad_demod_kernel, grid, block, 0, st, in, out, n_out, gain, atan_tab, vec);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// High-decimation direct form (FAST mode, decimations the tiled kernel does not take, few taps per
// polyphase component): y[n] = sum_k c[k] x[nD + k] costs ntaps/D MACs per INPUT sample, so at D = 20 and
// 400 taps the job is to stream the input through, not to feed the FMA pipes.
//  * a 256-lane workgroup stages the samples of Tn outputs ((Tn-1)D + ntaps of them, Tn even, as many as
//    50 KB of LDS hold) and every lane computes TWO adjacent outputs, so that one LDS read feeds two MACs
//    (taps c[k] and c[k-D]);
//  * LDS layout: sample u lives in sub-array u mod 2P (P = largest power of two dividing D, so every lane
//    of a wave reads the same sub-array for a given k) at index u / 2P: the lane stride is D/P, odd, hence
//    conflict-free 8-byte reads whatever the decimation;
//  * taps are wave-uniform scalar loads; 3 workgroups per CU cover each other's staging.
// Optional rotator-table multiply (freq_xlating) with the reference's unfused product.
// ===========================================================================
constexpr int HIDEC_LDS_SAMPLES = 6144;           // 48 KB of samples + the taps: three workgroups per CU
constexpr int HIDEC_NR = HIDEC_LDS_SAMPLES / 256 + 1;        // staging rounds of a 256-lane workgroup (upper bound)

__host__ __device__ inline int hidec_sub_log(int D)
{
    int v = 0;
    while (((D >> v) & 1) == 0 && v < 4) ++v;     // P = 2^v divides D
    return v + 1;                                  // sub-arrays: 2P
}

// Outputs per tile: every lane computes two adjacent outputs, and all 256 lanes work -- G lane groups of 256 / G lanes
// split the tap range between them (partial sums meet in LDS) -- so a tile is 2 * (256 / G) outputs, G the smallest
// count whose tile fits the LDS.
static int hidec_groups(int D, int ntaps)
{
    const int tn_max = (HIDEC_LDS_SAMPLES - ntaps - 64) / D;
    for (int G = 1; G <= 8; ++G)
        if (2 * (256 / G) <= tn_max) return G;
    return 0;
}

int hidec_outputs_per_tile(int D, int ntaps)
{
    const int G = hidec_groups(D, ntaps);
    return G ? 2 * (256 / G) : 0;
}

bool hidec_supported(int D, int ntaps)
{
    return D >= 3 && D <= 256 && ntaps >= 1 && ntaps <= 2048 && hidec_groups(D, ntaps) > 0;
}

// PREMIX (freq_xlating with a real prototype): y_bp[n0+m] = e^{-jw m D} sum_k proto[k] (x[u0+u] e^{jwu}), u = mD + k
// relative to the tile: the samples are mixed with etab[u] = e^{jwu} while they are staged, the taps are the
// real prototype (half the FMAs), and the lane's outputs are turned back with vtab[m] = e^{-jw m D}.
// Round 2: persistent workgroups walk the tiles with the next tile's samples (and rotator phases) requested one tile
// ahead into registers through a range-checked buffer descriptor; the pre-mix phasors of a lane's samples are tile
// independent and stay in registers; the taps sit in LDS (uniform-address reads: in-order with the sample reads, so the
// waits are counted -- as scalar loads they shared a counter that can only be waited to zero); all 256 lanes compute.
// DEMOD (pre-mix form only): the fused xlating -> quadrature demodulator of the other decimations.  As in the tiled and the
// matrix-core kernels the demodulator works directly on the pre-mixed accumulators (the pre-mix correction e^{-jwD} and
// the rotator step e^{+jwD} cancel in y[n] conj(y[n-1])): no rotator table, no intermediate in HBM.  Tiles overlap by one
// output pair (lane 0 computes it only to hand its second output on), the first output of a call takes the previous
// call's last composite output (y_prev) brought into the tile's frame, the call's last composite output goes to y_last.
struct HidecAtan {
    const float2 *p;
    __device__ __forceinline__ float2 operator[](int k) const { return p[k]; }
};

template <bool CTAPS, int V1, bool PREMIX, bool DEMOD = false>
__global__ void __launch_bounds__(256, 3)
fir_hidec_kernel(const float2 *__restrict__ x, long long n_in, const float *__restrict__ taps_g, int ntaps, int D,
                 int G, long long n_out, float2 *__restrict__ y, const float2 *__restrict__ gtab,
                 const float2 *__restrict__ etab, const float2 *__restrict__ vtab, long long ntiles,
                 float *__restrict__ d_out = nullptr, float gain = 0.f, const float2 *__restrict__ y_prev = nullptr,
                 float2 *__restrict__ y_last = nullptr, const float *__restrict__ atan_tab = nullptr,
                 int n_streams = 1, long long x_stride = 0, long long d_stride = 0, long long n_lo = 0)
{
    // (DEMOD, batched: tile ids run over (tile, stream) pairs, streams fastest; stream s reads x + s x_stride, whose first
    // n_lo items lie before the buffer and read as zero -- a fresh capture's history -- writes d_out + s d_stride and takes
    // / leaves its carry at y_prev[s] / y_last[s])
    static_assert(!DEMOD || (PREMIX && !CTAPS), "the fused demodulator belongs to the pre-mix form");
    extern __shared__ __attribute__((aligned(16))) unsigned char hidec_smem[];
    typedef float hd_f32x2 __attribute__((ext_vector_type(2)));
    typedef float hd_f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NSUB = 1 << V1, PM = NSUB - 1, SUB = ((HIDEC_LDS_SAMPLES >> V1) + 2) | 1;
    constexpr int TW = CTAPS ? 2 : 1;
    hd_f32x2 *xs = reinterpret_cast<hd_f32x2 *>(hidec_smem);                       // 2P sub-arrays of odd stride
    float *tl = reinterpret_cast<float *>(hidec_smem + (size_t)(NSUB * SUB) * 8);  // the padded taps: c[k] at tl[TW (D + k)]
    const int t = threadIdx.x;
    const int LG = 256 / G;                          // lanes per group
    const int grp = t / LG, tp = t - grp * LG;       // tap group, output pair inside the tile
    const bool idle = grp >= G;                      // (256 - G * LG lanes when G does not divide 256)
    const int Tn = 2 * LG;
    const int ns = (Tn - 1) * D + ntaps + NSUB;      // samples touched, the last tap group rounded up (all staged)
    const int ntl = (ntaps + 2 * D + 64) * TW;
    for (int i = t; i < ntl; i += 256) tl[i] = taps_g[i];

    // per-lane constants of the staging rounds: u = t + 256 i -> sub-array t & PM, index (t >> V1) + (256 >> V1) i
    const int st_slot = (t & PM) * SUB + (t >> V1);
    constexpr int ST_STEP = 256 >> V1;
    hd_f32x2 ph[PREMIX ? HIDEC_NR : 1];
    if (PREMIX) {
#pragma unroll
        for (int i = 0; i < HIDEC_NR; ++i) {
            const int u = t + 256 * i;
            const float2 e = etab[u < HIDEC_LDS_SAMPLES + 128 ? u : 0];
            ph[i] = hd_f32x2{e.x, e.y};
        }
    }
    hd_f32x2 v0c{1.f, 0.f}, v1c{1.f, 0.f};
    if (PREMIX) {
        const float2 a = vtab[2 * tp], b = vtab[2 * tp + 1];
        v0c = hd_f32x2{a.x, a.y}; v1c = hd_f32x2{b.x, b.y};
    }
    // the stream: items >= n_in read as zero (and move no bytes)
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(gtab), 0, gtab ? (int)(n_out * 8) : 0, 0x00020000);
    hd_f32x2 pre[HIDEC_NR];
    hd_f32x4 gq{1.f, 0.f, 1.f, 0.f};
    const int TNEW = DEMOD ? Tn - 2 : Tn;            // new outputs per tile
    auto request = [&](long long id) __attribute__((always_inline)) {
        int tq = t;
        asm volatile("" : "+v"(tq));                 // (offsets per tile: hoisted, they would be spilled)
        const long long tile = id / n_streams;
        const int sidx = (int)(id - tile * n_streams);
        // the stream from its first readable item on: items >= n_in read as zero (and move no bytes)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(x + (long long)sidx * x_stride + n_lo), 0,
                                                                           (int)((n_in - n_lo) * 8), 0x00020000);
        const long long u0 = (tile * TNEW - (DEMOD ? 2 : 0)) * D - n_lo;   // relative to the descriptor (DEMOD: the first tile starts two outputs early)
        const int first = u0 < 0 ? (int)-u0 : 0;     // samples before the stream: explicit out-of-range offsets
        const int vo = (int)(u0 * 8) + 8 * tq;
#pragma unroll
        for (int i = 0; i < HIDEC_NR; ++i) {
            if (256 * i < ns) {                      // (wave-uniform: rounds past the tile are skipped, not loaded out of range)
                const int off = (tq + 256 * i < ns && tq + 256 * i >= first) ? vo + 2048 * i : 0x7ffffff0;
                pre[i] = __builtin_bit_cast(hd_f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0));
            }
        }
        if (!DEMOD && gtab) {
            const long long n = tile * Tn + 2 * tp;
            // (two 8-byte loads: an odd n_out ends in the middle of the pair)
            const hd_f32x2 g0 = __builtin_bit_cast(hd_f32x2, __builtin_amdgcn_raw_buffer_load_b64(gr, (int)(n * 8), 0, 0));
            const hd_f32x2 g1 = __builtin_bit_cast(hd_f32x2, __builtin_amdgcn_raw_buffer_load_b64(gr, (int)(n * 8) + 8, 0, 0));
            gq = hd_f32x4{g0.x, g0.y, g1.x, g1.y};
        }
    };
    long long id = blockIdx.x;
    const long long nids = ntiles * n_streams;
    if (id < nids) request(id);
    __syncthreads();                                 // taps visible

    // tap range of this lane's group: groups of NSUB taps, k over [0, ntaps + D)
    const int ngroups = (ntaps + D + NSUB - 1) >> V1;
    const int gper = (ngroups + G - 1) / G;
    const int gk0 = grp * gper, gk1 = idle ? gk0 : (gk0 + gper < ngroups ? gk0 + gper : ngroups);
    const hd_f32x2 *xl = xs + ((2 * tp * D) >> V1);  // 2 tp D is a multiple of 2P

    for (; id < nids; id += gridDim.x) {
        const long long tile = id / n_streams;
        const int sidx = (int)(id - tile * n_streams);
        // ---- registers -> LDS (pre-mixed)
#pragma unroll
        for (int i = 0; i < HIDEC_NR; ++i) {
            if (256 * i >= ns) break;
            if (t + 256 * i < ns) {
                hd_f32x2 w = pre[i];
                if (PREMIX) { const float2 m = cmul_ref(make_float2(w.x, w.y), make_float2(ph[i].x, ph[i].y)); w = hd_f32x2{m.x, m.y}; }
                xs[st_slot + ST_STEP * i] = w;
            }
        }
        const hd_f32x4 gcur = gq;
        __syncthreads();
        if (id + gridDim.x < nids) request(id + gridDim.x);

        // ---- MACs: tap c[k] feeds the first output (k < ntaps), c[k-D] the second (k >= D); the tap table is zero padded
        // by D in front and behind (host), so no conditions are needed
        float a0x = 0.f, a0y = 0.f, a1x = 0.f, a1y = 0.f;
        const float *t0 = tl + TW * D;               // c[k] at t0[TW k]
        for (int gk = gk0; gk < gk1; ++gk) {
            const hd_f32x2 *xg = xl + gk;
            const int k = gk << V1;
#pragma unroll 8
            for (int j = 0; j < NSUB; ++j) {         // (8 at a time: 16 or 32 reads hoisted at once cost more registers than there are)
                const hd_f32x2 xv = xg[j * SUB];
                if (CTAPS) {
                    const float hr0 = t0[2 * (k + j)], hi0 = t0[2 * (k + j) + 1], hr1 = t0[2 * (k + j - D)], hi1 = t0[2 * (k + j - D) + 1];
                    a0x = __builtin_fmaf(hr0, xv.x, a0x); a0x = __builtin_fmaf(-hi0, xv.y, a0x);
                    a0y = __builtin_fmaf(hr0, xv.y, a0y); a0y = __builtin_fmaf(hi0, xv.x, a0y);
                    a1x = __builtin_fmaf(hr1, xv.x, a1x); a1x = __builtin_fmaf(-hi1, xv.y, a1x);
                    a1y = __builtin_fmaf(hr1, xv.y, a1y); a1y = __builtin_fmaf(hi1, xv.x, a1y);
                } else {
                    const float h0 = t0[k + j], h1 = t0[k + j - D];
                    a0x = __builtin_fmaf(h0, xv.x, a0x); a0y = __builtin_fmaf(h0, xv.y, a0y);
                    a1x = __builtin_fmaf(h1, xv.x, a1x); a1y = __builtin_fmaf(h1, xv.y, a1y);
                }
            }
        }
        // ---- the groups' partial sums meet in LDS (the sample area is free once every lane is through its MACs)
        hd_f32x4 *red = reinterpret_cast<hd_f32x4 *>(xs);
        float2 *carry = reinterpret_cast<float2 *>(xs) + 1024;          // DEMOD: second output of every pair (2 KB, behind the partial sums)
        float2 *atp = reinterpret_cast<float2 *>(xs) + 2048;            // DEMOD: the arctangent table as (tab[k], tab[k+1]) pairs
        if (G > 1 || DEMOD) {
            __syncthreads();
            if (grp > 0 && !idle) red[(grp - 1) * LG + tp] = hd_f32x4{a0x, a0y, a1x, a1y};
            if (DEMOD) atp[t] = make_float2(atan_tab[t], atan_tab[t + 1]);
            __syncthreads();
            if (grp == 0) {
                for (int g2 = 1; g2 < G; ++g2) {
                    const hd_f32x4 r = red[(g2 - 1) * LG + tp];
                    a0x += r[0]; a0y += r[1]; a1x += r[2]; a1y += r[3];
                }
            }
        }
        if (DEMOD) {
            if (grp == 0) carry[tp] = make_float2(a1x, a1y);
            __syncthreads();
            if (grp == 0) {
                const long long n = tile * TNEW - 2 + 2 * tp;           // outputs n, n + 1 (lane 0: the overlap pair)
                float2 prev = tp > 0 ? carry[tp - 1] : make_float2(0.f, 0.f);
                if (tile == 0 && tp == 1) {
                    // predecessor of the call's first output: the previous call's last composite output y_bp[-1], in this
                    // tile's frame a_m = e^{+jw m D} y_bp (m = 1 for output -1); a fresh stream has none: zero
                    prev = make_float2(0.f, 0.f);
                    if (y_prev) {
                        const float2 yp = y_prev[sidx], v1 = vtab[1];
                        prev = cmul_ref(yp, make_float2(v1.x, -v1.y));
                    }
                }
                const HidecAtan tabv{atp};
                const float2 a0 = make_float2(a0x, a0y), a1 = make_float2(a1x, a1y);
                const float d0 = quad_demod_fast(a0, prev, gain, tabv);
                const float d1 = quad_demod_fast(a1, a0, gain, tabv);
                if (tp > 0) {
                    float *dst = d_out + (long long)sidx * d_stride;
                    if (n + 1 < n_out && ((((uintptr_t)(dst + n)) & 7) == 0)) {
                        *reinterpret_cast<float2 *>(dst + n) = make_float2(d0, d1);
                    } else {
                        if (n < n_out) dst[n] = d0;
                        if (n + 1 < n_out) dst[n + 1] = d1;
                    }
                    // the call's last composite output, for the next call
                    if (y_last && (n == n_out - 1 || n + 1 == n_out - 1)) {
                        const bool second = n + 1 == n_out - 1;
                        y_last[sidx] = cmul_ref(second ? a1 : a0, second ? make_float2(v1c.x, v1c.y) : make_float2(v0c.x, v0c.y));
                    }
                }
            }
        } else if (grp == 0) {
            const long long n = tile * Tn + 2 * tp;
            float2 a0 = make_float2(a0x, a0y), a1 = make_float2(a1x, a1y);
            if (PREMIX) {
                a0 = cmul_ref(a0, make_float2(v0c.x, v0c.y));
                a1 = cmul_ref(a1, make_float2(v1c.x, v1c.y));
            }
            if (gtab) {
                a0 = cmul_ref(a0, make_float2(gcur[0], gcur[1]));
                a1 = cmul_ref(a1, make_float2(gcur[2], gcur[3]));
            }
            if (n + 1 < n_out && ((((uintptr_t)(y + n)) & 15) == 0)) {
                *reinterpret_cast<float4 *>(y + n) = make_float4(a0.x, a0.y, a1.x, a1.y);
            } else {
                if (n < n_out) y[n] = a0;
                if (n + 1 < n_out) y[n + 1] = a1;
            }
        }
        __syncthreads();                             // the sample area belongs to the next tile from here
    }
}

int launch_fir_hidec(bool ctaps, const float *taps_padded, int ntaps, int decim, const float2 *x, long long n_in,
                     float2 *y, long long n_out, const float2 *gtab, hipStream_t st, const float2 *etab, const float2 *vtab)
{
    if (n_out <= 0) return GRHIP_OK;
    if (!hidec_supported(decim, ntaps) || n_in < 1) return fail(GRHIP_EINVAL, "high-decimation FIR: unsupported shape");
    const bool premix = etab && vtab;
    if (premix && ctaps) return fail(GRHIP_EINVAL, "high-decimation FIR: pre-mix form takes real taps");
    if (n_in * 8 > 0x7fffffffLL || n_out * 8 > 0x7fffffffLL)
        return fail(GRHIP_EINVAL, "high-decimation FIR: more than 2 GB of items in one call");
    const int G = hidec_groups(decim, ntaps);
    const int Tn = 2 * (256 / G);
    const long long ntiles = (n_out + Tn - 1) / Tn;
    const int v1 = hidec_sub_log(decim);
    const int nsub = 1 << v1, sub = ((HIDEC_LDS_SAMPLES >> v1) + 2) | 1;
    const size_t lds = (size_t)nsub * sub * 8 + (size_t)(ntaps + 2 * decim + 64) * (ctaps ? 2 : 1) * 4;
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        n_cus = n > 0 ? n : 256;
    }
    const long long per_cu = lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1);
    const long long cap = per_cu * n_cus;
    const unsigned blocks = (unsigned)(ntiles < cap ? ntiles : cap);
#define GRHIP_HIDEC(C, V, P)                                                                                            \
    do {                                                                                                                \
        static size_t cfg = 0;                                                                                          \
        if (lds > 48 * 1024 && lds > cfg) {                                                                             \
            GRHIP_HIP(hipFuncSetAttribute((const void *)fir_hidec_kernel<C, V, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            cfg = lds;                                                                                                  \
        }                                                                                                               \
        hipLaunchKernelGGL((fir_hidec_kernel<C, V, P>), dim3(blocks), dim3(256), lds, st, x, n_in, taps_padded, ntaps, decim, \
                           G, n_out, y, gtab, etab, vtab, ntiles);                                                      \
    } while (0)
#define GRHIP_HIDEC_V(C, P)                                                                                  \
    switch (v1) { case 1: GRHIP_HIDEC(C, 1, P); break; case 2: GRHIP_HIDEC(C, 2, P); break;                  \
                  case 3: GRHIP_HIDEC(C, 3, P); break; case 4: GRHIP_HIDEC(C, 4, P); break;                  \
                  default: GRHIP_HIDEC(C, 5, P); break; }
    if (premix) { GRHIP_HIDEC_V(false, true) }
    else if (ctaps) { GRHIP_HIDEC_V(true, false) }
    else { GRHIP_HIDEC_V(false, false) }
#undef GRHIP_HIDEC_V
#undef GRHIP_HIDEC
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_fir_hidec_demod(const float *taps_padded, int ntaps, int decim, const float2 *x, long long n_in, float *d_out,
                           long long n_out, float gain, const float2 *y_prev, float2 *y_last, const float *atan_tab,
                           const float2 *etab, const float2 *vtab, hipStream_t st, int n_streams, long long x_stride,
                           long long d_stride, long long n_lo, int max_wg_per_cu, int max_cus)
{
    if (n_out <= 0) return GRHIP_OK;
    if (!hidec_supported(decim, ntaps) || n_in < 1 || !etab || !vtab || !atan_tab)
        return fail(GRHIP_EINVAL, "high-decimation FIR + demodulator: unsupported shape");
    if (n_in * 8 > 0x7fffffffLL) return fail(GRHIP_EINVAL, "high-decimation FIR: more than 2 GB of items in one call");
    const int G = hidec_groups(decim, ntaps);
    const int Tn = 2 * (256 / G);
    const long long ntiles = (n_out + (Tn - 2) - 1) / (Tn - 2);
    const int v1 = hidec_sub_log(decim);
    const int nsub = 1 << v1, sub = ((HIDEC_LDS_SAMPLES >> v1) + 2) | 1;
    const size_t lds = (size_t)nsub * sub * 8 + (size_t)(ntaps + 2 * decim + 64) * 4;
    int dev = 0, n_cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus < 1)
        n_cus = 256;
    long long per_cu = lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1);
    if (max_wg_per_cu > 0 && per_cu > max_wg_per_cu) per_cu = max_wg_per_cu;
    if (n_streams < 1 || n_lo < 0 || n_lo > n_in) return fail(GRHIP_EINVAL, "high-decimation FIR + demodulator: bad batch arguments");
    const long long cap = per_cu * (max_cus > 0 && max_cus < n_cus ? max_cus : n_cus);
    const long long nids = ntiles * n_streams;
    const unsigned blocks = (unsigned)(nids < cap ? nids : cap);
#define GRHIP_HIDEC_D(V)                                                                                                \
    do {                                                                                                                \
        static size_t cfg = 0;                                                                                          \
        if (lds > 48 * 1024 && lds > cfg) {                                                                             \
            GRHIP_HIP(hipFuncSetAttribute((const void *)fir_hidec_kernel<false, V, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            cfg = lds;                                                                                                  \
        }                                                                                                               \
        hipLaunchKernelGGL((fir_hidec_kernel<false, V, true, true>), dim3(blocks), dim3(256), lds, st, x, n_in, taps_padded, ntaps, decim, \
                           G, n_out, (float2 *)nullptr, (const float2 *)nullptr, etab, vtab, ntiles, d_out, gain, y_prev, y_last, atan_tab, \
                           n_streams, x_stride, d_stride, n_lo);                                                        \
    } while (0)
    switch (v1) { case 1: GRHIP_HIDEC_D(1); break; case 2: GRHIP_HIDEC_D(2); break; case 3: GRHIP_HIDEC_D(3); break;
                  case 4: GRHIP_HIDEC_D(4); break; default: GRHIP_HIDEC_D(5); break; }
#undef GRHIP_HIDEC_D
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// phasor tables of the pre-mix form (double precision angles): etab[u] = e^{jwu}, vtab[m] = e^{-jw m decim}
void hidec_premix_tables(double omega, int decim, std::vector<float> &etab, std::vector<float> &vtab)
{
    etab.resize(2 * (size_t)(HIDEC_LDS_SAMPLES + 128));
    for (int u = 0; u < HIDEC_LDS_SAMPLES + 128; ++u) {
        etab[2 * u] = (float)cos(omega * u); etab[2 * u + 1] = (float)sin(omega * u);
    }
    vtab.resize(2 * 512);
    for (int m = 0; m < 512; ++m) {
        const double a = -omega * (double)m * (double)decim;
        vtab[2 * m] = (float)cos(a); vtab[2 * m + 1] = (float)sin(a);
    }
}

// taps for launch_fir_hidec: correlation order, D zero taps in front and behind + one group of slack
void hidec_pad_taps(const float *taps_corr, int ntaps, int tw, int decim, std::vector<float> &out)
{
    out.assign((size_t)(ntaps + 2 * decim + 64) * tw, 0.f);
    for (int k = 0; k < ntaps * tw; ++k) out[(size_t)decim * tw + k] = taps_corr[k];
}

}  // namespace grhip
