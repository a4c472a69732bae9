// fir_mfma.hip -- FAST-mode engine of the long decimating FIRs with REAL taps on complex data
// (gr_fir_ccf, and gr_freq_xlating_fir_filter_ccc with a real prototype in its pre-mix form,
// with the fused quadrature demodulator): the FIR as a banded-Toeplitz product on the matrix
// cores.
//
// Why: at 256 taps the direct form is VALU-bound (28 flop/B against a ridge of
// 19.7): fir_tiled.hip runs the vector pipes 84 % busy and still reaches only 0.35 of the HBM
// roofline.  The contraction itself is dense -- 16 consecutive outputs of a decimating FIR are
// a 16 x (15 D + T) band matrix of taps times a matrix of samples whose columns are
// independent stream segments (mfma_tables.h) -- and v_mfma_f32_16x16x32_f16 retires it at
// 16x the fp32 vector rate.  binary16 alone has 11 significant bits, so both operands are
// split in two binary16 halves, x = xh + xl to ~22 bits (the taps on the host, the samples
// while they are staged, after a per-tile power-of-two scaling that puts the tile's largest
// sample at 2^14: the low halves stay normal numbers), and a product is three MFMAs
// (Ah Xh + Ah Xl + Al Xh) accumulated in f32: 3/16 of the fp32 time at 80 % band occupancy.
// Deviation from the f32 direct form ~3e-7 of sum|h||x| (tests: the same 1e-5 as before).
// What is left on the vector pipes is staging (pre-mix, split: ~7 operations per sample) and
// the demodulator; the kernel is bound by HBM.
//
//  * Persistent 256-lane workgroups, 2 per CU, tiles of NTC = 2000 computed outputs (1984
//    new) handed out as in fir_tiled.hip (two static tiles, then a queue).
//  * The tile is fetched with 16-byte raw buffer loads into registers one tile ahead (range
//    check = history zeros and stream end), scaled, pre-mixed, split and written to four
//    binary16 planes in LDS (re-hi, re-lo, im-hi, im-lo), sample u at byte 2u + 32 (u / Q),
//    Q = 64 D: the skew puts the 8 segments of a wave on different LDS slots, so the operand
//    reads (ds_read_b128, 8 consecutive samples of one segment per lane) are conflict-free.
//  * Wave w owns 8 segments of 64 outputs (columns = segment x {re, im}).  It slides along
//    them in chunks of 32 samples: chunk c is k-step c - 2b of block b (D = 4), so every
//    chunk is read from LDS once and multiplied against up to 5 k-steps of A held in
//    registers (80 VGPRs).
//  * Epilogue per 16-output block through a wave-private LDS scratch (no barrier): the
//    accumulator tile is written [segment][row][re,im] and read back as two consecutive
//    complex outputs per lane; the demodulator's predecessor is the neighbouring lane's
//    second output (DPP), the previous block's last output, or -- first output of a segment --
//    the previous segment's last output, handled after the last block.  The first block of
//    every wave is OVERLAP: it is computed only to hand its last output on, which makes the
//    waves of a tile and the tiles of a stream independent.
#include <cstdio>
#include <cstdlib>

#include "device_math.h"
#include "fir_kernels.h"
#include "grhip_internal.h"
#include "mfma_tables.h"

#ifndef GRHIP_LG_W2
#define GRHIP_LG_W2 0
#endif
#ifndef GRHIP_LG_ACC3
#define GRHIP_LG_ACC3 0
#endif
#ifndef GRHIP_LG_ORDER
#define GRHIP_LG_ORDER 1
#endif
#ifndef GRHIP_MF_WGS
#define GRHIP_MF_WGS 2            // workgroups per CU the shipped kernel is compiled and launched for (experiment: 3 with GRHIP_MF_NBLK=2)
#endif
#ifndef GRHIP_LG_SPF
#define GRHIP_LG_SPF 0              // 1: the staging's round phasors (scalar loads) requested a pair of rounds ahead, the first ones before the barrier:
                                    // 1.1859 / 1.1865 ms against 1.1854 / 1.1903 -- nothing
#endif
#ifndef GRHIP_LG_MAX4
#define GRHIP_LG_MAX4 0             // 1: four running maxima instead of one chain of 2 NI dependent v_max3: 1.205 / 1.176 ms against 1.192 / 1.178 -- nothing
#endif
#ifndef GRHIP_LG_EPI2
#define GRHIP_LG_EPI2 0             // 1: the epilogue's blocks in pairs (four demodulator chains for the scheduler): 1.208 / 1.205 ms against 1.204 / 1.202
#endif
#ifndef GRHIP_LG_STAGE2
#define GRHIP_LG_STAGE2 1         // the pre-mix staging two rounds at a time in a hand-ordered block (0: a round at a time, the compiler's order)
#endif
#ifndef GRHIP_LG_LEAN
#define GRHIP_LG_LEAN 0           // 1: the lean demodulator of device_math.h (seven vector instructions fewer per output, same values on cfg2) in the
                                  // shipped kernel's epilogue: 1.225 / 1.228 ms against 1.220 / 1.239 (same box, interleaved) -- no lever, like every
                                  // other cut of the vector work: the kernel is not bound by the vector pipe's throughput alone (profiles/r03_notes.md)
#endif
#ifndef GRHIP_LG_SPREAD
#define GRHIP_LG_SPREAD 0
#endif


namespace grhip {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) *cfloat_cp;

namespace {

__host__ __device__ constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

constexpr int SCR_SEG = 36;                          // floats per segment of the epilogue scratch (32 + pad)
constexpr int SCR_WAVE = mf::SEGS * SCR_SEG;         // floats per wave

template <int D, int KS> struct Geo {
    static constexpr int Q = mf::SEG_OUT * D;        // samples between segment starts
    static constexpr int LOGQ = ilog2c(Q);
    static constexpr int CB = mf::BLK * D / mf::CHUNK;   // chunks between block starts
    static constexpr int NCH = mf::chunks_per_seg(D, KS);
    static constexpr int SP = mf::tile_samples(D, KS);
    static constexpr int NI = mf::rounds(D, KS);
    static constexpr int PL = mf::plane_bytes(D, KS);
    // Epilogue scratch: none of its own.  Wave w is the only reader of the samples between what the wave before it
    // reads into w's range (HALO samples) and the next wave's range, so once its matrix phase is over it may reuse
    // that stretch of each plane: block b's accumulator tile goes to plane b there (SCR_WAVE floats each).
    static constexpr int HALO = NCH * mf::CHUNK - (mf::SEGS * mf::SEG_OUT - mf::WAVE_NEW) * D - (mf::SEGS - 1) * 0;
    static constexpr int OFF_ATAN = 4 * PL;
    static constexpr int OFF_MISC = OFF_ATAN + 256 * 8;
    static constexpr int OFF_G = OFF_MISC + 32;      // 4 wave maxima, the queue's hand-over slot; then the correction band G (freq_xlating)
    static constexpr int LDS = OFF_G + mf::NG * 64 * 16;
    static_assert((1 << LOGQ) == Q, "segment stride must be a power of two");
    static_assert(HALO >= 0 && HALO % 32 == 0, "halo");
    static_assert(mf::NBLK != 4 || mf::plane_pos(mf::WAVE_NEW * D - 1, D) + 2 - mf::plane_pos(HALO, D) >= SCR_WAVE * 4, "a wave's own stretch of a plane must hold one accumulator tile");
    static_assert(mf::ROUND % Q == 0, "a staging round must cover whole segment strides");
};

// max over the 64 lanes of a wave of a non-negative value (DPP, as wave_sum in fir_tiled.hip:
// rows masked out of the row_bcast steps contribute 0)
#define GRHIP_DPPF(v, ctrl, rmask, bc) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, (bc)))
__device__ __forceinline__ float wave_max_nonneg(float x)
{
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0xB1, 0xf, true));
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0x4E, 0xf, true));
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0x141, 0xf, true));
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0x140, 0xf, true));
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0x142, 0xa, false));
    x = __builtin_fmaxf(x, GRHIP_DPPF(x, 0x143, 0xc, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

__device__ __forceinline__ float wave_sum_f(float x)
{
    x += GRHIP_DPPF(x, 0xB1, 0xf, true);
    x += GRHIP_DPPF(x, 0x4E, 0xf, true);
    x += GRHIP_DPPF(x, 0x141, 0xf, true);
    x += GRHIP_DPPF(x, 0x140, 0xf, true);
    x += GRHIP_DPPF(x, 0x142, 0xa, false);
    x += GRHIP_DPPF(x, 0x143, 0xc, false);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// value of the lane `delta` places down the same row of 16 (DPP row_shr) / up (row_shl)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// low halves of a split pair: (binary16)(x - (float)hi), one mixed-precision FMA per element
// (x * 1.0 - hi with hi read as binary16, result rounded to binary16 into one half of the register)
__device__ __forceinline__ h16x2 split_lo(f32x2 x, h16x2 hi)
{
    unsigned lo;
    const unsigned h = __builtin_bit_cast(unsigned, hi);
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x.x), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x.y), "v"(h));
    return __builtin_bit_cast(h16x2, lo);
}

// both halves of a scaled pair in four mixed-precision FMAs: hi = (binary16)(x s), lo = (binary16)(x s - hi); s is a
// power of two, so x s is exact inside the FMA and each half is rounded once (the same values as a product, a
// conversion and split_lo, three instructions fewer per pair)
__device__ __forceinline__ void split_scaled(float x0, float x1, float s, h16x2 &hi, h16x2 &lo)
{
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = __builtin_bit_cast(h16x2, h);
    lo = __builtin_bit_cast(h16x2, l);
}

// Diagnostic build only (-DGRHIP_STAMP, `make stamp`): per-wave time shares of the phases of a
// tile (100 MHz real-time counter), written to a buffer of their own (never to an output).
#ifdef GRHIP_STAMP
__device__ unsigned long long *g_mf_stamp_buf = nullptr;
__device__ __forceinline__ unsigned long long mf_stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MF_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = mf_stamp_now(), st_begin = st_last
#define MF_STAMP(k) do { unsigned long long n__ = mf_stamp_now(); st_acc[k] += n__ - st_last; st_last = n__; } while (0)
#define MF_STAMP_OUT(nwaves) do { if ((threadIdx.x & 63) == 0 && g_mf_stamp_buf) { \
        unsigned long long *o__ = g_mf_stamp_buf + ((size_t)blockIdx.x * (nwaves) + (threadIdx.x >> 6)) * 10; \
        for (int k__ = 0; k__ < 8; ++k__) o__[k__] = st_acc[k__]; \
        o__[8] = st_begin; o__[9] = mf_stamp_now(); } } while (0)
#else
#define MF_STAMP_DECL
#define MF_STAMP(k)
#define MF_STAMP_OUT(nwaves)
#endif

struct AtanPairs {
    const f32x2 *p;
    __device__ __forceinline__ f32x2 operator[](int k) const { return p[k]; }
};

}  // namespace

template <int D, int KS, bool PREMIX, int EPI, bool TAPQ = false>
__global__ void __launch_bounds__(mf::THREADS, GRHIP_MF_WGS) fir_mfma_kernel(const FirMfmaArgs a)
{
    static_assert(!TAPQ || PREMIX, "the tap-angle correction belongs to freq_xlating's pre-mix form");
    using G = Geo<D, KS>;
    constexpr int NI = G::NI, PL = G::PL, LOGQ = G::LOGQ, CB = G::CB, SP = G::SP;
    constexpr bool DEMOD = EPI == EPI_DEMOD;
    constexpr bool ROT = EPI == EPI_ROTATE;
    static_assert(!DEMOD || PREMIX, "the fused demodulator belongs to the pre-mix form");
    static_assert(EPI == EPI_DEMOD || EPI == EPI_ROTATE || EPI == EPI_NONE, "epilogue");
    constexpr int NTE = mf::NTE, BLK = mf::BLK, NBLK = mf::NBLK;

    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const AtanPairs s_atan{reinterpret_cast<const f32x2 *>(smem + G::OFF_ATAN)};
    float *wmax = reinterpret_cast<float *>(smem + G::OFF_MISC);
    unsigned *sched_slot = reinterpret_cast<unsigned *>(smem + G::OFF_MISC + 16);

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int tiles_per_stream = (int)((a.n_out + NTE - 1) / NTE);

    // ---- launch constants ---------------------------------------------------------
    // band matrix of the taps, both halves, every k-step: registers for the whole launch
    h16x8 Ah[KS], Al[KS];
    {
        const h16x8 *Ag = reinterpret_cast<const h16x8 *>(a.A);
#pragma unroll
        for (int js = 0; js < KS; ++js) {
            Ah[js] = Ag[(js * 2 + 0) * 64 + lane];
            Al[js] = Ag[(js * 2 + 1) * 64 + lane];
        }
    }
    // freq_xlating: the correction band of the reference's tap-angle quantisation (mfma_tables.h), high halves of the
    // middle k-steps, behind A in the table; its sum T joins the accumulator with the factor j: re -= T(im), im += T(re)
    constexpr int JG0 = mf::g_first(KS);
    // (in LDS, 4 KB, read at lane * 16 when a middle k-step comes up: in registers it would be 16 more than the 256 there are)
    if (TAPQ) {
        const h16x8 *Gg = reinterpret_cast<const h16x8 *>(a.A) + (size_t)KS * 2 * 64;
        if (t < mf::NG * 64) reinterpret_cast<h16x8 *>(smem + G::OFF_G)[t] = Gg[t];
    }
    const float tq_sgn = __builtin_amdgcn_ldexpf((lane & 8) ? 1.0f : -1.0f, -mf::G_EXTRA_EXP);
    // pre-mix phasors of the lane's two samples of round 0: e^{jw(2t - off)}, e^{jw(2t + 1 - off)}
    f32x2 wl{1.f, 0.f}, wl1{1.f, 0.f};
    if (PREMIX) {
        const float2 v = a.wlane[t];
        wl = f32x2{v.x, v.y};
        wl1 = cmul_pk(wl, f32x2{a.wstep.x, a.wstep.y});
    }
    const cfloat_cp stab = (cfloat_cp)a.stab;
    if (DEMOD) {
        f32x2 *at = reinterpret_cast<f32x2 *>(smem + G::OFF_ATAN);
        for (int i = t; i < 256; i += mf::THREADS)
            at[i] = GRHIP_LG_LEAN ? f32x2{a.atan_tab[i], a.atan_tab[i + 1] - a.atan_tab[i]} : f32x2{a.atan_tab[i], a.atan_tab[i + 1]};
    }
    // staging store: sample u = 2t + 512 i  ->  byte 2u + 32 (u >> LOGQ) of each plane
    const int st_off = 4 * t + 32 * ((2 * t) >> LOGQ);
    constexpr int ST_STEP = 2 * mf::ROUND + 32 * (mf::ROUND >> LOGQ);
    // operand reads: lane = (column, k-group g); column = part * 8 + segment
    const int col = lane & 15, part = col >> 3, sl = col & 7, g = lane >> 4;
    const int rd_u = w * (mf::WAVE_NEW * D) + (sl << LOGQ) + 8 * g;       // sample of chunk 0
    const int rd_plane = part * 2 * PL;
    // epilogue scratch: written in the accumulator layout, read as 2 consecutive outputs per lane
    // (position of the first sample only this wave reads; every wave's range starts on a multiple of 32 samples)
    const int scw_u = w * (mf::WAVE_NEW * D) + G::HALO;
    float *scw = reinterpret_cast<float *>(smem + 2 * scw_u + 32 * (scw_u >> LOGQ));      // + b * PL bytes for block b
    const int sc_wr = sl * SCR_SEG + 8 * g + part;                      // + 2 i
    const int rsl = lane >> 3, r8 = lane & 7;
    const int sc_rd = rsl * SCR_SEG + 4 * r8;

    f32x4 pf[NI];

    // The stream is addressed through a raw buffer descriptor: lanes before the first real item
    // (the history zeros of a fresh flowgraph) or past the end get zeros from the hardware
    // range check.  The descriptor starts on a 16-byte boundary: when the first real item does
    // not (lead = 1), it starts one item earlier -- same 16-byte granule, so the same
    // allocation -- and that item is zeroed while staging (a 16-byte load at offset -8 would
    // be out of range as a whole).
    const int lead = a.off ^ (int)(a.n_lo & 1);
    auto tile_geom = [&](int s, int b, __amdgpu_buffer_rsrc_t &rsrc, int &voff) __attribute__((always_inline)) {
        const long long g0 = ((long long)b * NTE - BLK) * D - a.off - a.n_lo + lead;   // tile start relative to the descriptor
        const float2 *x = a.x + (long long)s * a.x_stride + a.n_lo - lead;
        const long long bytes = (a.n_in - a.n_lo + lead) * 8;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(x), 0, (int)bytes, 0x00020000);
        voff = (int)(g0 * 8) + 16 * t;              // negative = before the stream: out of range, zeros
    };
    // part < 0: all rounds; else the part-th quarter
    auto fetch = [&](__amdgpu_buffer_rsrc_t rsrc, int voff, int part_) __attribute__((always_inline)) {
        constexpr int PER = (NI + NBLK - 1) / NBLK;       // (a share of the next tile's loads per block)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (part_ >= 0 && i / PER != part_) continue;
            int vo = voff + i * (16 * mf::THREADS);
            if ((i + 1) * mf::ROUND > SP && 2 * t + i * mf::ROUND >= SP) vo = 0x7ffff000;   // past the tile: no traffic
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
            pf[i] = __builtin_bit_cast(f32x4, v);
        }
    };

    [[maybe_unused]] auto fetch_one = [&](__amdgpu_buffer_rsrc_t rsrc, int voff, int i) __attribute__((always_inline)) {
        int vo = voff + i * (16 * mf::THREADS);
        if ((i + 1) * mf::ROUND > SP && 2 * t + i * mf::ROUND >= SP) vo = 0x7ffff000;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
        pf[i] = __builtin_bit_cast(f32x4, v);
    };
    const unsigned total_tiles = (unsigned)tiles_per_stream * (unsigned)a.n_streams;
    const unsigned Gd = gridDim.x;
    unsigned cur = blockIdx.x, nxt = cur + Gd;
    auto decode = [&](unsigned id, int &s_, int &b_) __attribute__((always_inline)) {
        b_ = (int)(id / (unsigned)a.n_streams);
        s_ = (int)(id - (unsigned)b_ * (unsigned)a.n_streams);
    };
    int s = 0, bidx = 0;
    decode(cur, s, bidx);
    if (cur < total_tiles) {
        __amdgpu_buffer_rsrc_t rsrc; int voff;
        tile_geom(s, bidx, rsrc, voff);
        fetch(rsrc, voff, -1);
    }
    unsigned nn_q = 0;          // the queue's raw answer (ticket number): valid in lane 0 of the workgroup only
    int voff_cur = 0;
    if (lead) {
        __amdgpu_buffer_rsrc_t rsrc;
        tile_geom(s, bidx, rsrc, voff_cur);
    }

    MF_STAMP_DECL;
    while (cur < total_tiles) {
        int s_nxt, b_nxt;
        MF_STAMP(7);

#if GRHIP_LG_STAGE2
        // (the first two round phasors of the staging below, requested a phase and a barrier ahead of their use)
        f32x2 Sa_first{0.f, 0.f}, Sb_first{0.f, 0.f};
        if (PREMIX && GRHIP_LG_SPF) { Sa_first = f32x2{stab[0], stab[1]}; Sb_first = f32x2{stab[2], stab[3]}; }
#endif
        // ---- block floating point: the tile's largest |component| ---------------------
        {
            // (four running maxima: one would be a chain of 2 NI dependent instructions at the top of every tile)
            float m = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (GRHIP_LG_MAX4 && (i & 1)) {
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m2) : "v"(pf[i][0]), "v"(pf[i][1]));
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m3) : "v"(pf[i][2]), "v"(pf[i][3]));
                } else {
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(pf[i][0]), "v"(pf[i][1]));
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(GRHIP_LG_MAX4 ? m1 : m) : "v"(pf[i][2]), "v"(pf[i][3]));
                }
            }
            if (GRHIP_LG_MAX4) {
                asm("v_max3_f32 %0, %1, %2, %0" : "+v"(m) : "v"(m1), "v"(m2));   // (the same instruction: a non-finite sample is treated
                asm("v_max3_f32 %0, %1, %1, %0" : "+v"(m) : "v"(m3));            // as in the single chain)
            }
            m = wave_max_nonneg(m);
            if (!(m < __builtin_inff())) {          // an Inf / NaN sample: the scale comes from the finite ones (see fir_mfma_rs_kernel)
                float mf = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = __builtin_fabsf(pf[i][e]);
                        mf = av < __builtin_inff() ? __builtin_fmaxf(mf, av) : mf;
                    }
                m = wave_max_nonneg(mf);
            }
            if (lane == 0) wmax[w] = m;
            if (a.sched && t == 0) sched_slot[0] = nn_q + 2u * Gd;    // the queue's answer for the tile after next
        }
        MF_STAMP(0);
        __syncthreads();        // planes free (previous tile's operand reads done), maxima and slot visible
        MF_STAMP(1);
        if (a.sched && cur != blockIdx.x) nxt = sched_slot[0];
        // (wave-uniform by construction; saying so keeps the stream's buffer descriptor in SGPRs --
        // otherwise every load is wrapped in a waterfall loop)
        nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
        decode(nxt, s_nxt, b_nxt);
        float scale, inv_scale;
        {
            const float mt = __builtin_fmaxf(__builtin_fmaxf(wmax[0], wmax[1]), __builtin_fmaxf(wmax[2], wmax[3]));
            int k = 14 - __builtin_amdgcn_frexp_expf(mt);   // |x| e^{jw} components stay below 2^15
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            scale = __builtin_amdgcn_ldexpf(1.0f, k);
            inv_scale = __builtin_amdgcn_ldexpf(1.0f, -k - a.kexp);
        }

        // ---- stage: registers -> (scale, pre-mix, split) -> LDS planes ---------------------
        if (lead && bidx == 0) {
            // the item in front of a stream that does not start on a 16-byte boundary reads as zero
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (voff_cur + i * (16 * mf::THREADS) == 0) { pf[i][0] = 0.f; pf[i][1] = 0.f; }
        }
#if defined(GRHIP_DIAG) && defined(GRHIP_MF_EXP)      // timing experiments of diagnostic builds (wrong results): 2 = no staging,
        const bool exp_off = a.n_out < 0;                  // 1 = no matrix phase / epilogue, 4 = no epilogue; never true at run time
#define MF_EXP_SKIP(bit) if (!((GRHIP_MF_EXP & (bit)) && !exp_off))
#else
#define MF_EXP_SKIP(bit)
#endif
#if GRHIP_LG_STAGE2
        // The pre-mix form's rounds two at a time in ONE hand-ordered block: a round is a chain of ten dependent packed
        // instructions (phasor = lane phasor x round phasor, sample x phasor, convert, split), hipcc 7.2 emits the rounds one
        // after the other with every instruction waiting for the one before it (the register file is full: nothing to
        // interleave with), and with two waves per SIMD the staging phase was a third of a wave's time (stamps,
        // profiles/r03_notes.md).  Here the four chains of two rounds (two samples each) advance in step, so every
        // instruction's operands are four issues old; the round phasors come as scalar pairs.
        if (PREMIX) MF_EXP_SKIP(2)
        {
            const f32x2 ws0 = wl * scale, ws1 = wl1 * scale;
            unsigned char *dst = smem + st_off;
            auto store_round = [&](int i, unsigned rh, unsigned rlo, unsigned ih, unsigned ilo) __attribute__((always_inline)) {
                if ((i + 1) * mf::ROUND <= SP || 2 * t + i * mf::ROUND < SP) {
                    unsigned char *d = dst + i * ST_STEP;
                    *reinterpret_cast<unsigned *>(d) = rh;
                    *reinterpret_cast<unsigned *>(d + PL) = rlo;
                    *reinterpret_cast<unsigned *>(d + 2 * PL) = ih;
                    *reinterpret_cast<unsigned *>(d + 3 * PL) = ilo;
                }
            };
            // cmul(a, b) = v_pk_mul t, a, b [1,1][1,0] neg_lo:[0,1]; v_pk_fma r, a, b, t [0,1,1]   (device_math.h cmul_pk)
#define MF_CM1(t, a, b) "v_pk_mul_f32 " t ", " a ", " b " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
#define MF_CM2(r, a, b, t) "v_pk_fma_f32 " r ", " a ", " b ", " t " op_sel_hi:[0,1,1]\n\t"
            int i = 0;
            // (the round phasors are scalar loads: those of the next pair of rounds are requested before this pair's blocks,
            // or every pair would start with a wait for the scalar cache)
            f32x2 Sa = Sa_first, Sb = Sb_first;
            if (!GRHIP_LG_SPF) { Sa = f32x2{stab[0], stab[1]}; Sb = f32x2{stab[2], stab[3]}; }
#pragma unroll
            for (; i + 1 < NI; i += 2) {
                f32x2 Sa_n = Sa, Sb_n = Sb;
                if (GRHIP_LG_SPF) {
                    if (i + 2 < NI) Sa_n = f32x2{stab[2 * i + 4], stab[2 * i + 5]};
                    if (i + 3 < NI) Sb_n = f32x2{stab[2 * i + 6], stab[2 * i + 7]};
                } else if (i > 0) {
                    Sa = f32x2{stab[2 * i], stab[2 * i + 1]}; Sb = f32x2{stab[2 * i + 2], stab[2 * i + 3]};
                }
                f32x2 a0{pf[i][0], pf[i][1]}, a1{pf[i][2], pf[i][3]}, b0{pf[i + 1][0], pf[i + 1][1]}, b1{pf[i + 1][2], pf[i + 1][3]};
                f32x2 t0, t1, t2, t3, p0, p1, p2, p3;
                unsigned arh, aih, arl, ail, brh, bih, brl, bil;
                asm(MF_CM1("%0", "%12", "%14") MF_CM1("%1", "%13", "%14") MF_CM1("%2", "%12", "%15") MF_CM1("%3", "%13", "%15")
                    MF_CM2("%4", "%12", "%14", "%0") MF_CM2("%5", "%13", "%14", "%1") MF_CM2("%6", "%12", "%15", "%2") MF_CM2("%7", "%13", "%15", "%3")
                    MF_CM1("%0", "%8", "%4") MF_CM1("%1", "%9", "%5") MF_CM1("%2", "%10", "%6") MF_CM1("%3", "%11", "%7")
                    MF_CM2("%8", "%8", "%4", "%0") MF_CM2("%9", "%9", "%5", "%1") MF_CM2("%10", "%10", "%6", "%2") "v_pk_fma_f32 %11, %11, %7, %3 op_sel_hi:[0,1,1]"
                    : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3),
                      "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                    : "v"(ws0), "v"(ws1), "s"(Sa), "s"(Sb));
                // (re, im) of the two samples of each round -> high halves, then low halves, the four registers in step
                asm("v_cvt_pk_f16_f32 %0, %8, %10\n\t"
                    "v_cvt_pk_f16_f32 %1, %9, %11\n\t"
                    "v_cvt_pk_f16_f32 %4, %12, %14\n\t"
                    "v_cvt_pk_f16_f32 %5, %13, %15\n\t"
                    "v_fma_mixlo_f16 %2, %8, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixlo_f16 %3, %9, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixlo_f16 %6, %12, 1.0, -%4 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixlo_f16 %7, %13, 1.0, -%5 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixhi_f16 %2, %10, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixhi_f16 %3, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixhi_f16 %6, %14, 1.0, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixhi_f16 %7, %15, 1.0, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                    : "=&v"(arh), "=&v"(aih), "=&v"(arl), "=&v"(ail), "=&v"(brh), "=&v"(bih), "=&v"(brl), "=&v"(bil)
                    : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(b0.x), "v"(b0.y), "v"(b1.x), "v"(b1.y));
                store_round(i, arh, arl, aih, ail);
                store_round(i + 1, brh, brl, bih, bil);
                if (GRHIP_LG_SPF) { Sa = Sa_n; Sb = Sb_n; }
            }
            if (i < NI) {                          // the odd round out: the two chains of one round
                const f32x2 S = GRHIP_LG_SPF ? Sa : f32x2{stab[2 * i], stab[2 * i + 1]};
                f32x2 e0{pf[i][0], pf[i][1]}, e1{pf[i][2], pf[i][3]};
                e0 = cmul_pk(e0, cmul_pk(ws0, S));
                e1 = cmul_pk(e1, cmul_pk(ws1, S));
                const f32x2 re{e0.x, e1.x}, im{e0.y, e1.y};
                const h16x2 rh = __builtin_convertvector(re, h16x2), ih = __builtin_convertvector(im, h16x2);
                const h16x2 rlo = split_lo(re, rh), ilo = split_lo(im, ih);
                store_round(i, __builtin_bit_cast(unsigned, rh), __builtin_bit_cast(unsigned, rlo), __builtin_bit_cast(unsigned, ih),
                            __builtin_bit_cast(unsigned, ilo));
            }
#undef MF_CM1
#undef MF_CM2
        }
        else
#endif
        MF_EXP_SKIP(2)
        {
            const f32x2 ws0 = wl * scale, ws1 = wl1 * scale;
            unsigned char *dst = smem + st_off;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x2 e0{pf[i][0], pf[i][1]}, e1{pf[i][2], pf[i][3]};
                if (PREMIX) {
#if defined(GRHIP_DIAG) && defined(GRHIP_PROBE) && (GRHIP_PROBE & 2)
                    // attribution probe: every sample's phasor from double precision instead of the product of two
                    // binary32 phasors
                    const double ang0 = a.omega * (double)(2 * t + mf::ROUND * i - a.off);
                    const f32x2 p0{(float)cos(ang0) * scale, (float)sin(ang0) * scale};
                    const f32x2 p1{(float)cos(ang0 + a.omega) * scale, (float)sin(ang0 + a.omega) * scale};
                    e0 = cmul_pk(e0, p0);
                    e1 = cmul_pk(e1, p1);
#else
                    const f32x2 S{stab[2 * i], stab[2 * i + 1]};
                    e0 = cmul_pk(e0, cmul_pk(ws0, S));
                    e1 = cmul_pk(e1, cmul_pk(ws1, S));
#endif
                } else {
                    e0 = e0 * scale;
                    e1 = e1 * scale;
                }
                const f32x2 re{e0.x, e1.x}, im{e0.y, e1.y};
                const h16x2 rh = __builtin_convertvector(re, h16x2), ih = __builtin_convertvector(im, h16x2);
                const h16x2 rlo = split_lo(re, rh), ilo = split_lo(im, ih);
                if ((i + 1) * mf::ROUND <= SP || 2 * t + i * mf::ROUND < SP) {
#if GRHIP_LG_W2
                    // two stores per round, each to two planes a multiple of 256 bytes apart (round 3: the LDS store path
                    // takes 2 cycles per source register: 6 + 6 instead of 4 x 4 per round)
                    const int ad = st_off + i * ST_STEP;            // LDS byte address (dynamic LDS starts at 0)
                    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:%5\n\t"
                                 "ds_write2st64_b32 %0, %3, %4 offset0:%6 offset1:%7"
                                 :: "v"(ad), "v"(rh), "v"(rlo), "v"(ih), "v"(ilo),
                                    "n"(PL / 256), "n"(2 * (PL / 256)), "n"(3 * (PL / 256)) : "memory");
#else
                    unsigned char *d = dst + i * ST_STEP;
                    *reinterpret_cast<h16x2 *>(d) = rh;
                    *reinterpret_cast<h16x2 *>(d + PL) = rlo;
                    *reinterpret_cast<h16x2 *>(d + 2 * PL) = ih;
                    *reinterpret_cast<h16x2 *>(d + 3 * PL) = ilo;
#endif
                }
            }
        }
        // the carry of the previous call, for the stream's first tile (frame of the composite FIR
        // output, fir_kernels.h), brought into this tile's frame and scale
        float2 ypf = make_float2(0.f, 0.f);
        if (DEMOD && bidx == 0 && a.y_prev) {
            const float2 yp = a.y_prev[s];
            const float2 vm = a.vtab[BLK - 1];
            const float2 q = cmul_fma(yp, make_float2(vm.x, -vm.y));
            const float sc2 = __builtin_amdgcn_ldexpf(scale, a.kexp);
            ypf = make_float2(q.x * sc2, q.y * sc2);
        }
        MF_STAMP(2);
        __syncthreads();
        MF_STAMP(3);

        // next tile's HBM traffic flies under this tile's matrix phase and epilogue
        __amdgpu_buffer_rsrc_t rsrc_n; int voff_n;
        tile_geom(s_nxt, b_nxt, rsrc_n, voff_n);
        if (nxt >= total_tiles) voff_n = 0x7ffff000 - NI * 16 * mf::THREADS;     // out of range: zeros, no traffic
        // (the ticket is used as it comes back, a tile later: arithmetic on it here would make wave 0 wait for the round trip)
        if (a.sched && t == 0) nn_q = atomicAdd(a.sched, 1u);

        // output of this tile's stream, through a buffer descriptor: a store whose offset is out of
        // range (past the end of the stream; or made so for the lanes that own no output) is dropped
        // by the hardware, so the epilogue has no branches and can be interleaved with the MFMAs
        constexpr int OOB = 0x7ffffff0;
        __amdgpu_buffer_rsrc_t orsrc, grsrc;
        if (DEMOD) {
            orsrc = __builtin_amdgcn_make_buffer_rsrc(a.d_out + (long long)s * a.d_stride, 0, (int)(a.n_out * 4), 0x00020000);
        } else {
            orsrc = __builtin_amdgcn_make_buffer_rsrc(a.y_out + (long long)s * a.y_stride, 0, (int)(a.n_out * 8), 0x00020000);
            if (ROT) grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.gtab), 0, (int)(a.n_out * 8), 0x00020000);
        }
        // tile-local output index of this lane's first output of block b: jt + 16 b
        const int jt = w * mf::WAVE_NEW + rsl * mf::SEG_OUT + 2 * r8;
        const int n_base = bidx * NTE - BLK + jt;          // stream index of that output (block 0); < 2^28
        const bool first_of_stream = bidx == 0 && t == 7;   // lane (segment 0, rows 14/15) of wave 0
        float ypendx = 0.f, ypendy = 0.f;           // first output of the segment, waits for its predecessor
        float d1_pend = 0.f;
        float y1x = 0.f, y1y = 0.f;
        const float ypfx = ypf.x, ypfy = ypf.y;

        // one 16-output block: two consecutive outputs per lane (v: re0, im0, re1, im1) -> demodulator / store
        // (plain floats for everything that is selected per lane: a select between members of two float2 objects
        // becomes a select between their ADDRESSES, which pins both -- and every variable captured next to them --
        // in scratch memory)
        auto epilogue = [&](int b, const f32x4 &v) __attribute__((always_inline)) {
            const float y0x = v[0], y0y = v[1];
            const float y1px = y1x, y1py = y1y;          // previous block's second output
            y1x = v[2]; y1y = v[3];
            const float2 y0 = make_float2(y0x, y0y);
            const int n = n_base + BLK * b;
            const bool own = !(b == 0 && rsl == 0) && n >= 0;       // the wave's overlap block stores nothing
            if (DEMOD) {
                if (b == 0) {
                    y1x = first_of_stream ? ypfx : y1x;
                    y1y = first_of_stream ? ypfy : y1y;
                }
                // predecessor of y0: the lane before (same segment), or the previous block's last output
                float px = dpp_row<0x111>(y1x), py = dpp_row<0x111>(y1y);      // row_shr:1
                if (b > 0) {
                    const float cx = dpp_row<0x107>(y1px), cy = dpp_row<0x107>(y1py);   // row_shl:7: lane + 7
                    px = r8 == 0 ? cx : px;
                    py = r8 == 0 ? cy : py;
                } else {
                    ypendx = y0x; ypendy = y0y;
                }
                const float2 prev = make_float2(px, py), y1 = make_float2(y1x, y1y);
                const float d0 = GRHIP_LG_LEAN ? quad_demod_lean(y0, prev, a.gain, s_atan) : quad_demod_fast(y0, prev, a.gain, s_atan);
                const float d1 = GRHIP_LG_LEAN ? quad_demod_lean(y1, y0, a.gain, s_atan) : quad_demod_fast(y1, y0, a.gain, s_atan);
                if (b == 0) d1_pend = d1;
                const bool st_ok = own && !(b == 0 && r8 == 0);     // (a segment's first pair waits for its predecessor)
                const f32x2 dd{d0, d1};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, dd), orsrc, st_ok ? 4 * n : OOB, 0, 0);
            } else {
                float2 o0 = y0, o1 = make_float2(y1x, y1y);
                if (PREMIX) {
                    const float4 vv = *reinterpret_cast<const float4 *>(a.vtab + jt + BLK * b);   // e^{-jw j D}
                    o0 = cmul_fma(o0, make_float2(vv.x, vv.y));
                    o1 = cmul_fma(o1, make_float2(vv.z, vv.w));
                }
                o0.x *= inv_scale; o0.y *= inv_scale; o1.x *= inv_scale; o1.y *= inv_scale;
                if (ROT) {
                    const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(grsrc, own ? 8 * n : OOB, 0, 0);
                    const f32x4 gq = __builtin_bit_cast(f32x4, gv);
                    o0 = cmul_ref(o0, make_float2(gq[0], gq[1]));                   // gr_rotator: z = in * d_phase
                    o1 = cmul_ref(o1, make_float2(gq[2], gq[3]));
                }
                const f32x2 oa{o0.x, o0.y}, ob{o1.x, o1.y};
                const int so = own ? 8 * n : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oa), orsrc, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), orsrc, so + 8, 0, 0);
            }
        };

        // ---- matrix phase, block by block (30 MFMAs each) ----
        // Operand chunks are read one chunk ahead of their use (an LDS round trip is ~100 cycles).
        auto chunk_ptr = [&](int c) __attribute__((always_inline)) {
            const int u = rd_u + mf::CHUNK * c;
            return smem + rd_plane + 2 * u + 32 * (u >> LOGQ);
        };
        f32x4 acc[NBLK];
#if defined(GRHIP_DIAG) && defined(GRHIP_MF_EXP) && (GRHIP_MF_EXP & 1)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch(rsrc_n, voff_n, q);
        if (exp_off)
#endif
        {
        h16x8 Bh_n = *reinterpret_cast<const h16x8 *>(chunk_ptr(0));
        h16x8 Bl_n = *reinterpret_cast<const h16x8 *>(chunk_ptr(0) + PL);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
#if GRHIP_LG_ACC3
            // three accumulator tiles per block, added once at its end: the high-half products of the even and of the
            // odd k-steps and the two low-half products.  An output then sees a third of the f32 accumulation roundings
            // at about half the partial-sum magnitude: the FAST demodulator's per-element deviation from the reference
            // went from 2.40e-5 to 1.74e-5 on cfg2 (round 3)
            f32x4 m0{0.f, 0.f, 0.f, 0.f}, m1{0.f, 0.f, 0.f, 0.f}, lo{0.f, 0.f, 0.f, 0.f};
#else
            acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            f32x4 tq{0.f, 0.f, 0.f, 0.f};
            // The k-steps of a block are taken from the band's ends towards its middle (0, KS-1, 1, KS-2, ...): a low-pass
            // has its large taps in the middle, so the partial sum stays small until the last two or three k-steps and
            // only those accumulations round at the output's magnitude (in index order every accumulation from the
            // middle on does).  Same MFMAs, same registers; the FAST demodulator's largest per-element deviation from
            // the reference on cfg2: 2.40e-5 in index order (round 3).
            auto kstep = [](int q) constexpr { return GRHIP_LG_ORDER ? ((q & 1) ? KS - 1 - (q >> 1) : (q >> 1)) : q; };
            h16x8 Gcur{}, Gnext{};
#pragma unroll
            for (int jq = 0; jq < KS; ++jq) {
                const int j = kstep(jq);
                const h16x8 Bh = Bh_n, Bl = Bl_n;
                if (TAPQ) {
                    // the correction band's operand of the NEXT k-step that needs one, a k-step ahead of its MFMA (read at the
                    // point of use it cost an LDS round trip per middle k-step: 1.44 against 1.22 ms).  The address goes through
                    // an empty asm: a plain loop-invariant LDS read is hoisted out of the tile loop, and G is back in sixteen
                    // registers the kernel does not have.
                    Gcur = Gnext;
                    const int jn = jq + 1 < KS ? kstep(jq + 1) : (b + 1 < NBLK ? kstep(0) : -1);
                    if (jn >= JG0 && jn < JG0 + mf::NG) {
                        // (an LDS OFFSET goes through the asm and comes back as an LDS pointer: a laundered generic pointer
                        // is read with flat_load, whose wait is vmcnt(0) -- it drained the tile prefetch: 1.38 against 1.21 ms)
                        unsigned goff = (unsigned)(G::OFF_G + 16 * lane + (jn - JG0) * 1024);
                        asm volatile("" : "+v"(goff));
                        Gnext = *reinterpret_cast<const __attribute__((address_space(3))) h16x8 *>(goff);
                    }
                }
                const int cn = jq + 1 < KS ? CB * b + kstep(jq + 1) : CB * (b + 1) + kstep(0);       // next chunk: this block's, or the next block's first
                if (jq + 1 < KS || b + 1 < NBLK) {
                    const unsigned char *src = chunk_ptr(cn);
                    Bh_n = *reinterpret_cast<const h16x8 *>(src);
                    Bl_n = *reinterpret_cast<const h16x8 *>(src + PL);
                }
#if GRHIP_LG_ACC3
                // (the even / odd split only where the registers hold it without spilling: the headline shape and the short bands)
                constexpr bool EVEN_ODD = GRHIP_LG_ACC3 == 1 && (KS <= 6 || (D == 4 && EPI == EPI_DEMOD));
                if (EVEN_ODD && (j & 1)) m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bh, m1, 0, 0, 0);
                else m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bh, m0, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bl, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[j], Bh, lo, 0, 0, 0);
#else
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bh, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bl, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[j], Bh, acc[b], 0, 0, 0);
#endif
                if (TAPQ && j >= JG0 && j < JG0 + mf::NG) {
                    // the other part's chunk of the same segment sits in the lane 8 places round the row of 16 (column ^ 8)
                    const u32x4 bw = __builtin_bit_cast(u32x4, Bh);
                    u32x4 sw;
#pragma unroll
                    for (int e = 0; e < 4; ++e)          // row_ror:8 (every lane has a source: `old` is never used, and naming the source saves a move of zero)
                        sw[e] = (unsigned)__builtin_amdgcn_update_dpp((int)bw[e], (int)bw[e], 0x128, 0xf, 0xf, false);
                    tq = __builtin_amdgcn_mfma_f32_16x16x32_f16(Gcur, __builtin_bit_cast(h16x8, sw), tq, 0, 0, 0);
                }
#if GRHIP_LG_SPREAD
                {   // one of the next tile's loads behind every second k-step (instead of a quarter of them per block)
                    constexpr int STEP = (NBLK * KS) / NI > 0 ? (NBLK * KS) / NI : 1;
                    const int pos = b * KS + jq;
                    if (pos % STEP == 0 && pos / STEP < NI) fetch_one(rsrc_n, voff_n, pos / STEP);
                }
#else
                if (jq == 1) fetch(rsrc_n, voff_n, b);       // a quarter of the next tile's loads per block
#endif
            }
#if GRHIP_LG_ACC3
            acc[b] = GRHIP_LG_ACC3 == 1 ? (m0 + m1) + lo : m0 + lo;
#endif
            if (TAPQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[b][e] = __builtin_fmaf(tq_sgn, tq[e], acc[b][e]);
            }
        }
        }
        MF_STAMP(4);
        MF_EXP_SKIP(5)
        {
        // The epilogues run BEHIND the matrix phase, fenced off from it.  Letting hipcc 7.2 interleave
        // them with the MFMAs of the following block (it does so eagerly: the code is branch-free) bought
        // nothing (1.26 ms either way, same box) and, in the rotate epilogue, gave one 16-output block in
        // thirty a stale real part (bit-reproducible per binary; gone with this fence, with a fence in
        // front of the scratch stores only, and with 16 s_nop behind the output store).  Two suspects are
        // ruled out by micro-tests kept in tools/dbg/: a vector write to an MFMA source register right
        // behind the MFMA (mfma_war.hip: harmless), and too few wait states between the MFMA and the LDS
        // store of its result (mfma_lds_raw*.hip: 6 suffice, alone or beside an MFMA-issuing partner wave;
        // hipcc pads 8).  Unresolved; the fence costs nothing.
        __builtin_amdgcn_sched_barrier(0);
        // all four accumulator tiles through the wave's own stretch of the planes in ONE round trip: accumulator layout
        // in ([segment][row][re, im]), two consecutive complex outputs per lane out.  (LDS executes a wave's operations
        // in order; the wavefront-scope fences order the compiler and emit nothing.)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            float *sb = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(scw) + b * PL);
#pragma unroll
            for (int i = 0; i < 4; ++i) sb[sc_wr + 2 * i] = acc[b][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // block b+1 is read back while block b is demodulated (all four at once would not fit the register budget)
        auto read_block = [&](int b) __attribute__((always_inline)) {
            return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(scw) + b * PL + 4 * sc_rd);
        };
        f32x4 yv_n = read_block(0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const f32x4 yv = yv_n;
            if (b + 1 < NBLK) yv_n = read_block(b + 1);
            epilogue(b, yv);
            // (GRHIP_LG_EPI2: the scheduler may interleave the blocks of a pair -- four demodulator chains in step instead of two)
            if (!GRHIP_LG_EPI2 || (b & 1) || !DEMOD) __builtin_amdgcn_sched_barrier(0);
        }
        if (DEMOD) {
            // first output of every segment but the wave's first: predecessor = last output of the
            // segment before, i.e. the second output of the lane before, after the last block
            const float px = __shfl_up(y1x, 1), py = __shfl_up(y1y, 1);
            const float dp = GRHIP_LG_LEAN ? quad_demod_lean(make_float2(ypendx, ypendy), make_float2(px, py), a.gain, s_atan)
                                           : quad_demod_fast(make_float2(ypendx, ypendy), make_float2(px, py), a.gain, s_atan);
            const bool st_ok = r8 == 0 && rsl != 0 && n_base >= 0;
            const f32x2 dd{dp, d1_pend};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, dd), orsrc, st_ok ? 4 * n_base : OOB, 0, 0);
            // carry for the next call: the composite FIR output of the stream's last output, computed
            // directly (f32, composite taps) by the first wave of the workgroup that owns the last tile
            if (a.y_last && bidx == tiles_per_stream - 1 && w == 0) {
                const long long item0 = (a.n_out - 1) * D - a.n_lo + lead;       // relative to the stream's descriptor
                __amdgpu_buffer_rsrc_t xr; int vdummy;
                tile_geom(s, bidx, xr, vdummy);
                float sx = 0.f, sy = 0.f;
                for (int i = lane; i < a.T; i += 64) {
                    const long long it = item0 + i;
                    const int vo = (it < 0 || (lead && it == 0)) ? OOB : (int)(it * 8);
                    const u32x2 xv = __builtin_amdgcn_raw_buffer_load_b64(xr, vo, 0, 0);
                    const f32x2 xf = __builtin_bit_cast(f32x2, xv);
                    const float2 c = a.ctaps[i];
                    sx = __builtin_fmaf(c.x, xf.x, sx); sx = __builtin_fmaf(-c.y, xf.y, sx);
                    sy = __builtin_fmaf(c.x, xf.y, sy); sy = __builtin_fmaf(c.y, xf.x, sy);
                }
                sx = wave_sum_f(sx); sy = wave_sum_f(sy);
                if (lane == 0) a.y_last[s] = make_float2(sx, sy);
            }
        }

        }
        MF_STAMP(5);
        cur = nxt;
        if (!a.sched) nxt = nxt + Gd;
        s = s_nxt; bidx = b_nxt;
        voff_cur = voff_n;
    }
    // the last workgroup out re-arms the queue for the next launch
    if (a.sched && t == 0) {
        __threadfence();
        if (atomicAdd(a.sched + 1, 1u) == Gd - 1) {
            a.sched[0] = 0;
            a.sched[1] = 0;
            __threadfence();
        }
    }
#ifdef GRHIP_STAMP
    if (lane == 0 && g_mf_stamp_buf) {
        unsigned long long *o = g_mf_stamp_buf + ((size_t)blockIdx.x * mf::WAVES + w) * 10;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
        o[8] = st_begin; o[9] = mf_stamp_now();
    }
#endif
}

#ifdef GRHIP_STAMP
// debug hook (not part of the ABI): where the stamp sums go; needs 10 u64 per wave
extern "C" __attribute__((visibility("default"))) int grdbg_set_stamp_buffer_mfma(void *d_buf)
{
    unsigned long long *p = (unsigned long long *)d_buf;
    GRHIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mf_stamp_buf), &p, sizeof(p)));
    return GRHIP_OK;
}
#endif


#if defined(GRHIP_DIAG) && GRHIP_MF_NBLK == 4       // (diagnostic builds only: see launch_mfma_inst)
// =================================================================================================
// fir_mfma_rs_kernel -- the same engine with the waves of a workgroup in two ROLES (round 3)
// =================================================================================================
// What bounded the kernel above at 0.59 of the HBM peak was bytes in flight: a wave's
// 17 loads land in the registers it stages from, so they fly for only half a tile period, and the
// period is the SUM of a wave's phases (stage, matrix, epilogue) beside one partner wave doing the
// same.  Here one 768-lane workgroup per CU splits the work by role:
//   * 8 STAGER waves (two per SIMD): loads two tiles ahead into TWO register sets (9 x 16 bytes per
//     lane each, so a tile's loads fly for about one and a half periods), block-floating-point
//     maximum, scale / pre-mix / split into the binary16 planes, and the epilogue (demodulator,
//     stores) of the tile the matrix waves finished a period ago;
//   * 4 MATRIX waves (one per SIMD): operand reads and MFMAs only, accumulator tiles back into
//     their own stretch of the planes.  The band matrix (80 VGPRs) lives only here.
// The planes are double buffered (2 x 4 x 17.3 KB); the two roles meet at two barriers per period:
//     period p    stagers                                         matrix waves
//     1st half    loads of tile p;  epilogue of tile p-3;         blocks 0, 1 of tile p-2
//                 maximum of tile p-1
//     -- barrier (maxima visible; the epilogue has left buffer (p-1)&1) --
//     2nd half    stage tile p-1 into buffer (p-1)&1              blocks 2, 3; accumulators -> buffer p&1
//     -- barrier --
// A SIMD then holds one MFMA stream and two vector streams instead of two waves that alternate
// between both.  Every block's epilogue is independent: the demodulator's predecessor of a
// segment's first output is read from the accumulator tiles in LDS (previous block, or the last
// block of the segment before) instead of being carried from block to block.
// Accuracy: a block's 30 MFMAs go to three accumulators (Ah Xh of even / odd k-steps, the two
// low-half products) that are added once at the end, so an output sees a third of the f32
// accumulation roundings at half the partial-sum magnitude.
// Diagnostic builds only (-DGRHIP_RS_ABL=mask, wrong results): 1 = no operand reads / MFMAs, 2 = no epilogue,
// 4 = no conversion / plane stores, 8 = no tile loads, 16 = no block-floating-point maximum.  Which unit pins the period?
#ifndef GRHIP_RS_ABL
#define GRHIP_RS_ABL 0
#endif
namespace rs {
constexpr int NSTG = 8, NMAT = mf::WAVES;
constexpr int THREADS = 64 * (NSTG + NMAT);
constexpr int STG_T = 64 * NSTG;
constexpr int ROUND = 2 * STG_T;        // samples per staging round of the stagers
#ifndef GRHIP_RS_PD
#define GRHIP_RS_PD 2
#endif
constexpr int PD = GRHIP_RS_PD;         // operand chunks read ahead of their MFMAs
static_assert(mf::NBLK == 4 && NSTG == 2 * NMAT, "two stagers share a matrix wave's four blocks");
}  // namespace rs

namespace {
template <int D, int KS> struct GeoRS {
    using G = Geo<D, KS>;
    static constexpr int NI = (G::SP + rs::ROUND - 1) / rs::ROUND;
    static constexpr int BUF = 4 * G::PL;
    static constexpr int OFF_ATAN = 2 * BUF;
    static constexpr int OFF_MISC = OFF_ATAN + 256 * 8;
    static constexpr int LDS = OFF_MISC + 64;
    static_assert(rs::ROUND % G::Q == 0, "a staging round must cover whole segment strides");
    static_assert(2 * (NI - 1) < G::NI, "the stagers' round phasors are every other entry of the 512-sample table");
    static_assert(LDS <= 160 * 1024, "LDS");
};

// LDS writes of this wave done, then the workgroup's barrier.  Not __syncthreads(): its fences may
// wait for the vector-memory counter, which would drain the loads that are meant to stay in flight.
__device__ __forceinline__ void rs_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
}  // namespace

template <int D, int KS, bool PREMIX, int EPI>
__global__ void __launch_bounds__(rs::THREADS, 1) fir_mfma_rs_kernel(const FirMfmaArgs a)
{
    using G = Geo<D, KS>;
    using R = GeoRS<D, KS>;
    constexpr int NI = R::NI, PL = G::PL, LOGQ = G::LOGQ, CB = G::CB, SP = G::SP;
    constexpr bool DEMOD = EPI == EPI_DEMOD;
    constexpr bool ROT = EPI == EPI_ROTATE;
    static_assert(!DEMOD || PREMIX, "the fused demodulator belongs to the pre-mix form");
    constexpr int NTE = mf::NTE, BLK = mf::BLK, NBLK = mf::NBLK;

    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const AtanPairs s_atan{reinterpret_cast<const f32x2 *>(smem + R::OFF_ATAN)};
    float *wmax = reinterpret_cast<float *>(smem + R::OFF_MISC);

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_per_stream = (int)((a.n_out + NTE - 1) / NTE);
    const unsigned total_tiles = (unsigned)tiles_per_stream * (unsigned)a.n_streams;
    const unsigned Gd = gridDim.x;
    // tiles blockIdx.x, blockIdx.x + Gd, ...: q-th tile of this workgroup = blockIdx.x + q Gd
    const int n_my = (int)((total_tiles - blockIdx.x + Gd - 1) / Gd);
    const int P = n_my + 3;

    // ---- shared by both roles ----
    const int ts = t;                       // stager thread index (0 .. STG_T - 1 in the stager waves)
    constexpr int OOB = 0x7ffffff0;
    const int lead = a.off ^ (int)(a.n_lo & 1);
    int *kring = reinterpret_cast<int *>(smem + R::OFF_MISC + 32);      // block-floating-point exponent of tile q at [q & 3]
    auto decode = [&](int q, int &s_, int &b_) __attribute__((always_inline)) {
        const unsigned id = blockIdx.x + (unsigned)q * Gd;
        b_ = (int)(id / (unsigned)a.n_streams);
        s_ = (int)(id - (unsigned)b_ * (unsigned)a.n_streams);
    };
    auto tile_geom = [&](int s, int b, __amdgpu_buffer_rsrc_t &rsrc, int &voff) __attribute__((always_inline)) {
        const long long g0 = ((long long)b * NTE - BLK) * D - a.off - a.n_lo + lead;   // tile start relative to the descriptor
        const float2 *x = a.x + (long long)s * a.x_stride + a.n_lo - lead;
        const long long bytes = (a.n_in - a.n_lo + lead) * 8;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(x), 0, (int)bytes, 0x00020000);
        voff = (int)(g0 * 8) + 16 * ts;             // negative = before the stream: out of range, zeros
    };
    // ---- E: one 16-output block of a matrix wave's accumulator tiles (buffer q & 1): demodulator / rotator, stores.
    // Blocks 2 and 3 of a matrix wave are finished by that wave itself right behind its accumulator stores (it would
    // otherwise wait at the barrier); blocks 0 and 1 by the two stagers that belong to it, in the next period.
    const int rsl = lane >> 3, r8 = lane & 7;
    struct EpiRegs { f32x4 yv; f32x2 pv; };
    auto epilogue_read = [&](int q, int mw, int b, EpiRegs &er) __attribute__((always_inline)) {
        const int scw_u = mw * (mf::WAVE_NEW * D) + G::HALO;
        const unsigned char *scr = smem + (q & 1) * R::BUF + 2 * scw_u + 32 * (scw_u >> LOGQ);
        er.yv = *reinterpret_cast<const f32x4 *>(scr + b * PL + 4 * (rsl * SCR_SEG + 4 * r8));
        er.pv = f32x2{0.f, 0.f};
        if (DEMOD) {
            // predecessor of a segment's first output of the block: row 15 of the block before, or -- block 0 --
            // of the last block of the segment before (the wave's first segment has none: its block 0 is overlap)
            const int pb = b == 0 ? NBLK - 1 : b - 1;
            const int ps = b == 0 ? (rsl > 0 ? rsl - 1 : 0) : rsl;
            er.pv = *reinterpret_cast<const f32x2 *>(scr + pb * PL + 4 * (ps * SCR_SEG + 30));
        }
    };
    auto epilogue_block = [&](int q, int kx, int mw, int b, bool carry_out, const EpiRegs &er) __attribute__((always_inline)) {
        int s, bidx;
        decode(q, s, bidx);
        const float inv_scale = __builtin_amdgcn_ldexpf(1.0f, -kx - a.kexp);
        // the carry of the previous call, for the stream's first tile (frame of the composite FIR output,
        // fir_kernels.h), brought into this tile's frame and scale
        float ypfx = 0.f, ypfy = 0.f;
        if (DEMOD && bidx == 0 && b == 1 && a.y_prev) {
            const float2 yp = a.y_prev[s];
            const float2 vm = a.vtab[BLK - 1];
            const float2 qq = cmul_fma(yp, make_float2(vm.x, -vm.y));
            const float sc2 = __builtin_amdgcn_ldexpf(1.0f, kx + a.kexp);
            ypfx = qq.x * sc2; ypfy = qq.y * sc2;
        }
        __amdgpu_buffer_rsrc_t orsrc, grsrc;
        if (DEMOD) {
            orsrc = __builtin_amdgcn_make_buffer_rsrc(a.d_out + (long long)s * a.d_stride, 0, (int)(a.n_out * 4), 0x00020000);
        } else {
            orsrc = __builtin_amdgcn_make_buffer_rsrc(a.y_out + (long long)s * a.y_stride, 0, (int)(a.n_out * 8), 0x00020000);
            if (ROT) grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.gtab), 0, (int)(a.n_out * 8), 0x00020000);
        }
        const int jt = mw * mf::WAVE_NEW + rsl * mf::SEG_OUT + 2 * r8;       // tile-local index of the lane's first output of block 0
        const int n_base = bidx * NTE - BLK + jt;                          // its stream index; < 2^28
        {
            const float y0x = er.yv[0], y0y = er.yv[1], y1x = er.yv[2], y1y = er.yv[3];
            const int n = n_base + BLK * b;
            const bool own = !(b == 0 && rsl == 0) && n >= 0;       // the wave's overlap block stores nothing
            if (DEMOD) {
                float px = dpp_row<0x111>(y1x), py = dpp_row<0x111>(y1y);      // row_shr:1: the lane before, same segment
                const bool from_carry = bidx == 0 && mw == 0 && b == 1 && rsl == 0;    // output 0 of the stream
                const float qx = from_carry ? ypfx : er.pv[0], qy = from_carry ? ypfy : er.pv[1];
                px = r8 == 0 ? qx : px;
                py = r8 == 0 ? qy : py;
                const float d0 = quad_demod_fast(make_float2(y0x, y0y), make_float2(px, py), a.gain, s_atan);
                const float d1 = quad_demod_fast(make_float2(y1x, y1y), make_float2(y0x, y0y), a.gain, s_atan);
                const f32x2 dd{d0, d1};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, dd), orsrc, own ? 4 * n : OOB, 0, 0);
            } else {
                float2 o0 = make_float2(y0x, y0y), o1 = make_float2(y1x, y1y);
                if (PREMIX) {
                    const float4 vv = *reinterpret_cast<const float4 *>(a.vtab + jt + BLK * b);   // e^{-jw j D}
                    o0 = cmul_fma(o0, make_float2(vv.x, vv.y));
                    o1 = cmul_fma(o1, make_float2(vv.z, vv.w));
                }
                o0.x *= inv_scale; o0.y *= inv_scale; o1.x *= inv_scale; o1.y *= inv_scale;
                if (ROT) {
                    const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(grsrc, own ? 8 * n : OOB, 0, 0);
                    const f32x4 gq = __builtin_bit_cast(f32x4, gv);
                    o0 = cmul_ref(o0, make_float2(gq[0], gq[1]));                   // gr_rotator: z = in * d_phase
                    o1 = cmul_ref(o1, make_float2(gq[2], gq[3]));
                }
                const f32x2 oa{o0.x, o0.y}, ob{o1.x, o1.y};
                const int so = own ? 8 * n : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oa), orsrc, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), orsrc, so + 8, 0, 0);
            }
        }
        // carry for the next call: the composite FIR output of the stream's last output, computed directly (f32,
        // composite taps) by the wave the caller names, for the stream's last tile
        if (carry_out && DEMOD && a.y_last && bidx == tiles_per_stream - 1) {
            const long long item0 = (a.n_out - 1) * D - a.n_lo + lead;       // relative to the stream's descriptor
            __amdgpu_buffer_rsrc_t xr; int vdummy;
            tile_geom(s, bidx, xr, vdummy);
            float sx = 0.f, sy = 0.f;
            for (int i = lane; i < a.T; i += 64) {
                const long long it = item0 + i;
                const int vo = (it < 0 || (lead && it == 0)) ? OOB : (int)(it * 8);
                const u32x2 xv = __builtin_amdgcn_raw_buffer_load_b64(xr, vo, 0, 0);
                const f32x2 xf = __builtin_bit_cast(f32x2, xv);
                const float2 c = a.ctaps[i];
                sx = __builtin_fmaf(c.x, xf.x, sx); sx = __builtin_fmaf(-c.y, xf.y, sx);
                sy = __builtin_fmaf(c.x, xf.y, sy); sy = __builtin_fmaf(c.y, xf.x, sy);
            }
            sx = wave_sum_f(sx); sy = wave_sum_f(sy);
            if (lane == 0) a.y_last[s] = make_float2(sx, sy);
        }
    };

    if (w >= rs::NSTG) {
        // =============================== matrix role ===============================
        const int mw = w - rs::NSTG;
#if defined(GRHIP_RS_PRIO) && GRHIP_RS_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        h16x8 Ah[KS], Al[KS];
        {
            const h16x8 *Ag = reinterpret_cast<const h16x8 *>(a.A);
#pragma unroll
            for (int js = 0; js < KS; ++js) {
                Ah[js] = Ag[(js * 2 + 0) * 64 + lane];
                Al[js] = Ag[(js * 2 + 1) * 64 + lane];
            }
        }
        // operand reads: lane = (column, k-group g); column = part * 8 + segment
        const int col = lane & 15, part = col >> 3, sl = col & 7, g = lane >> 4;
        const int wu = mw * (mf::WAVE_NEW * D);                         // first sample of the wave's range (a multiple of 32)
        // byte of sample wu + 32 c + (sl << LOGQ) + 8 g: the lane's part and the chunk's part add up without a carry
        // into the skew term ((wu + 32 c) mod Q is a multiple of 32 and 8 g < 32)
        const int rd_lane = part * 2 * PL + 2 * ((sl << LOGQ) + 8 * g) + 32 * sl;
        auto chunk_off = [&](int c) __attribute__((always_inline)) {
            const int u = wu + mf::CHUNK * c;
            return 2 * u + 32 * (u >> LOGQ);
        };
        const int scw_u = wu + G::HALO;
        const int scw_off = 2 * scw_u + 32 * (scw_u >> LOGQ);             // + b * PL for block b
        const int sc_wr = sl * SCR_SEG + 8 * g + part;                  // + 2 i

        MF_STAMP_DECL;
        constexpr int NCH = G::NCH;                   // operand chunks of a segment; chunk c is k-step c - CB b of block b
        constexpr int PDK = KS > 6 ? 1 : rs::PD;     // (long band: A takes 80 registers, one chunk ahead is what fits beside the wave's share of the epilogue)
        for (int p = 0; p < P; ++p) {
            const bool act = p >= 2 && p - 2 < n_my;
            unsigned char *buf = smem + (p & 1) * R::BUF;
            // CHUNK-major: every chunk is read once (32 instead of 80 16-byte reads per tile) and multiplied into every
            // block whose band covers it, PDK chunks ahead of its use.  The order is pinned with sched_barriers:
            // left alone, hipcc 7.2 merges the block-major form's repeated reads by itself but issues each read right
            // in front of its first MFMA, an LDS round trip per chunk (1.2 us of matrix phase per half period instead of 0.5).
            f32x4 m0[NBLK], m1[NBLK], lo[NBLK];
            h16x8 Bh[PDK + 1], Bl[PDK + 1];
            auto ld = [&](int c) __attribute__((always_inline)) {
                const unsigned char *src = buf + rd_lane + chunk_off(c);
                Bh[c % (PDK + 1)] = *reinterpret_cast<const h16x8 *>(src);
                Bl[c % (PDK + 1)] = *reinterpret_cast<const h16x8 *>(src + PL);
            };
            auto half = [&](int h) __attribute__((always_inline)) {
                constexpr int HC = NCH / 8;               // barrier A comes early in the period (the stagers only read their accumulator tiles in front of it)
                if (h == 0) {
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        m0[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                        m1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                        lo[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int c = 0; c < PDK; ++c) ld(c);
                }
#pragma unroll
                for (int c = h * HC; c < (h ? NCH : HC); ++c) {
                    if (c + PDK < NCH) ld(c + PDK);
                    __builtin_amdgcn_sched_barrier(0);
                    const int sl_ = c % (PDK + 1);
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int j = c - CB * b;
                        if (j < 0 || j >= KS) continue;
                        if (j & 1) m1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bh[sl_], m1[b], 0, 0, 0);
                        else m0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bh[sl_], m0[b], 0, 0, 0);
                    }
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int j = c - CB * b;
                        if (j < 0 || j >= KS) continue;
                        lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[j], Bl[sl_], lo[b], 0, 0, 0);
                    }
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int j = c - CB * b;
                        if (j < 0 || j >= KS) continue;
                        lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[j], Bh[sl_], lo[b], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            MF_STAMP(7);
            if (act && !(GRHIP_RS_ABL & 1)) half(0);
            MF_STAMP(0);
            rs_barrier();
            MF_STAMP(3);
            if (act) {
                if (!(GRHIP_RS_ABL & 1)) half(1);
                else {
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) { m0[b] = f32x4{0.f, 0.f, 0.f, 0.f}; m1[b] = m0[b]; lo[b] = m0[b]; }
                }
                // accumulator layout in ([segment][row][re, im]): block b into plane b of the wave's own stretch
                // (nobody else reads it: Geo::HALO), read by the stagers after the next barrier
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const f32x4 accf = (m0[b] + m1[b]) + lo[b];
                    float *sb = reinterpret_cast<float *>(buf + scw_off + b * PL);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sb[sc_wr + 2 * i] = accf[i];
                }
                // blocks 2 and 3 straight away (own tiles only: LDS keeps a wave's operations in order)
                const int q = p - 2;
                const int kx = kring[q & 3];
#pragma unroll
                for (int b = 2; b < NBLK; ++b) {
                    if (GRHIP_RS_ABL & 2) break;
                    EpiRegs er;
                    epilogue_read(q, mw, b, er);
                    epilogue_block(q, kx, mw, b, false, er);
                }
            }
            MF_STAMP(4);
            rs_barrier();
            MF_STAMP(5);
        }
        MF_STAMP_OUT(rs::NSTG + rs::NMAT);
        return;
    }

    // =============================== stager role ===============================
    MF_STAMP_DECL;
#if defined(GRHIP_RS_PRIO) && GRHIP_RS_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
    // pre-mix phasors of the lane's two samples of every staging round, e^{jw(2 ts + ROUND i - off)} and the next one:
    // tile independent, registers for the whole launch (the tile's scale goes into the binary16 conversion instead)
    // (the second sample's phasor is one more product with e^{jw}: eighteen registers fewer than keeping both)
    f32x2 W0[NI];
    const f32x2 wstep{a.wstep.x, a.wstep.y};
    const cfloat_cp stab = (cfloat_cp)a.stab;
    if (PREMIX) {
        const float2 v = a.wlane[ts];
        const f32x2 wl{v.x, v.y};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x2 S{stab[4 * i], stab[4 * i + 1]};          // e^{jw ROUND i}: every other entry of the 512-sample table
            W0[i] = cmul_pk(wl, S);
        }
    }
    if (DEMOD) {
        f32x2 *at = reinterpret_cast<f32x2 *>(smem + R::OFF_ATAN);
        for (int i = ts; i < 256; i += rs::STG_T) at[i] = f32x2{a.atan_tab[i], a.atan_tab[i + 1]};
    }
    // staging store: sample u = 2 ts + ROUND i  ->  byte 2u + 32 (u >> LOGQ) of each plane
    const int st_off = 4 * ts + 32 * ((2 * ts) >> LOGQ);
    constexpr int ST_STEP = 2 * rs::ROUND + 32 * (rs::ROUND >> LOGQ);
    // the item in front of a stream that does not start on a 16-byte boundary (lead = 1: the descriptor starts one
    // item early) reads as zero: it sits in the first tile's round 0 (the tile starts BLK D + off + n_lo - lead < ROUND
    // samples before the stream's first item) and is dropped where the values are used, not in the registers
    static_assert(BLK * D + 1 + 32 * KS < rs::ROUND, "the stream's first item lies in round 0 of its first tile");
    auto lead_item_here = [&](int q) __attribute__((always_inline)) -> bool {
        if (!lead) return false;
        int s, b;
        decode(q, s, b);
        __amdgpu_buffer_rsrc_t rsrc; int voff;
        tile_geom(s, b, rsrc, voff);
        return b == 0 && voff == 0;
    };

    // ---- L: the tile's loads into one of the two register sets.  Hand-issued (inline asm) and hand-waited: a set is
    // loaded in one trip of the period loop and used in the next, and for such loads hipcc 7.2's wait-count bookkeeping
    // falls back to "all but the youngest 8 ... 0 operations" at the first use -- it drained the loads that had just been
    // issued (or, with the request at the period's start, let nothing fly across the loop's back edge).  The compiler
    // does not know these registers have writes pending: every use sits behind loads_landed() below, which waits
    // (vmcnt counts in order: leaving the `younger` most recent operations in flight) and redefines the registers for it.
    struct LoadCtx { u32x4 rs; int voff; };
    auto load_ctx = [&](int q) __attribute__((always_inline)) -> LoadCtx {
        int s, b;
        decode(q, s, b);
        const long long g0 = ((long long)b * NTE - BLK) * D - a.off - a.n_lo + lead;
        const unsigned long long xb = (unsigned long long)(a.x + (long long)s * a.x_stride + a.n_lo - lead);
        LoadCtx c;                                   // the words __builtin_amdgcn_make_buffer_rsrc(x, 0, bytes, 0x00020000) holds
        c.rs[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        c.rs[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffffu));
        c.rs[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)((a.n_in - a.n_lo + lead) * 8));
        c.rs[3] = 0x00020000u;
        c.voff = (int)(g0 * 8) + 16 * ts;
        return c;
    };
    auto issue_round = [&](const LoadCtx &c, f32x4 &dst, int i) __attribute__((always_inline)) {
        int vo = c.voff + i * (16 * rs::STG_T);
        if ((i + 1) * rs::ROUND > SP && 2 * ts + i * rs::ROUND >= SP) vo = 0x7ffff000;   // past the tile: no traffic
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(vo), "s"(c.rs));
    };
    auto issue_loads = [&](int q, f32x4 (&pf)[NI]) __attribute__((always_inline)) {
        const LoadCtx c = load_ctx(q);
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_round(c, pf[i], i);
    };
    auto loads_landed = [&](f32x4 (&pf)[NI], bool younger_set) __attribute__((always_inline)) {
        if (younger_set) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI));
        else asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(pf[i]));
    };
    // ---- M: block floating point, the wave's largest |component| of the tile ----
    auto tile_max = [&](int q, f32x4 (&pf)[NI]) __attribute__((always_inline)) {
        const bool lz = lead_item_here(q);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float x0 = (i == 0 && lz) ? 0.f : pf[i][0], x1 = (i == 0 && lz) ? 0.f : pf[i][1];
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(x0), "v"(x1));
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(pf[i][2]), "v"(pf[i][3]));
        }
        m = wave_max_nonneg(m);
        if (!(m < __builtin_inff())) {
            // an Inf / NaN sample (rare; the branch is wave-uniform): the scale comes from the finite samples, so that
            // the damage stays where the reference has it -- the outputs whose window holds the sample (and, here, the
            // rest of their 16-output block: zeros of the band matrix times Inf) -- instead of flushing the whole tile
            float mf = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = __builtin_fabsf(pf[i][e]);
                    if (!(i == 0 && e < 2 && lz)) mf = av < __builtin_inff() ? __builtin_fmaxf(mf, av) : mf;
                }
            m = wave_max_nonneg(mf);
        }
        if (lane == 0) wmax[w] = m;
    };
    // ---- C: registers -> (pre-mix, scale, split) -> registers: the four binary16 pairs of every round (vector work) ----
    // ---- C + S: registers -> (pre-mix, scale, split) -> the four planes of buffer q & 1, round by round; behind every
    // round (q_load >= 0) the same round of tile q_load is requested into the registers it has just left -- one load
    // between two rounds of vector work instead of nine in a row, which held the wave at the memory pipeline's queue for
    // 0.5-1 us per period.  Two stores per round, each to two planes a multiple of 256 bytes apart.
    static_assert(PL % 256 == 0 && 3 * (PL / 256) < 256, "plane distance as a ds_write2st64_b32 offset");
    auto convert = [&](int q, f32x4 (&pf)[NI], int &kslot, int q_load) __attribute__((always_inline)) {
        const LoadCtx lc = load_ctx(q_load >= 0 ? q_load : q);
        float mt = wmax[0];
#pragma unroll
        for (int i = 1; i < rs::NSTG; ++i) mt = __builtin_fmaxf(mt, wmax[i]);
        int k = 14 - __builtin_amdgcn_frexp_expf(mt);   // |x| e^{jw} components stay below 2^15
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        kslot = k;
        if (ts == 0) kring[q & 3] = k;                  // for the matrix waves' share of the epilogue (a period later)
        const float scale = __builtin_amdgcn_ldexpf(1.0f, k);
        const bool lz = lead_item_here(q);
        const int base = (q & 1) * R::BUF + st_off;     // LDS byte address (dynamic LDS starts at 0)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x2 e0{pf[i][0], pf[i][1]}, e1{pf[i][2], pf[i][3]};
            if (i == 0 && lz) e0 = f32x2{0.f, 0.f};
            if (PREMIX) {
                e0 = cmul_pk(e0, W0[i]);
                e1 = cmul_pk(e1, cmul_pk(W0[i], wstep));
            }
            h16x2 rh, rlo, ih, ilo;
            split_scaled(e0.x, e1.x, scale, rh, rlo);
            split_scaled(e0.y, e1.y, scale, ih, ilo);
            if ((i + 1) * rs::ROUND <= SP || 2 * ts + i * rs::ROUND < SP) {
                const int ad = base + i * ST_STEP;
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:%5\n\t"
                             "ds_write2st64_b32 %0, %3, %4 offset0:%6 offset1:%7"
                             :: "v"(ad), "v"(rh), "v"(rlo), "v"(ih), "v"(ilo),
                                "n"(PL / 256), "n"(2 * (PL / 256)), "n"(3 * (PL / 256)) : "memory");
            }
            if (q_load >= 0) issue_round(lc, pf[i], i);
        }
    };
    // one period: pfL takes the loads of tile p, pfS holds tile p - 1 (complete: every period ends with the wave's
    // vector-memory counter at zero, and its maxima were published before the last barrier); kslot: the exponent of
    // tile p - 3 on entry (the epilogue's), of tile p - 1 on exit.
    //   first half   (vector work)   loads of tile p out; epilogue of tile p - 3; tile p - 1 converted in registers
    //   -- barrier: the epilogue has left buffer (p-1)&1 --
    //   second half  (LDS stores)    tile p - 1 into the planes; wait for tile p; its maxima
    //   -- barrier --
    // The two halves are bound by different units -- vector issue and the LDS store path (a stager's 36 four-byte
    // stores per tile took as long as its 120 conversion instructions) -- so each gets a half to itself and the matrix
    // waves' MFMAs run beside both.  The loads fly for the whole period and are waited for ONCE, with a wait the
    // compiler's counter bookkeeping understands (__builtin_amdgcn_s_waitcnt): a set that is loaded in one trip of the
    // loop and used in the next makes hipcc 7.2 wait for "all but the 8 youngest" operations at the first use, i.e.
    // for the loads just issued.
    // One period (pfL: tile p, in flight; pfS: tile p - 1, complete; kslot: the exponent of tile p - 3 on entry -- the
    // epilogue's --, of tile p - 1 on exit):
    //   the wave's block of tile p - 3 out of the accumulator tiles (LDS reads)
    //   -- barrier A: from here the planes of buffer (p-1)&1 may be written --
    //   that block's demodulator / stores; tile p - 1 converted and stored round by round, tile p + 1 requested round by
    //   round into the registers that come free; wait for tile p (leaving tile p + 1 in flight), its maxima
    //   -- barrier B --
    // Tile p + 1's loads fly for a period and a half; some tile's loads are in flight at every moment.
    auto period = [&](int p, f32x4 (&pfL)[NI], f32x4 (&pfS)[NI], int &kslot) __attribute__((always_inline)) {
        MF_STAMP(7);
        static_assert(NBLK == 4 && rs::NSTG == 2 * rs::NMAT, "stager w finishes block w & 1 of matrix wave w / 2");
        const bool eact = p >= 3 && p - 3 < n_my && !(GRHIP_RS_ABL & 2);
        const int kx = kslot;
        EpiRegs er;
        if (eact) epilogue_read(p - 3, w >> 1, w & 1, er);
        MF_STAMP(0);
        rs_barrier();
        MF_STAMP(3);
        // Half of the stagers convert first and finish their block of the epilogue afterwards, the other half the other
        // way round: the CU's loads (requested round by round inside the conversion) are then spread over the whole
        // period.  With all eight stagers in step the requests came in one burst per period, the memory pipeline's queue
        // ran empty between two bursts, and transfer time and vector time ADDED UP (compute alone 0.83 ms, loads alone
        // 0.80 ms, both 1.28 ms per 64 x 10 M samples).
        const bool sact = p >= 1 && p - 1 < n_my;
        const bool lact = p + 1 < n_my && !(GRHIP_RS_ABL & 8);
        const bool convert_first = (w & 4) == 0;
        if (!convert_first && eact) epilogue_block(p - 3, kx, w >> 1, w & 1, w == 0, er);
        MF_STAMP(1);
        if (sact && !(GRHIP_RS_ABL & 4)) convert(p - 1, pfS, kslot, lact ? p + 1 : -1);   // (pfS becomes tile p + 1's set)
        else if (lact) issue_loads(p + 1, pfS);
        MF_STAMP(2);
        if (convert_first && eact) epilogue_block(p - 3, kx, w >> 1, w & 1, w == 0, er);
        MF_STAMP(4);
        if (p < n_my) {
            loads_landed(pfL, lact);                // tile p is there; tile p + 1 stays in flight
            if (!(GRHIP_RS_ABL & 16)) tile_max(p, pfL);
        }
        MF_STAMP(6);
        rs_barrier();
        MF_STAMP(5);
    };
    f32x4 pf0[NI], pf1[NI];
    int k0 = 0, k1 = 0;
    if (!(GRHIP_RS_ABL & 8)) issue_loads(0, pf0);
    for (int p = 0; p < P; p += 2) {
        period(p, pf0, pf1, k1);
        if (p + 1 < P) period(p + 1, pf1, pf0, k0);
    }
    MF_STAMP_OUT(rs::NSTG + rs::NMAT);
}

#endif  // GRHIP_DIAG

static int g_mf_cus = 0;

template <int D, int KS, bool PREMIX, int EPI, bool TAPQ = false>
static int launch_mfma_inst(const FirMfmaArgs &a, hipStream_t st)
{
    using G = Geo<D, KS>;
    auto kern = fir_mfma_kernel<D, KS, PREMIX, EPI, TAPQ>;
    static bool configured = false;
    if (!configured) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        configured = true;
    }
    if (g_mf_cus == 0) {
        int dev = 0, n = 0;
        GRHIP_HIP(hipGetDevice(&dev));
        GRHIP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_mf_cus = n > 0 ? n : 256;
    }
    const long long tiles = ((a.n_out + mf::NTE - 1) / mf::NTE) * a.n_streams;
    // The role-split kernel (one 768-lane workgroup per CU, the whole register file) wherever the FIR has its CUs to
    // itself; a caller that runs another kernel on the same CUs (max_wg_per_cu = 1: the chain's clock recovery beside
    // the FIR on shared CUs) keeps the kernel above, which leaves half of every CU free.
    // The role-split kernel (fir_mfma_rs_kernel above: one 768-lane workgroup per CU, stager and matrix waves) exists in
    // diagnostic builds only, selected with GRHIP_MF_RS=1: ten versions of it tied with or lost to the kernel above by 0-6 %
    // (DESIGN 4.0a has the A/B table, the stamps and the ablations that say why).
    bool role_split = false;
#if defined(GRHIP_DIAG) && GRHIP_MF_NBLK == 4
    if (const char *e = getenv("GRHIP_MF_RS")) role_split = atoi(e) != 0 && a.max_wg_per_cu != 1;
    if (role_split) {
        using R = GeoRS<D, KS>;
        auto kern_rs = fir_mfma_rs_kernel<D, KS, PREMIX, EPI>;
        static bool configured_rs = false;
        if (!configured_rs) {
            GRHIP_HIP(hipFuncSetAttribute((const void *)kern_rs, hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS));
            configured_rs = true;
        }
        long long grid = a.max_cus > 0 && a.max_cus < g_mf_cus ? a.max_cus : g_mf_cus;
        if (grid > tiles) grid = tiles;
        hipLaunchKernelGGL(kern_rs, dim3((unsigned)grid), dim3(rs::THREADS), R::LDS, st, a);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
#endif
    (void)role_split;
    int wgs = (160 * 1024) / (G::LDS + 256);
    if (wgs > GRHIP_MF_WGS) wgs = GRHIP_MF_WGS;
    if (a.max_wg_per_cu > 0 && wgs > a.max_wg_per_cu) wgs = a.max_wg_per_cu;
    if (wgs < 1) wgs = 1;
    long long grid = (long long)wgs * (a.max_cus > 0 && a.max_cus < g_mf_cus ? a.max_cus : g_mf_cus);
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(mf::THREADS), G::LDS, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int D, int KS>
static int launch_mfma_dk(bool premix, int epi, const FirMfmaArgs &a, hipStream_t st)
{
    if (premix) {
        if (a.tapq) {
            if (epi == EPI_DEMOD) return launch_mfma_inst<D, KS, true, EPI_DEMOD, true>(a, st);
            if (epi == EPI_ROTATE) return launch_mfma_inst<D, KS, true, EPI_ROTATE, true>(a, st);
        }
        if (epi == EPI_DEMOD) return launch_mfma_inst<D, KS, true, EPI_DEMOD>(a, st);
        if (epi == EPI_ROTATE) return launch_mfma_inst<D, KS, true, EPI_ROTATE>(a, st);
        return fail(GRHIP_EINVAL, "matrix FIR: the pre-mix form needs a rotate or demod epilogue");
    }
    if (epi != EPI_NONE) return fail(GRHIP_EINVAL, "matrix FIR: real taps without pre-mix have no rotator");
    return launch_mfma_inst<D, KS, false, EPI_NONE>(a, st);
}

bool mfma_supported(int decim, int ntaps) { return mf::supported(decim, ntaps); }

// alignment the kernel needs from a launch: 16-byte loads per lane pair
bool mfma_launch_ok(const FirMfmaArgs &a)
{
    if ((a.x_stride & 1) && a.n_streams > 1) return false;
    return true;
}

int launch_fir_mfma(int decim, int ntaps, bool premix, int epi, const FirMfmaArgs &a_in, hipStream_t st)
{
    if (a_in.n_out <= 0 || a_in.n_streams <= 0) return GRHIP_OK;
    if (!mf::supported(decim, ntaps)) return fail(GRHIP_EINVAL, "matrix FIR: unsupported shape");
    if (!mfma_launch_ok(a_in)) return fail(GRHIP_EINVAL, "matrix FIR: odd stream stride");
    if ((a_in.n_in - a_in.n_lo) * 8 >= (1ll << 31) - (1ll << 20))
        return fail(GRHIP_EINVAL, "matrix FIR: more than 2 GiB of input per stream in one call; call work() in pieces");
    const int KS = mf::ksteps_inst(decim, ntaps);
    switch (decim * 100 + KS) {
    case 406: return launch_mfma_dk<4, 6>(premix, epi, a_in, st);
    case 410: return launch_mfma_dk<4, 10>(premix, epi, a_in, st);
    case 206: return launch_mfma_dk<2, 6>(premix, epi, a_in, st);
    case 210: return launch_mfma_dk<2, 10>(premix, epi, a_in, st);
    }
    return fail(GRHIP_EINVAL, "matrix FIR: unsupported decimation %d", decim);
}

}  // namespace grhip
