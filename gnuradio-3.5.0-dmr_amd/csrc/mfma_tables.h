// mfma_tables.h -- host-side geometry and operand tables of the matrix-core FIR kernel
// (fir_mfma.hip).  Plain C++ (no HIP): the geometry is unit-tested on the CPU
// (tests/test_mfma_tables.py compiles this header with g++).
//
// The decimating FIR y[n] = sum_{i<T} h[i] x[nD + i] (h real, x complex) for 16 consecutive
// outputs n0 .. n0+15 is one banded-Toeplitz product
//
//      Y[a][col] = sum_k A[a][k] X[k][col],   A[a][k] = h[k - D a - off]  (0 outside the band)
//
// with X[k][col] the samples x[n0 D - off + k] of 16 "columns" (8 stream segments x {re, im}).
// A is constant: it lives in registers for the whole launch, split in two binary16 halves
// (A = Ah + Al to ~22 bits); the samples are split the same way while they are staged, and
// a product is three v_mfma_f32_16x16x32_f16 (Ah Xh + Ah Xl + Al Xh, f32 accumulation).
// `off` (0 or 1) is the parity of the stream's 16-byte alignment: the staged tile starts on
// a 16-byte boundary of the stream and the band is shifted by one sample instead.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace grhip {
namespace mf {

constexpr int THREADS = 256, WAVES = 4;
constexpr int SEGS = 8;             // stream segments per wave = MFMA column pairs
constexpr int BLK = 16;             // outputs per MFMA block (rows of A)
#ifndef GRHIP_MF_NBLK
#define GRHIP_MF_NBLK 4             // (experiment: 2 = tiles of half the size)
#endif
constexpr int NBLK = GRHIP_MF_NBLK; // blocks per segment
constexpr int SEG_OUT = BLK * NBLK; // 64
constexpr int WAVE_OUT = SEGS * SEG_OUT;        // 512 outputs computed per wave
constexpr int WAVE_NEW = WAVE_OUT - BLK;        // 496: the first block of a wave only hands its last output on
constexpr int NTC = WAVE_NEW * (WAVES - 1) + WAVE_OUT;   // 2000 outputs computed per tile
constexpr int NTE = WAVE_NEW * WAVES;                    // 1984 of them new
constexpr int CHUNK = 32;           // samples per MFMA k-step
constexpr int ROUND = 2 * THREADS;  // samples per staging round of the workgroup (16 bytes per lane)

// k-steps needed for T taps at decimation D (band of the last row ends at 15 D + off + T)
constexpr int ksteps_for(int D, int T) { return (15 * D + 1 + T + CHUNK - 1) / CHUNK; }
// chunks a segment slides over, samples per tile, staging rounds, LDS plane size
constexpr int chunks_per_seg(int D, int KS) { return (NBLK - 1) * (BLK * D / CHUNK) + KS; }
constexpr int tile_samples(int D, int KS)
{
    return ((WAVES - 1) * WAVE_NEW + (SEGS - 1) * SEG_OUT) * D + chunks_per_seg(D, KS) * CHUNK;
}
constexpr int rounds(int D, int KS) { return (tile_samples(D, KS) + ROUND - 1) / ROUND; }
// byte position of sample u inside a plane: 2 bytes per sample + 32 bytes of skew per segment
// stride, which puts the 8 segments of a wave on different LDS slots (fir_mfma.hip)
constexpr int plane_pos(int u, int D) { return 2 * u + 32 * (u / (SEG_OUT * D)); }
constexpr int plane_bytes(int D, int KS)
{
    return (plane_pos(tile_samples(D, KS) - 1, D) + 2 + 255) & ~255;
}

inline bool supported(int D, int T)
{
    if (!(D == 2 || D == 4)) return false;
    if (T < 1) return false;
    return ksteps_for(D, T) <= 10;
}
// the instantiated k-step counts
inline int ksteps_inst(int D, int T) { int k = ksteps_for(D, T); return k <= 6 ? 6 : 10; }

// ---- binary16 (round to nearest even), host side ---------------------------------
inline uint16_t f32_to_f16(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           // rounds to infinity
    if (x < 0x33000001u) return (uint16_t)sign;                          // below half the smallest subnormal
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                          // subnormal halves lose more bits
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    if (e >= -14) return (uint16_t)(sign | (uint32_t)(((e + 15) << 10) + (q - 0x400u)));   // carry walks into the exponent
    return (uint16_t)(sign | q);
}
inline float f16_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    int e = (h >> 10) & 31;
    uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    return sign ? -v : v;
}

// Scale of the taps: the largest |h| lands in [2^13, 2^14), so that its low half is a normal
// binary16 and nothing overflows.  Returns k with scale = 2^k.
inline int tap_scale_exp(const float *h, int T)
{
    float m = 0.f;
    for (int i = 0; i < T; ++i) m = std::fmax(m, std::fabs(h[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 0;
    int e;
    std::frexp(m, &e);          // m = f 2^e, f in [0.5, 1)
    int k = 14 - e;
    if (k > 100) k = 100;
    if (k < -100) k = -100;
    return k;
}

// A operand of every k-step, both halves, in the lane order of v_mfma_f32_16x16x32_f16:
// lane l holds A[row = l & 15][k = 8 (l >> 4) + j], j = 0..7.
// layout: [KS][2 (hi, lo)][64 lanes][8] binary16.
inline void build_A(const float *h, int T, int D, int KS, int off, int kexp, std::vector<uint16_t> &out)
{
    out.assign((size_t)KS * 2 * 64 * 8, 0);
    for (int js = 0; js < KS; ++js)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int a = l & 15, k = CHUNK * js + 8 * (l >> 4) + j;
                const int i = k - D * a - off;
                if (i < 0 || i >= T) continue;
                const float v = std::ldexp(h[i], kexp);
                const uint16_t hi = f32_to_f16(v);
                const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                out[(((size_t)js * 2 + 0) * 64 + l) * 8 + j] = hi;
                out[(((size_t)js * 2 + 1) * 64 + l) * 8 + j] = lo;
            }
}

// ---- the reference's tap-angle quantisation (freq_xlating only) -----------------------------------------
// gr_freq_xlating_fir_filter builds its composite taps as proto[i] * exp(j * (float)(i * fwT0)) with the product i * fwT0
// rounded to binary32 (filter/gr_freq_xlating_fir_filter_XXX.cc.t:79): tap i carries an angle error e_i of up to half an
// ulp of a number as large as ntaps * fwT0 (7.6e-6 rad at 256 taps and fwT0 = pi/4).  The pre-mix form evaluates exact
// angles; on a signal whose phase turns through the filter's span that difference is what separates its demodulator
// output from the reference's (1.64e-5 of 1.69e-5 per element on cfg2).  To first order
//      h[i] e^{j e_i} = h[i] + j g[i],   g[i] = h[i] e_i,
// so the reference's output is the real-tap sum plus j times a second real-tap sum with the taps g -- a band matrix G
// like A, needed to ~10 bits only (high halves, one MFMA per k-step), and only for the NG middle k-steps that hold the
// prototype's main lobe for every row of a block.
constexpr int NG = 4;                                   // k-steps that carry the correction
constexpr int g_first(int KS) { return KS / 2 - NG / 2; }
constexpr int G_EXTRA_EXP = 10;                         // g is ~2^-18 of h: scaled by 2^10 more, taken out when T is added
inline double tap_angle_error(int i, float fwT0) { return (double)((float)i * fwT0) - (double)i * (double)fwT0; }
// layout [NG][64 lanes][8] binary16, appended to build_A's table
inline void build_G(const float *h, int T, int D, int KS, int off, int kexp, float fwT0, std::vector<uint16_t> &out)
{
    const size_t base = out.size();
    out.resize(base + (size_t)NG * 64 * 8, 0);
    for (int q = 0; q < NG; ++q)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int js = g_first(KS) + q;
                const int a = l & 15, k = CHUNK * js + 8 * (l >> 4) + j;
                const int i = k - D * a - off;
                if (i < 0 || i >= T || js < 0 || js >= KS) continue;
                const double g = (double)h[i] * tap_angle_error(i, fwT0);
                out[base + ((size_t)q * 64 + l) * 8 + j] = f32_to_f16((float)std::ldexp(g, kexp + G_EXTRA_EXP));
            }
}

}  // namespace mf
}  // namespace grhip
