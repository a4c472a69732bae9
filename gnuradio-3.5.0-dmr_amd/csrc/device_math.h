// device_math.h -- device-side restatements of the scalar helpers on the path.
// The whole library is compiled with -ffp-contract=off: every '*' '+' '-' below
// is one IEEE binary32 operation, as in the reference's x86-64 SSE scalar code;
// fused multiply-adds appear only where __builtin_fmaf is written out.
#pragma once
#include <hip/hip_runtime.h>

namespace grhip {

// complex<float> product as libgcc's __mulsc3 computes it for finite operands
// (used by gr_rotator::rotate, gr_quadrature_demod_cf::work, gr_fir_ccc_generic).
__device__ __forceinline__ float2 cmul_ref(float2 a, float2 b)
{
    float ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
    return make_float2(ac - bd, ad + bc);
}

// fused form for the fast kernels
__device__ __forceinline__ float2 cmul_fma(float2 a, float2 b)
{
    return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// the same product as cmul_fma in two packed instructions: the halves of the operands
// are routed with op_sel and the one negation is an operand modifier, so nothing is
// moved or sign-flipped beforehand
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t cmul_pk(f32x2_t a, f32x2_t b)
{
    f32x2_t t, r;
    // t = (-(a.y * b.y), a.y * b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    // r = (fma(a.x, b.x, t.x), fma(a.x, b.y, t.y))
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

// gr_fast_atan2f (gnuradio-core/src/lib/general/gr_fast_atan2f.cc:125-198), written
// without branches (on a GPU every lane of a wave sits in a different octant, so each
// branch of the original would be executed by every wave anyway).  With EXACT_DIV the
// result is bit-identical to the reference for every input:
//  * :140  z = min(|y|,|x|) / max(|y|,|x|): the same IEEE division whichever branch the
//    reference takes;
//  * :147  `z < TAN_MAP_RES` compares in double against 0.003921569; for a binary32 z that
//    is exactly `z < 0x1.010104p-8f` (0x3b808082, the smallest float >= the constant);
//  * :151  `z * (REAL)256 - .5`: the product is exact (power of two) and the double
//    subtraction is exact before narrowing, i.e. one float subtraction;
//  * :161-195  the eight octant cases are q + s*base, negated for y < 0, with
//    q in {0, pi, pi/2}: a - b == -(b - a) and 0 + b == b hold bit for bit in IEEE
//    arithmetic, so the selects reproduce the reference's additions and subtractions.
// EXACT_DIV = false (FAST-mode fused epilogue only) replaces the division by
// ldexp(num, -e) * rcp(mantissa(den)): 1 ulp, any magnitude, six instructions fewer.
// `tab`: a pointer to the 257 floats, or a view that returns the pair (tab[i], tab[i+1]).
__device__ __forceinline__ void atan_pair(const float *tab, int i, float &t0, float &t1) { t0 = tab[i]; t1 = tab[i + 1]; }
template <class PairView>
__device__ __forceinline__ void atan_pair(PairView tab, int i, float &t0, float &t1)
{
    const auto p = tab[i];
    t0 = p.x; t1 = p.y;
}
template <bool EXACT_DIV = true, class Tab>
__device__ __forceinline__ float fast_atan2f(float y, float x, Tab tab)
{
    const float y_abs = __builtin_fabsf(y), x_abs = __builtin_fabsf(x);
    const bool big = x_abs > y_abs;                        // :161 (and :140 `y_abs < x_abs`)
    const float num = big ? y_abs : x_abs, den = big ? x_abs : y_abs;
    float z;
    if (EXACT_DIV) {
        z = num / den;                                     // :140-143 (IEEE divide)
    } else {
        const int e = __builtin_amdgcn_frexp_expf(den);
        z = __builtin_amdgcn_ldexpf(num, -e) * __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(den));
    }
    float alpha = z * 256.0f - 0.5f;                       // :151
    int index = (int)alpha;
    index = index < 0 ? 0 : (index > 255 ? 255 : index);   // no-op for finite z >= TAN_MAP_RES; keeps the rest in bounds
    if (EXACT_DIV) alpha -= (float)index;
    else alpha = __builtin_amdgcn_fractf(alpha);           // same value wherever the interpolation is used (alpha >= 0)
    float t0, t1;
    atan_pair(tab, index, t0, t1);
    float interp = t0;
    interp += (t1 - t0) * alpha;                           // :156-157, unfused
    const float base_angle = z < __builtin_bit_cast(float, 0x3b808082u) ? z : interp;   // :147

    const float PI_F = (float)3.14159265358979323846;
    const float HALF_PI_F = (float)1.57079632679489661923;
    const bool xpos = x >= 0.0f, ypos = y >= 0.0f;
    const float q = big ? (xpos ? 0.0f : PI_F) : HALF_PI_F;
    const float sb = (big != xpos) ? -base_angle : base_angle;
    const float ap = q + sb;
    float angle = ypos ? ap : -ap;
    return (den == 0.0f) ? 0.0f : angle;                   // :133 (y == 0 && x == 0  <=>  max(|y|,|x|) == 0)
}

// one output of gr_quadrature_demod_cf::work (general/gr_quadrature_demod_cf.cc:57-59)
template <class Tab>
__device__ __forceinline__ float quad_demod_one(float2 cur, float2 prev, float gain, Tab tab)
{
    float2 product = cmul_ref(cur, make_float2(prev.x, -prev.y));   // in[i] * conj(in[i-1])
    return gain * fast_atan2f<true>(product.y, product.x, tab);
}

// FAST-mode fused epilogue: fused product, 1-ulp division
template <class Tab>
__device__ __forceinline__ float quad_demod_fast(float2 cur, float2 prev, float gain, Tab tab)
{
    const f32x2_t product = cmul_pk(f32x2_t{cur.x, cur.y}, f32x2_t{prev.x, -prev.y});
#if defined(GRHIP_DIAG) && defined(GRHIP_PROBE) && (GRHIP_PROBE & 1)     // attribution probe: the IEEE divide in the FAST epilogue
    return gain * fast_atan2f<true>(product.y, product.x, tab);
#else
    return gain * fast_atan2f<false>(product.y, product.x, tab);
#endif
}

// The matrix-core engine's demodulator (GRHIP_LG_LEAN, fir_mfma.hip): the same table, seven instructions fewer per output.
//  * z = num * rcp(den): the engine's samples are block floating point -- what lies 2^-22 below its tile's largest
//    component is already lost in the split into two binary16 halves -- so a den whose reciprocal overflows (below 2^-126)
//    is a quotient of rounding noise; den == 0 is selected away as before.  In range the value is the scaled form's, bit for bit
//    (rcp of a power-of-two multiple is that multiple of the rcp);
//  * the table holds (tab[i], tab[i+1] - tab[i]): one fused multiply-add for the interpolation;
//  * the index is clamped with one v_med3.
template <class PairView>
__device__ __forceinline__ float fast_atan2f_lean(float y, float x, PairView tab)
{
    const float y_abs = __builtin_fabsf(y), x_abs = __builtin_fabsf(x);
    const bool big = x_abs > y_abs;
    const float num = __builtin_fminf(y_abs, x_abs), den = __builtin_fmaxf(y_abs, x_abs);
    const float z = num * __builtin_amdgcn_rcpf(den);
    float alpha = __builtin_fmaf(z, 256.0f, -0.5f);
    int index = (int)alpha;
    index = index < 0 ? 0 : (index > 255 ? 255 : index);
    alpha = __builtin_amdgcn_fractf(alpha);
    const auto p = tab[index];                                // (tab[i], tab[i+1] - tab[i])
    const float interp = __builtin_fmaf(p.y, alpha, p.x);
    const float base_angle = z < __builtin_bit_cast(float, 0x3b808082u) ? z : interp;   // :147
    const float PI_F = (float)3.14159265358979323846;
    const float HALF_PI_F = (float)1.57079632679489661923;
    const bool xpos = x >= 0.0f, ypos = y >= 0.0f;
    const float q = big ? (xpos ? 0.0f : PI_F) : HALF_PI_F;
    const float sb = (big != xpos) ? -base_angle : base_angle;
    const float ap = q + sb;
    const float angle = ypos ? ap : -ap;
    return (den == 0.0f) ? 0.0f : angle;
}
template <class PairView>
__device__ __forceinline__ float quad_demod_lean(float2 cur, float2 prev, float gain, PairView tab)
{
    const f32x2_t product = cmul_pk(f32x2_t{cur.x, cur.y}, f32x2_t{prev.x, -prev.y});
    return gain * fast_atan2f_lean(product.y, product.x, tab);
}

// gr_branchless_clip (general/gr_math.h:63-69)
__device__ __forceinline__ float branchless_clip(float x, float clip)
{
    float x1 = __builtin_fabsf(x + clip);
    float x2 = __builtin_fabsf(x - clip);
    x1 -= x2;
    return (float)(0.5 * (double)x1);
}

}  // namespace grhip
