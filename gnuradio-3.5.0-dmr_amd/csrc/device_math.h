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

// gr_fast_atan2f (gnuradio-core/src/lib/general/gr_fast_atan2f.cc:125-198).
// REAL = float but the literals are double: the comparisons and the "- .5"
// are evaluated in double and narrowed on assignment, as written there.
// `tab` is anything indexable with 0..256: a pointer, or a view of a table stored with gaps
template <class Tab>
__device__ __forceinline__ float fast_atan2f(float y, float x, Tab tab)
{
    float x_abs, y_abs, z;
    float alpha, angle, base_angle;
    int index;

    if ((y == 0.0f) && (x == 0.0f)) return 0.0f;          // :133

    y_abs = __builtin_fabsf(y);
    x_abs = __builtin_fabsf(x);
    if (y_abs < x_abs) z = y_abs / x_abs;                  // :140 (IEEE divide)
    else               z = x_abs / y_abs;

    // :147 `z < TAN_MAP_RES` compares in double against 0.003921569.  For a binary32 z
    // that is exactly `z < 0x1.010104p-8f` (0x3b808082, the smallest float >= the
    // constant), so no double arithmetic is needed.
    // :151 `z * (REAL)256 - .5`: the product is exact (power of two) and the double
    // subtraction is exact before narrowing, i.e. one float subtraction.
    if (z < __builtin_bit_cast(float, 0x3b808082u)) {
        base_angle = z;
    } else {
        alpha = z * 256.0f - 0.5f;
        index = (int)alpha;
        index = index < 0 ? 0 : (index > 255 ? 255 : index);   // no-op for finite input; keeps NaN in bounds
        alpha -= (float)index;
        float t0 = tab[index], t1 = tab[index + 1];
        base_angle = t0;
        base_angle += (t1 - t0) * alpha;                   // :156-157, unfused
    }

    const float PI_F = (float)3.14159265358979323846;
    const float HALF_PI_F = (float)1.57079632679489661923;
    if (x_abs > y_abs) {                                   // :161
        if (x >= 0.0f) {
            angle = (y >= 0.0f) ? base_angle : -base_angle;
        } else {
            angle = PI_F;
            if (y >= 0.0f) angle -= base_angle;
            else           angle = base_angle - angle;
        }
    } else {
        if (y >= 0.0f) {
            angle = HALF_PI_F;
            if (x >= 0.0f) angle -= base_angle;
            else           angle += base_angle;
        } else {
            angle = -HALF_PI_F;
            if (x >= 0.0f) angle += base_angle;
            else           angle -= base_angle;
        }
    }
    return angle;
}

// one output of gr_quadrature_demod_cf::work (general/gr_quadrature_demod_cf.cc:57-59)
template <class Tab>
__device__ __forceinline__ float quad_demod_one(float2 cur, float2 prev, float gain, Tab tab)
{
    float2 product = cmul_ref(cur, make_float2(prev.x, -prev.y));   // in[i] * conj(in[i-1])
    return gain * fast_atan2f(product.y, product.x, tab);
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
