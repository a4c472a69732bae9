// Micro-test: does a VALU write to an MFMA's A/B source VGPR, issued right behind the MFMA, corrupt that MFMA?
// (v_mfma_f32_16x16x32_f16 on gfx950: 4 VGPRs per operand.)  Build: hipcc --offload-arch=gfx950 -O2 mfma_war.hip -o mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, int WHICH>
__global__ void k(const h16x8 *a, const h16x8 *b, f32x4 *out)
{
    int l = threadIdx.x;
    h16x8 A = a[l], B = b[l];
    f32x4 acc = {0, 0, 0, 0};
    // the MFMA, then NOPS s_nop 0, then a VALU overwrite of one source register (WHICH: 0..3 = dword of B, 4..7 = dword of A)
    asm volatile(
        "s_nop 7\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
        ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 %3, 0x7e007e00\n\t"      // NaN halves
        "s_nop 7\n\ts_nop 7\n\ts_nop 7"
        : "+v"(acc), "+v"(A), "+v"(B)
        : "v"(0), "i"(NOPS));
    (void)WHICH;
    out[l] = acc;
}

// variant that really overwrites a chosen dword of B or A
#define KERNEL(NAME, REGEXPR)                                                                          \
    template <int NOPS> __global__ void NAME(const h16x8 *a, const h16x8 *b, f32x4 *out)               \
    {                                                                                                  \
        int l = threadIdx.x;                                                                           \
        h16x8 A = a[l], B = b[l];                                                                      \
        f32x4 acc = {0, 0, 0, 0};                                                                      \
        asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t.rept %c3\n\ts_nop 0\n\t.endr\n\t" \
                     "v_mov_b32 " REGEXPR ", 0x7e007e00\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                \
                     : "+v"(acc), "+v"(A), "+v"(B) : "i"(NOPS));                                      \
        out[l] = acc;                                                                                  \
    }
// %2 is a 4-register tuple v[n:n+3]; there is no syntax for "its k-th dword" in inline asm, so use 4 scalars instead
template <int NOPS, int DW, bool ISB>
__global__ void k2(const unsigned *a, const unsigned *b, f32x4 *out)
{
    int l = threadIdx.x;
    unsigned a0 = a[l * 4], a1 = a[l * 4 + 1], a2 = a[l * 4 + 2], a3 = a[l * 4 + 3];
    unsigned b0 = b[l * 4], b1 = b[l * 4 + 1], b2 = b[l * 4 + 2], b3 = b[l * 4 + 3];
    f32x4 acc = {0, 0, 0, 0};
    // force consecutive registers by naming them: v[100:103] = A, v[104:107] = B, v[108:111] = acc
    asm volatile(
        "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t"
        "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\t"
        "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "v_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t"
        ".rept %c9\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 v%c10, 0x7e007e00\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        "v_mov_b32 %0, v108\n\t"
        : "=v"(acc[0])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "i"(NOPS), "i"((ISB ? 104 : 100) + DW)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    asm volatile("v_mov_b32 %0, v109\n\tv_mov_b32 %1, v110\n\tv_mov_b32 %2, v111" : "=v"(acc[1]), "=v"(acc[2]), "=v"(acc[3]) :: "v109", "v110", "v111");
    out[l] = acc;
}

template <int NOPS, int DW, bool ISB>
int run(const unsigned *da, const unsigned *db, f32x4 *dout, const std::vector<float> &ref)
{
    hipMemset(dout, 0, 64 * sizeof(f32x4));
    hipLaunchKernelGGL((k2<NOPS, DW, ISB>), dim3(1), dim3(64), 0, 0, da, db, dout);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0, firstlane = -1;
    for (int i = 0; i < 256; ++i)
        if (!(o[i] == ref[i])) { ++bad; if (firstlane < 0) firstlane = i / 4; }
    printf("%s dword %d, %d nops: %d wrong values%s\n", ISB ? "B" : "A", DW, NOPS, bad, bad ? "" : " (clean)");
    return bad;
}

int main()
{
    std::vector<_Float16> A(64 * 8), B(64 * 8);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)((i * 7 % 13) - 6); B[i] = (_Float16)((i * 5 % 11) - 5); }
    unsigned *da, *db; f32x4 *dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dout, 64 * sizeof(f32x4));
    hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice);
    // reference: 8 nops of distance and an overwrite of a register that is not an operand (v112)
    std::vector<float> ref(256);
    {
        hipLaunchKernelGGL((k2<7, 12, false>), dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(ref.data(), dout, 1024, hipMemcpyDeviceToHost);
        // host check of the reference itself
        double worst = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int row = 4 * (l >> 4) + r, col = l & 15; double s = 0;
            for (int k = 0; k < 32; ++k) s += (double)A[(16 * (k / 8) + row) * 8 + k % 8] * (double)B[(16 * (k / 8) + col) * 8 + k % 8];
            worst = fmax(worst, fabs(s - ref[l * 4 + r]));
        }
        printf("reference vs host: worst %g\n", worst);
    }
    run<0, 0, true>(da, db, dout, ref); run<0, 1, true>(da, db, dout, ref); run<0, 2, true>(da, db, dout, ref); run<0, 3, true>(da, db, dout, ref);
    run<1, 0, true>(da, db, dout, ref); run<1, 3, true>(da, db, dout, ref); run<2, 0, true>(da, db, dout, ref); run<2, 3, true>(da, db, dout, ref);
    run<3, 3, true>(da, db, dout, ref); run<4, 3, true>(da, db, dout, ref);
    run<0, 0, false>(da, db, dout, ref); run<0, 1, false>(da, db, dout, ref); run<0, 2, false>(da, db, dout, ref); run<0, 3, false>(da, db, dout, ref);
    run<1, 3, false>(da, db, dout, ref); run<2, 3, false>(da, db, dout, ref); run<3, 3, false>(da, db, dout, ref);
    return 0;
}
