// Micro-test 2: MFMA -> ds_write wait states with a SECOND wave on the same SIMD issuing MFMAs back to back.
// 512-thread block = 8 waves = 2 per SIMD.  Waves 0-3 run the measured sequence REPS times; waves 4-7 spam MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, bool SPAM>
__global__ void __launch_bounds__(512) k(const unsigned *a, const unsigned *b, const float *ref, int *nbad, int reps)
{
    __shared__ float lds[8 * 64 * 4];
    int t = threadIdx.x, l = t & 63, w = t >> 6;
    unsigned a0 = a[l * 4], a1 = a[l * 4 + 1], a2 = a[l * 4 + 2], a3 = a[l * 4 + 3];
    unsigned b0 = b[l * 4], b1 = b[l * 4 + 1], b2 = b[l * 4 + 2], b3 = b[l * 4 + 3];
    unsigned addr = (w * 64 + l) * 16;
    float r0 = ref[l * 4], r1 = ref[l * 4 + 1], r2 = ref[l * 4 + 2], r3 = ref[l * 4 + 3];
    int bad = 0;
    if (w >= 4) {
        if (SPAM) {
            for (int it = 0; it < reps * 3; ++it)
                asm volatile(
                    "v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\t"
                    "v_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7\n\t"
                    "s_nop 4\n\t"
                    ".rept 12\n\tv_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t"
                    "v_mfma_f32_16x16x32_f16 v[112:115], v[100:103], v[104:107], v[112:115]\n\t.endr\n\t"
                    :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
                    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
        }
        return;
    }
    for (int it = 0; it < reps; ++it) {
        asm volatile(
            "v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\t"
            "v_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7\n\t"
            "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            ".rept 3\n\tv_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t.endr\n\t"
            ".rept %c9\n\ts_nop 0\n\t.endr\n\t"
            "ds_write_b128 %8, v[108:111]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            :
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(addr), "i"(NOPS)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory");
        const float *p = &lds[(w * 64 + l) * 4];
        volatile const float *vp = p;
        if (vp[0] != 3 * r0 || vp[1] != 3 * r1 || vp[2] != 3 * r2 || vp[3] != 3 * r3) ++bad;
    }
    if (bad) atomicAdd(nbad, bad);
}

template <int NOPS, bool SPAM>
void run(const unsigned *da, const unsigned *db, const float *dref, int *dn)
{
    hipMemset(dn, 0, 4);
    hipLaunchKernelGGL((k<NOPS, SPAM>), dim3(256), dim3(512), 0, 0, da, db, dref, dn, 2000);
    int n = 0;
    hipMemcpy(&n, dn, 4, hipMemcpyDeviceToHost);
    printf("%s partner, %2d wait states: %d wrong lane-results of %d\n", SPAM ? "MFMA-issuing" : "idle        ", NOPS, n, 256 * 4 * 64 * 2000);
}

int main()
{
    std::vector<_Float16> A(512), B(512);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)((i * 7 % 13) - 6); B[i] = (_Float16)((i * 5 % 11) - 5); }
    unsigned *da, *db; float *dref; int *dn;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dref, 1024); hipMalloc(&dn, 4);
    hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice);
    std::vector<float> ref(256);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15; double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)A[(16 * (k / 8) + row) * 8 + k % 8] * (double)B[(16 * (k / 8) + col) * 8 + k % 8];
        ref[l * 4 + r] = (float)s;
    }
    hipMemcpy(dref, ref.data(), 1024, hipMemcpyHostToDevice);
    run<4, false>(da, db, dref, dn); run<6, false>(da, db, dref, dn); run<8, false>(da, db, dref, dn);
    run<4, true>(da, db, dref, dn); run<6, true>(da, db, dref, dn); run<8, true>(da, db, dref, dn); run<10, true>(da, db, dref, dn);
    run<12, true>(da, db, dref, dn); run<16, true>(da, db, dref, dn); run<20, true>(da, db, dref, dn); run<24, true>(da, db, dref, dn);
    run<32, true>(da, db, dref, dn); run<40, true>(da, db, dref, dn);
    return 0;
}
