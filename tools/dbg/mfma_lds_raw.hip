// Micro-test: wait states needed between v_mfma_f32_16x16x32_f16 and a ds_write that reads its result (gfx950).
// hipcc 7.2 pads this pair with 8 states (s_nop 7); this test finds what the hardware needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, int CHAIN>
__global__ void k(const unsigned *a, const unsigned *b, float *out)
{
    __shared__ float lds[64 * 4];
    int l = threadIdx.x;
    unsigned a0 = a[l * 4], a1 = a[l * 4 + 1], a2 = a[l * 4 + 2], a3 = a[l * 4 + 3];
    unsigned b0 = b[l * 4], b1 = b[l * 4 + 1], b2 = b[l * 4 + 2], b3 = b[l * 4 + 3];
    unsigned addr = (unsigned)(size_t)(&lds[l * 4]);      // LDS byte address (low 32 bits of the generic pointer's offset)
    addr = l * 16;
    asm volatile(
        "v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\t"
        "v_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7\n\t"
        "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        ".rept %c10\n\tv_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t.endr\n\t"
        ".rept %c9\n\ts_nop 0\n\t.endr\n\t"
        "ds_write_b128 %8, v[108:111]\n\t"
        "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        :
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(addr), "i"(NOPS), "i"(CHAIN)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = lds[l * 4 + i];
}

template <int NOPS, int CHAIN>
void run(const unsigned *da, const unsigned *db, float *dout, const std::vector<float> &ref1)
{
    hipMemset(dout, 0, 1024);
    hipLaunchKernelGGL((k<NOPS, CHAIN>), dim3(1), dim3(64), 0, 0, da, db, dout);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0; int lanes[4] = {0, 0, 0, 0}, regs[4] = {0, 0, 0, 0};
    for (int i = 0; i < 256; ++i)
        if (o[i] != ref1[i] * CHAIN) { ++bad; lanes[(i / 4) >> 4]++; regs[i & 3]++; }
    printf("chain %d, %2d wait states: %3d wrong (by lane group %d %d %d %d; by register %d %d %d %d)\n", CHAIN, NOPS, bad,
           lanes[0], lanes[1], lanes[2], lanes[3], regs[0], regs[1], regs[2], regs[3]);
}

int main()
{
    std::vector<_Float16> A(512), B(512);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)((i * 7 % 13) - 6); B[i] = (_Float16)((i * 5 % 11) - 5); }
    unsigned *da, *db; float *dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dout, 1024);
    hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice);
    std::vector<float> ref(256);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15; double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)A[(16 * (k / 8) + row) * 8 + k % 8] * (double)B[(16 * (k / 8) + col) * 8 + k % 8];
        ref[l * 4 + r] = (float)s;
    }
    run<0, 1>(da, db, dout, ref); run<2, 1>(da, db, dout, ref); run<4, 1>(da, db, dout, ref); run<6, 1>(da, db, dout, ref);
    run<8, 1>(da, db, dout, ref); run<10, 1>(da, db, dout, ref); run<12, 1>(da, db, dout, ref); run<14, 1>(da, db, dout, ref);
    run<16, 1>(da, db, dout, ref); run<18, 1>(da, db, dout, ref); run<20, 1>(da, db, dout, ref); run<24, 1>(da, db, dout, ref);
    run<0, 3>(da, db, dout, ref); run<4, 3>(da, db, dout, ref); run<8, 3>(da, db, dout, ref); run<10, 3>(da, db, dout, ref);
    run<12, 3>(da, db, dout, ref); run<14, 3>(da, db, dout, ref); run<16, 3>(da, db, dout, ref); run<18, 3>(da, db, dout, ref);
    run<20, 3>(da, db, dout, ref); run<24, 3>(da, db, dout, ref); run<32, 3>(da, db, dout, ref);
    return 0;
}
