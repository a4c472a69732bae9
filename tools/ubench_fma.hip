// micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 at 1/2/4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef ITER_SCALE
#define ITER_SCALE 1
#endif
typedef float float2_ __attribute__((ext_vector_type(2)));

template <bool PK, bool SGPR_OP>
__global__ void __launch_bounds__(256) k(float *out, const float *taps, int iters, unsigned long long *ticks)
{
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    float2_ a[16];
    for (int i = 0; i < 16; ++i) a[i] = (float2_){(float)threadIdx.x * 1e-3f + i, 1.0f};
    float2_ x = (float2_){1.0001f, 0.9999f};
    float h0 = taps[0], h1 = taps[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (PK) {
                    float hh = (i & 1) ? h0 : h1;
                    a[i] = __builtin_elementwise_fma((float2_){hh, hh}, x, a[i]);
                } else {
                    float hh = (i & 1) ? h0 : h1;
                    a[i].x = __builtin_fmaf(hh, x.x, a[i].x);
                    a[i].y = __builtin_fmaf(hh, x.y, a[i].y);
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool PK>
void run(const char *name, int blocks_per_cu)
{
    int ncu = 256;
    float *out, *taps;
    hipMalloc(&out, 4 * 256 * ncu * 8);
    hipMalloc(&taps, 64);
    float h[2] = {0.999f, 1.001f};
    hipMemcpy(taps, h, 8, hipMemcpyHostToDevice);
    unsigned long long *ticks;
    hipMalloc(&ticks, 8);
    int iters = 20000 * ITER_SCALE;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<PK, true><<<ncu * blocks_per_cu, 256>>>(out, taps, 100, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<PK, true><<<ncu * blocks_per_cu, 256>>>(out, taps, iters, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double inst_per_wave = (double)iters * 64 * (PK ? 1 : 2);
    double waves_per_simd = blocks_per_cu;         // 256 threads = 4 waves = 1 per SIMD
    double lane_fma = (double)iters * 64 * 2 * 256.0 * ncu * blocks_per_cu;
    printf("%-10s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD, %.1f TFLOP/s (fma=2)\n", name,
           blocks_per_cu, ms, ms * 1e6 / (inst_per_wave * waves_per_simd), lane_fma * 2 / (ms * 1e-3) / 1e12);
    unsigned long long tk = 0;
    hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
    printf("           s_memtime: %llu ticks = %.1f MHz; %.2f ticks per wave-instr of wave 0\n", tk, tk / (ms * 1e3),
           (double)tk / inst_per_wave);
    hipFree(out); hipFree(taps); hipFree(ticks);
}

static int ITER_SCALE_dummy;
int main()
{
    for (int b : {1, 2, 4}) run<true>("pk_fma", b);
    for (int b : {1, 2}) run<false>("fma", b);
    return 0;
}
