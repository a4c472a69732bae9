// Micro-benchmark: what does a read-only stream reach on this part, by access shape?  (context for roofline.frac of
// the FIR kernels: they read 8 B and write 1 B per sample)   build: hipcc --offload-arch=gfx950 -O3 readbw.hip -o readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride over 16-byte items, U loads in flight per lane
template <int U>
__global__ void __launch_bounds__(256) stream_read(const f4 *__restrict__ x, size_t n, float *out)
{
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&x[i + u * stride]);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// tile-shaped: persistent workgroups, each takes contiguous tiles of TILE_KB, all of a tile's loads issued at once
// (the FIR kernel's shape: 17 x 16 B per lane, 256 lanes)
template <int ROUNDS, bool NT>
__global__ void __launch_bounds__(256) tile_read(const f4 *__restrict__ x, size_t n_tiles, float *out)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const f4 *p = x + t * (size_t)(ROUNDS * 256) + threadIdx.x;
        f4 v[ROUNDS];
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) v[u] = NT ? __builtin_nontemporal_load(&p[u * 256]) : p[u * 256];
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <class F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 30; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20;
}

int main()
{
    const size_t bytes = 5ull << 30;
    f4 *x; float *o;
    hipMalloc(&x, bytes); hipMalloc(&o, 64);
    hipMemset(x, 1, bytes);
    const size_t n = bytes / 16;
    auto rep = [&](const char *name, float ms) { printf("%-52s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); };
    rep("grid-stride, 4 in flight, 2048 WGs", timeit([&] { hipLaunchKernelGGL(stream_read<4>, dim3(2048), dim3(256), 0, 0, x, n, o); }));
    rep("grid-stride, 8 in flight, 2048 WGs", timeit([&] { hipLaunchKernelGGL(stream_read<8>, dim3(2048), dim3(256), 0, 0, x, n, o); }));
    rep("grid-stride, 16 in flight, 1024 WGs", timeit([&] { hipLaunchKernelGGL(stream_read<16>, dim3(1024), dim3(256), 0, 0, x, n, o); }));
    rep("grid-stride, 16 in flight, 512 WGs", timeit([&] { hipLaunchKernelGGL(stream_read<16>, dim3(512), dim3(256), 0, 0, x, n, o); }));
    const size_t nt17 = n / (17 * 256), nt8 = n / (8 * 256), nt4 = n / (4 * 256);
    rep("tiles of 68 KB (17 rounds), 512 WGs", timeit([&] { hipLaunchKernelGGL((tile_read<17, false>), dim3(512), dim3(256), 0, 0, x, nt17, o); }));
    rep("tiles of 68 KB (17 rounds), 512 WGs, nontemporal", timeit([&] { hipLaunchKernelGGL((tile_read<17, true>), dim3(512), dim3(256), 0, 0, x, nt17, o); }));
    rep("tiles of 68 KB (17 rounds), 1024 WGs", timeit([&] { hipLaunchKernelGGL((tile_read<17, false>), dim3(1024), dim3(256), 0, 0, x, nt17, o); }));
    rep("tiles of 32 KB (8 rounds), 1024 WGs", timeit([&] { hipLaunchKernelGGL((tile_read<8, false>), dim3(1024), dim3(256), 0, 0, x, nt8, o); }));
    rep("tiles of 32 KB (8 rounds), 2048 WGs", timeit([&] { hipLaunchKernelGGL((tile_read<8, false>), dim3(2048), dim3(256), 0, 0, x, nt8, o); }));
    rep("tiles of 16 KB (4 rounds), 2048 WGs", timeit([&] { hipLaunchKernelGGL((tile_read<4, false>), dim3(2048), dim3(256), 0, 0, x, nt4, o); }));
    return 0;
}
