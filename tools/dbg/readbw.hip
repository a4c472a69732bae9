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

// the FIR kernel's load path, feature by feature: MODE bit 0 = raw buffer loads through a descriptor, bit 1 = two
// workgroup barriers per tile (+ a wave max through LDS), bit 2 = tiles overlap by 320 samples (advance 63488 B),
// bit 3 = tiles dealt stream-fastest over 64 streams of 80 MB
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256, 2) fir_like_read(const f4 *__restrict__ x, size_t n_tiles, float *out)
{
    __shared__ float wm[4];
    __shared__ float big[(MODE & 16) ? 19000 : 1];       // bit 4: 76 KB of LDS per workgroup (two workgroups per CU, no more)
    f4 acc = {0, 0, 0, 0};
    const int t = threadIdx.x;
    if (MODE & 16) big[t * 70] = (float)t;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        size_t byte0;
        if (MODE & 8) {
            const size_t s = tile % 64, b = tile / 64;
            byte0 = s * 80000000ull + b * ((MODE & 4) ? 63488ull : 69632ull);
        } else {
            byte0 = tile * ((MODE & 4) ? 63488ull : 69632ull);
        }
        f4 v[17];
        if (MODE & 1) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(x), 0, 0x7fffff00, 0x00020000);
            const char *base = (const char *)x + byte0;
            r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7fffff00, 0x00020000);
#pragma unroll
            for (int u = 0; u < 17; ++u) v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, 16 * t + u * 4096, 0, 0));
        } else {
            const f4 *p = (const f4 *)((const char *)x + byte0) + t;
#pragma unroll
            for (int u = 0; u < 17; ++u) v[u] = p[u * 256];
        }
        float m = 0;
#pragma unroll
        for (int u = 0; u < 17; ++u) { acc += v[u]; m = fmaxf(m, v[u][0]); }
        if (MODE & 2) {
            if ((t & 63) == 0) wm[t >> 6] = m;
            __syncthreads();
            acc[0] += wm[0] + wm[1] + wm[2] + wm[3];
            __syncthreads();
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

__global__ void fill_random(unsigned *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // a float in (-1, 1) with random mantissa
        p[i] = (h & 0x807fffffu) | 0x3f000000u;
    }
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
    {   // sustained: two seconds of back-to-back launches first (the clocks of a loaded chip), then the same timing
        const size_t nt = 64 * 1150;
        for (int i = 0; i < 2500; ++i) hipLaunchKernelGGL((fir_like_read<15>), dim3(512), dim3(256), 0, 0, x, nt, o);
        hipDeviceSynchronize();
        float ms = timeit([&] { hipLaunchKernelGGL((fir_like_read<15>), dim3(512), dim3(256), 0, 0, x, nt, o); });
        printf("%-60s %.3f ms  %.2f TB/s\n", "fir-like, all of it, after 2 s of load", ms, nt * 69632.0 / ms / 1e9);
    }
    for (int pass = 0; pass < 1; ++pass) {
        if (pass == 1) {
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned *)x, bytes / 4);
            hipDeviceSynchronize();
            printf("--- the same with random float data in the buffer ---\n");
        }
        const size_t nt = 64 * 1150;         // 64 streams x 1150 tiles: ~5 GB either way
        auto rp = [&](const char *name, float ms, double tile_bytes) { printf("%-60s %.3f ms  %.2f TB/s\n", name, ms, nt * tile_bytes / ms / 1e9); };
#define RUN(MODE, NAME, TB) rp(NAME, timeit([&] { hipLaunchKernelGGL((fir_like_read<MODE>), dim3(512), dim3(256), 0, 0, x, nt, o); }), TB)
        RUN(0, "fir-like: global loads, contiguous tiles", 69632.0);
        RUN(1, "fir-like: + buffer descriptor loads", 69632.0);
        RUN(2, "fir-like: global loads + 2 barriers per tile", 69632.0);
        RUN(3, "fir-like: buffer loads + 2 barriers", 69632.0);
        RUN(7, "fir-like: buffer + barriers + overlapping tiles", 69632.0);
        RUN(15, "fir-like: all of it, tiles dealt over 64 streams", 69632.0);
        RUN(8, "fir-like: global loads, tiles dealt over 64 streams", 69632.0);
        RUN(31, "fir-like: all of it + 76 KB LDS per workgroup (2 per CU)", 69632.0);
        RUN(16, "fir-like: global loads + 76 KB LDS per workgroup", 69632.0);
    }
    return 0;
}
