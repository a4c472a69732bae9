// capi_pfbdec.hip -- gr_pfb_decimator_ccf: kernel and C ABI.
//
// Reference: gnuradio-core/src/lib/filter/gr_pfb_decimator_ccf.cc:43-68 (constructor), 77-111 (set_taps),
// 130-180 (work).  out[i] = sum_j f_j[i] * exp(+2*pi*i*j*chan/M), f_j[i] = gr_fir_ccf(d_taps[j]).filter(&in_{M-1-j}[i]),
// d_taps[j][t] = taps[j + t*M] (the reference gets the sum from an M-point backward FFT and keeps bin `chan`).
//
// pfb_dec_kernel: one workgroup = 1024 outputs, four ADJACENT outputs per lane: eight samples slide through
// registers (two ds_read_b128 per four taps), every wave-uniform tap (scalar load) feeds four packed FMAs, so
// the inner loop is VALU work, not LDS traffic (one read per MAC ran at the LDS rate: 34 TFLOP/s).  The M
// input streams are staged through LDS one after the other (1024 + taps_per_filter + 3 samples each); the
// partial sum of a stream is multiplied by its rotator and accumulated in registers, so traffic is the
// algorithmic 8*M B in + 8 B out per output and nothing else.
#include <cmath>
#include <vector>

#include "grhip_internal.h"

using namespace grhip;

namespace grhip {

constexpr int PFBDEC_TILE = 1024;
constexpr int PFBDEC_MAX_TPF = 1024;

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256)
pfb_dec_kernel(const float2 *__restrict__ in, long long stride, int M, int tpf4, long long avail,
               const float *__restrict__ ftaps_g, const float2 *__restrict__ rot_g, float2 *__restrict__ out, long long nout)
{
    __shared__ __attribute__((aligned(16))) float2 xs[PFBDEC_TILE + PFBDEC_MAX_TPF + 8];
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp ftaps = (cfp)ftaps_g;              // [M][tpf4], reversed and zero padded: ftaps[j][k] multiplies in[i + k]
    const cfp rot = (cfp)(const float *)rot_g;   // [M] (re, im)
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * PFBDEC_TILE;
    const int ns = PFBDEC_TILE + tpf4 + 4;       // samples a tile touches (taps beyond the real ones are zero)
    f32x2 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (f32x2){0.f, 0.f};

    for (int s = 0; s < M; ++s) {
        const int j = M - 1 - s;                 // stream s feeds filter j (.cc:146-149)
        const float2 *x = in + (long long)s * stride + base;
        if (s) __syncthreads();
        for (int ub = t; ub < ns; ub += 256 * 8) {          // eight independent loads in flight per lane
            float2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = ub + 256 * i;
                v[i] = make_float2(0.f, 0.f);
                if (u < ns && base + u < avail) v[i] = x[u];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = ub + 256 * i;
                if (u < ns) xs[u] = v[i];
            }
        }
        __syncthreads();
        // lane t: outputs 4t .. 4t+3; w[] slides over x[4t + k .. 4t + k + 7], four taps per step
        f32x2 f[4], w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = (f32x2){0.f, 0.f};
        const float4 *xq = reinterpret_cast<const float4 *>(xs + 4 * t);
        {
            const float4 a = xq[0], b = xq[1];
            w[4] = (f32x2){a.x, a.y}; w[5] = (f32x2){a.z, a.w}; w[6] = (f32x2){b.x, b.y}; w[7] = (f32x2){b.z, b.w};
        }
        const cfp h = ftaps + (long long)j * tpf4;
        for (int k0 = 0; k0 < tpf4; k0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = w[q + 4];
            const float4 a = xq[(k0 >> 1) + 2], b = xq[(k0 >> 1) + 3];
            w[4] = (f32x2){a.x, a.y}; w[5] = (f32x2){a.z, a.w}; w[6] = (f32x2){b.x, b.y}; w[7] = (f32x2){b.z, b.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float c = h[k0 + kk];
                const f32x2 cv = (f32x2){c, c};
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] = __builtin_elementwise_fma(cv, w[r + kk], f[r]);
            }
        }
        const float wr = rot[2 * j], wi = rot[2 * j + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ax = __builtin_fmaf(f[r].x, wr, __builtin_fmaf(-f[r].y, wi, acc[r].x));
            const float ay = __builtin_fmaf(f[r].x, wi, __builtin_fmaf(f[r].y, wr, acc[r].y));
            acc[r] = (f32x2){ax, ay};
        }
    }
    const long long i = base + 4 * t;
    if (i + 3 < nout) {
        float4 *o = reinterpret_cast<float4 *>(out + i);      // out + base is 32-byte aligned when out is 16-byte aligned
        if ((((uintptr_t)out) & 15) == 0) {
            o[0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            o[1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (i + r < nout) out[i + r] = make_float2(acc[r].x, acc[r].y);
}

}  // namespace grhip

struct grhip_pfb_decimator_ccf : HandleBase {
    unsigned M = 0, chan = 0, taps_per_filter = 0;
    bool updated = false;
    DevBuf d_ftaps, d_rot;

    // set_taps (filter/gr_pfb_decimator_ccf.cc:77-111)
    int set_taps(const float *taps, size_t ntaps)
    {
        const unsigned tpf = (unsigned)ceil((double)ntaps / (double)M);
        if (tpf > (unsigned)PFBDEC_MAX_TPF) return fail(GRHIP_EINVAL, "pfb_decimator_ccf: more than 1024 taps per filter");
        const size_t tot = (size_t)M * tpf;
        const unsigned tpf4 = (tpf + 3u) & ~3u;              // rows zero padded to whole steps of the kernel
        std::vector<float> tmp(tot ? tot : 1, 0.f), ft((size_t)M * tpf4 + 4, 0.f);
        for (size_t i = 0; i < ntaps; ++i) tmp[i] = taps[i];
        for (unsigned i = 0; i < M; i++)
            for (unsigned j = 0; j < tpf; j++) ft[(size_t)i * tpf4 + (tpf - 1 - j)] = tmp[i + (size_t)j * M];   // gr_fir reverses
        int rc = bind();
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(setter_mutex);
        GRHIP_HIP(hipDeviceSynchronize());
        if ((rc = d_ftaps.reserve(ft.size() * 4))) return rc;
        GRHIP_HIP(hipMemcpy(d_ftaps.p, ft.data(), ft.size() * 4, hipMemcpyHostToDevice));
        taps_per_filter = tpf;
        updated = true;
        return GRHIP_OK;
    }
};

extern "C" {

int grhip_pfb_decimator_ccf_create(grhip_pfb_decimator_ccf **h, unsigned decim, const float *taps, size_t ntaps,
                                   unsigned channel, int device)
{
    if (!h || (!taps && ntaps)) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (decim == 0 || decim > 4096) return fail(GRHIP_EINVAL, "pfb_decimator_ccf: decimation must be in 1..4096");
    auto *b = new (std::nothrow) grhip_pfb_decimator_ccf();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->M = decim; b->chan = channel;
    int rc = b->init_device(device);
    if (!rc) {
        // the rotators the backward FFT applies to bin `channel`: exp(+2*pi*i*j*chan/M) (.cc:57, 165-173)
        std::vector<float> rot(2 * (size_t)decim);
        for (unsigned j = 0; j < decim; ++j) {
            const double a = 2.0 * M_PI * (double)(((unsigned long long)j * channel) % decim) / (double)decim;
            rot[2 * j] = (float)cos(a); rot[2 * j + 1] = (float)sin(a);
        }
        rc = b->d_rot.reserve(rot.size() * 4);
        if (!rc && hipMemcpy(b->d_rot.p, rot.data(), rot.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GRHIP_ERUNTIME, "hipMemcpy failed");
    }
    if (!rc) rc = b->set_taps(taps, ntaps);
    if (rc) { grhip_pfb_decimator_ccf_destroy(b); return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_pfb_decimator_ccf_destroy(grhip_pfb_decimator_ccf *h)
{
    if (!h) return;
    (void)h->bind();
    h->d_ftaps.release(); h->d_rot.release();
    h->destroy_base();
    delete h;
}

int grhip_pfb_decimator_ccf_set_taps(grhip_pfb_decimator_ccf *h, const float *taps, size_t ntaps)
{
    if (!h || (!taps && ntaps)) return fail(GRHIP_EINVAL, "null argument");
    return h->set_taps(taps, ntaps);
}

int grhip_pfb_decimator_ccf_history(const grhip_pfb_decimator_ccf *h)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    return (int)h->taps_per_filter;                         // set_history(d_taps_per_filter), .cc:108
}

int grhip_pfb_decimator_ccf_work_device(grhip_pfb_decimator_ccf *h, int noutput_items, const void *d_in,
                                        size_t stream_stride_items, void *d_out, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    unsigned tpf;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if (h->updated) { h->updated = false; return 0; }    // .cc:138-141
        tpf = h->taps_per_filter;
    }
    if (noutput_items == 0) return 0;
    if (!d_in || !d_out) return fail(GRHIP_EINVAL, "null buffer");
    if (h->M > 1 && stream_stride_items < (size_t)noutput_items + (tpf ? tpf - 1 : 0))
        return fail(GRHIP_EINVAL, "pfb_decimator_ccf: stream stride shorter than noutput_items + history - 1");
    const unsigned blocks = (unsigned)(((long long)noutput_items + PFBDEC_TILE - 1) / PFBDEC_TILE);
    hipLaunchKernelGGL(pfb_dec_kernel, dim3(blocks), dim3(256), 0, h->pick(stream), (const float2 *)d_in,
                       (long long)stream_stride_items, (int)h->M, (int)((tpf + 3u) & ~3u),
                       (long long)noutput_items + (tpf ? tpf - 1 : 0), h->d_ftaps.as<float>(),
                       h->d_rot.as<float2>(), (float2 *)d_out, (long long)noutput_items);
    GRHIP_HIP(hipGetLastError());
    return noutput_items;
}

int grhip_pfb_decimator_ccf_work(grhip_pfb_decimator_ccf *h, int noutput_items, const void *const *ins, void *out)
{
    if (!h || !ins) return fail(GRHIP_EINVAL, "null argument");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative noutput_items");
    int rc = h->bind();
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if (h->updated) { h->updated = false; return 0; }
    }
    if (noutput_items == 0) return 0;
    if (!out) return fail(GRHIP_EINVAL, "null buffer");
    const size_t tpf = h->taps_per_filter, per = (size_t)noutput_items + (tpf ? tpf - 1 : 0);
    if ((rc = h->stage_in.reserve(per * h->M * 8))) return rc;
    if ((rc = h->stage_out.reserve((size_t)noutput_items * 8))) return rc;
    hipStream_t st = h->own_stream;
    for (unsigned j = 0; j < h->M; ++j)
        GRHIP_H2D(h, h->stage_in.as<float2>() + (size_t)j * per, ins[j], per * 8, st);
    rc = grhip_pfb_decimator_ccf_work_device(h, noutput_items, h->stage_in.p, per, h->stage_out.p, st);
    if (rc < 0) return rc;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)noutput_items * 8, st);
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}

}  // extern "C"
