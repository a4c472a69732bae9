// capi_mmcc.hip -- digital_clock_recovery_mm_cc: kernel and C ABI.
//
// Reference: gr-digital/lib/digital_clock_recovery_mm_cc.cc:49-215, include/digital_clock_recovery_mm_cc.h:67-110,
// filter/gri_mmse_fir_interpolator_cc.cc:33-80 (the 8-tap / 129-phase table through gr_fir_ccf).
// Same plan as mm_kernel (digital_kernels.hip): the loop is serial and data dependent, so one wavefront per
// stream; all lanes stage a window of complex input into LDS, lanes 0..7 each form one tap product (re, im)
// and DPP row shifts add them in gr_fir_ccf_generic's order (N_UNROLL = 2: a0 = ((0+p0)+p2)+p4)+p6,
// a1 likewise over the odd taps, out = a0 + a1); the loop state is computed redundantly by every lane, the
// sample position is a wave-uniform scalar; outputs go through LDS and are written coalesced per window.
// Every float operation is a single unfused IEEE op in the reference's order: bit-exact.
#include <cmath>

#include "device_math.h"
#include "grhip_internal.h"

using namespace grhip;

namespace grhip {

constexpr int MMC_CH = 2048;        // complex samples per window
constexpr int MMC_NTAPS = 8;
constexpr int MMC_NSTEPS = 128;
constexpr int MMC_OUT = 1024;
constexpr int MMC_FUDGE = 16;       // .cc:36

struct MMccState {
    float mu, omega, min_omega, omega_mid, max_omega;
    float gain_omega, gain_mu, omega_relative_limit;
    float2 p_2T, p_1T, p_0T, c_2T, c_1T, c_0T;
    int produced, consumed;
};

template <int N>
__device__ __forceinline__ float row_shl_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// gr_fir_ccf_generic::filter over 8 taps, lane k holds product k; the result is valid in lane 0
__device__ __forceinline__ float ccf8_sum(float p)
{
    float s = p + row_shl_f<2>(p);                 // lane 0: p0+p2, lane 1: p1+p3
    s = s + row_shl_f<4>(p);
    s = s + row_shl_f<6>(p);
    s = s + row_shl_f<1>(s);                       // a0 + a1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0));
}

template <bool ERR>
__global__ void __launch_bounds__(64)
mmcc_kernel(MMccState *__restrict__ state, int noutput_items, int ninput_items, const float2 *__restrict__ x,
            float2 *__restrict__ y, float *__restrict__ foptr, const float *__restrict__ mmse_rev)
{
    __shared__ float2 s_in[MMC_CH];
    __shared__ float s_taps[MMC_NTAPS * (MMC_NSTEPS + 1)];
    __shared__ float2 s_out[MMC_OUT];
    __shared__ float s_err[MMC_OUT];
    const int lane = threadIdx.x;
    for (int i = lane; i < MMC_NTAPS * (MMC_NSTEPS + 1); i += 64) s_taps[i] = mmse_rev[i];

    const MMccState st = *state;
    float mu = st.mu, omega = st.omega;
    float2 p2 = st.p_2T, p1 = st.p_1T, p0 = st.p_0T, c2 = st.c_2T, c1 = st.c_1T, c0 = st.c_0T;
    const float omega_mid = st.omega_mid, gain_omega = st.gain_omega, gain_mu = st.gain_mu, rel = st.omega_relative_limit;
    const float lim = ERR ? 4.0f : 1.0f;           // .cc:153 / .cc:185
    int ii = 0, oo = 0;
    const int ni = ninput_items - MMC_NTAPS - MMC_FUDGE;     // .cc:130
    const int k = lane & 7;
    const float *tapcol = &s_taps[k * (MMC_NSTEPS + 1)];
    bool done = !(oo < noutput_items && ii < ni);

    while (!done) {
        const int base = ii, obase = oo;
        for (int ib = lane; ib < MMC_CH; ib += 64 * 16) {    // sixteen independent loads in flight per lane
            float2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long long g = (long long)base + ib + 64 * q;
                v[q] = make_float2(0.f, 0.f);
                if (g >= 0 && g < ninput_items) v[q] = x[g];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) s_in[ib + 64 * q] = v[q];
        }
        __syncthreads();
        const int lim_i = base + MMC_CH - MMC_NTAPS;
        const float2 *xk = &s_in[k - base];
        while (oo < noutput_items && ii < ni && ii <= lim_i && ii >= base && oo - obase < MMC_OUT) {
            p2 = p1; p1 = p0;
            int imu = (int)__builtin_rintf(mu * (float)MMC_NSTEPS);
            imu = imu < 0 ? 0 : (imu > MMC_NSTEPS ? MMC_NSTEPS : imu);
            const float tap = tapcol[imu];
            const float2 v = xk[ii];
            p0.x = ccf8_sum(__builtin_fmaf(tap, v.x, 0.0f));        // 0 + tap*x, product rounded once
            p0.y = ccf8_sum(__builtin_fmaf(tap, v.y, 0.0f));
            c2 = c1; c1 = c0;
            c0 = make_float2(p0.x > 0 ? 1.0f : 0.0f, p0.y > 0 ? 1.0f : 0.0f);     // slicer_0deg
            // x = (c0 - c2) * conj(p1); y = (p0 - p2) * conj(c1); mm_val = real(y - x)   (.cc:147-150)
            const float ar = c0.x - c2.x, ai = c0.y - c2.y, nb = -p1.y;
            const float xr = ar * p1.x - ai * nb;
            const float br = p0.x - p2.x, bi = p0.y - p2.y, nc = -c1.y;
            const float yr = br * c1.x - bi * nc;
            float mm_val = yr - xr;
            s_out[oo - obase] = p0;
            mm_val = branchless_clip(mm_val, lim);
            omega = omega + gain_omega * mm_val;
            omega = omega_mid + branchless_clip(omega - omega_mid, rel);
            mu = mu + omega + gain_mu * mm_val;
            const float fl = __builtin_floorf(mu);
            ii = __builtin_amdgcn_readfirstlane(ii + (int)fl);
            mu = mu - fl;
            if (ERR) s_err[oo - obase] = mm_val;
            oo++;
            if (ii < 0) ii = 0;                                      // .cc:165-166
        }
        __syncthreads();
        for (int i = lane; i < oo - obase; i += 64) {
            y[obase + i] = s_out[i];
            if (ERR) foptr[obase + i] = s_err[i];
        }
        done = !(oo < noutput_items && ii < ni);
        __syncthreads();
    }
    if (lane == 0) {
        MMccState so = st;
        so.mu = mu; so.omega = omega;
        so.p_2T = p2; so.p_1T = p1; so.p_0T = p0; so.c_2T = c2; so.c_1T = c1; so.c_0T = c0;
        so.produced = oo; so.consumed = ii;
        *state = so;
    }
}

}  // namespace grhip

struct grhip_clock_recovery_mm_cc : HandleBase {
    MMccState host{};            // parameters; the loop state proper lives on the device
    DevBuf d_state;
    const DeviceTables *tabs = nullptr;
    bool dirty = true;           // host parameters newer than the device copy

    void set_omega(float omega)  // .h:76-81, mixed float/double arithmetic as written there
    {
        host.omega = omega;
        host.min_omega = omega * (1.0 - host.omega_relative_limit);
        host.max_omega = omega * (1.0 + host.omega_relative_limit);
        host.omega_mid = 0.5 * (host.min_omega + host.max_omega);
    }
    // bring the device state to the host (mu, omega and the sample history change every call)
    int pull(hipStream_t st)
    {
        if (dirty) return GRHIP_OK;
        GRHIP_HIP(hipMemcpyAsync(&host, d_state.p, sizeof(host), hipMemcpyDeviceToHost, st));
        GRHIP_HIP(hipStreamSynchronize(st));
        return GRHIP_OK;
    }
    int push(hipStream_t st)
    {
        if (!dirty) return GRHIP_OK;
        GRHIP_HIP(hipMemcpyAsync(d_state.p, &host, sizeof(host), hipMemcpyHostToDevice, st));
        GRHIP_HIP(hipStreamSynchronize(st));
        dirty = false;
        return GRHIP_OK;
    }
};

extern "C" {

int grhip_clock_recovery_mm_cc_create(grhip_clock_recovery_mm_cc **h, float omega, float gain_omega, float mu,
                                      float gain_mu, float omega_relative_limit, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (omega <= 0.0) return fail(GRHIP_ERANGE, "clock rate must be > 0");                        // .cc:62-63
    if (gain_mu < 0 || gain_omega < 0) return fail(GRHIP_ERANGE, "Gains must be non-negative");   // .cc:64-65
    auto *b = new (std::nothrow) grhip_clock_recovery_mm_cc();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    memset(&b->host, 0, sizeof(b->host));
    b->host.mu = mu; b->host.gain_omega = gain_omega; b->host.gain_mu = gain_mu;
    b->host.omega_relative_limit = omega_relative_limit;
    b->set_omega(omega);
    int rc = b->init_device(device);
    if (!rc) rc = get_device_tables(device, &b->tabs);
    if (!rc) rc = b->d_state.reserve(sizeof(MMccState));
    if (rc) { grhip_clock_recovery_mm_cc_destroy(b); return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_clock_recovery_mm_cc_destroy(grhip_clock_recovery_mm_cc *h)
{
    if (!h) return;
    (void)h->bind();
    h->d_state.release();
    h->destroy_base();
    delete h;
}

int grhip_clock_recovery_mm_cc_forecast(grhip_clock_recovery_mm_cc *h, int noutput_items)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    int rc = h->bind();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->setter_mutex);
    if ((rc = h->pull(h->own_stream))) return rc;
    return (int)ceil((noutput_items * h->host.omega) + MMC_NTAPS) + MMC_FUDGE;                    // .cc:82-83
}

int grhip_clock_recovery_mm_cc_history(const grhip_clock_recovery_mm_cc *) { return 3; }          // .cc:69

#define MMCC_GET(name, field)                                                             \
    int grhip_clock_recovery_mm_cc_##name(grhip_clock_recovery_mm_cc *h, float *v)        \
    {                                                                                     \
        if (!h || !v) return fail(GRHIP_EINVAL, "null argument");                        \
        int rc = h->bind();                                                               \
        if (rc) return rc;                                                                \
        std::lock_guard<std::mutex> lk(h->setter_mutex);                                  \
        if ((rc = h->pull(h->own_stream))) return rc;                                     \
        *v = h->host.field;                                                               \
        return GRHIP_OK;                                                                  \
    }
MMCC_GET(mu, mu)
MMCC_GET(omega, omega)
MMCC_GET(gain_mu, gain_mu)
MMCC_GET(gain_omega, gain_omega)

#define MMCC_SET(name, stmt)                                                              \
    int grhip_clock_recovery_mm_cc_set_##name(grhip_clock_recovery_mm_cc *h, float v)     \
    {                                                                                     \
        if (!h) return fail(GRHIP_EINVAL, "null handle");                                 \
        int rc = h->bind();                                                               \
        if (rc) return rc;                                                                \
        std::lock_guard<std::mutex> lk(h->setter_mutex);                                  \
        if ((rc = h->pull(h->own_stream))) return rc;                                     \
        stmt;                                                                             \
        h->dirty = true;                                                                  \
        return GRHIP_OK;                                                                  \
    }
MMCC_SET(mu, h->host.mu = v)                    // .h:75
MMCC_SET(omega, h->set_omega(v))                // .h:76-81
MMCC_SET(gain_mu, h->host.gain_mu = v)          // .h:73
MMCC_SET(gain_omega, h->host.gain_omega = v)    // .h:74

// d_in / d_out / d_err are device pointers; d_err may be null (the reference's second output is optional and
// selects the clip limit, .cc:137-139).  Returns the items produced, *consumed as consume_each() would get.
static int mmcc_run(grhip_clock_recovery_mm_cc *h, int noutput_items, int ninput_items, const void *d_in, void *d_out,
                    float *d_err, int *consumed, hipStream_t st)
{
    int rc;
    {
        std::lock_guard<std::mutex> lk(h->setter_mutex);
        if ((rc = h->push(st))) return rc;
    }
    if (d_err)
        hipLaunchKernelGGL(mmcc_kernel<true>, dim3(1), dim3(64), 0, st, h->d_state.as<MMccState>(), noutput_items,
                           ninput_items, (const float2 *)d_in, (float2 *)d_out, d_err, h->tabs->mmse_rev);
    else
        hipLaunchKernelGGL(mmcc_kernel<false>, dim3(1), dim3(64), 0, st, h->d_state.as<MMccState>(), noutput_items,
                           ninput_items, (const float2 *)d_in, (float2 *)d_out, (float *)nullptr, h->tabs->mmse_rev);
    GRHIP_HIP(hipGetLastError());
    int pc[2];
    GRHIP_HIP(hipMemcpyAsync(pc, (char *)h->d_state.p + offsetof(MMccState, produced), sizeof(pc), hipMemcpyDeviceToHost, st));
    GRHIP_HIP(hipStreamSynchronize(st));
    if (consumed) *consumed = pc[1] > 0 ? pc[1] : 0;
    return pc[0];
}

int grhip_clock_recovery_mm_cc_general_work_device(grhip_clock_recovery_mm_cc *h, int noutput_items, int ninput_items,
                                                   const void *d_in, void *d_out, float *d_err, int *consumed,
                                                   void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || ninput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    if ((noutput_items && !d_out) || (ninput_items && !d_in)) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    return mmcc_run(h, noutput_items, ninput_items, d_in, d_out, d_err, consumed, h->pick(stream));
}

int grhip_clock_recovery_mm_cc_general_work(grhip_clock_recovery_mm_cc *h, int noutput_items, int ninput_items,
                                            const void *in, void *out, float *err, int *consumed)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0 || ninput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    if ((noutput_items && !out) || (ninput_items && !in)) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->own_stream;
    if ((rc = h->stage_in.reserve((size_t)ninput_items * 8 + 8))) return rc;
    if ((rc = h->stage_out.reserve((size_t)noutput_items * 12 + 16))) return rc;
    if (ninput_items) GRHIP_H2D(h, h->stage_in.p, in, (size_t)ninput_items * 8, st);
    float *d_err = err ? (float *)((char *)h->stage_out.p + (size_t)noutput_items * 8) : nullptr;
    int n = mmcc_run(h, noutput_items, ninput_items, h->stage_in.p, h->stage_out.p, d_err, consumed, st);
    if (n <= 0) return n;
    GRHIP_D2H(h, out, h->stage_out.p, (size_t)n * 8, st);
    if (err) GRHIP_HIP(hipMemcpyAsync(err, d_err, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    GRHIP_HIP(hipStreamSynchronize(st));
    return n;
}

}  // extern "C"
